#!/bin/bash
# is the K loop of the 256x256 split-f16 kernel LDS-bound?  LDS activity counters of the layer's four products (three passes), one MI355X
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5/lds; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/lds_$n -- python $R/tools/gemm_epi_bench.py 252160 > /dev/null 2>&1
  db=$(find /tmp/lds_$n -name "*.db" | head -1); python $R/tools/pmc_summary.py $db gemm_nt_f16x3_v3i > $O/$n.txt 2>&1; rm -rf /tmp/lds_$n
  cat $O/$n.txt | cut -c1-200
done
