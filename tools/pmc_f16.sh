#!/bin/bash
# SQ counters of the single-pass f16 GEMM kernel (tools/gemm_f16_bench.py): run on the GPU box from the repo root.
# usage: tools/pmc_f16.sh <out tag> [kernel substring]
TAG=${1:-r5}; SUB=${2:-gemm_nt_f16_p8}      # BENCH_ONLY=<shape name substring> restricts the bench to one shape
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_f16_$i -- python $R/tools/gemm_f16_bench.py > $O/pmc_f16_$i.log 2>&1
  echo "pass $i rc=$?"
  db=$(find /tmp/pmc_f16_$i -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $db $SUB > $O/pmc_f16_$i.txt 2>&1
  rm -rf /tmp/pmc_f16_$i
done
cat $O/pmc_f16_*.txt
