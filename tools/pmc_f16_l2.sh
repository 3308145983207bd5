#!/bin/bash
# L2 hit rate and fabric traffic of the f16 GEMM kernels (tools/gemm_f16_bench.py); one counter group per pass (TCC slots: MI355X_MICROARCH.md)
# usage (GPU box, repo root): BENCH_ONLY=<shape> tools/pmc_f16_l2.sh <tag> [kernel substring]
TAG=${1:-r5l2}; SUB=${2:-gemm_nt_f16_pp}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_l2_$i -- python $R/tools/gemm_f16_bench.py > $O/pmc_l2_$i.log 2>&1
  echo "pass $i rc=$?"
  db=$(find /tmp/pmc_l2_$i -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $db $SUB > $O/pmc_l2_$i.txt 2>&1
  rm -rf /tmp/pmc_l2_$i
done
cat $O/pmc_l2_*.txt
