#!/bin/bash
# Fabric traffic of the dominant GEMM (gemm_nt_f16x3_v3i_kernel) per launch: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (counters only with --kernel-trace: MI355X_MICROARCH.md, HBM section) over tools/gemm_epi_bench.py at the token-matrix size of the pass.
# usage (GPU box, repo root): tools/pmc_gemm_traffic.sh <round tag> [M ...]
TAG=${1:-r4}; shift; MS=${@:-252160}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in $MS; do for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_${C}_$M -- python $R/tools/gemm_epi_bench.py $M > $O/pmc_${C}_$M.log 2>&1
  echo "$C M=$M rc=$?"
  db=$(find /tmp/pmc_${C}_$M -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $db gemm_nt_f16x3_v3i > $O/pmc_${C}_$M.txt 2>&1
  rm -rf /tmp/pmc_${C}_$M
done; done
cat $O/pmc_*.txt
