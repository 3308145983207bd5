#!/bin/bash
# fabric traffic of the dominant GEMM at the default bench pass size (32 images per pass: M = 403 456), default tile order, last build;
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (tools/r5_traffic.sh did M = 252 160)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5/traffic32; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in 403456 252160; do
  timeout 300 python $R/tools/gemm_epi_bench.py $M > $O/time_$M.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f_$M -- python $R/tools/gemm_epi_bench.py $M > /dev/null 2>&1
  db=$(find /tmp/pmc_f_$M -name "*.db" | head -1); python $R/tools/pmc_summary.py $db gemm_nt_f16x3_v3i > $O/fetch_$M.txt 2>&1; rm -rf /tmp/pmc_f_$M
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w_$M -- python $R/tools/gemm_epi_bench.py $M > /dev/null 2>&1
  db=$(find /tmp/pmc_w_$M -name "*.db" | head -1); python $R/tools/pmc_summary.py $db gemm_nt_f16x3_v3i > $O/write_$M.txt 2>&1; rm -rf /tmp/pmc_w_$M
  echo "== M=$M"; tail -6 $O/time_$M.txt; tail -8 $O/fetch_$M.txt; tail -8 $O/write_$M.txt
done
