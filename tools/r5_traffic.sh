#!/bin/bash
# verdict item 5(b): fabric traffic of the dominant GEMM (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes) on the round-5 build, and
# the same under other tile orders (RLCF_X3_GROUP = M tiles per scheduling group; >= 100: N-fastest inside a group) with the launch times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5/traffic; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for G in 0 4 16 104 108 116 132; do
  export RLCF_X3_GROUP=$G
  timeout 300 python $R/tools/gemm_epi_bench.py 252160 > $O/time_g$G.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f_$G -- python $R/tools/gemm_epi_bench.py 252160 > /dev/null 2>&1
  db=$(find /tmp/pmc_f_$G -name "*.db" | head -1); python $R/tools/pmc_summary.py $db gemm_nt_f16x3_v3i > $O/fetch_g$G.txt 2>&1; rm -rf /tmp/pmc_f_$G
  echo "== GROUP=$G"; grep -i "us\|TF" $O/time_g$G.txt | tail -6; cat $O/fetch_g$G.txt | tail -8
done
unset RLCF_X3_GROUP
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -- python $R/tools/gemm_epi_bench.py 252160 > /dev/null 2>&1
db=$(find /tmp/pmc_w -name "*.db" | head -1); python $R/tools/pmc_summary.py $db gemm_nt_f16x3_v3i > $O/write.txt 2>&1; cat $O/write.txt | tail -8
