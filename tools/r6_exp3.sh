#!/bin/bash
# round 6, experiment 3 (one lease): start-time cohorts of the persistent f16 GEMM (RLCF_F16_PP_DESYNC = P, round 5's switch) re-measured with the
# round-6 intra-epilogue stamps: does a cohort's store burst get shorter when only 1 / P of the CUs store at a time?
O=gpurun_out/r6; mkdir -p $O
for P in 1 2 4 8 -2 1; do
  echo "== RLCF_F16_PP_DESYNC=$P (timing)"; RLCF_F16_PP_DESYNC=$P timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep -v amdgpu.ids | head -4
done > $O/exp3_desync.txt 2>&1
for P in 1 4 8; do
  echo "== RLCF_F16_PP_DESYNC=$P (trace, in_proj)"; BENCH_ONLY=in_proj RLCF_F16_PP_DESYNC=$P RLCF_F16_PP_TRACE=1 timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep "pp trace" | grep -v "K-tile pairs" | grep -A1 "wg   8 \|wg 100\|wg 255" | tail -12
done >> $O/exp3_desync.txt 2>&1
cat $O/exp3_desync.txt
