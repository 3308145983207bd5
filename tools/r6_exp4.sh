#!/bin/bash
# round 6, experiment 4 (one lease): full-line stores through a per-wave LDS slab (RLCF_F16_PP_TSTORE=1, gemm_nt_f16_pp_kernel<.., TS = 1>) against
# the 32-row x 32-B stores, each with start-time cohorts (RLCF_F16_PP_DESYNC = 1 / 2 / 4); then the intra-epilogue trace of in_proj / c_fc
O=gpurun_out/r6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round6.py -q -k "deferred_stores" 2>&1 | tail -2 > $O/exp4_tstore.txt
for r in 1 2; do for T in 0 1; do for P in 1 2 4; do
  echo "== round $r TSTORE=$T DESYNC=$P"; RLCF_F16_PP_TSTORE=$T RLCF_F16_PP_DESYNC=$P timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep -v amdgpu.ids | head -4
done; done; done >> $O/exp4_tstore.txt 2>&1
for T in 0 1; do for P in 1 4; do for s in in_proj c_fc; do
  echo "== trace TSTORE=$T DESYNC=$P $s"; BENCH_ONLY=$s RLCF_F16_PP_TSTORE=$T RLCF_F16_PP_DESYNC=$P RLCF_F16_PP_TRACE=1 timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep "pp trace" | grep -v "K-tile pairs" | grep -A1 "wg 100\|wg 255" | tail -8
done; done; done >> $O/exp4_tstore.txt 2>&1
cat $O/exp4_tstore.txt
