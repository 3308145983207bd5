#!/bin/bash
# round 6, experiment 1 (one lease): the f16 GEMM of round 5 (tools/ab/librlcf_hip_r5gemm.so: round-5 gemm_f16.hip linked against this
# round's other objects) vs this round's kernel with all stores in the epilogue (DEFER=0) vs half of them deferred (DEFER=1), op level,
# A/B/C/A/B/C; then the s_memtime trace of in_proj / c_fc in both forms
O=gpurun_out/r6; mkdir -p $O
for r in 1 2; do
  echo "== round $r: r5 kernel";        RLCF_LIB_PATH=$PWD/tools/ab/librlcf_hip_r5gemm.so timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep -v "amdgpu.ids" | head -5
  echo "== round $r: r6 kernel, DEFER=0"; RLCF_F16_PP_DEFER=0 timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep -v "amdgpu.ids" | head -5
  echo "== round $r: r6 kernel, DEFER=1"; RLCF_F16_PP_DEFER=1 timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep -v "amdgpu.ids" | head -5
done > $O/exp1_ab.txt 2>&1
for d in 0 1; do
  for s in in_proj c_fc; do
    echo "== trace DEFER=$d $s"; BENCH_ONLY=$s RLCF_F16_PP_TRACE=1 RLCF_F16_PP_DEFER=$d timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep "pp trace" | grep -v "K-tile pairs" | sort | uniq -c | sort -rn | head -12
  done
done > $O/exp1_trace.txt 2>&1
cat $O/exp1_ab.txt
head -60 $O/exp1_trace.txt
