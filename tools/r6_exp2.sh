#!/bin/bash
# round 6, experiment 2 (one lease): tile order of the persistent f16 GEMM (RLCF_X3_GROUP = M tiles per scheduling group; >= 100: N-fastest
# inside a group) on the four layer products, op level; the default (0 -> 8, M-fastest for products wider than 4 tiles) first and last
O=gpurun_out/r6; mkdir -p $O
for G in 0 2 4 16 32 104 108 116 0; do
  echo "== RLCF_X3_GROUP=$G"; RLCF_X3_GROUP=$G timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep -v "amdgpu.ids" | head -4
done > $O/exp2_tile_order.txt 2>&1
cat $O/exp2_tile_order.txt
