#!/bin/bash
set -u
O=gpurun_out/r5/exp7; mkdir -p $O
for pp in 1 0; do echo "RLCF_F16_PP=$pp"; RLCF_F16_PP=$pp BENCH_ONLY="->f16" timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep "\[p" | tee $O/gemm_pp$pp.txt; done
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "^\[|passed|failed|FAILED|Error|error" | tail -40 > $O/pytest_gpu_full.txt; cat $O/pytest_gpu_full.txt
