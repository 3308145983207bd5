#!/bin/bash
# ResNet every-parameter tuning on checkpoint-grid weights: tests + timing (RN50, N = 64)
set -u
O=gpurun_out/r5/exp18; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > $O/pytest_r5.txt 2>&1; tail -8 $O/pytest_r5.txt
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q -m gpu -k "resnet or bn or rn" > $O/pytest_rn.txt 2>&1; tail -3 $O/pytest_rn.txt
for g in 0 1; do
  echo "== RN50 GRID=$g"; GRID=$g REWARD_ARCH=ViT-B/16 timeout 600 python tools/time_ln_path.py RN50 1000 1 1 full 2>&1 | grep -v "^$" | tail -2
done 2>&1 | tee $O/times.txt
echo "== RN50 GRID=1 RLCF_X3_WLO0=0" | tee -a $O/times.txt; GRID=1 RLCF_X3_WLO0=0 REWARD_ARCH=ViT-B/16 timeout 600 python tools/time_ln_path.py RN50 1000 1 1 full 2>&1 | tail -2 | tee -a $O/times.txt
