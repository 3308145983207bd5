#!/bin/bash
# every-parameter tuning on checkpoint-grid weights: two passes again after every reset (the large forward / backward products of a sample)
set -u
O=gpurun_out/r5/exp17; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > $O/pytest_r5.txt 2>&1; tail -5 $O/pytest_r5.txt
for arch in ViT-B/16 ViT-L/14; do
  for g in 0 1; do
    echo "== $arch GRID=$g"; GRID=$g timeout 600 python tools/time_ln_path.py $arch 1000 1 1 full 2>&1 | grep -v "^$" | tail -3
  done
  echo "== $arch GRID=1 RLCF_X3_WLO0=0"; GRID=1 RLCF_X3_WLO0=0 timeout 600 python tools/time_ln_path.py $arch 1000 1 1 full 2>&1 | tail -2
done 2>&1 | tee $O/times.txt
