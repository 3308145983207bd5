#!/bin/bash
# round 6, experiment 7 (one lease): attention forward, the wave's output block as whole 128-byte lines through LDS (RLCF_ATTN_LINEST=1) against
# 16-byte pieces per lane (=0): bit-identity tests, the op at the benchmark's shape with the arms interleaved, then the driver's step in both modes
O=gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py -q -m gpu -x -k "attention" 2>&1 | tail -3 > $O/exp7_attn_lines.txt
timeout 600 python tools/r6_attn_lines.py 9 >> $O/exp7_attn_lines.txt 2>&1
for P in f16x3 f16; do for r in 1 2 3; do for T in 0 1; do
  echo "== $P round $r ATTN_LINEST=$T"; RLCF_ATTN_LINEST=$T timeout 400 python bench.py --precision $P --steps 20 --warmup 5 --no-cpu-baseline --no-harness-leg --no-f16-line --no-roofline --timed-repeats 1 --sustain-seconds 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'])"
done; done; done >> $O/exp7_attn_lines.txt 2>&1
cat $O/exp7_attn_lines.txt
