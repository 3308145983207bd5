#!/bin/bash
# round 6, experiment 8 (one lease): non-temporal output stores of the persistent f16 GEMM now that they are whole lines (RLCF_F16_PP_NT=1; read once
# per process): op level, arms interleaved three times; then the f16 mode's step
O=gpurun_out/r6; mkdir -p $O
for r in 1 2 3; do for T in 0 1; do
  echo "== round $r NT=$T"; RLCF_F16_PP_NT=$T timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep "^\[pp\]" | head -4
done; done > $O/exp8_f16_nt.txt 2>&1
for r in 1 2; do for T in 0 1; do
  echo "== step round $r NT=$T"; RLCF_F16_PP_NT=$T timeout 400 python bench.py --precision f16 --steps 20 --warmup 5 --no-cpu-baseline --no-harness-leg --no-f16-line --no-roofline --timed-repeats 1 --sustain-seconds 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'])"
done; done >> $O/exp8_f16_nt.txt 2>&1
cat $O/exp8_f16_nt.txt
