#!/bin/bash
# round 6, experiment 5 (one lease): line-complete pair stores of the parity kernel's in_proj / c_fc epilogues (RLCF_X3_LINEST=1) against
# the 8-byte hi / lo stores (=0): the parity tests first (bit-identical results required), then the driver's step, arms interleaved
O=gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -m gpu -x 2>&1 | tail -3 > $O/exp5_linest.txt
for r in 1 2 3; do for T in 0 1; do
  echo "== round $r LINEST=$T"; RLCF_X3_LINEST=$T timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-harness-leg --no-f16-line --no-roofline --timed-repeats 1 --sustain-seconds 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'])"
done; done >> $O/exp5_linest.txt 2>&1
cat $O/exp5_linest.txt
