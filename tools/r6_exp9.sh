#!/bin/bash
# round 6, experiment 9 (one lease): non-temporal forms of the whole-line stores of LayerNorm (RLCF_LN_LINEST=2) and of the attention forward
# (RLCF_ATTN_LINEST=2) against the default policy (=1): the attention op, then the driver's step in both modes, arms interleaved
O=gpurun_out/r6; mkdir -p $O
for P in f16x3 f16; do for r in 1 2 3; do for T in "1 1" "2 1" "1 2" "2 2"; do
  set -- $T
  echo -n "$P round $r LN=$1 ATTN=$2 "; RLCF_LN_LINEST=$1 RLCF_ATTN_LINEST=$2 timeout 400 python bench.py --precision $P --steps 40 --warmup 20 --no-cpu-baseline --no-harness-leg --no-f16-line --no-roofline --timed-repeats 1 --sustain-seconds 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', round(d['value'], 2))"
done; done; done | tee $O/exp9_ln_attn_nt.txt
