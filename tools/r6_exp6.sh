#!/bin/bash
# round 6, experiment 6 (one lease): LayerNorm's interleaved pair rows as whole-line stores (RLCF_LN_LINEST=1) against half-line hi / lo
# stores (=0): the LayerNorm / tower parity tests first, then the driver's step with the arms interleaved, then the op itself under rocprofv3
O=gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round6.py -q -m gpu -x 2>&1 | tail -3 > $O/exp6_ln_linest.txt
for r in 1 2 3; do for T in 0 1; do
  echo "== round $r LN_LINEST=$T"; RLCF_LN_LINEST=$T timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-harness-leg --no-f16-line --no-roofline --timed-repeats 1 --sustain-seconds 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'])"
done; done >> $O/exp6_ln_linest.txt 2>&1
export TMPDIR=/tmp
for T in 0 1; do
  RLCF_LN_LINEST=$T timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ln$T -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-harness-leg --no-f16-line --no-roofline --timed-repeats 1 --sustain-seconds 0 > /dev/null 2>&1
  db=$(find /tmp/prof_ln$T -name "*_results.db" | head -1)
  echo "== kernel table LN_LINEST=$T"; python tools/prof_summary.py "$db" "LN_LINEST=$T" 45 | grep -i "layernorm_fwd\|total kernel"
  rm -rf /tmp/prof_ln$T
done >> $O/exp6_ln_linest.txt 2>&1
cat $O/exp6_ln_linest.txt
