#!/bin/bash
# kernel table of the single-pass f16 mode (RLCF_PREC_F16) at the driver's pass size + A/B of the parity mode's non-temporal stores
set -u
O=gpurun_out/r5/f16prof; mkdir -p $O
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_f16 -- python bench.py --precision f16 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --no-f16-line --no-harness-leg > $O/prof_f16.log 2>&1
db=$(find /tmp/prof_f16 -name "*_results.db" | head -1)
[ -n "$db" ] && python tools/prof_summary.py "$db" "round 5: rocprofv3 --kernel-trace --stats -- python bench.py --precision f16 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --no-f16-line --no-harness-leg (RLCF_PREC_F16, NOT parity-grade; set-up pass of 20 + 5 warm-up + 20 timed + 20 profiled images = 65)" 65 > $O/kernel_stats_f16.txt
tail -c 1500 $O/prof_f16.log > $O/prof_f16.tail; rm -f $O/prof_f16.log; rm -rf /tmp/prof_f16
head -60 $O/kernel_stats_f16.txt
for nt in 1 0 1 0; do RLCF_X3_NT=$nt timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f16-line --no-harness-leg --no-roofline > $O/ab_nt$nt.json 2>/dev/null; python -c "
import json,sys; d=json.loads(open('$O/ab_nt$nt.json').read().strip().splitlines()[-1]); print('NT=$nt', d['value'], d.get('sustained',{}))"; done
