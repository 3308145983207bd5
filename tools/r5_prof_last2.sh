#!/bin/bash
# after making the finalize-time grid check cheap: the grid tests, the driver command line's kernel table and the default bench line
set -u
O=gpurun_out/r5/proflast2; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > $O/pytest_r5.txt 2>&1; tail -3 $O/pytest_r5.txt
C="--no-cpu-baseline --sustain-seconds 0 --no-f16-line --no-harness-leg"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c1 -- python bench.py --steps 20 --warmup 5 $C > $O/prof_c1.log 2>&1
db=$(find /tmp/prof_c1 -name "*_results.db" | head -1)
[ -n "$db" ] && python tools/prof_summary.py "$db" "round 5 last build: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 $C (BASELINE configs[1], float32 weights; 65 images)" 65 > $O/kernel_stats_c1.txt
head -12 $O/kernel_stats_c1.txt | cut -c1-150; grep -n grid_check $O/kernel_stats_c1.txt | cut -c1-150
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/proflast2/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.items() if k.startswith('secondary')})
PY
