#!/bin/bash
set -u
O=gpurun_out/r5/exp10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q -k "mirror or harness or cache or reset or hint or track" 2>&1 | tail -4
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f16-line > $O/bench_harness.json 2>$O/err.txt; python -c "
import json; d=json.loads(open('$O/bench_harness.json').read().strip().splitlines()[-1]); print(d['value']); print({k:v for k,v in d['harness'].items() if k!='what'})"
bash tools/env_switch_smoke.sh 2>&1 | tee $O/env_smoke.txt
