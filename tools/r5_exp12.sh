#!/bin/bash
set -u
O=gpurun_out/r5/exp12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -k "grid" 2>&1 | tail -8
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-harness-leg > $O/bench.json 2>$O/err.txt; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('headline', d['value']); print(json.dumps(d['secondary_checkpoint_grid_weights'], indent=1)); print(d['secondary_f16_single_pass']['images_per_s'])" || tail -5 $O/err.txt
