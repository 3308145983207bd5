#!/bin/bash
set -u
O=gpurun_out/r5/exp14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -k "grid" 2>&1 | tail -8; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f16-line --no-harness-leg --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"headline\", d[\"value\"], d[\"sustained\"][\"images_per_s_mean\"])"
timeout 1200 python -m pytest tests -m gpu -x -q -k "rn or resnet or config5 or ens or RN" 2>&1 | tail -3
for w in fp16grid fp32; do timeout 900 python bench.py --config 4 --weights $w --no-cpu-baseline --sustain-seconds 0 --no-roofline > $O/c4_$w.json 2>$O/err.txt; python -c "
import json; d=json.loads(open('$O/c4_$w.json').read().strip().splitlines()[-1]); print('config4 $w', round(d['value'],2), round(d['ms_per_step'],2))" || tail -3 $O/err.txt; done
RLCF_CONV_GRID=0 timeout 900 python bench.py --config 4 --weights fp16grid --no-cpu-baseline --sustain-seconds 0 --no-roofline > $O/c4_grid_folded.json 2>$O/err.txt; python -c "
import json; d=json.loads(open('$O/c4_grid_folded.json').read().strip().splitlines()[-1]); print('config4 fp16grid RLCF_CONV_GRID=0', round(d['value'],2), round(d['ms_per_step'],2))"
