#!/bin/bash
set -u
O=gpurun_out/r5/exp16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -k "grid" 2>&1 | tail -5
for pk in 1 0 1 0; do RLCF_X3_WPK=$pk timeout 300 python bench.py --weights fp16grid --steps 20 --warmup 5 --no-cpu-baseline --no-f16-line --no-harness-leg > $O/grid_pk$pk.json 2>$O/err.txt; python -c "
import json; d=json.loads(open('$O/grid_pk$pk.json').read().strip().splitlines()[-1]); r=d['roofline']; print('WPK=$pk', round(d['value'],2), round(d['sustained']['images_per_s_mean'],2), [(e['kernel'][:12],round(e.get('tflops',0)), round(e.get('avg_ms',0),3)) for e in r['per_kernel'][:4]])" || tail -3 $O/err.txt; done
