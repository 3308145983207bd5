#!/bin/bash
set -u
O=gpurun_out/r5/exp11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -k "f16" 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --precision f16 --steps 20 --warmup 5 --no-cpu-baseline --no-f16-line --no-harness-leg > $O/bench_f16_$i.json 2>$O/err.txt; python -c "
import json; d=json.loads(open('$O/bench_f16_$i.json').read().strip().splitlines()[-1]); print('f16 images/s', d['value'], d['sustained']['images_per_s_mean']); r=d['roofline']; print([(e['kernel'][:12],round(e.get('tflops',0)), round(e.get('avg_ms',0),3)) for e in r['per_kernel']])" || tail -5 $O/err.txt; done
