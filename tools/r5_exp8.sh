#!/bin/bash
set -u
O=gpurun_out/r5/exp8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -3
for t in 1 0 1 0; do RLCF_F16_PP_TOUCH=$t timeout 600 python bench.py --precision f16 --steps 20 --warmup 5 --no-cpu-baseline --no-f16-line --no-harness-leg > $O/bench_f16_touch$t.json 2>$O/err.txt; python -c "
import json; d=json.loads(open('$O/bench_f16_touch$t.json').read().strip().splitlines()[-1]); print('f16 TOUCH=$t images/s', d['value'], d['sustained']['images_per_s_mean']); r=d['roofline']; print([(e['kernel'][:12],round(e.get('tflops',0)), round(e.get('avg_ms',0),3)) for e in r['per_kernel']])" || tail -5 $O/err.txt; done
