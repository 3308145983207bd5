#!/bin/bash
# last captures of the round on the final build: full GPU suite, the driver's command line (with both secondary lines and the harness leg),
# configs[4] on checkpoint-grid weights
set -u
O=gpurun_out/r5/final3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmdline.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench_driver_cmdline.json').read().strip().splitlines()[-1]); print('headline', d['value'], 'grid', d['secondary_checkpoint_grid_weights']['images_per_s'], 'f16', d['secondary_f16_single_pass']['images_per_s'], {k:round(v,1) for k,v in d['harness'].items() if isinstance(v,float)})"
timeout 900 python bench.py --config 4 --weights fp16grid --no-cpu-baseline > $O/c4_fp16grid.json 2>$O/err.txt; python -c "
import json; d=json.loads(open('$O/c4_fp16grid.json').read().strip().splitlines()[-1]); print('config4 fp16grid', round(d['value'],2), round(d['ms_per_step'],2))"
