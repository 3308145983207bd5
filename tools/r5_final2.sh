#!/bin/bash
# refresh of the round's captures on the last build (packed hi-only W in the two-pass kernels): full GPU suite, the driver's command line, the
# checkpoint-grid-weight line of every single-GPU config
set -u
O=gpurun_out/r5/final4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmdline.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench_driver_cmdline.json').read().strip().splitlines()[-1]); print('headline', d['value'], 'grid', d['secondary_checkpoint_grid_weights']['images_per_s'], 'f16', d['secondary_f16_single_pass']['images_per_s'], {k:round(v,1) for k,v in d['harness'].items() if isinstance(v,float)})"
run() { name=$1; shift; timeout 900 python bench.py "$@" --weights fp16grid --no-cpu-baseline > $O/$name.json 2> $O/$name.err; python -c "
import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$name', round(d['value'],2), 'ms', round(d['ms_per_step'],3), 'dom TF', round(r.get('achieved',0),1), 'harness', {k:round(v,1) for k,v in (d.get('harness') or {}).items() if isinstance(v,float)})" || tail -3 $O/$name.err; }
run grid_driver --steps 20 --warmup 5
run grid_default
run grid_c0 --config 0
run grid_c2 --config 2
run grid_c4 --config 4
run grid_c5 --config 5
run grid_batch1 --batch 1
