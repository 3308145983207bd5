#!/bin/bash
set -u
O=gpurun_out/r5/exp13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -k "grid" 2>&1 | tail -5
run() { name=$1; shift; timeout 900 python bench.py "$@" --weights fp16grid --no-cpu-baseline > $O/$name.json 2> $O/$name.err; python -c "
import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$name', round(d['value'],2), 'ms', round(d['ms_per_step'],3), 'dom TF', round(r.get('achieved',0),1), 'harness', {k:round(v,1) for k,v in (d.get('harness') or {}).items() if isinstance(v,float)})" || tail -3 $O/$name.err; }
run grid_driver --steps 20 --warmup 5
run grid_default
run grid_c0 --config 0
run grid_c2 --config 2
run grid_c4 --config 4
run grid_c5 --config 5
run grid_batch1 --batch 1
