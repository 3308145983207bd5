#!/bin/bash
# round-5 state capture (one MI355X): f16 GEMM op-level table, attention op-level table, round-5 GPU tests, the driver's bench line
set -u
O=gpurun_out/r5/state; mkdir -p $O
timeout 300 python tools/gemm_f16_bench.py > $O/gemm_f16.txt 2>&1; tail -12 $O/gemm_f16.txt
ATTN_BENCH_ONLY=1 timeout 300 python tools/attn_bench.py > $O/attn.txt 2>&1; tail -12 $O/attn.txt
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > $O/pytest_r5.txt 2>&1; tail -3 $O/pytest_r5.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmdline.json 2> $O/bench_driver_cmdline.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/state/bench_driver_cmdline.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
print(json.dumps(d.get('secondary_f16_single_pass'), indent=1)[:3000])
print(json.dumps(d.get('harness'), indent=1)[:1500])
PY
