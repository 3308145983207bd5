#!/bin/bash
# one-image mode after the few-row kernel rewrite: parity subset + bench
timeout 900 python -m pytest tests -m gpu -x -q -k "skinny or tta or sparse or stream or one_image or retrieval or harness" 2>&1 | tail -3
timeout 600 python bench.py --batch 1 --steps 20 --warmup 5 --no-harness-leg --no-cpu-baseline --no-f16-line > gpurun_out/b1_new.json 2> gpurun_out/b1_new.err; echo rc=$?
python - <<PY
import json
d=json.loads(open("gpurun_out/b1_new.json").read().strip().splitlines()[-1])
print("batch1", d["value"], d["ms_per_step"])
for r in d["roofline"]["per_kernel_all_launches"]:
    if r["dims"][0] <= 256: print(round(r["total_ms_per_image"],3), r["launches"], round(r.get("tflops",0),1), r["dims"], r["kernel"][:40], round(1e3*r["total_ms_per_image"]/r["launches"],1),"us")
PY
