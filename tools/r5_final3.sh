#!/bin/bash
# last tree of the round: whole GPU suite, smoke(), the driver's command line and the default bench line
set -u
O=gpurun_out/r5/final3; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmdline.json 2> $O/bench_driver_cmdline.err
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
for n in ("bench_driver_cmdline", "bench_default"):
    try:
        d = json.loads(open(f"gpurun_out/r5/final3/{n}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(n, round(d["value"], 1), "ms", round(d["ms_per_step"], 2), "frac", round(r["frac"], 4), "traffic", r["traffic"], "busy", r["mfma_busy_counter"],
              "f16", (d.get("secondary_f16_single_pass") or {}).get("images_per_s"), "grid", (d.get("secondary_checkpoint_grid_weights") or {}).get("images_per_s"),
              "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "unreadable:", e)
PY
