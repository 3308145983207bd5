#!/bin/bash
set -u
O=gpurun_out/r5/exp22; mkdir -p $O
for r in 1 2; do
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --no-f16-line > $O/bench$r.json 2> $O/bench$r.err
python - $r <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5/exp22/bench{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(d["value"]); print({k.replace("images_per_s_one_image_per_pass_", ""): (round(v, 1) if isinstance(v, float) else v) for k, v in d["harness"].items() if k != "what"})
g = d.get("harness_checkpoint_grid_weights"); print(g and {k.replace("images_per_s_one_image_per_pass_", ""): (round(v, 1) if isinstance(v, float) else v) for k, v in g.items() if k != "what"})
PY
done
