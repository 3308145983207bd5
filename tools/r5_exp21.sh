#!/bin/bash
# samples in flight through the mirror (test_time_adapt_eval(in_flight=2)): tests + the harness leg of the bench line
set -u
O=gpurun_out/r5/exp21; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "in_flight or conveniences" > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --no-f16-line > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/exp21/bench.json").read().strip().splitlines()[-1])
print(d["value"]); print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in d["harness"].items() if k != "what"})
g = d.get("harness_checkpoint_grid_weights"); print(g and {k: (round(v, 1) if isinstance(v, float) else v) for k, v in g.items() if k != "what"})
PY
