#!/bin/bash
# 192x256 tile form: bit-identity against the other tile shapes, timing, and the single-image / default bench A/B
mkdir -p gpurun_out/mt3
RLCF_X3_MT3=0 timeout 600 python tools/gemm_mt3_check.py > gpurun_out/mt3/off.txt 2>&1; echo "off rc=$?"
RLCF_X3_MT3=1 timeout 600 python tools/gemm_mt3_check.py > gpurun_out/mt3/on.txt 2>&1; echo "on rc=$?"
grep ^SIG gpurun_out/mt3/off.txt > /tmp/a.txt; grep ^SIG gpurun_out/mt3/on.txt > /tmp/b.txt
if diff /tmp/a.txt /tmp/b.txt > gpurun_out/mt3/sigdiff.txt; then echo "BIT-IDENTICAL"; else echo "SIG DIFF"; head -20 gpurun_out/mt3/sigdiff.txt; fi
grep "^MT3" gpurun_out/mt3/off.txt | head -40; grep "^MT3" gpurun_out/mt3/on.txt | head -40
for m in 0 1; do
  RLCF_X3_MT3=$m timeout 900 python bench.py --batch 1 --steps 20 --warmup 5 --no-harness-leg --no-cpu-baseline > gpurun_out/mt3/b1_$m.json 2> gpurun_out/mt3/b1_$m.err; echo "b1 mt3=$m rc=$?"
  python - <<PY
import json
d=json.loads(open("gpurun_out/mt3/b1_$m.json").read().strip().splitlines()[-1])
print("batch1 mt3=$m", d["value"], d["ms_per_step"])
PY
done
for m in 0 1; do
  RLCF_X3_MT3=$m timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-harness-leg --no-cpu-baseline > gpurun_out/mt3/b20_$m.json 2> gpurun_out/mt3/b20_$m.err; echo "b20 mt3=$m rc=$?"
  python - <<PY
import json
d=json.loads(open("gpurun_out/mt3/b20_$m.json").read().strip().splitlines()[-1])
print("driver-cmdline mt3=$m", d["value"], d["ms_per_step"])
PY
done
