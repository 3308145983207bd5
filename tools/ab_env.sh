#!/bin/bash
# A/B of one environment switch on bench.py: tools/ab_env.sh VAR "v1 v2 ..." [bench args...]; prints value / ms_per_step per setting, twice
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v timeout 600 python bench.py --no-cpu-baseline --no-f16-line --sustain-seconds 0 "$@" 2>/dev/null | tail -1 > /tmp/ab_env.json
  python - "$VAR" "$v" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab_env.json').read())
print(f"{sys.argv[1]}={sys.argv[2]}: {d['value']:.2f} {d['unit']}  {d['ms_per_step']:.3f} ms/step")
PY
done; done
