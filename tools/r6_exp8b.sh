mkdir -p gpurun_out/r6
for r in 1 2 3 4; do for T in 0 2; do
  echo -n "round $r NT=$T "; RLCF_F16_PP_NT=$T timeout 400 python bench.py --precision f16 --steps 40 --warmup 20 --no-cpu-baseline --no-harness-leg --no-f16-line --no-roofline --timed-repeats 1 --sustain-seconds 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', round(d['value'], 2))"
done; done | tee gpurun_out/r6/exp8b_f16_nt_step.txt
