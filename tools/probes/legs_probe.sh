mkdir -p gpurun_out/r6
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f16-line --no-roofline --sustain-seconds 0 --timed-repeats 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); h = d['harness']
        for k, v in h.items():
            if 'legs' in k or k.startswith('images_per_s_one_image_per_pass'): print(k, v)
" | tee gpurun_out/r6/legs_probe.txt
