# do more hardware queues (GPU_MAX_HW_QUEUES, read by the HIP runtime at start-up; default 4) steady the in-flight legs?
mkdir -p gpurun_out/r6
for Q in 4 8 4 8; do
  echo "== GPU_MAX_HW_QUEUES=$Q"
  GPU_MAX_HW_QUEUES=$Q timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f16-line --no-roofline --sustain-seconds 0 --timed-repeats 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); h = d['harness']
        print('value', round(d['value'], 2), 'one/call', round(h['images_per_s_one_image_per_pass_staged_views'], 2))
        for k, v in h.items():
            if k.endswith('_legs'): print('  ', k[32:], v)
"
done 2>&1 | tee gpurun_out/r6/hwq_probe.txt
