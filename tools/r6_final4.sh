O=gpurun_out/r6/final4; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmdline.json 2> $O/bench_driver_cmdline.err; echo "rc=$?"
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"
