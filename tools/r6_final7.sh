#!/bin/bash
# round 6, the very last tree: GPU suite on the NULL stream, then the driver's bench command line
O=gpurun_out/r6/final7; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/suite_null_stream.txt 2>&1; tail -2 $O/suite_null_stream.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmdline.json 2> $O/bench_driver_cmdline.err; echo "bench rc=$?"
