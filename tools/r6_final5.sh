#!/bin/bash
# round 6, last verification of the final tree: the GPU suite (non-blocking stream mode), smoke(), the driver's bench command line
O=gpurun_out/r6/final5; mkdir -p $O
RLCF_TEST_STREAM=nonblocking timeout 1500 python -m pytest tests -m gpu -q > $O/suite_nonblocking.txt 2>&1; tail -2 $O/suite_nonblocking.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmdline.json 2> $O/bench_driver_cmdline.err; echo "bench rc=$?"
