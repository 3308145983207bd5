#!/bin/bash
# round 6, last GPU run: the f16 stream reports against the COMPLETE fp16-autocast fixture (32 samples), the driver's bench command line on the final
# tree, smoke(), and the tests touched since the last full-suite run
O=gpurun_out/r6/final2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round4.py -q -s -k "stream" 2>&1 | grep -E "^\[|sample [0-9]+:|passed|failed" > $O/stream_reports.txt; tail -1 $O/stream_reports.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmdline.json 2> $O/bench_driver_cmdline.err; echo "bench rc=$?"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_views.py tests/test_gpu_round6.py tests/test_gpu_round5.py -q -m gpu 2>&1 | tail -2
