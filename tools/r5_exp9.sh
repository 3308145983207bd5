#!/bin/bash
set -u
O=gpurun_out/r5/exp9; mkdir -p $O
for s in 0 2 3 4 8 0; do echo "STAGGER=$s"; RLCF_X3_STAGGER=$s timeout 300 python tools/gemm_epi_bench.py 252160 2>&1 | grep "M=252160" | tee -a $O/stagger_$s.txt; done
for s in 0 4 0 4; do RLCF_X3_STAGGER=$s timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f16-line --no-harness-leg --no-roofline > $O/bench_s$s.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_s$s.json').read().strip().splitlines()[-1]); print('STAGGER=$s headline', d['value'], d['sustained']['images_per_s_mean'])"; done
