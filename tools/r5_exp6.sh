#!/bin/bash
set -u
O=gpurun_out/r5/exp6; mkdir -p $O
timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep "\[pp\]" | tee $O/gemm.txt
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -5
for f in 1 0; do RLCF_F16_LNFOLD=$f timeout 600 python bench.py --precision f16 --steps 20 --warmup 5 --no-cpu-baseline --no-f16-line --no-harness-leg > $O/bench_f16_fold$f.json 2>$O/bench_f16_fold$f.err; python -c "
import json; d=json.loads(open('$O/bench_f16_fold$f.json').read().strip().splitlines()[-1]); print('f16 LNFOLD=$f images/s', d['value'], d['sustained']['images_per_s_mean']); r=d['roofline']; print([(e['kernel'][:12],round(e.get('tflops',0)), round(e.get('avg_ms',0),3)) for e in r['per_kernel']])" || tail -5 $O/bench_f16_fold$f.err; done
timeout 900 python -m pytest tests -m gpu -x -q -k "f16 or F16 or single" 2>&1 | tail -5
