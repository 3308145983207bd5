#!/bin/bash
set -u
O=gpurun_out/r5/exp3; mkdir -p $O
for abl in 0 1 2; do echo "ABL=$abl (1 = no stores, 2 = no epilogue)"; RLCF_F16_PP_ABL=$abl BENCH_ONLY="->f16" timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep "\[pp\]" | tee $O/abl_$abl.txt; done
for bk in 1 0 1 0; do RLCF_CONV_BOUND_KERNEL=$bk timeout 600 python bench.py --config 4 --no-cpu-baseline --sustain-seconds 0 --no-roofline > $O/bench_c4_bk$bk.json 2>$O/bench_c4_bk$bk.err; python -c "
import json; d=json.loads(open('$O/bench_c4_bk$bk.json').read().strip().splitlines()[-1]); print('config4 bound_kernel=$bk', d['value'], d['ms_per_step'])"; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f16-line --no-harness-leg --no-roofline > $O/bench_c1.json 2>$O/bench_c1.err; python -c "
import json; d=json.loads(open('$O/bench_c1.json').read().strip().splitlines()[-1]); print('headline', d['value'], d['sustained']['images_per_s_mean'])"
timeout 900 python -m pytest tests -m gpu -x -q -k "rn or resnet or config5 or ens or RN" 2>&1 | tail -3
