#!/bin/bash
# f16 persistent GEMM: op-level table, s_memtime tile trace (K loop / line-up / epilogue ticks), start-time cohorts, round-5 tests
set -u
O=gpurun_out/r5/f16gemm; mkdir -p $O
timeout 300 python tools/gemm_f16_bench.py > $O/table.txt 2>&1; grep "\[pp\]" $O/table.txt
for shp in in_proj c_fc; do
  BENCH_ONLY=$shp RLCF_F16_PP_TRACE=1 timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep "pp trace" | tail -6 > $O/trace_$shp.txt; cat $O/trace_$shp.txt
done
for d in 2 4 -2; do echo "DESYNC=$d"; RLCF_F16_PP_DESYNC=$d BENCH_ONLY="->f16" timeout 300 python tools/gemm_f16_bench.py 2>&1 | grep "\[pp\]" | tee $O/desync_$d.txt; done
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --precision f16 --steps 20 --warmup 5 --no-cpu-baseline --no-f16-line --no-harness-leg > $O/bench_f16.json 2>$O/bench_f16.err; python -c "
import json; d=json.loads(open('$O/bench_f16.json').read().strip().splitlines()[-1]); print('f16 images/s', d['value']); r=d['roofline']; print([(e['kernel'],round(e.get('tflops',0))) for e in r['per_kernel']])"
# ResNet tower after the max|.| arena + in-launch bound scale: tests, then configs[4]
timeout 1200 python -m pytest tests -m gpu -x -q -k "rn or resnet or config5 or ens or RN" 2>&1 | tail -3
for bk in 0 1; do RLCF_CONV_BOUND_KERNEL=$bk timeout 600 python bench.py --config 4 --no-cpu-baseline --sustain-seconds 0 --no-roofline > $O/bench_c4_bk$bk.json 2>$O/bench_c4_bk$bk.err; python -c "
import json; d=json.loads(open('$O/bench_c4_bk$bk.json').read().strip().splitlines()[-1]); print('config4 bound_kernel=$bk', d['value'], d['ms_per_step'])"; done
