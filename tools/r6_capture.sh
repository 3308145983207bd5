#!/bin/bash
# end-of-round captures (one MI355X lease): bench lines of every single-GPU config, rocprofv3 kernel tables of the driver's command line
# (parity mode and the f16 mode), fabric traffic + SQ counters of the dominant GEMM, the GPU suite in BOTH stream modes; -> gpurun_out/r6/final
set -u
O=gpurun_out/r6/final; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run bench_driver_cmdline --steps 20 --warmup 5
run bench_default --no-cpu-baseline
run bench_c0 --config 0 --no-cpu-baseline
run bench_c2 --config 2 --no-cpu-baseline
run bench_c4 --config 4 --no-cpu-baseline
run bench_c5 --config 5 --no-cpu-baseline
run bench_batch1 --batch 1 --no-cpu-baseline
export TMPDIR=/tmp
prof() { name=$1; nimg=$2; title=$3; shift 3; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python bench.py "$@" > $O/prof_$name.log 2>&1
         db=$(find /tmp/prof_$name -name "*_results.db" | head -1)
         if [ -n "$db" ]; then python tools/prof_summary.py "$db" "$title" $nimg > $O/kernel_stats_$name.txt; else echo "prof $name: no database"; fi
         tail -c 1500 $O/prof_$name.log > $O/prof_$name.tail; rm -f $O/prof_$name.log; rm -rf /tmp/prof_$name; }
C="--no-cpu-baseline --sustain-seconds 0 --no-f16-line --no-harness-leg"
prof c1 85 "round 6 final build: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 $C (BASELINE configs[1]; set-up pass of 20 + 5 warm-up + 2 x 20 timed + 20 profiled images = 85)" --steps 20 --warmup 5 $C
prof f16 85 "round 6 final build: rocprofv3 --kernel-trace --stats -- python bench.py --precision f16 --steps 20 --warmup 5 $C (RLCF_PREC_F16 default form, NOT parity-grade; 85 images)" --precision f16 --steps 20 --warmup 5 $C
# fabric traffic of the dominant GEMM (separate --pmc passes), M = 252 160 (20 images per pass) and 403 456 (32)
R=$PWD
for M in 252160 403456; do
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f_$M -- python $R/tools/gemm_epi_bench.py $M > /dev/null 2>&1)
  db=$(find /tmp/pmc_f_$M -name "*.db" | head -1); python tools/pmc_summary.py $db gemm_nt_f16x3_v3i > $O/traffic_fetch_$M.txt 2>&1; rm -rf /tmp/pmc_f_$M
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w_$M -- python $R/tools/gemm_epi_bench.py $M > /dev/null 2>&1)
  db=$(find /tmp/pmc_w_$M -name "*.db" | head -1); python tools/pmc_summary.py $db gemm_nt_f16x3_v3i > $O/traffic_write_$M.txt 2>&1; rm -rf /tmp/pmc_w_$M
  timeout 300 python tools/gemm_epi_bench.py $M > $O/traffic_time_$M.txt 2>&1
done
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES -d /tmp/sq_x3 -- python $R/tools/gemm_epi_bench.py 252160 > /dev/null 2>&1)
db=$(find /tmp/sq_x3 -name "*.db" | head -1); python tools/pmc_summary.py $db gemm_nt_f16x3_v3i > $O/sq_x3_busy.txt 2>&1; rm -rf /tmp/sq_x3
for shp in in_proj c_fc; do
  (cd /tmp; BENCH_ONLY=$shp timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES -d /tmp/sq_f16_$shp -- python $R/tools/gemm_f16_bench.py > /dev/null 2>&1)
  db=$(find /tmp/sq_f16_$shp -name "*.db" | head -1); python tools/pmc_summary.py $db gemm_nt_f16_pp > $O/sq_f16_${shp}_busy.txt 2>&1; rm -rf /tmp/sq_f16_$shp
done
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round4.py -q -s -k "stream" 2>&1 | grep -E "^\[|sample [0-9]+:|passed|failed" > $O/stream_reports.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/suite_null_stream.txt 2>&1; tail -3 $O/suite_null_stream.txt
RLCF_TEST_STREAM=nonblocking timeout 1500 python -m pytest tests -m gpu -q > $O/suite_nonblocking.txt 2>&1; tail -3 $O/suite_nonblocking.txt
for i in 1 2 3 4 5 6 7 8; do RLCF_TEST_STREAM=nonblocking timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py -q -k "in_flight or lanes" 2>&1 | tail -1; done > $O/in_flight_8x.txt; cat $O/in_flight_8x.txt
ls -la $O | head -60
