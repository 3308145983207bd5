#!/bin/bash
# end-of-round captures (one MI355X): bench lines of every single-GPU config + rocprofv3 kernel tables; outputs under gpurun_out/r5/final
set -u
O=gpurun_out/r5/final; mkdir -p $O
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
prof c1 65 "round 5 final build: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 $C (BASELINE configs[1]; set-up pass of 20 + 5 warm-up + 20 timed + 20 profiled images = 65)" --steps 20 --warmup 5 $C
prof f16 65 "round 5 final build: rocprofv3 --kernel-trace --stats -- python bench.py --precision f16 --steps 20 --warmup 5 $C (RLCF_PREC_F16, NOT parity-grade; 65 images)" --precision f16 --steps 20 --warmup 5 $C
prof c2 120 "round 5 final build: rocprofv3 --kernel-trace --stats -- python bench.py --config 2 --no-cpu-baseline --sustain-seconds 0 (BASELINE configs[2]: ViT-L/14 + ViT-L/14, LayerNorm tuning, N = 64, 20 images per pass; set-up + warm-up + timed + profiled passes)" --config 2 --no-cpu-baseline --sustain-seconds 0
prof c4 56 "round 5 final build: rocprofv3 --kernel-trace --stats -- python bench.py --config 4 --no-cpu-baseline --sustain-seconds 0 (BASELINE configs[4]: RN50x64 @448 student + ViT-L/14 reward, N = 32, 8 images per pass)" --config 4 --no-cpu-baseline --sustain-seconds 0
prof c5 100 "round 5 final build: rocprofv3 --kernel-trace --stats -- python bench.py --config 5 --no-cpu-baseline --sustain-seconds 0 (rlcf-prompt.sh: ViT-B/16 student + ViT-L/14 reward, 3 steps, N = 64, 20 images per pass; set-up 20 + warm-up 20 + timed 40 + profiled 20 images)" --config 5 --no-cpu-baseline --sustain-seconds 0
prof b1 26 "round 5 final build: rocprofv3 --kernel-trace --stats -- python bench.py --batch 1 --steps 20 --warmup 5 $C --no-roofline (one image per pass: set-up + 5 warm-up + 20 timed images)" --batch 1 --steps 20 --warmup 5 $C --no-roofline
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round4.py -q -s -k "stream" 2>&1 | grep -E "^\[|sample [0-9]+:|passed|failed" > $O/stream_reports.txt; cat $O/stream_reports.txt
ls -la $O | head -40
