#!/bin/bash
# end-of-round captures (one MI355X): bench lines of every single-GPU BASELINE config + rocprofv3 kernel tables; outputs under gpurun_out/r4/final
set -u
O=gpurun_out/r4/final; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(tail -c 200 $O/$name.json | head -c 0)"; }
run bench_driver_cmdline --steps 20 --warmup 5
run bench_default --no-cpu-baseline
run bench_c0 --config 0
run bench_c2 --config 2
run bench_c4 --config 4
run bench_batch1 --batch 1 --no-cpu-baseline
export TMPDIR=/tmp
# (the rocpd databases are tens of MB each and gpurun_out/ travels back only below 64 MiB: summarise on the box, drop the database)
prof() { name=$1; nimg=$2; title=$3; shift 3; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python bench.py "$@" > $O/prof_$name.log 2>&1
         db=$(find /tmp/prof_$name -name "*_results.db" | head -1)
         if [ -n "$db" ]; then python tools/prof_summary.py "$db" "$title" $nimg > $O/kernel_stats_$name.txt; else echo "prof $name: no database"; fi
         tail -c 1500 $O/prof_$name.log > $O/prof_$name.tail; rm -f $O/prof_$name.log; rm -rf /tmp/prof_$name; }
prof c1 65 "round 4 final build: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --no-f16-line --no-harness-leg (BASELINE configs[1]; set-up pass of 20 + 5 warm-up + 20 timed + 20 profiled images = 65)" --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --no-f16-line --no-harness-leg
prof c2 120 "round 4 final build: rocprofv3 --kernel-trace --stats -- python bench.py --config 2 --no-cpu-baseline --sustain-seconds 0 (BASELINE configs[2]: ViT-L/14 + ViT-L/14, LayerNorm tuning, N = 64, 20 images per pass; set-up + warm-up + timed + profiled passes)" --config 2 --no-cpu-baseline --sustain-seconds 0
prof c4 56 "round 4 final build: rocprofv3 --kernel-trace --stats -- python bench.py --config 4 --no-cpu-baseline --sustain-seconds 0 (BASELINE configs[4]: RN50x64 @448 student + ViT-L/14 reward, N = 32, 8 images per pass; set-up + warm-up + timed + profiled passes)" --config 4 --no-cpu-baseline --sustain-seconds 0
prof b1 26 "round 4 final build: rocprofv3 --kernel-trace --stats -- python bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --no-f16-line --no-harness-leg --no-roofline (one image per pass: set-up + 5 warm-up + 20 timed images)" --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --no-f16-line --no-harness-leg --no-roofline
ls -la $O | head -40
# row a-R: RN50 student, BatchNorm tuning
REWARD_ARCH=ViT-B/16 timeout 300 python tools/time_ln_path.py RN50 1000 1 1 > $O/bn_rn50.txt 2>&1
REWARD_ARCH=ViT-B/16 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bn -- python tools/time_ln_path.py RN50 1000 1 1 > $O/prof_bn.log 2>&1
db=$(find /tmp/prof_bn -name "*_results.db" | head -1)
[ -n "$db" ] && python tools/prof_summary.py "$db" "row a-R, final build: RN50 student, BatchNorm tuning (rlcf_tta_sample_ln), N=64 views, C=1000, 1 step, ViT-B/16 reward; REWARD_ARCH=ViT-B/16 rocprofv3 --kernel-trace --stats -- python tools/time_ln_path.py RN50 1000 1 1 (6 test images incl. 2 warm-up)" 6 > $O/kernel_stats_bn.txt
rm -rf /tmp/prof_bn $O/prof_bn.log
tail -2 $O/bn_rn50.txt

# every-parameter tuning of a ResNet student (round 4)
REWARD_ARCH=ViT-B/16 timeout 600 python tools/time_ln_path.py RN50 1000 1 1 full > $O/rnfull_rn50.txt 2>&1; tail -2 $O/rnfull_rn50.txt
# ViT-B/16 every-parameter tuning, 3 steps (rlcf-tune.sh), and LayerNorm tuning one image per pass
timeout 600 python tools/time_ln_path.py ViT-B/16 1000 1 3 full > $O/full_b16.txt 2>&1; tail -2 $O/full_b16.txt
