#!/bin/bash
# one-image mode under rocprofv3: kernel table + dependent-launch gaps
export TMPDIR=/tmp
mkdir -p gpurun_out/b1prof
rm -rf /tmp/prof_b1
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_b1 -- python bench.py --batch 1 --steps 20 --warmup 5 --no-harness-leg --no-cpu-baseline --no-f16-line --sustain-seconds 0 --no-roofline > gpurun_out/b1prof/log.txt 2>&1
db=$(find /tmp/prof_b1 -name "*_results.db" | head -1)
python tools/prof_summary.py "$db" "bench.py --batch 1 --steps 20 --warmup 5 --no-roofline (one image per pass: set-up + 5 warm-up + 20 timed images), rocprofv3 --kernel-trace --stats" 26 > gpurun_out/b1prof/kernel_stats_b1.txt
head -48 gpurun_out/b1prof/kernel_stats_b1.txt | cut -c1-150
python tools/gap_analysis.py "$db" 22 --ours 2>&1 | cut -c1-150 | head -60
