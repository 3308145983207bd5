#!/bin/bash
# rocprofv3 kernel tables of the LAST build: the driver's command line (float32 weights), configs[2] and one image per pass on checkpoint-grid weights
set -u
O=gpurun_out/r5/proflast; mkdir -p $O
export TMPDIR=/tmp
prof() { name=$1; nimg=$2; title=$3; shift 3; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python bench.py "$@" > $O/prof_$name.log 2>&1
         db=$(find /tmp/prof_$name -name "*_results.db" | head -1)
         if [ -n "$db" ]; then python tools/prof_summary.py "$db" "$title" $nimg > $O/kernel_stats_$name.txt; else echo "prof $name: no database"; fi
         rm -f $O/prof_$name.log; rm -rf /tmp/prof_$name; head -8 $O/kernel_stats_$name.txt | cut -c1-150; }
C="--no-cpu-baseline --sustain-seconds 0 --no-f16-line --no-harness-leg"
prof c1 65 "round 5 last build: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 $C (BASELINE configs[1], float32 weights; 65 images)" --steps 20 --warmup 5 $C
prof c2grid 120 "round 5 last build: rocprofv3 --kernel-trace --stats -- python bench.py --config 2 --weights fp16grid --no-cpu-baseline --sustain-seconds 0 (BASELINE configs[2] on checkpoint-grid weights: two MFMA passes; 120 images)" --config 2 --weights fp16grid --no-cpu-baseline --sustain-seconds 0
prof b1grid 26 "round 5 last build: rocprofv3 --kernel-trace --stats -- python bench.py --batch 1 --weights fp16grid --steps 20 --warmup 5 $C --no-roofline (one image per pass on checkpoint-grid weights; 26 images)" --batch 1 --weights fp16grid --steps 20 --warmup 5 $C --no-roofline
