#!/bin/bash
# the f16 mode's kernel table on the final tree (non-temporal whole-line stores on): rocprofv3 --kernel-trace --stats over the driver's command line
O=gpurun_out/r6/final6; mkdir -p $O; export TMPDIR=/tmp
C="--no-cpu-baseline --sustain-seconds 0 --no-f16-line --no-harness-leg"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_f16 -- python bench.py --precision f16 --steps 20 --warmup 5 $C > $O/prof_f16.log 2>&1
db=$(find /tmp/prof_f16 -name "*_results.db" | head -1)
python tools/prof_summary.py "$db" "round 6 final tree: rocprofv3 --kernel-trace --stats -- python bench.py --precision f16 --steps 20 --warmup 5 $C (RLCF_PREC_F16 default form, NOT parity-grade; 85 images)" 85 > $O/kernel_stats_f16.txt
head -12 $O/kernel_stats_f16.txt | cut -c1-150
