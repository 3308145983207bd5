#!/bin/bash
# end-of-round captures (one MI355X): bench lines of every single-GPU BASELINE config + rocprofv3 kernel tables; outputs under gpurun_out/r3/final
set -u
O=gpurun_out/r3/final; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(tail -c 200 $O/$name.json | head -c 0)"; }
run bench_driver_cmdline --steps 20 --warmup 5
run bench_default --no-cpu-baseline
run bench_c0 --config 0
run bench_c2 --config 2
run bench_c4 --config 4
run bench_batch1 --batch 1 --no-cpu-baseline
export TMPDIR=/tmp
prof() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$name -- python bench.py "$@" > $O/prof_$name.log 2>&1; db=$(find $O/prof_$name -name "*_results.db" | head -1); echo "prof $name db=$db"; }
prof c1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --no-f16-line
prof c2 --config 2 --no-cpu-baseline --sustain-seconds 0
prof c4 --config 4 --no-cpu-baseline --sustain-seconds 0
ls -la $O | head -40
