#!/bin/bash
# evidence for the two-pass products on checkpoint-grid weights: rocprofv3 kernel table of the driver's command line with --weights fp16grid,
# and the MFMA-busy counters of the WLO0 kernel on the layer's four products (weights on the grid)
set -u
O=gpurun_out/r5/gridprof; mkdir -p $O
export TMPDIR=/tmp
C="--no-cpu-baseline --sustain-seconds 0 --no-f16-line --no-harness-leg"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_grid -- python bench.py --weights fp16grid --steps 20 --warmup 5 $C > $O/prof.log 2>&1
db=$(find /tmp/prof_grid -name "*_results.db" | head -1)
[ -n "$db" ] && python tools/prof_summary.py "$db" "round 5 final build: rocprofv3 --kernel-trace --stats -- python bench.py --weights fp16grid --steps 20 --warmup 5 $C (BASELINE configs[1] with the GEMM weights on the fp16 grid, as a released checkpoint holds them: two MFMA passes per product; 65 images)" 65 > $O/kernel_stats_grid.txt
rm -rf /tmp/prof_grid; tail -c 800 $O/prof.log > $O/prof.tail; rm -f $O/prof.log
head -12 $O/kernel_stats_grid.txt | cut -c1-160
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA -d /tmp/sq_grid -- python $GRAFT_REPO_ROOT/bench.py --weights fp16grid --steps 20 --warmup 5 $C --no-roofline > /dev/null 2>&1
db=$(find /tmp/sq_grid -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/pmc_summary.py $db gemm_nt_f16x3_v3i > $GRAFT_REPO_ROOT/$O/sq_grid.txt 2>&1; rm -rf /tmp/sq_grid
cat $GRAFT_REPO_ROOT/$O/sq_grid.txt | cut -c1-200
