#!/bin/bash
# SQ counter evidence of the round-5 build (run on the GPU box from the repo root): MFMA-busy / CU-busy / MFMA instruction counts
# (one pass) and the effective shader clock (GRBM_GUI_ACTIVE, second pass) of
#   (a) the parity-mode dominant GEMM gemm_nt_f16x3_v3i_kernel on the four products of a ViT-B/16 layer at 20 images per pass (tools/gemm_epi_bench.py)
#   (b) attention_fwd_pair_kernel, split-f16 and single-pass f16, 1 280 ViT-B/16 sequences (tools/attn_bench.py)
#   (c) the single-pass f16 GEMM kernels (gemm_nt_f16_pp_kernel: in_proj / out_proj / c_fc / c_proj -> f16; tools/gemm_f16_bench.py)
# -> gpurun_out/<tag>/sq_*.txt (tools/pmc_summary.py tables); tools/sq_counters_report.py turns them into profiles/r5_sq_counters.{txt,json}
TAG=${1:-r5sq}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() { # name, kernel substring, counters, command...
  name=$1; sub=$2; ctr=$3; shift 3
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/sq_$name -- "$@" > $O/sq_$name.log 2>&1
  echo "$name rc=$?"
  db=$(find /tmp/sq_$name -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $db $sub > $O/sq_$name.txt 2>&1
  rm -rf /tmp/sq_$name; tail -c 600 $O/sq_$name.log > $O/sq_$name.tail; rm -f $O/sq_$name.log
}
C1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES"
C2="GRBM_GUI_ACTIVE"
pass x3_busy gemm_nt_f16x3_v3i "$C1" python $R/tools/gemm_epi_bench.py 252160
pass x3_clk gemm_nt_f16x3_v3i "$C2" python $R/tools/gemm_epi_bench.py 252160
ATTN_BENCH_ONLY=1 ATTN_BENCH_KERNELS=p2,p1 pass attn_busy attention_fwd_pair "$C1" python $R/tools/attn_bench.py
ATTN_BENCH_ONLY=1 ATTN_BENCH_KERNELS=p2,p1 pass attn_clk attention_fwd_pair "$C2" python $R/tools/attn_bench.py
for shp in in_proj out_proj-\> c_fc c_proj-\>; do
  n=$(echo $shp | tr -d '>-')
  BENCH_ONLY=$shp pass f16_${n}_busy gemm_nt_f16_pp "$C1" python $R/tools/gemm_f16_bench.py
  BENCH_ONLY=$shp pass f16_${n}_clk gemm_nt_f16_pp "$C2" python $R/tools/gemm_f16_bench.py
done
grep -h "TF\|us" $O/*.tail | head -40
