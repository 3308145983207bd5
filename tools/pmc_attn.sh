# SQ counters of the attention forward kernels (tools/attn_bench.py): run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  ATTN_BENCH_ONLY=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/r3/pmc_at$i -- python $R/tools/attn_bench.py > $R/gpurun_out/r3/pmc_at$i.log 2>&1
  echo "pass $i rc=$?"
  db=$(find $R/gpurun_out/r3/pmc_at$i -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $db attention > $R/gpurun_out/r3/pmc_at$i.txt 2>&1
  rm -rf $R/gpurun_out/r3/pmc_at$i
done
