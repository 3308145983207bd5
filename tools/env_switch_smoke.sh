for kv in RLCF_ATTN_OLD=1 RLCF_NO_OVERLAP=1 RLCF_BATCH_PARTS=2 RLCF_X3_NOFASTEPI=1 RLCF_X3_KERNEL=2 RLCF_X3_KERNEL=3 RLCF_X3_NOV2S=1 RLCF_X3_NOSPLITK=1 RLCF_TEXT_NOPACK=1 RLCF_X3_GROUP=4 RLCF_X3_V4=1 RLCF_ATTN_VAR=0 RLCF_X3_SK=1 RLCF_X3_MT3=0 RLCF_X3_MT3=2 RLCF_X3_SPLIT24=0 RLCF_X3_V2MIN=256 RLCF_SAVE_NOPAIRS=1 RLCF_V2S8=0 RLCF_SKINNY=0 RLCF_F32_SMALL=1 RLCF_X3_NT=0 RLCF_X3_STAGGER=4; do
  out=$(env $kv timeout 300 python bench.py --steps 8 --warmup 8 --batch 8 --no-cpu-baseline --no-f16-line --no-harness-leg --sustain-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['top1_first'])" 2>&1 | tail -1)
  echo "$kv -> $out"
done
for kv in RLCF_ATTN_BWD_F32=1 RLCF_ATTN_BWD_ATOMIC=1 RLCF_ATTN_BWD_OLD=1; do
  out=$(env $kv timeout 300 python bench.py --config 2 --steps 2 --warmup 2 --classes 100 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['top1_first'])" 2>&1 | tail -1)
  echo "$kv (config 2) -> $out"
done
for kv in RLCF_CONV_IM2COL=1 RLCF_RN_NOFUSE=1 RLCF_CONV_BOUND_KERNEL=0; do
  out=$(env $kv timeout 300 python bench.py --config 4 --steps 2 --warmup 1 --classes 100 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['top1_first'])" 2>&1 | tail -1)
  echo "$kv (config 4) -> $out"
done
# round 5: switches of the single-pass f16 mode (same top-1 expected from every variant but the ablations)
for kv in RLCF_F16_LNFOLD=0 RLCF_F16_PP=0 RLCF_F16_P8=0 RLCF_F16_PP_NT=1 RLCF_F16_PP_DESYNC=2 RLCF_F16_RESADD=0; do
  out=$(env $kv timeout 300 python bench.py --precision f16 --steps 8 --warmup 8 --batch 8 --no-cpu-baseline --no-f16-line --no-harness-leg --sustain-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['top1_first'])" 2>&1 | tail -1)
  echo "$kv (f16 mode) -> $out"
done
