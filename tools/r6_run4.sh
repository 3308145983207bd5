#!/bin/bash
# round 6, GPU run 4 (one lease): the driver's bench command line; the f16 stream reports; the in-flight tests repeated; the attention A/B
# with the ablation library; the GPU suite on the NULL stream
set -u
O=gpurun_out/r6; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmdline.json 2> $O/bench_driver_cmdline.err; echo "bench rc=$?"; tail -3 $O/bench_driver_cmdline.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r6/bench_driver_cmdline.json").read().strip().splitlines()[-1])
    keys = ["value", "value_sustained", "ms_per_step", "timed_regions_seconds", "roofline_frac", "attention_fwd_frac_of_hbm_roofline", "f16_images_per_s", "f16_in_proj_frac", "f16_c_fc_frac",
            "f16_c_proj_frac", "f16_out_proj_frac", "f16_attention_frac", "f16_attention_frac_of_hbm_roofline", "f16_top1_agreement", "f16_lnfold_images_per_s", "f16_lnfold_in_proj_frac",
            "f16_lnfold_c_fc_frac", "grid_weights_images_per_s", "harness_one_image_per_call", "harness_one_image_per_call_views_in_loop", "harness_three_in_flight",
            "harness_three_in_flight_views_in_loop", "harness_three_in_flight_legs_spread", "view_generation_ms_per_image", "cpu_baseline_images_per_s"]
    for k in keys: print(f"  {k}: {d.get(k)}")
    h = d.get("harness", {})
    for k, v in h.items():
        if k.endswith("_legs") or "prefetched" in k: print("  harness", k, v)
except Exception as e:
    print("bench line unreadable:", e)
PY
timeout 900 python -m pytest tests/test_gpu_round2.py -q -s -k "f16_single_pass_mode_b16_stream" 2>&1 | grep -E "^\[|sample [0-9]+:|passed|failed|Error|assert" > $O/f16_stream_reports.txt; cat $O/f16_stream_reports.txt
for i in 1 2 3 4 5 6 7 8; do RLCF_TEST_STREAM=nonblocking timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py -q -k "in_flight or lanes" 2>&1 | tail -1; done > $O/in_flight_8x.txt; cat $O/in_flight_8x.txt
RLCF_LIB_PATH=$PWD/tools/ab/librlcf_hip_abl.so timeout 600 python tools/r6_attn.py 7 > $O/attn_ab.txt 2>&1; cat $O/attn_ab.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/suite_null_stream.txt 2>&1; tail -5 $O/suite_null_stream.txt
