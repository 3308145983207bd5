"""Round 6, verdict item 5: the attention forward of the image towers (attention_fwd_pair_kernel) at the benchmark's layer shape —
1 280 sequences x 12 heads x 197 tokens — in both operand forms (split-f16 pairs = parity mode, plain f16 = RLCF_PREC_F16), in ONE process:
  * the shipped kernel against the ablation builds of the SAME kernel (needs the ABLATION library: RLCF_LIB_PATH=tools/ab/librlcf_hip_abl.so;
    8 = no MFMAs, 32 = no K / V DMA, 40 = neither, 64 = no per-block arithmetic: pure streaming) -> what each part costs when it is taken away;
  * single-pass form: 64-key stages (shipped) against 128-key stages (RLCF_ATTN_SK=128: one barrier per 128 keys), interleaved.
Median of `rounds` x 5 launches per build; HBM roofline of the launch printed next to it."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n_seq, tok, W = 1280, 197, 768
T = n_seq * tok
abl = "abl" in os.environ.get("RLCF_LIB_PATH", "")
qkv = torch.randn(T, 3 * W, device=dev)
seqs = torch.tensor([[i * tok, tok, 0, 0] for i in range(n_seq)], dtype=torch.int32, device=dev)
flops = 4.0 * tok * tok * 64 * (W // 64) * n_seq
for mode, prec, bytes_per in (("pair (parity mode)", L.PREC_F16X3, 16.0), ("single-pass f16", L.PREC_F16, 8.0)):
    pairs = torch.empty(T, 3 * W, device=dev)
    L.check(lib.rlcf_split_pairs(qkv.data_ptr(), pairs.data_ptr(), T * 3 * W, prec, st()))
    op = torch.empty(T, W, device=dev)

    def run():
        L.check(lib.rlcf_attention_fwd_pairs(pairs.data_ptr(), seqs.data_ptr(), n_seq, tok, W, None, op.data_ptr(), None, prec, st()))
    builds = [("shipped", 1, None)]
    sk_abl = None
    if abl:
        builds += [("no MFMAs", 8, sk_abl), ("no K/V DMA", 32, sk_abl), ("no DMA, no MFMAs", 40, sk_abl), ("streaming only", 64, sk_abl)]
    if prec == L.PREC_F16:
        # (each stage size twice, interleaved: the build that runs FIRST in a round measured ~4 % slow on two leases whichever it was)
        builds += [("128-key stages", 1, "128"), ("64-key stages (2nd)", 1, None), ("128-key stages (2nd)", 1, "128"), ("4-wave WGs, 128-key", 1, "nw4"),
                   ("4-wave WGs, 64-key", 1, "nw4_64"), ("64-key stages (3rd)", 1, None)]
    times = {b[0]: [] for b in builds}
    for r in range(rounds + 1):
        for name, var, sk in builds:
            lib.rlcf_attention_debug(0, var)
            os.environ.pop("RLCF_ATTN_SK", None)                     # (default: 64-key stages)
            os.environ.pop("RLCF_ATTN_NW", None)
            if sk in ("128", "nw4"):
                os.environ["RLCF_ATTN_SK"] = "128"
            if sk in ("nw4", "nw4_64"):
                os.environ["RLCF_ATTN_NW"] = "4"
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[name].append(e0.elapsed_time(e1) / 5 * 1e3)
    lib.rlcf_attention_debug(0, 1)
    os.environ.pop("RLCF_ATTN_SK", None)
    os.environ.pop("RLCF_ATTN_NW", None)
    # the structure variants (not the ablations) must give the shipped kernel's bits
    ref_out = None
    for name, var, sk in builds:
        if var != 1:
            continue
        lib.rlcf_attention_debug(0, 1)
        os.environ.pop("RLCF_ATTN_SK", None)
        os.environ.pop("RLCF_ATTN_NW", None)
        if sk in ("128", "nw4"):
            os.environ["RLCF_ATTN_SK"] = "128"
        if sk in ("nw4", "nw4_64"):
            os.environ["RLCF_ATTN_NW"] = "4"
        op.fill_(float("nan"))
        run()
        torch.cuda.synchronize()
        if ref_out is None:
            ref_out = op.clone()
        else:
            print(f"   [{name}] == shipped: bit-equal {torch.equal(op.view(torch.int32), ref_out.view(torch.int32))}")
    os.environ.pop("RLCF_ATTN_SK", None)
    os.environ.pop("RLCF_ATTN_NW", None)
    hbm_us = T * W * bytes_per / 8000e9 * 1e6
    print(f"== {mode}: algorithmic bytes {T * W * bytes_per / 1e9:.3f} GB -> HBM roofline {hbm_us:.0f} us at 8 TB/s; {flops / 1e9:.1f} GFLOP")
    for name, _, _ in builds:
        med = statistics.median(times[name])
        print(f"   {name:18s} median {med:8.1f} us  min {min(times[name]):8.1f}  max {max(times[name]):8.1f}   {flops / med / 1e6:6.1f} TF  "
              f"frac of HBM roofline {hbm_us / med:.3f}", flush=True)
    del pairs, op
