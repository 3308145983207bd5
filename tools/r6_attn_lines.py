"""Round 6: attention forward of the image towers at the benchmark's layer shape (1 280 sequences x 12 heads x 197 tokens), both operand
forms, output rows as 16-byte pieces per lane (RLCF_ATTN_LINEST=0) against whole 128-byte lines through the wave's LDS slab (=1, default).
ONE process, arms interleaved (the switch is read per launch), median of `rounds` x 5 launches per arm and position."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 9
n_seq, tok, W = 1280, 197, 768
T = n_seq * tok
qkv = torch.randn(T, 3 * W, device=dev)
seqs = torch.tensor([[i * tok, tok, 0, 0] for i in range(n_seq)], dtype=torch.int32, device=dev)
for mode, prec, bytes_per in (("pair (parity mode)", L.PREC_F16X3, 16.0), ("single-pass f16", L.PREC_F16, 8.0)):
    pairs = torch.empty(T, 3 * W, device=dev)
    L.check(lib.rlcf_split_pairs(qkv.data_ptr(), pairs.data_ptr(), T * 3 * W, prec, st()))
    op = torch.empty(T, W, device=dev)

    def run():
        L.check(lib.rlcf_attention_fwd_pairs(pairs.data_ptr(), seqs.data_ptr(), n_seq, tok, W, None, op.data_ptr(), None, prec, st()))
    arms = [("pieces", "0"), ("lines", "1"), ("pieces (2nd)", "0"), ("lines (2nd)", "1")]
    times = {a[0]: [] for a in arms}
    outs = {}
    for r in range(rounds + 1):
        for name, v in arms:
            os.environ["RLCF_ATTN_LINEST"] = v
            run()
            torch.cuda.synchronize()
            if r == 0:
                outs[v] = op.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[name].append(e0.elapsed_time(e1) / 5 * 1e3)
    same = torch.equal(outs["0"].view(torch.int32), outs["1"].view(torch.int32))
    roof = T * W * bytes_per / 8e12 * 1e6
    print(f"{mode}: outputs bit-identical: {same}; HBM roofline {roof:.0f} us")
    for name, _ in arms:
        m = statistics.median(times[name])
        print(f"   {name:14s} {m:7.1f} us   {roof / m:.3f} of roofline")
os.environ.pop("RLCF_ATTN_LINEST", None)
