"""Micro-benchmark of the attention forward kernels through the C ABI (GPU only): ViT-B/16 (197 tokens, 12 heads) and
ViT-L/14 (257 tokens, 16 heads) image-tower shapes, plus packed text sequences.  Kernels: the round-2 forward on f32 qkv
(rlcf_attention_fwd, precision 0 = f32 MFMA, 2 = split-f16, 1 = single-pass f16) and the round-3 forward on producer-emitted operand
pairs (rlcf_attention_fwd_pairs: LDS-DMA staging + ds_read_b64_tr_b16), 'p2' = split-f16, 'p1' = single-pass f16."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
big = int(os.environ.get("ATTN_BENCH_SEQS", "1280"))
cases = [("vit-b16 512 views", 512, 197, 768, 0), (f"vit-b16 {big} views", big, 197, 768, 0), ("vit-l14 64 views", 64, 257, 1024, 0),
         ("vit-l14 384 views", 384, 257, 1024, 0), ("text 4095 rows", 1000, 9, 512, 1)]
if os.environ.get("ATTN_BENCH_ONLY"): cases = cases[1:2]      # counter passes: the benchmark-shaped ViT-B/16 case only
for name, n_seq, tok, W, causal in cases:
    T = n_seq * tok
    qkv = torch.randn(T, 3 * W, device=dev)
    seqs = torch.tensor([[i * tok, tok, 0, 0] for i in range(n_seq)], dtype=torch.int32, device=dev)
    outs = {}
    variants = [0, 2, 1] + ([] if causal else ["p2", "p1"])
    if os.environ.get("ATTN_BENCH_KERNELS"): variants = [v for v in variants if str(v) in os.environ["ATTN_BENCH_KERNELS"].split(",")]
    for prec in variants:
        out = torch.empty(T, W, device=dev)
        if isinstance(prec, str):
            p = L.PREC_F16X3 if prec == "p2" else L.PREC_F16
            pairs = torch.empty(T, 3 * W, device=dev)
            L.check(lib.rlcf_split_pairs(qkv.data_ptr(), pairs.data_ptr(), T * 3 * W, p, st()))
            op = torch.empty(T, W, device=dev)
            # the engine's call: operand pairs in, operand pairs out (no f32 output)
            run = lambda: L.check(lib.rlcf_attention_fwd_pairs(pairs.data_ptr(), seqs.data_ptr(), n_seq, tok, W, None, op.data_ptr(), None, p, st()))
            fin = lambda: L.check(lib.rlcf_attention_fwd_pairs(pairs.data_ptr(), seqs.data_ptr(), n_seq, tok, W, out.data_ptr(), None, None, p, st()))
        else:
            run = lambda: L.check(lib.rlcf_attention_fwd(qkv.data_ptr(), seqs.data_ptr(), n_seq, tok, W, causal, out.data_ptr(), None, prec, st()))
            fin = run
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        flops = 4.0 * tok * tok * 64 * (W // 64) * n_seq * (0.5 if causal else 1.0)
        fin(); torch.cuda.synchronize()
        outs[prec] = out.clone()
        print(f"{name}: kernel={prec} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF", flush=True)
    # reference on a few sequences
    q, k, v = qkv[: 2 * tok].double().split(W, dim=1)
    H = W // 64
    ref = torch.empty(2 * tok, W, dtype=torch.float64, device=dev)
    for s in range(2):
        for hd in range(H):
            sl = slice(s * tok, (s + 1) * tok); cs = slice(hd * 64, hd * 64 + 64)
            sc = q[sl, cs] @ k[sl, cs].t() / 8
            if causal: sc = sc.masked_fill(torch.ones(tok, tok, device=dev).triu(1).bool(), float("-inf"))
            ref[sl, cs] = torch.softmax(sc, -1) @ v[sl, cs]
    for prec in variants:
        print(f"   maxerr kernel={prec}: {(outs[prec][:2 * tok].double() - ref).abs().max().item():.2e}")
