"""The 256x256 split-f16 GEMM with the epilogues of a ViT layer (interleaved operands, what the engine runs): in_proj (bias, f32 out),
out_proj / c_proj (bias + residual in place), c_fc (bias + QuickGELU -> operand pair).  args: [M] ; RLCF_X3_NOFASTEPI=1 selects the
generic per-row epilogue for an A/B on the same box.  Checks 64 sampled rows against f64."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0"); st = lambda: torch.cuda.current_stream().cuda_stream
M = int(sys.argv[1]) if len(sys.argv) > 1 else 252160
W = 768
def il(hi, lo):
    R, K = hi.shape
    return torch.stack([hi.view(R, K // 32, 32), lo.view(R, K // 32, 32)], dim=2).reshape(R, 2 * K).contiguous()
def split(x):
    R, K = x.shape
    h = torch.empty(R, K, dtype=torch.float16, device=dev); l = torch.empty_like(h)
    L.check(lib.rlcf_split_f16x2(x.data_ptr(), h.data_ptr(), l.data_ptr(), R * K, st()))
    return il(h, l)
rows = torch.randint(0, M, (64,), device=dev)
tot_ms = 0.0
for name, N, K, epi, res, pair in [("in_proj", 3 * W, W, 0, False, False), ("out_proj+res", W, W, 0, True, False),
                                   ("c_fc+gelu->pair", 4 * W, W, 1, False, True), ("c_proj+res", W, 4 * W, 0, True, False)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev) * 0.1
    a2, w2 = split(a), split(w)
    x = torch.randn(M, N, device=dev) if res else None
    c = x.clone() if res else (None if pair else torch.empty(M, N, device=dev))
    ch = torch.empty(M, N, dtype=torch.float16, device=dev) if pair else None
    cl = torch.empty_like(ch) if pair else None
    ref = a[rows].double() @ w.double().t() + b.double()
    if epi == 1: ref = ref * torch.sigmoid(1.702 * ref)
    if res: ref = ref + x[rows].double()
    def run(resid):
        L.check(lib.rlcf_gemm_f16x3(a2.data_ptr(), a2.data_ptr() + 64, 2 * K, w2.data_ptr(), w2.data_ptr() + 64, 2 * K, b.data_ptr(),
                                    resid.data_ptr() if res else None, N, None, 0, c.data_ptr() if c is not None else None, N,
                                    ch.data_ptr() if pair else None, cl.data_ptr() if pair else None, N, M, N, K, 1.0, epi, st()))
    run(x)                                     # correctness: residual read from x, result in c
    torch.cuda.synchronize()
    got = (ch[rows].double() + cl[rows].double()) if pair else c[rows].double()
    err = (got - ref).abs().max().item()
    for _ in range(2): run(c if res else None)  # timing: in place like the engine (values drift, irrelevant)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps): run(c if res else None)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tot_ms += ms
    print(f"M={M} {name:16s} N={N:5d} K={K:5d}: {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:7.1f} TF  maxerr={err:.2e}", flush=True)
print(f"layer GEMMs total {tot_ms:.3f} ms  ({'generic' if os.environ.get('RLCF_X3_NOFASTEPI') == '1' else 'fast'} epilogue)")
