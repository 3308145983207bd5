"""192 x 256 tile form of the split-f16 GEMM (gemm_f16x3.hip, MT = 3): the same products in the same K order and the same epilogue
arithmetic as the 256 x 256 / 256 x 128 kernels, so its output must be BIT-IDENTICAL to theirs.  Runs every epilogue kind on shapes the
launcher routes to it (one image's token matrix; ragged M; N not a multiple of 256), prints a checksum of the output bits, the f64
error on a row sample, and the time.  Run twice — RLCF_X3_MT3=0 and =1 — and diff the checksum columns (tools/r4_mt3.sh does).
args: [M ...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0"); st = lambda: torch.cuda.current_stream().cuda_stream
Ms = [int(a) for a in sys.argv[1:]] or [12608, 12500, 6304, 23640]
W = 768
def il(hi, lo):
    R, K = hi.shape
    return torch.stack([hi.view(R, K // 32, 32), lo.view(R, K // 32, 32)], dim=2).reshape(R, 2 * K).contiguous()
def split(x):
    R, K = x.shape
    h = torch.empty(R, K, dtype=torch.float16, device=dev); l = torch.empty_like(h)
    L.check(lib.rlcf_split_f16x2(x.data_ptr(), h.data_ptr(), l.data_ptr(), R * K, st()))
    return il(h, l)
def bits(t):
    return int(t.contiguous().view(torch.int16 if t.dtype == torch.float16 else torch.int32).to(torch.int64).sum().item())
mode = os.environ.get("RLCF_X3_MT3", "1")
torch.manual_seed(0)
for M in Ms:
    for name, N, K, epi, res, f32o, pair, aux in [("out_proj+res", W, W, 0, True, True, False, False), ("c_proj+res", W, 4 * W, 0, True, True, False, False),
                                                  ("f32", W, W, 0, False, True, False, False), ("gelu->pair", W, W, 1, False, False, True, False),
                                                  ("pair", W, W, 0, False, False, True, False), ("f32+pair+res", W, W, 0, True, True, True, False),
                                                  ("gelu_bwd(aux)", W, 4 * W, 2, False, True, False, True), ("relu+res", W, W, 3, True, True, False, False), ("N=640 f32", 640, W, 0, False, True, False, False),
                                                  ("N=1024 res", 1024, 1024, 0, True, True, False, False)]:
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev) * 0.1
        a2, w2 = split(a), split(w)
        x = torch.randn(M, N, device=dev) if res else None
        ax = torch.randn(M, N, device=dev) if aux else None
        c = torch.empty(M, N, device=dev) if f32o else None
        ch = torch.empty(M, N, dtype=torch.float16, device=dev) if pair else None
        cl = torch.empty_like(ch) if pair else None
        def run():
            L.check(lib.rlcf_gemm_f16x3(a2.data_ptr(), a2.data_ptr() + 64, 2 * K, w2.data_ptr(), w2.data_ptr() + 64, 2 * K, b.data_ptr(),
                                        x.data_ptr() if res else None, N, ax.data_ptr() if aux else None, N if aux else 0,
                                        c.data_ptr() if f32o else None, N, ch.data_ptr() if pair else None, cl.data_ptr() if pair else None, N,
                                        M, N, K, 1.0, epi, st()))
        run(); torch.cuda.synchronize()
        var = lib.rlcf_last_gemm_variant() if hasattr(lib, "rlcf_last_gemm_variant") else -1
        got = c if f32o else (ch.float() + cl.float())
        rows = torch.cat([torch.arange(0, M, 61, device=dev), torch.arange(max(0, M - 200), M, device=dev)]).unique()
        ref = a[rows].double() @ w.double().t() + b.double()
        if epi == 1: ref = ref * torch.sigmoid(1.702 * ref)
        if epi == 2:
            s = torch.sigmoid(1.702 * ax[rows].double()); ref = ref * (s * (1 + 1.702 * ax[rows].double() * (1 - s)))
        if res: ref = ref + x[rows].double()
        if epi == 3: ref = ref.clamp_min(0)
        err = (got[rows].double() - ref).abs().max().item()
        sig = (bits(c) if f32o else 0) ^ ((bits(ch) * 31 + bits(cl)) if pair else 0)
        for _ in range(2): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); reps = 10
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"SIG M={M} {name} N={N} K={K} {sig:x}", flush=True)
        print(f"MT3={mode} M={M:6d} {name:14s} N={N:5d} K={K:5d} variant={var}: {ms * 1e3:8.1f} us {2 * M * N * K / ms / 1e9:7.1f} TF  maxerr={err:.2e}", flush=True)
        assert err < 3e-4, err
