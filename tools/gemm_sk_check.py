"""Stream-K tail of the 256x256 split-f16 GEMM (gemm_f16x3.hip): correctness against f64 on EVERY element of a few tiles-of-interest
shapes (grids that do not fill their last round: one image's token matrix, the reward tower of a pass, the 20-image pass), run-to-run
bit-reproducibility, and timing with the tail on / off (RLCF_X3_SK=0 in the environment switches it off: run the script twice).
args: [M ...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0"); st = lambda: torch.cuda.current_stream().cuda_stream
Ms = [int(a) for a in sys.argv[1:]] or [12608, 23640, 252160]
W = 768
def il(hi, lo):
    R, K = hi.shape
    return torch.stack([hi.view(R, K // 32, 32), lo.view(R, K // 32, 32)], dim=2).reshape(R, 2 * K).contiguous()
def split(x):
    R, K = x.shape
    h = torch.empty(R, K, dtype=torch.float16, device=dev); l = torch.empty_like(h)
    L.check(lib.rlcf_split_f16x2(x.data_ptr(), h.data_ptr(), l.data_ptr(), R * K, st()))
    return il(h, l)
sk = os.environ.get("RLCF_X3_SK", "1")
for M in Ms:
    tot = 0.0
    for name, N, K, epi, res, pair in [("in_proj", 3 * W, W, 0, False, False), ("out_proj+res", W, W, 0, True, False),
                                       ("c_fc+gelu->pair", 4 * W, W, 1, False, True), ("c_proj+res", W, 4 * W, 0, True, False)]:
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev) * 0.1
        a2, w2 = split(a), split(w)
        x = torch.randn(M, N, device=dev) if res else None
        c = x.clone() if res else (None if pair else torch.empty(M, N, device=dev))
        ch = torch.empty(M, N, dtype=torch.float16, device=dev) if pair else None
        cl = torch.empty_like(ch) if pair else None
        def run(resid):
            L.check(lib.rlcf_gemm_f16x3(a2.data_ptr(), a2.data_ptr() + 64, 2 * K, w2.data_ptr(), w2.data_ptr() + 64, 2 * K, b.data_ptr(),
                                        resid.data_ptr() if res else None, N, None, 0, c.data_ptr() if c is not None else None, N,
                                        ch.data_ptr() if pair else None, cl.data_ptr() if pair else None, N, M, N, K, 1.0, epi, st()))
        run(x); torch.cuda.synchronize()
        got = (ch.float() + cl.float()) if pair else c.clone()
        # full check in f32 chunks against an f64 reference on a strided row sample + EVERY row of the last 3 M tiles (the stream-K tiles)
        rows = torch.cat([torch.arange(0, M, 97, device=dev), torch.arange(max(0, M - 3 * 256), M, device=dev)]).unique()
        ref = a[rows].double() @ w.double().t() + b.double()
        if epi == 1: ref = ref * torch.sigmoid(1.702 * ref)
        if res: ref = ref + x[rows].double()
        err = (got[rows].double() - ref).abs().max().item()
        if res: c.copy_(x)
        run(x); torch.cuda.synchronize()
        got2 = (ch.float() + cl.float()) if pair else c
        same = bool(torch.equal(got, got2))
        for _ in range(2): run(c if res else None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); reps = 10
        for _ in range(reps): run(c if res else None)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps; tot += ms
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        print(f"SK={sk} M={M:6d} {name:16s} N={N:5d} K={K:5d} tiles={tiles:5d} ({tiles / 256:6.2f} rounds): {ms * 1e3:8.1f} us {2 * M * N * K / ms / 1e9:7.1f} TF  "
              f"maxerr={err:.2e} bit-reproducible={same}", flush=True)
        assert err < 2e-4 and same
    print(f"SK={sk} M={M}: layer GEMMs total {tot:.3f} ms", flush=True)
