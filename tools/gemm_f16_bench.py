"""Single-pass f16 GEMM (rlcf_gemm_f16) on the four products of a ViT-B/16 layer at the token-matrix size of a pass, plus two large
square shapes (the structural ceiling of the kernel without K = 768's prologue / epilogue share).  args: [M]; RLCF_F16_P8=0 selects the
K/2 alias of the pair kernels (A/B in two processes).  Prints TF/s algorithmic and the fraction of the 2.5 PF dense f16 peak."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0"); st = lambda: torch.cuda.current_stream().cuda_stream
M0 = int(sys.argv[1]) if len(sys.argv) > 1 else 252160
W = 768
rows_n = 64
tag = "alias" if os.environ.get("RLCF_F16_P8") == "0" else ("p8" if os.environ.get("RLCF_F16_PP") == "0" else "pp")
tot_ms, tot_fl = 0.0, 0.0
only = os.environ.get('BENCH_ONLY', '')          # substring of the shape names to run (PMC passes: one shape per pass)
for name, M, N, K, epi, res, f16o in [("in_proj->f16", M0, 3 * W, W, 0, False, True), ("out_proj->f16", M0, W, W, 0, False, True),
                                      ("c_fc+gelu->f16", M0, 4 * W, W, 1, False, True), ("c_proj->f16", M0, W, 4 * W, 0, False, True),
                                      ("out_proj+res f32", M0, W, W, 0, True, False), ("c_proj+res f32", M0, W, 4 * W, 0, True, False),
                                      ("square8k f16out", 8192, 8192, 8192, 0, False, True), ("square8k f32out", 8192, 8192, 8192, 0, False, False),
                                      ("M=65536 N=4096 K=4096 f16", 65536, 4096, 4096, 0, False, True)]:
    if only and only not in name: continue
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); b = torch.randn(N, device=dev) * 0.1
    x = torch.randn(M, N, device=dev) if res else None
    c = x.clone() if res else (None if f16o else torch.empty(M, N, device=dev))
    c16 = torch.empty(M, N, dtype=torch.float16, device=dev) if f16o else None
    rows = torch.randint(0, M, (rows_n,), device=dev)
    ref = a[rows].double() @ w.double().t() + b.double()
    if epi == 1: ref = ref * torch.sigmoid(1.702 * ref)
    if res: ref = ref + x[rows].double()
    def run():
        L.check(lib.rlcf_gemm_f16(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), c.data_ptr() if res else None, N,
                                  c.data_ptr() if c is not None else None, N, c16.data_ptr() if f16o else None, N, M, N, K, 1.0, epi, st()))
    run(); torch.cuda.synchronize()
    got = c16[rows].double() if f16o else c[rows].double()
    err = ((got - ref).abs() / (ref.abs() + 1.0)).max().item()
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2 * M * N * K / ms / 1e9
    if M == M0 and f16o: tot_ms += ms; tot_fl += 2.0 * M * N * K
    print(f"[{tag}] {name:28s} M={M:6d} N={N:5d} K={K:5d}: {ms*1e3:8.1f} us {tf:7.1f} TF  frac {tf/2500:.3f}  relerr={err:.2e}", flush=True)
if tot_ms > 0: print(f"[{tag}] layer GEMMs total {tot_ms:.3f} ms = {tot_fl/tot_ms/1e9:.1f} TF  frac {tot_fl/tot_ms/1e9/2500:.3f}")
