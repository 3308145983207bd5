"""Micro-benchmark of the GEMM kernels through the C ABI (GPU only): ViT-B/16 layer shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
shapes = [(12608, 2304, 768), (12608, 768, 768), (12608, 3072, 768), (12608, 768, 3072), (4095, 1536, 512), (4095, 2048, 512), (4095, 512, 2048),
          (8192, 8192, 4096)] if len(sys.argv) < 2 else [tuple(int(x) for x in s.split("x")) for s in sys.argv[1:]]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5
    c = torch.empty(M, N, device=dev)
    for prec in (0, 2):
        def run():
            L.check(lib.rlcf_gemm_nt(a.data_ptr(), K, w.data_ptr(), K, None, None, 0, None, 0, c.data_ptr(), N, M, N, K, 1.0, 0, prec, st()))
        if prec == 2:   # time the pre-split kernel only
            ah = torch.empty(M, K, dtype=torch.float16, device=dev); al = torch.empty_like(ah)
            wh = torch.empty(N, K, dtype=torch.float16, device=dev); wl = torch.empty_like(wh)
            L.check(lib.rlcf_split_f16x2(a.data_ptr(), ah.data_ptr(), al.data_ptr(), M * K, st()))
            L.check(lib.rlcf_split_f16x2(w.data_ptr(), wh.data_ptr(), wl.data_ptr(), N * K, st()))
            def run():
                L.check(lib.rlcf_gemm_f16x3(ah.data_ptr(), al.data_ptr(), K, wh.data_ptr(), wl.data_ptr(), K, None, None, 0, None, 0,
                                            c.data_ptr(), N, None, None, 0, M, N, K, 1.0, 0, st()))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        ref = (a[:64].double() @ w.double().t()).float()
        err = (c[:64] - ref).abs().max().item()
        print(f"M={M} N={N} K={K} prec={prec}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TF  maxerr={err:.2e}", flush=True)
