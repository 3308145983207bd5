"""Small-M GEMMs of the one-image path (sparse text forward / backward: M = 239 rows; reward tower: M = 1182): us per launch of
rlcf_gemm_nt in f32 mode (RLCF_F32_SMALL=1 routes them to the LDS-tiled 64x64 kernel) — A/B by environment."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0"); st = lambda: torch.cuda.current_stream().cuda_stream
for M, N, K in [(239, 2048, 512), (239, 512, 2048), (239, 512, 512), (239, 1536, 512), (239, 512, 1536), (1182, 768, 768)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev); c = torch.empty(M, N, device=dev)
    run = lambda: L.check(lib.rlcf_gemm_nt(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), None, 0, None, 0, c.data_ptr(), N, M, N, K, 1.0, 0, L.PREC_F32, st()))
    run(); torch.cuda.synchronize()
    err = (c.double() - (a.double() @ w.double().t() + b.double())).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"RLCF_F32_SMALL={os.environ.get('RLCF_F32_SMALL', '0')} [{M},{N},{K}]: {us:7.1f} us  {2 * M * N * K / us / 1e6:6.1f} TF  maxerr {err:.1e}")
