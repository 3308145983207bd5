"""Few-row split-f16 products of the one-image path through the C ABI (rlcf_gemm_skinny): M = 239 rows against the text tower's
weights, forward (operand scale 1), with a producer's max|A| and with the in-kernel per-workgroup max; us per launch (HIP events over 200
back-to-back launches on W matrices that rotate through 16 copies, so that W comes from HBM as it does in the step), max error vs f64.
args: [MxNxK ...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0"); st = lambda: torch.cuda.current_stream().cuda_stream
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(239, 512, 512), (239, 2048, 512), (239, 1536, 512), (239, 512, 2048), (239, 512, 1536), (64, 512, 512)]
NW = 16
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev); ws = [torch.randn(N, K, device=dev) * K ** -0.5 for _ in range(NW)]
    b = torch.randn(N, device=dev) * 0.1
    wp = []
    for w in ws:
        p = torch.empty(N, K, device=dev)           # 4 bytes per element: the pair layout
        L.check(lib.rlcf_split_pairs(w.data_ptr(), p.data_ptr(), N * K, L.PREC_F16X3, st()))
        wp.append(p)
    c = torch.empty(M, N, device=dev)
    amax = a.abs().max().reshape(1).clone()
    for mode, (am, loc, scale) in {"scale 1": (None, 0, 1.0), "amax_in": (amax, 0, 1e-5), "local": (None, 1, 1e-5)}.items():
        x = (a * scale).contiguous(); amx = (amax * scale).contiguous() if am is not None else None
        run = lambda i: L.check(lib.rlcf_gemm_skinny(x.data_ptr(), K, wp[i % NW].data_ptr(), b.data_ptr(), None, 0, None, 0, c.data_ptr(), N, M, N, K,
                                                      1.0, 0, amx.data_ptr() if amx is not None else None, loc, st()))
        run(0); torch.cuda.synchronize()
        ref = x.double() @ ws[0].double().t() + b.double()
        err = ((c.double() - ref).abs().max() / ref.abs().max()).item()
        for i in range(20): run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200): run(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        print(f"[{M},{N},{K}] {mode:8s}: {us:6.1f} us/launch  {2 * M * N * K / us / 1e6:6.1f} TF  rel err {err:.1e}", flush=True)
        assert err < 3e-6, err
