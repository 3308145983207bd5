"""Mid-size products of the one-image path on the 128x128 split-f16 kernel (reward tower of one image: M = 1182; final text pass:
M = 4095) through rlcf_gemm_f16x3 with interleaved operands; W rotates through 16 copies so that it comes from HBM as in the step.
(No K slices here: the op-level call has no workspace.)  args: [MxNxK ...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0"); st = lambda: torch.cuda.current_stream().cuda_stream
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a] or [
    (1182, 2304, 768), (1182, 768, 768), (1182, 3072, 768), (1182, 768, 3072), (4095, 1536, 512), (4095, 512, 512), (4095, 2048, 512), (4095, 512, 2048)]
NW = 16
torch.manual_seed(0)
def pairs(x):
    R, K = x.shape
    p = torch.empty(R, K, device=dev)
    L.check(lib.rlcf_split_pairs(x.data_ptr(), p.data_ptr(), R * K, L.PREC_F16X3, st()))
    return p
tot = 0.0
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev); ws = [torch.randn(N, K, device=dev) * K ** -0.5 for _ in range(NW)]
    ap = pairs(a); wp = [pairs(w) for w in ws]; c = torch.empty(M, N, device=dev)
    def run(i):
        L.check(lib.rlcf_gemm_f16x3(ap.data_ptr(), ap.data_ptr() + 64, 2 * K, wp[i % NW].data_ptr(), wp[i % NW].data_ptr() + 64, 2 * K, None, None, 0, None, 0,
                                    c.data_ptr(), N, None, None, 0, M, N, K, 1.0, 0, st()))
    run(0); torch.cuda.synchronize()
    err = (c[:64].double() - a[:64].double() @ ws[0].double().t()).abs().max().item()
    for i in range(20): run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200): run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3; tot += us
    sig = int(c.view(torch.int32).to(torch.int64).sum().item())
    print(f"SIG [{M},{N},{K}] {sig:x}")
    print(f"[{M},{N},{K}]: {us:7.1f} us  {2 * M * N * K / us / 1e6:6.1f} TF  maxerr {err:.1e}", flush=True)
print(f"sum {tot:.1f} us")
