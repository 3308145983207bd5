"""Round 6, verdict item 1: the persistent single-pass f16 GEMM with half of every tile's stores DEFERRED under the next tile's K loop
(gemm_nt_f16_pp_kernel<.., DEFER = 1>, rlcf_amd/csrc/gemm_f16.hip) against the same kernel with all 16 stores of a wave in the epilogue
(RLCF_F16_PP_DEFER=0; the switch is read per launch, so both run in ONE process on ONE lease, interleaved A/B/A/B).

  1. bit-equality of the two forms on whole matrices (both epilogues, K = 768 and 3072, an M that is not a multiple of 256);
  2. timing of the four products of a ViT-B/16 layer at the pass size of the driver's command line (20 images x 64 views x 197 tokens),
     `reps` launches back to back per arm, arms interleaved `rounds` times; per shape the mean per arm and the ratio.
args: [M] [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
M0 = int(sys.argv[1]) if len(sys.argv) > 1 else 252160
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W = 768


def gemm(a, w, b, c16, epi):
    M, K = a.shape
    N = w.shape[0]
    L.check(lib.rlcf_gemm_f16(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), None, N, None, N, c16.data_ptr(), N, M, N, K, 1.0, epi, st()))


def arm(defer):
    os.environ["RLCF_F16_PP_DEFER"] = "1" if defer else "0"


# ---- 1. bit-equality
for (M, N, K, epi) in [(70000 + 37, 768, 768, 0), (70000 + 37, 2304, 768, 0), (66000, 3072, 768, 1), (70000 + 37, 768, 3072, 0), (66816, 1024, 1024, 0)]:
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.randn(N, device=dev) * 0.1
    outs = []
    for d in (0, 1, 1):
        arm(d)
        c = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
        gemm(a, w, b, c, epi)
        torch.cuda.synchronize()
        outs.append(c)
    rows = torch.randint(0, M, (256,), device=dev)
    ref = a[rows].double() @ w.double().t() + b.double()
    if epi == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    err = ((outs[1][rows].double() - ref).abs() / (ref.abs() + 1.0)).max().item()
    same = torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    print(f"[defer check] M={M} N={N} K={K} epi={epi}: deferred == epilogue-only stores: {same}; nan left: {int(torch.isnan(outs[1]).sum())}; relerr vs f64 {err:.2e}", flush=True)
    assert same and err < 2e-2 and not torch.isnan(outs[1]).any()
    del a, w, b, outs

# ---- 2. timing, interleaved arms
shapes = [("in_proj", M0, 3 * W, W, 0), ("out_proj", M0, W, W, 0), ("c_fc+gelu", M0, 4 * W, W, 1), ("c_proj", M0, W, 4 * W, 0)]
res = {(n, d): [] for n, *_ in shapes for d in (0, 1)}
bufs = {}
for name, M, N, K, epi in shapes:
    bufs[name] = (torch.randn(M, K, device=dev).half(), (torch.randn(N, K, device=dev) * K ** -0.5).half(), torch.randn(N, device=dev) * 0.1,
                  torch.empty(M, N, dtype=torch.float16, device=dev))
reps = 20
for r in range(ROUNDS):
    for d in (0, 1):
        arm(d)
        for name, M, N, K, epi in shapes:
            a, w, b, c = bufs[name]
            for _ in range(3):
                gemm(a, w, b, c, epi)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                gemm(a, w, b, c, epi)
            e1.record()
            torch.cuda.synchronize()
            res[(name, d)].append(e0.elapsed_time(e1) / reps * 1e3)
tot = {0: 0.0, 1: 0.0}
fl = 0.0
for name, M, N, K, epi in shapes:
    m0, m1 = sum(res[(name, 0)]) / ROUNDS, sum(res[(name, 1)]) / ROUNDS
    tot[0] += m0
    tot[1] += m1
    fl += 2.0 * M * N * K
    f = 2.0 * M * N * K / 1e6
    print(f"[defer A/B] {name:10s} M={M} N={N:5d} K={K:5d}: epilogue-only {m0:8.1f} us ({f / m0 / 2500:.3f} of peak)  deferred {m1:8.1f} us ({f / m1 / 2500:.3f})  "
          f"ratio {m0 / m1:.3f}   legs {['%.0f/%.0f' % (x, y) for x, y in zip(res[(name, 0)], res[(name, 1)])]}", flush=True)
print(f"[defer A/B] layer total: epilogue-only {tot[0] / 1e3:.3f} ms ({fl / 1e6 / tot[0] / 2500:.3f})  deferred {tot[1] / 1e3:.3f} ms ({fl / 1e6 / tot[1] / 2500:.3f})  "
      f"ratio {tot[0] / tot[1]:.3f}")
