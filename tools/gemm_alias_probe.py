"""Does fabric traffic limit the 256x256 split-f16 GEMM?  Same launch, same MFMA / DMA instruction stream, but every A row aliases
ONE 256-row panel (A rows taken modulo 256 through a 256-row buffer and lda as usual is not possible, so: M rows all read from a
buffer of 256 rows by giving the kernel M = 256-row tiles of the SAME memory via lda = 0 is degenerate) -> instead A is a [256, K]
buffer and the call runs N x M swapped: C^T tiles... kept simple: three variants of A footprint —
  full : A [M, K] (775 MB at M = 252160, K = 768)
  small: A rows r -> buffer row (r % 8192) emulated by launching M/8192 GEMMs of 8192 rows on the SAME A slab, back to back
Reports TF for both; equal TF => the A-panel re-reads (fabric / MALL) are not what limits the kernel."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0"); st = lambda: torch.cuda.current_stream().cuda_stream
def il(hi, lo):
    R, K = hi.shape
    return torch.stack([hi.view(R, K // 32, 32), lo.view(R, K // 32, 32)], dim=2).reshape(R, 2 * K).contiguous()
M, N, K = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "252160x2304x768").split("x"))
a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5
ah = torch.empty(M, K, dtype=torch.float16, device=dev); al = torch.empty_like(ah)
wh = torch.empty(N, K, dtype=torch.float16, device=dev); wl = torch.empty_like(wh)
L.check(lib.rlcf_split_f16x2(a.data_ptr(), ah.data_ptr(), al.data_ptr(), M * K, st()))
L.check(lib.rlcf_split_f16x2(w.data_ptr(), wh.data_ptr(), wl.data_ptr(), N * K, st()))
ail, wil = il(ah, al), il(wh, wl)
del a, ah, al
c = torch.empty(M, N, device=dev)
def run_full():
    L.check(lib.rlcf_gemm_f16x3(ail.data_ptr(), ail.data_ptr() + 64, 2 * K, wil.data_ptr(), wil.data_ptr() + 64, 2 * K, None, None, 0, None, 0,
                                c.data_ptr(), N, None, None, 0, M, N, K, 1.0, 0, st()))
def run_c_small():        # full A, but C written into ONE 65536-row slab over and over (C footprint 604 MB -> fits nothing either; control)
    S = 65536
    for m0 in range(0, M, S):
        rows = min(S, M - m0)
        L.check(lib.rlcf_gemm_f16x3(ail.data_ptr() + m0 * 4 * K, ail.data_ptr() + m0 * 4 * K + 64, 2 * K, wil.data_ptr(), wil.data_ptr() + 64, 2 * K, None, None, 0, None, 0,
                                    c.data_ptr(), N, None, None, 0, rows, N, K, 1.0, 0, st()))
def run_a_small():        # A from ONE 65536-row slab (201 MB: Infinity-Cache resident), C written in full
    S = 65536
    for m0 in range(0, M, S):
        rows = min(S, M - m0)
        L.check(lib.rlcf_gemm_f16x3(ail.data_ptr(), ail.data_ptr() + 64, 2 * K, wil.data_ptr(), wil.data_ptr() + 64, 2 * K, None, None, 0, None, 0,
                                    c.data_ptr() + m0 * 4 * N, N, None, None, 0, rows, N, K, 1.0, 0, st()))
def run_a_tiny():         # A from ONE 8192-row slab (25 MB: L2 + MALL resident)
    S = 8192
    for m0 in range(0, M, 65536):
        rows = min(65536, M - m0)
        for s0 in range(0, rows, S):
            L.check(lib.rlcf_gemm_f16x3(ail.data_ptr(), ail.data_ptr() + 64, 2 * K, wil.data_ptr(), wil.data_ptr() + 64, 2 * K, None, None, 0, None, 0,
                                        c.data_ptr() + (m0 + s0) * 4 * N, N, None, None, 0, min(S, rows - s0), N, K, 1.0, 0, st()))
import statistics
res = {k: [] for k in ("full", "slabs_fullA", "slabs_A201MB", "slabs_A25MB")}
fns = dict(full=run_full, slabs_fullA=run_c_small, slabs_A201MB=run_a_small, slabs_A25MB=run_a_tiny)
for r in range(6):
    for k, fn in fns.items():
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); torch.cuda.synchronize()
        if r: res[k].append(e0.elapsed_time(e1) / 3)
for k, v in res.items():
    ms = statistics.median(v)
    print(f"{M}x{N}x{K} {k:14s}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:6.1f} TF", flush=True)
