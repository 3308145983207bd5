"""Interleaved [hi32|lo32] operand layout (128-B lines per row per K tile, what the engine uses) vs separate hi/lo arrays (the
plain C-ABI form) through rlcf_gemm_f16x3.  args: [MxNxK ...] [il]   ("il": interleaved layout only, few launches: PMC passes)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0"); st = lambda: torch.cuda.current_stream().cuda_stream
def il(hi, lo):
    R, K = hi.shape
    return torch.stack([hi.view(R, K // 32, 32), lo.view(R, K // 32, 32)], dim=2).reshape(R, 2 * K).contiguous()
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(100864, 3072, 768), (100864, 768, 3072), (100864, 2304, 768)]
il_only = "il" in sys.argv[1:]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; c = torch.empty(M, N, device=dev)
    ah = torch.empty(M, K, dtype=torch.float16, device=dev); al = torch.empty_like(ah)
    wh = torch.empty(N, K, dtype=torch.float16, device=dev); wl = torch.empty_like(wh)
    L.check(lib.rlcf_split_f16x2(a.data_ptr(), ah.data_ptr(), al.data_ptr(), M * K, st()))
    L.check(lib.rlcf_split_f16x2(w.data_ptr(), wh.data_ptr(), wl.data_ptr(), N * K, st()))
    ail, wil = il(ah, al), il(wh, wl)
    ref = (a[:64].double() @ w.double().t()).float()
    for name, (p_ah, p_al, lda, p_wh, p_wl, ldw) in {"separate": (ah.data_ptr(), al.data_ptr(), K, wh.data_ptr(), wl.data_ptr(), K),
                                                      "interleaved": (ail.data_ptr(), ail.data_ptr() + 64, 2 * K, wil.data_ptr(), wil.data_ptr() + 64, 2 * K)}.items():
        if il_only and name != "interleaved":
            continue
        def run():
            L.check(lib.rlcf_gemm_f16x3(p_ah, p_al, lda, p_wh, p_wl, ldw, None, None, 0, None, 0, c.data_ptr(), N, None, None, 0, M, N, K, 1.0, 0, st()))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 4 if il_only else 20
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"M={M} N={N} K={K} {name:12s}: {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:7.1f} TF maxerr={(c[:64]-ref).abs().max().item():.2e}", flush=True)
