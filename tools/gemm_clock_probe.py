"""Sustained shader clock / socket power while one GEMM variant runs back to back (c_proj shape of ViT-B/16: M x 768 x 3072 + residual).
RLCF_X3_V4 selects the variant (0 = 8-wave v3i, 1 = 4-wave v4, 2 = v4 without DMA, 3 = v4 without MFMAs).  Samples rocm-smi at ~4 Hz."""
import os, subprocess, sys, threading, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0"); st = lambda: torch.cuda.current_stream().cuda_stream
M, N, K = 403456, 768, 3072
def il(hi, lo):
    R, K_ = hi.shape
    return torch.stack([hi.view(R, K_ // 32, 32), lo.view(R, K_ // 32, 32)], dim=2).reshape(R, 2 * K_).contiguous()
def split(x):
    R, K_ = x.shape
    h = torch.empty(R, K_, dtype=torch.float16, device=dev); l = torch.empty_like(h)
    L.check(lib.rlcf_split_f16x2(x.data_ptr(), h.data_ptr(), l.data_ptr(), R * K_, st()))
    return il(h, l)
a2 = split(torch.randn(M, K, device=dev)); w2 = split(torch.randn(N, K, device=dev) * K ** -0.5); b = torch.randn(N, device=dev) * 0.1
c = torch.randn(M, N, device=dev) * 1e-3
def run():
    L.check(lib.rlcf_gemm_f16x3(a2.data_ptr(), a2.data_ptr() + 64, 2 * K, w2.data_ptr(), w2.data_ptr() + 64, 2 * K, b.data_ptr(), c.data_ptr(), N, None, 0,
                                c.data_ptr(), N, None, None, N, M, N, K, 1e-3, 0, st()))
samples, stop = [], False
def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(o); card = next(iter(d.values()))
            samples.append({k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()})
        except Exception as e:
            samples.append({"err": str(e)})
        time.sleep(0.2)
for _ in range(3): run()
torch.cuda.synchronize()
th = threading.Thread(target=poll); th.start()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = int(os.environ.get("REPS", "700"))
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
ms = e0.elapsed_time(e1) / reps
print(f"variant RLCF_X3_V4={os.environ.get('RLCF_X3_V4', '0')}: {ms*1e3:.1f} us per launch, {2*M*N*K/ms/1e9:.1f} TF over {reps} launches; {len(samples)} samples")
mid = samples[len(samples) // 4: max(len(samples) // 4 + 1, 3 * len(samples) // 4)]
for s_ in mid[:6]: print("  ", s_)
