"""One image per pass, K samples in flight: K engines (own weights, own scratch) on K streams driven from K host threads, each running
rlcf_tta_sample back to back on its own test images — does the chip overlap one sample's few-row tail (reward tower, sparse text passes,
final text pass) with another sample's 64-view student forward?  Prints aggregate images/s for K = 1, 2, 3.
env GRID=1: GEMM weights on the fp16 grid.  args: [images per engine]"""
import os, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L, synth
from rlcf_amd.engine import Engine, TTAConfig
dev = torch.device("cuda:0")
geo = synth.GEOMETRIES["ViT-B/16"]
ssd, rsd = synth.make_state_dict(geo, 11, device=dev), synth.make_state_dict(geo, 23, device=dev)
if os.environ.get("GRID") == "1":
    ssd, rsd = synth.to_fp16_grid(ssd), synth.to_fp16_grid(rsd)
tokens = synth.make_token_bank(geo, 1000, seed=7, n_ctx=4)
ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(geo, 4), device=dev)].clone()
cfg = TTAConfig(selection_p=0.1, tta_steps=1, sample_k=3)
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 24
KMAX = 3
engines = []
for k in range(KMAX):
    e = Engine(geo, geo, 64, 1000, L.PREC_F16X3)
    e.load_state_dict(L.STUDENT, ssd); e.load_state_dict(L.REWARD, rsd); e.finalize()
    e.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
    engines.append(e)
views = [synth.make_views(1113 + i, 64, 224, device=dev) for i in range(8)]
ref = [engines[0].tta_sample(v, cfg, want_intermediates=False)["final_logits"].clone() for v in views]
torch.cuda.synchronize()

def work(k, n, out):
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        last = None
        for i in range(n):
            last = engines[k].tta_sample(views[(i + k) % 8], cfg, want_intermediates=False)["final_logits"]
        torch.cuda.current_stream().synchronize()
        out[k] = (last.clone(), (n - 1 + k) % 8)

for K in (1, 2, 3, 1, 2):
    out = {}
    for k in range(K): work(k, 2, out)                     # warm
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k, n_img, out)) for k in range(K)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = all(torch.equal(out[k][0], ref[out[k][1]]) for k in range(K))
    print(f"{K} in flight: {K * n_img / dt:6.1f} images/s ({dt / (K * n_img) * 1e3:.2f} ms/image), results equal to the serial run: {ok}", flush=True)
