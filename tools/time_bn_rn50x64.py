"""Row a-R at the largest geometry: RN50x64 @448^2 student, BatchNorm tuning (rlcf_tta_sample_ln), N = 32 views, ViT-L/14 reward,
train-mode BatchNorm and --prior_strength 16.  Prints ms/image and the device memory in use (z / y of every unit are kept for the backward)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L, synth
from rlcf_amd.engine import Engine, TTAConfig
dev = torch.device("cuda:0")
geo, rgeo = synth.GEOMETRIES["RN50x64"], synth.GEOMETRIES["ViT-L/14"]
ssd, rsd = synth.make_state_dict(geo, 11, device=dev), synth.make_state_dict(rgeo, 23, device=dev)
N, C = 32, 200
eng = Engine(geo, rgeo, N, C, L.PREC_F16X3)
eng.load_state_dict(L.STUDENT, ssd); eng.load_state_dict(L.REWARD, rsd); eng.finalize()
tokens = synth.make_token_bank(geo, C, seed=7, n_ctx=4)
ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(geo, 4), device=dev)].clone()
eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
cfg = TTAConfig(selection_p=0.1, tta_steps=1, sample_k=3, lr=1e-5, weight_decay=5e-4)
for prior in (-1, 16):
    eng.set_bn_prior_strength(prior)
    v = [synth.make_views(1000 + i, N, geo.image_resolution, device=dev) for i in range(3)]
    o = eng.tta_sample_ln(v[0], cfg); torch.cuda.synchronize(); t0 = time.perf_counter()
    for x in v[1:]: o = eng.tta_sample_ln(x, cfg)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    free, total = torch.cuda.mem_get_info()
    print(f"RN50x64 @448 BatchNorm tuning, N={N}, prior_strength={prior}: {dt*1e3:.1f} ms/image, |grad|={o['ln_grad'].norm().item():.3e}, "
          f"finite={bool(torch.isfinite(o['final_logits']).all())}, {(total - free)/2**30:.1f} GB of device memory in use")
