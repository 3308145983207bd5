"""Timing of BASELINE configs[4] (RN50x64 student + ViT-L/14 reward, N=32, 1000 classes) and of the bare RN50x64 image tower."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L, synth
from rlcf_amd.engine import Engine, TTAConfig
dev = torch.device("cuda:0")
sg, rg = synth.GEOMETRIES["RN50x64"], synth.GEOMETRIES["ViT-L/14"]
ssd = synth.make_state_dict(sg, 11, device=dev); rsd = synth.make_state_dict(rg, 23, device=dev)
tokens = synth.make_token_bank(sg, 1000, seed=7, n_ctx=4)
ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(sg, 4), device=dev)].clone()
views = synth.make_views(1000, 32, 448, device=dev)
for prec, name in [(L.PREC_F16X3, "f16x3"), (L.PREC_F32, "f32")][: 1 if os.environ.get("ONLY_X3") else 2]:
    eng = Engine(sg, rg, 32, 1000, prec)
    eng.load_state_dict(L.STUDENT, ssd); eng.load_state_dict(L.REWARD, rsd); eng.finalize()
    eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
    cfg = TTAConfig(selection_p=0.1)
    for fn, label in ((lambda: eng.encode_image(L.STUDENT, views), "RN50x64 encode 32 views"), (lambda: eng.tta_sample(views, cfg, want_intermediates=False), "config5 tta_sample")):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"{name}: {label}: {dt*1e3:.1f} ms  (flops {eng.last_flops()/1e12:.2f} TF -> {eng.last_flops()/dt/1e12:.1f} TF/s)", flush=True)
    eng.close()
