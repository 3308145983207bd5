"""Find view seeds whose RLCF step has non-trivial rewards (positive CLIP scores) on the synthetic ViT-B/16 pair."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L, synth
from rlcf_amd.engine import Engine, TTAConfig
dev = torch.device("cuda:0")
geo = synth.GEOMETRIES["ViT-B/16"]
ssd, rsd = synth.make_state_dict(geo, 11, device=dev), synth.make_state_dict(geo, 23, device=dev)
eng = Engine(geo, geo, 64, 1000, L.PREC_F16X3)
eng.load_state_dict(L.STUDENT, ssd); eng.load_state_dict(L.REWARD, rsd); eng.finalize()
tokens = synth.make_token_bank(geo, 1000, seed=7, n_ctx=4)
ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(geo, 4), device=dev)].clone()
eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
cfg = TTAConfig(selection_p=0.1)
for seed in range(1000, 1120):
    o = eng.tta_sample(synth.make_views(seed, 64, 224, device=dev), cfg)
    pos = int((o["clip_score"] > 0).sum())
    if pos >= 1:
        print(seed, pos, float(o["ctx_grad"].norm()), [round(x, 4) for x in o["clip_score"].tolist()])
