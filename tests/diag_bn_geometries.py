"""(diagnostic, run by hand: python tests/diag_bn_geometries.py tiny w64r64 ...) BatchNorm-tuning gradient of the HIP path vs the oracle (CPU) per layer, on ad-hoc ResNet geometries: bisects accuracy problems."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rlcf_ref as RR
from rlcf_amd import synth, _lib as L
from rlcf_amd.engine import Engine, TTAConfig
from rlcf_amd.synth import ClipGeometry
dev = torch.device("cuda:0")
CASES = {
    "tiny": (ClipGeometry(64, 64, (1, 2, 1, 1), 16, None, 77, 1024, 64, 1, 2), 8),
    "w64r64": (ClipGeometry(64, 64, (1, 1, 1, 1), 64, None, 77, 1024, 64, 1, 2), 16),
    "w64r224": (ClipGeometry(64, 224, (1, 1, 1, 1), 64, None, 77, 1024, 64, 1, 2), 16),
    "w16r224": (ClipGeometry(64, 224, (1, 1, 1, 1), 16, None, 77, 1024, 64, 1, 2), 16),
}
rg = synth.GEOMETRIES["tiny-r"]
for name in sys.argv[1:]:
    sg, N = CASES[name]
    n_cls = 16
    ssd = synth.make_state_dict(sg, 11, device=dev)
    rsd = synth.make_state_dict(rg, 23, device=dev)
    tokens = synth.make_token_bank(sg, n_cls, seed=7, n_ctx=4)
    views = synth.make_views(1000, N, sg.image_resolution, device=dev)
    cfg = TTAConfig(selection_p=0.5, lr=1e-3, tta_steps=1)
    hp = RR.TTAHyper(selection_p=0.5, tta_steps=1, sample_k=cfg.sample_k, lr=1e-3, weight_decay=cfg.weight_decay)
    # the reward model sees 32x32 inputs: views are resampled by both sides alike
    ref = RR.tta_sample_ln({k: v.cpu() for k, v in ssd.items()}, {k: v.cpu() for k, v in rsd.items()}, views.cpu(), tokens, hp)
    gr = ref["ln_grad"]
    for prec in (0, 2):
        eng = Engine(sg, rg, N, n_cls, prec)
        eng.load_state_dict(L.STUDENT, ssd); eng.load_state_dict(L.REWARD, rsd); eng.finalize()
        ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(sg, 4), device=dev)].clone()
        eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
        o = eng.tta_sample_ln(views, cfg)
        og = o["ln_grad"].cpu()
        print(f"{name} prec {prec}: sel {o['selected_idx'].tolist() == ref['selected_idx'].tolist()} total rel {((og - gr).norm() / gr.norm()).item():.3e} "
              f"logits {(o['logits'].cpu() - ref['logits']).abs().max().item():.2e}")
        off = 0
        for k in RR.visual_bn_keys({k: v for k, v in ssd.items()}):
            n = ssd[k].numel()
            e = ((og[off:off+n] - gr[off:off+n]).norm() / gr[off:off+n].norm()).item()
            print(f"   {k:36s} {e:.2e}" if k.endswith("weight") else f"  bias {e:.2e}", end="" if k.endswith("weight") else "\n")
            off += n
        eng.close()
