"""(diagnostic, run by hand: python tests/diag_bn_fixture.py [fixture]) Per-BatchNorm relative error of the gradient rlcf_tta_sample_ln returns for a ResNet student, against a bn_* reference fixture."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_parity import load_golden, _cfg_from_meta
from oracle import rlcf_ref as RR
from rlcf_amd import synth, _lib as L
from rlcf_amd.engine import Engine
name = sys.argv[1] if len(sys.argv) > 1 else "bn_rn50_train"
dev = torch.device("cuda:0")
g, meta = load_golden(name)
sg, rg = synth.GEOMETRIES[meta["student"]], synth.GEOMETRIES[meta["reward"]]
ssd = synth.make_state_dict(sg, meta["student_seed"], device=dev)
rsd = synth.make_state_dict(rg, meta["reward_seed"], device=dev)
for prec in (0, 2):
    eng = Engine(sg, rg, meta["n_views"], meta["n_cls"], prec)
    eng.load_state_dict(L.STUDENT, ssd); eng.load_state_dict(L.REWARD, rsd); eng.finalize()
    tokens = synth.make_token_bank(sg, meta["n_cls"], seed=meta["bank_seed"], n_ctx=meta["n_ctx"])
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(sg, meta["n_ctx"]), device=dev)].clone()
    eng.set_class_bank(tokens, meta["n_ctx"], ctx0, L.TEXT_SHARED)
    eng.set_bn_prior_strength(meta["prior_strength"])
    views = synth.make_views(meta["view_seed"], meta["n_views"], sg.image_resolution, device=dev)
    o = eng.tta_sample_ln(views, _cfg_from_meta(meta))
    og, gr = o["ln_grad"].cpu(), g["ln_grad"]
    import numpy as np
    z64 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name + "_f64.npz"))
    g64 = torch.from_numpy(z64["ln_grad"])
    print(f"prec {prec}: vs f64: HIP {((og.double() - g64).norm() / g64.norm()).item():.3e}  reference {float(z64['ref_err']):.3e}   "
          f"final logits vs f64: HIP {(o['final_logits'].cpu().double() - torch.from_numpy(z64['final_logits'])).abs().max().item():.2e} "
          f"reference {(g['final_logits'].double() - torch.from_numpy(z64['final_logits'])).abs().max().item():.2e}")
    if "logits" in z64.files:
        l64 = torch.from_numpy(z64["logits"])
        print(f"prec {prec}: FIRST-PASS logits vs f64: HIP {(o['logits'].cpu().double() - l64).abs().max().item():.2e}  reference {float(z64['ref_logit_err']):.2e}")
    print(f"prec {prec}: total rel {((og - gr).norm() / gr.norm()).item():.3e}  logits {(o['logits'].cpu() - g['logits']).abs().max().item():.2e} "
          f"final {(o['final_logits'].cpu() - g['final_logits']).abs().max().item():.2e} "
          f"stats {(eng.bn_stats().cpu() - g['bn_stats_after']).abs().max().item():.2e}")
    off = 0
    for k in (RR.visual_bn_keys(ssd) if os.environ.get("BN_DIAG_LAYERS") else []):
        n = ssd[k].numel()
        if k.endswith("weight"):
            print(f"   {k:40s} {((og[off:off+n] - gr[off:off+n]).norm() / gr[off:off+n].norm()).item():.2e}", end="")
        else:
            print(f"   bias {((og[off:off+n] - gr[off:off+n]).norm() / gr[off:off+n].norm()).item():.2e}")
        off += n
    eng.close()
