"""float64 companion of the bn_* reference fixtures: the ORACLE (oracle/rlcf_ref.py, pinned to the reference's float32 runs by
tests/test_oracle_golden.py::test_bn_tuning_oracle_matches_reference) re-run in double precision on the same seeded inputs.

Why: with train-mode BatchNorms the float32 gradient of the reference itself is only good to ~1e-3 .. 1e-2 of its norm (channels whose
batch variance is small against eps amplify the rounding noise of z - mean by 1/sqrt(var + eps), and the noise then rides down the whole
backward pass).  A float32 implementation with another summation order lands somewhere else inside that noise band, so the HIP path's
BatchNorm gradient is judged against the float64 value, next to the reference's own distance from it (`ref_err`), and against the
float32 fixture at the band's width.  Writes tests/golden/<case>_f64.npz: ln_grad, final_logits, bn_stats_after (as float64).

    python tests/golden/make_bn_f64.py [case ...]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import rlcf_ref as R          # noqa: E402
from rlcf_amd import synth               # noqa: E402

CASES = ["bn_tiny_train", "bn_tiny_train_s3", "bn_tiny_prior0", "bn_tiny_prior16_s3", "bn_rn50_train", "bn_rn50_prior16"]
# every-parameter tuning of a ModifiedResNet student (CLIPCLS_TTA(only_norm=False), tests/golden/make_golden.py --only rnvis,rnvisrn50): the same
# float64 re-run of the oracle; what is kept is what the float32 fixtures keep — per-tensor gradient norms, every 7th gradient element
# where the fixture has them, the first-pass and the final logits — plus the reference's own distance from each (round 5)
VIS_CASES = ["rnvis_tiny_s1", "rnvis_rn50"]


def main(names):
    torch.Tensor.float = lambda self, *a, **k: self.double()        # the oracle's explicit .float() casts follow the run's precision
    dbl = lambda sd: {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    for name in names:
        z = np.load(os.path.join(HERE, name + ".npz"))
        meta = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
        sg, rg = synth.GEOMETRIES[meta["student"]], synth.GEOMETRIES[meta["reward"]]
        ssd, rsd = synth.make_state_dict(sg, meta["student_seed"]), synth.make_state_dict(rg, meta["reward_seed"])
        tokens = synth.make_token_bank(sg, meta["n_cls"], seed=meta["bank_seed"], n_ctx=meta["n_ctx"])
        views = synth.make_views(meta["view_seed"], meta["n_views"], sg.image_resolution)
        hp = R.TTAHyper(selection_p=meta["selection_p"], tta_steps=meta["tta_steps"], sample_k=meta["sample_k"], lr=meta["lr"],
                        weight_decay=meta["weight_decay"])
        o = R.tta_sample_ln(dbl(ssd), dbl(rsd), views.double(), tokens, hp, prior_strength=meta["prior_strength"])
        assert o["selected_idx"].tolist() == z["selected_idx"].tolist() and o["topk_idx"].reshape(-1).tolist() == z["topk_idx"].reshape(-1).tolist()
        g64, g32 = o["ln_grad"].numpy(), z["ln_grad"].astype(np.float64)
        ref_err = float(np.linalg.norm(g32 - g64) / np.linalg.norm(g64))
        # (round 4: + the FIRST-PASS logits of all views — forward only, no gradient involved — and the reference's own distance from them)
        l64 = o["logits"].numpy()
        ref_logit_err = float(np.abs(z["logits"].astype(np.float64) - l64).max())
        np.savez_compressed(os.path.join(HERE, name + "_f64.npz"), ln_grad=g64, final_logits=o["final_logits"].numpy(),
                            bn_stats_after=o["bn_stats_after"].numpy(), ref_err=np.float64(ref_err), logits=l64,
                            ref_logit_err=np.float64(ref_logit_err))
        print(f"{name}: |reference f32 - f64| / |f64| = {ref_err:.3e}   first-pass logits {ref_logit_err:.2e}   "
              f"final logits {np.abs(z['final_logits'] - o['final_logits'].numpy()).max():.2e}", flush=True)


def main_vis(names):
    torch.Tensor.float = lambda self, *a, **k: self.double()
    dbl = lambda sd: {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    for name in names:
        z = np.load(os.path.join(HERE, name + ".npz"))
        meta = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
        sg, rg = synth.GEOMETRIES[meta["student"]], synth.GEOMETRIES[meta["reward"]]
        ssd, rsd = synth.make_state_dict(sg, meta["student_seed"]), synth.make_state_dict(rg, meta["reward_seed"])
        tokens = synth.make_token_bank(sg, meta["n_cls"], seed=meta["bank_seed"], n_ctx=meta["n_ctx"])
        views = synth.make_views(meta["view_seed"], meta["n_views"], sg.image_resolution)
        hp = R.TTAHyper(selection_p=meta["selection_p"], tta_steps=meta["tta_steps"], sample_k=meta["sample_k"], lr=meta["lr"],
                        weight_decay=meta["weight_decay"])
        o = R.tta_sample_ln(dbl(ssd), dbl(rsd), views.double(), tokens, hp, only_norm=False)
        assert o["selected_idx"].tolist() == z["selected_idx"].tolist() and o["topk_idx"].reshape(-1).tolist() == z["topk_idx"].reshape(-1).tolist()
        keys = R.visual_param_keys(ssd)
        g64 = o["ln_grad"]
        norms, off = [], 0
        for k in keys:
            n = ssd[k].numel()
            norms.append(float(g64[off:off + n].norm())); off += n
        norms = np.array(norms)
        ref_norm_err = np.abs(z["vis_grad_l2"].astype(np.float64) - norms) / np.maximum(norms, 1e-300)
        l64, f64 = o["logits"].numpy(), o["final_logits"].numpy()
        out = dict(vis_grad_l2=norms, ref_grad_l2_relerr=ref_norm_err, logits=l64, final_logits=f64,
                   ref_logit_err=np.float64(np.abs(z["logits"].astype(np.float64) - l64).max()),
                   ref_final_err=np.float64(np.abs(z["final_logits"].astype(np.float64) - f64).max()), bn_stats_after=o["bn_stats_after"].numpy())
        if "vis_grad_sample" in z.files:
            gs = g64[::7].numpy()
            out["vis_grad_sample"] = gs
            out["ref_sample_err"] = np.float64(np.linalg.norm(z["vis_grad_sample"].astype(np.float64) - gs) / np.linalg.norm(gs))
        np.savez_compressed(os.path.join(HERE, name + "_f64.npz"), **out)
        print(f"{name}: reference f32 vs f64 — per-tensor gradient norms worst {ref_norm_err.max():.3e} (median {np.median(ref_norm_err):.3e}), "
              f"first-pass logits {out['ref_logit_err']:.2e}, final logits {out['ref_final_err']:.2e}", flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    vis = [a for a in args if a.startswith("rnvis")]
    bn = [a for a in args if not a.startswith("rnvis")]
    if vis: main_vis(vis)
    if bn or not args: main(bn or CASES)
