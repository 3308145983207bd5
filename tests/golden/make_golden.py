"""Generate the golden fixtures under tests/golden/ by running the REFERENCE
itself (imported read-only from /root/reference/TPT) on this repo's seeded
synthetic weights / views / token banks.

Runs only in the build container (the reference does not travel).  The fixtures
hold inputs that cannot be regenerated from a seed (none — everything comes from
rlcf_amd.synth) and the reference's OUTPUTS, as small .npz files.

    python tests/golden/make_golden.py [--only tiny,small,ops,b16n8,b16n64,modules]

Import recipe: SURVEY.md Appendix B (stub torchvision/ftfy, open the
DOWNLOAD_ROOT gate, replace clip.load by a factory that builds the reference's
own `CLIP` class and loads our state dict, replace the BPE `tokenize` by a
lookup into the synthetic token bank).
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import types
import warnings

import numpy as np
import torch
import torch.utils._python_dispatch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from rlcf_amd import synth  # noqa: E402
from oracle import rlcf_ref as RR  # noqa: E402  (only for the key-order assertion of the only_norm=False cases)

REF = "/root/reference/TPT"


def import_reference():
    sys.dont_write_bytecode = True
    tv = types.ModuleType("torchvision")
    tv.__path__ = []
    tvt = types.ModuleType("torchvision.transforms")
    tvd = types.ModuleType("torchvision.datasets")
    for n in ["Compose", "Resize", "CenterCrop", "ToTensor", "Normalize", "RandomResizedCrop",
              "RandomHorizontalFlip"]:
        setattr(tvt, n, type(n, (), {"__init__": lambda s, *a, **k: None}))
    tvt.InterpolationMode = type("InterpolationMode", (), {"BICUBIC": "bicubic"})
    tv.transforms, tv.datasets = tvt, tvd
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.datasets": tvd})
    ft = types.ModuleType("ftfy")
    ft.fix_text = lambda s: s
    sys.modules["ftfy"] = ft
    _exists = os.path.exists
    os.path.exists = lambda p: True if p == "/YOUR/PATH" else _exists(p)
    sys.path.insert(0, REF)
    import clip  # noqa
    import clip.clip as cc
    import clip.custom_clip as custom
    import clip_reward
    from clip import model as refmodel
    import tpt_cls_rl
    import tune_cls_rl                      # (_modified_bn_forward, the BatchNorm forward `--prior_strength` installs)
    os.path.exists = _exists
    return types.SimpleNamespace(clip=clip, cc=cc, custom=custom, clip_reward=clip_reward,
                                 refmodel=refmodel, tpt=tpt_cls_rl, tune=tune_cls_rl)


class Bank:
    """Synthetic tokenizer: 'a photo of a c<i>.' -> row i of the token bank."""

    def __init__(self, geo, n_cls, n_ctx=4, seed=7):
        self.geo, self.n_ctx = geo, n_ctx
        self.tokens = synth.make_token_bank(geo, n_cls, seed=seed, n_ctx=n_ctx)
        self.ctx_ids = synth.ctx_token_ids_default(geo, n_ctx)
        self.classnames = [f"c{i}" for i in range(n_cls)]

    def tokenize(self, texts, context_length=77, truncate=False):
        if isinstance(texts, str):
            texts = [texts]
        rows = []
        for t in texts:
            t = t.strip()
            if t.endswith("."):
                rows.append(self.tokens[int(t.rstrip(".").split("c")[-1])])
            else:  # the ctx_init words
                r = torch.zeros(self.geo.context_length, dtype=torch.int64)
                ids = [self.geo.vocab_size - 2, *self.ctx_ids, self.geo.vocab_size - 1]
                r[: len(ids)] = torch.tensor(ids)
                rows.append(r)
        return torch.stack(rows)


class _FastHalfMM(torch.utils._python_dispatch.TorchDispatchMode):
    """torch's CPU kernel for a float16 mm whose SECOND operand is a contiguous [K, N] matrix (the `grad_out @ W` of every nn.Linear's
    backward) is a scalar loop, 150x slower than the same product against a transposed view (measured here: 10.1 s vs 0.07 s at
    3850 x 2048 x 512) — hours per test image at 1000 classes.  The fp16-autocast fixture therefore runs the reference under this
    dispatch mode, which hands such products to the fast kernel (same operands, same mathematical product, fp32 accumulation inside the
    library either way); nothing of the reference's code is touched."""

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        if func is torch.ops.aten.mm.default and args[0].dtype == torch.float16 and args[1].is_contiguous():
            return func(args[0], args[1].t().contiguous().t())
        return func(*args, **(kwargs or {}))


def install_models(ref, sds):
    """sds: arch name -> (geometry, state dict).  Makes every `load` in the
    reference return the reference's own CLIP class with our weights."""

    def fake_load(name, device="cpu", jit=False, download_root=None):
        geo, sd = sds[name]
        m = ref.refmodel.CLIP(*geo.as_tuple())
        m.load_state_dict({k: v.clone() for k, v in sd.items()})
        return m.eval().float(), geo.embed_dim, None

    ref.cc.load = ref.clip.load = ref.custom.load = fake_load
    ref.clip_reward.clip.load = fake_load


def run_reference_tta(ref, student, reward, n_views, n_cls, hp, view_seed=1000, n_ctx=4):
    """The harness body TPT/tpt_cls_rl.py:251-262 around the reference's own
    test_time_tuning, with taps on its intermediates."""
    ensemble = "+" in reward
    grid = bool(hp.pop("fp16_grid", 0))         # GEMM weights rounded to fp16 values, as a released checkpoint holds them (synth.to_fp16_grid)
    # fp16_autocast: the arithmetic of the reference's GPU path (TPT/tpt_cls_rl.py:52,261 wrap every model call in
    # torch.cuda.amp.autocast(); :127 GradScaler(init_scale=1000)) reproduced on this CPU-only box: the SAME reference code with
    # torch.cuda.amp.autocast bound to torch.autocast("cpu", float16) and an ENABLED GradScaler (torch.amp.GradScaler("cpu")).
    amp16 = bool(hp.pop("fp16_autocast", 0))
    s_geo = synth.GEOMETRIES[student]
    s_sd = synth.make_state_dict(s_geo, seed=11)
    if grid:
        s_sd = synth.to_fp16_grid(s_sd)
    if not ensemble:
        r_geo = synth.GEOMETRIES[reward]
        r_sd = synth.make_state_dict(r_geo, seed=23)
        if grid:
            r_sd = synth.to_fp16_grid(r_sd)
    install_models(ref, {student: (s_geo, s_sd)})
    bank = Bank(s_geo, n_cls, n_ctx)
    ref.custom.tokenize = bank.tokenize
    # name_lens of the reference come from its BPE tokenizer (custom_clip.py:127); the synthetic bank's class names have the length the
    # token bank gives them: 'front' / 'middle' prompts split the suffix there
    ref.custom._tokenizer = types.SimpleNamespace(
        encode=lambda name: [0] * (int(bank.tokens[int(name.split("c")[-1])].argmax()) - 1 - n_ctx - 1))
    model = ref.custom.ClipTestTimeTuning("cpu", bank.classnames, None, arch=student, n_ctx=n_ctx,
                                          ctx_init=hp.get("ctx_init", "a_photo_of_a"), ctx_position=hp.get("ctx_position", "end"))
    for name, p in model.named_parameters():
        if "prompt_learner" not in name:
            p.requires_grad_(False)
    optimizer = torch.optim.AdamW(model.prompt_learner.parameters(), hp["lr"], weight_decay=hp["weight_decay"])
    args = types.SimpleNamespace(tta_steps=hp["tta_steps"], selection_p=hp["selection_p"],
                                 min_entropy_reg=hp.get("min_entropy_reg", 0),
                                 min_entropy_w=hp.get("min_entropy_w", 0.2), gpu=None, tpt=True)
    if ensemble:
        # CLIPRewardsMultiple (clip_reward.py:180-307) over three of the arch names its CONFIDECES table knows, each bound
        # to a seeded synthetic CLIP
        members = synth.reward_members(reward, hp["reward_seeds"])
        names = hp.get("reward_archs", "+".join(ENSEMBLE_NAMES)).split("+")
        install_models(ref, {a: m for a, m in zip(names, members)})
        rm = ref.clip_reward.CLIPRewardsMultiple("cpu", arch=names[: len(members)], classification=True,
                                                 amplify_rewards=hp.get("reward_amplify", False), sample_k=hp["sample_k"],
                                                 reward_process=hp.get("reward_process", True),
                                                 process_batch=hp.get("process_batch", False),
                                                 weighted_scores=bool(hp.get("weighted_scores", 1)),
                                                 default_resolutions=s_geo.image_resolution)
    else:
        install_models(ref, {reward: (r_geo, r_sd)})      # student and reward may share an arch name
        rm = ref.clip_reward.CLIPRewards("cpu", arch=reward, classification=True,
                                         amplify_rewards=hp.get("reward_amplify", False), sample_k=hp["sample_k"],
                                         reward_process=hp.get("reward_process", True),
                                         process_batch=hp.get("process_batch", False),
                                         **({"default_resolutions": s_geo.image_resolution} if s_geo.image_resolution != 224 and
                                            r_geo.image_resolution == 224 else {}))     # views at 448 (RN50x64 student), reward model at 224
    assert torch.equal(model.prompt_learner.tokenized_prompts, bank.tokens)
    rm.set_class_features(tokenized_classes=model.prompt_learner.tokenized_prompts)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        scaler = torch.amp.GradScaler("cpu", init_scale=1000) if amp16 else torch.cuda.amp.GradScaler(init_scale=1000)
    views = synth.make_views(view_seed, n_views, s_geo.image_resolution)
    orig_autocast = torch.cuda.amp.autocast
    if amp16:
        torch.cuda.amp.autocast = lambda *a, **k: torch.autocast("cpu", dtype=torch.float16)

    taps = {}
    orig_select = ref.tpt.select_confident_samples

    def tap_select(logits, top):
        out, idx = orig_select(logits, top)
        if "logits" not in taps:
            taps["logits"] = logits.detach().clone()
            taps["selected_idx"] = idx.clone()
        return out, idx

    orig_score, orig_post = rm.CLIPScore, rm.rewards_post_process

    def tap_score(*a, **k):
        s = orig_score(*a, **k)
        if "clip_score" not in taps:
            taps["clip_score"] = s.detach().clone()
            taps["topk_idx"] = k["class_index"].clone()
        return s

    def tap_post(x):
        r = orig_post(x)
        taps.setdefault("rewards", r.detach().clone())
        return r

    ref.tpt.select_confident_samples = tap_select
    rm.CLIPScore, rm.rewards_post_process = tap_score, tap_post
    grads = []
    model.prompt_learner.ctx.register_hook(lambda g: grads.append(g.detach().clone()))

    model.eval()
    with torch.no_grad():
        model.reset()
    t0 = time.time()
    import contextlib
    with warnings.catch_warnings(), (_FastHalfMM() if amp16 else contextlib.nullcontext()):
        warnings.simplefilter("ignore")
        ref.tpt.test_time_tuning(model, views, optimizer, scaler, args, reward_model=rm)
        t1 = time.time()
        with torch.no_grad():
            with torch.cuda.amp.autocast():            # (tpt_cls_rl.py:260-262; a disabled context on this box unless amp16)
                final = model(views[:1])
    t2 = time.time()
    torch.cuda.amp.autocast = orig_autocast
    final = final.float()
    ref.tpt.select_confident_samples = orig_select
    lg = taps["logits"].float()
    lp = lg.log_softmax(1)
    taps["clip_score"], taps["rewards"] = taps["clip_score"].float(), taps["rewards"].float()
    out = dict(
        logits=lg, entropy=-(lp.exp() * lp).sum(1), selected_idx=taps["selected_idx"],
        topk_idx=taps["topk_idx"].reshape(-1, hp["sample_k"]), clip_score=taps["clip_score"],
        rewards=taps["rewards"], ctx_grad=grads[0], ctx_after=model.prompt_learner.ctx.detach().clone(),
        final_logits=final, top5=torch.topk(final, min(5, n_cls), dim=-1).indices[0],
        ref_seconds=torch.tensor([t1 - t0, t2 - t1]),
    )
    if ensemble:
        for i in range(rm.n_model):
            out[f"reward_image_features_{i}"] = rm.image_features[i].clone()
            # (large banks: every 25th class row — the fixture stays small; the test compares the same rows)
            out[f"reward_class_features_{i}"] = rm.class_features[i].clone() if n_cls <= 64 else rm.class_features[i][::25].clone()
        out["reward_weights"] = torch.tensor(rm.weights)
    else:
        out.update(reward_image_features=rm.image_features.clone(), reward_class_features=rm.class_features.clone())
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


def run_reference_ln(ref, student, reward, n_views, n_cls, hp, view_seed=1000, n_ctx=4, only_norm=True, full_vectors=True,
                     prior_strength=None):
    """TPT/tune_cls_rl.py harness body (:206-227) around the reference's own CLIPCLS_TTA(only_norm=...) and
    test_time_tuning, with taps on the intermediates.  only_norm=False (the `--tune_norm 0` default, scripts/rlcf-tune.sh):
    every visual parameter is tuned; the gradient / adapted-parameter vectors are then stored per tensor as L2 norms
    (`vis_grad_l2`, `vis_delta_l2` = |after - pristine|) and, with full_vectors, as every 7th element of the concatenated vectors."""
    s_geo, r_geo = synth.GEOMETRIES[student], synth.GEOMETRIES[reward]
    s_sd = synth.make_state_dict(s_geo, seed=11)
    r_sd = synth.make_state_dict(r_geo, seed=23)
    install_models(ref, {student: (s_geo, s_sd)})
    bank = Bank(s_geo, n_cls, n_ctx)
    ref.custom.tokenize = bank.tokenize
    model = ref.custom.CLIPCLS_TTA("cpu", bank.classnames, arch=student, prompt_prefix="a_photo_of_a", only_visual=True,
                                   momentum_update=False, only_norm=only_norm)
    assert torch.equal(model.tokenized_prompts, bank.tokens)
    # ModifiedResNet student: `--prior_strength s` (s >= 0) swaps the BatchNorm forward exactly as tune_cls_rl.py:73-76 does
    bn_cls = torch.nn.BatchNorm2d
    bn_forward_orig = bn_cls.forward
    if prior_strength is not None and prior_strength >= 0:
        bn_cls.prior = float(prior_strength) / float(prior_strength + 1)
        bn_cls.forward = ref.tune._modified_bn_forward
    trainable = model.parameters()
    names = [n for n, p in model.clip_model.visual.named_parameters() if not only_norm or "ln" in n or "bn" in n]
    if not only_norm:
        assert ["visual." + n for n in names] == RR.visual_param_keys(s_sd), "oracle key order != named_parameters order"
    pristine = {n: p.detach().clone() for n, p in model.clip_model.visual.named_parameters()}
    optimizer = torch.optim.AdamW(trainable, hp["lr"], weight_decay=hp["weight_decay"])
    args = types.SimpleNamespace(tta_steps=hp["tta_steps"], selection_p=hp["selection_p"], min_entropy_reg=0, min_entropy_w=0.2,
                                 gpu=None, tpt=True)
    install_models(ref, {reward: (r_geo, r_sd)})
    rm = ref.clip_reward.CLIPRewards("cpu", arch=reward, classification=True, amplify_rewards=False, sample_k=hp["sample_k"],
                                     reward_process=True, process_batch=False)
    rm.set_class_features(tokenized_classes=model.tokenized_prompts)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        scaler = torch.cuda.amp.GradScaler(init_scale=1000)
    views = synth.make_views(view_seed, n_views, s_geo.image_resolution)
    taps = {}
    orig_select = ref.tpt.select_confident_samples

    def tap_select(logits, top):
        out, idx = orig_select(logits, top)
        if "logits" not in taps:
            taps["logits"] = logits.detach().clone()
            taps["selected_idx"] = idx.clone()
        return out, idx

    orig_score, orig_post = rm.CLIPScore, rm.rewards_post_process

    def tap_score(*a, **k):
        s = orig_score(*a, **k)
        if "clip_score" not in taps:
            taps["clip_score"] = s.detach().clone()
            taps["topk_idx"] = k["class_index"].clone()
        return s

    def tap_post(x):
        r = orig_post(x)
        taps.setdefault("rewards", r.detach().clone())
        return r

    ref.tpt.select_confident_samples = tap_select
    rm.CLIPScore, rm.rewards_post_process = tap_score, tap_post
    pmap = dict(model.clip_model.visual.named_parameters())
    first_grads = {}

    def keep_first(n):
        def hook(g):                         # must return None: a returned tensor would REPLACE the gradient
            first_grads.setdefault(n, g.detach().clone())
        return hook

    for n in names:
        pmap[n].register_hook(keep_first(n))
    model.reset()
    model.train()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref.tpt.test_time_tuning(model, views, optimizer, scaler, args, reward_model=rm)
    model.eval()
    with torch.no_grad():
        final = model(views[:1])
    ref.tpt.select_confident_samples = orig_select
    bn_cls.forward = bn_forward_orig
    bn_stats = None
    if s_geo.is_resnet:            # the running statistics the final inference used (train-mode passes update them in place)
        vsd = model.clip_model.visual.state_dict()
        bns = RR.visual_bn_stat_keys({"visual." + k: v for k, v in vsd.items()})
        bn_stats = torch.cat([torch.cat([vsd[b[len("visual."):] + ".running_mean"].reshape(-1), vsd[b[len("visual."):] + ".running_var"].reshape(-1)]) for b in bns])
        if only_norm:
            assert ["visual." + n for n in names] == RR.visual_bn_keys(s_sd), "oracle key order != named_parameters order"
    out = dict(logits=taps["logits"], selected_idx=taps["selected_idx"], topk_idx=taps["topk_idx"].reshape(-1, hp["sample_k"]),
               clip_score=taps["clip_score"], rewards=taps["rewards"],
               final_logits=final, top5=torch.topk(final, min(5, n_cls), dim=-1).indices[0])
    if bn_stats is not None:
        out["bn_stats_after"] = bn_stats
    if only_norm:
        out.update(ln_grad=torch.cat([first_grads[n].reshape(-1) for n in names]),
                   ln_after=torch.cat([pmap[n].detach().reshape(-1) for n in names]))
    elif full_vectors:      # every 7th element of the concatenated vectors (keeps the fixture small)
        out.update(vis_grad_sample=torch.cat([first_grads[n].reshape(-1) for n in names])[::7].clone(),
                   vis_after_sample=torch.cat([pmap[n].detach().reshape(-1) for n in names])[::7].clone())
    if not only_norm:
        out.update(vis_grad_l2=torch.stack([first_grads[n].double().norm() for n in names]).float(),
                   vis_delta_l2=torch.stack([(pmap[n].detach() - pristine[n]).double().norm() for n in names]).float())
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


def run_reference_ln_momentum(ref, student, reward, n_views, n_cls, hp, n_samples=3, n_ctx=4, only_norm=True):
    """TPT/tune_cls_rl.py:206-240 over several consecutive samples with CLIPCLS_TTA(momentum_update=True): per sample reset ->
    test_time_tuning -> clean-view logits -> momentum_update_model (custom_clip.py:460-475).  only_norm=False: every visual
    parameter tuned; the per-sample vectors are stored as per-tensor L2 norms of (value - checkpoint)."""
    import copy
    s_geo, r_geo = synth.GEOMETRIES[student], synth.GEOMETRIES[reward]
    s_sd = synth.make_state_dict(s_geo, seed=11)
    r_sd = synth.make_state_dict(r_geo, seed=23)
    install_models(ref, {student: (s_geo, s_sd)})
    bank = Bank(s_geo, n_cls, n_ctx)
    ref.custom.tokenize = bank.tokenize
    model = ref.custom.CLIPCLS_TTA("cpu", bank.classnames, arch=student, prompt_prefix="a_photo_of_a", only_visual=True,
                                   momentum_update=True, update_freq=hp["update_freq"], update_w=hp["update_w"],
                                   momentum=hp["momentum"], only_norm=only_norm)
    names = [n for n, p in model.clip_model.visual.named_parameters() if not only_norm or "ln" in n or "bn" in n]
    pristine = {n: p.detach().clone() for n, p in model.clip_model.visual.named_parameters()}
    optimizer = torch.optim.AdamW(model.parameters(), hp["lr"], weight_decay=hp["weight_decay"])
    optim_state = copy.deepcopy(optimizer.state_dict())
    args = types.SimpleNamespace(tta_steps=hp["tta_steps"], selection_p=hp["selection_p"], min_entropy_reg=0, min_entropy_w=0.2,
                                 gpu=None, tpt=True)
    install_models(ref, {reward: (r_geo, r_sd)})
    rm = ref.clip_reward.CLIPRewards("cpu", arch=reward, classification=True, amplify_rewards=False, sample_k=hp["sample_k"],
                                     reward_process=True, process_batch=False)
    rm.set_class_features(tokenized_classes=model.tokenized_prompts)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        scaler = torch.cuda.amp.GradScaler(init_scale=1000)
    out = {}
    for i in range(n_samples):
        views = synth.make_views(1000 + i, n_views, s_geo.image_resolution)
        model.reset()
        optimizer.load_state_dict(optim_state)
        model.train()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref.tpt.test_time_tuning(model, views, optimizer, scaler, args, reward_model=rm)
        model.eval()
        with torch.no_grad():
            out[f"final_logits_{i}"] = model(views[:1]).clone()
        pmap = dict(model.clip_model.visual.named_parameters())
        if only_norm:
            out[f"ln_after_{i}"] = torch.cat([pmap[n].detach().reshape(-1) for n in names]).clone()
        else:
            out[f"vis_delta_l2_{i}"] = torch.stack([(pmap[n].detach() - pristine[n]).double().norm() for n in names]).float()
        model.momentum_update_model()
        if only_norm:
            out[f"ln_reset_{i}"] = torch.cat([model.initial_state_dict[n].reshape(-1) for n in names]).clone()
        else:
            out[f"vis_reset_delta_l2_{i}"] = torch.stack([(model.initial_state_dict[n] - pristine[n]).double().norm() for n in names]).float()
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


TOKENIZER_STRINGS = [
    "a photo of a tench.", "a_photo_of_a great white shark.", "a photo of a hen-of-the-woods.", "itap of a toilet tissue.",
    "a bad photo of the CD player.", "a photo of a 3D-printed #42 T-shirt's logo!", "art of the  maillot   (tank suit).",
    "a photo of a", "X X X X goldfish.", "a origami Bernese mountain dog.", "a photo of the large jack-o'-lantern.",
    "a photo of a café au lait, naïve façade.", "graffiti of a potter's wheel.", "a photo of a person riding a horse & cart.",
    "a photo of 1 2 3 go-karts?", "the embroidered cock-a-doodle-doo.", "a photo of a \u4e2d\u6587 sign.", "<|startoftext|> nested <|endoftext|>",
    "a photo of a web site, website, internet site, site.", "a tattoo of the I'll've we're it's don't'd.",
]


def gen_tokenizer(ref):
    """clip.tokenize of the reference (TPT/clip/clip.py:197-233) on TOKENIZER_STRINGS."""
    toks = ref.cc.tokenize(TOKENIZER_STRINGS)
    return {"tokens": toks.numpy()}


VIS_CASES = {      # CLIPCLS_TTA(only_norm=False): name -> (student, reward, views, classes, overrides, store full vectors)
    "vis_tiny_s1": ("tiny", "tiny-r", 8, 16, dict(lr=1e-4), True),
    "vis_tiny_s3": ("tiny", "tiny-r", 8, 16, dict(lr=1e-4, tta_steps=3), False),
    "vis_tinyp6_s3": ("tiny-p6", "tiny-r", 8, 16, dict(lr=1e-4, tta_steps=3), False),        # padded conv1 columns (as ViT-L/14)
    "vis_small_s1": ("small", "small", 16, 40, dict(lr=1e-4, selection_p=0.25), False),
    "vis_b16_s3": ("ViT-B/16", "ViT-B/16", 8, 1000, dict(lr=1e-5, tta_steps=3, selection_p=0.25), False),   # rlcf-tune.sh: lr 1e-5, 3 steps
}
RNVIS_CASES = {    # ModifiedResNet student, CLIPCLS_TTA(only_norm=False) — the parser defaults of tune_cls_rl.py (`--arch RN50 --tune_norm 0`)
    "rnvis_tiny_s1": ("tiny-rn", "tiny-r", 8, 16, dict(lr=1e-4), True),
    "rnvis_tiny_s3": ("tiny-rn", "tiny-r", 8, 16, dict(lr=1e-4, tta_steps=3), False),
    "rnvis_rn50": ("RN50", "ViT-B/16", 16, 40, dict(lr=1e-5, selection_p=0.5, sample_k=6), False),
}
LN_CASES = {
    "ln_tiny_s1": ("tiny", "tiny-r", 8, 16, dict(lr=1e-3)),
    "ln_tiny_s3": ("tiny", "tiny-r", 8, 16, dict(lr=1e-3, tta_steps=3)),
    "ln_small_s1": ("small", "small", 16, 40, dict(lr=1e-3, selection_p=0.25)),
    "ln_b16_n8": ("ViT-B/16", "ViT-B/16", 8, 1000, dict(lr=1e-4)),
    "ln_l14_n8": ("ViT-L/14", "ViT-L/14", 8, 1000, dict(lr=1e-4)),          # BASELINE configs[2] geometry (N=8)
    "ln_l14_n64": ("ViT-L/14", "ViT-L/14", 64, 1000, dict(lr=1e-4, selection_p=0.1)),   # BASELINE configs[2] at FULL size: N=64 views, 6 selected
}


def save(name, arrays, meta):
    if "reward_class_features" in arrays and arrays["reward_class_features"].shape[0] > 64:
        arrays["reward_class_features"] = arrays["reward_class_features"][:64]     # 64 classes pin the reward bank; keeps the file small
    path = os.path.join(HERE, name + ".npz")
    tmp = path + ".tmp.npz"                     # (written next to the target and renamed: a reader never sees half a file)
    np.savez_compressed(tmp, **arrays, **{"meta_" + k: np.asarray(v) for k, v in meta.items()})
    os.replace(tmp, path)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KB)")


ENSEMBLE_NAMES = ["ViT-L/14@336px", "ViT-L/14", "ViT-B/16"]      # CONFIDECES 10, 5, 1 -> weights [0.62, 0.31, 0.06]

BN_CASES = {       # ModifiedResNet student, CLIPCLS_TTA(only_norm=True): name -> (student, reward, views, classes, overrides, prior_strength)
    "bn_tiny_train": ("tiny-rn", "tiny-r", 8, 16, dict(lr=1e-3), -1),
    "bn_tiny_train_s3": ("tiny-rn", "tiny-r", 8, 16, dict(lr=1e-3, tta_steps=3), -1),
    "bn_tiny_prior0": ("tiny-rn", "tiny-r", 8, 16, dict(lr=1e-3), 0),
    "bn_tiny_prior16_s3": ("tiny-rn", "tiny-r", 8, 16, dict(lr=1e-3, tta_steps=3), 16),
    # (RN50 geometry; K = 6 of 8 selected views: ~a quarter of the synthetic reward's scores are positive, the rest clamp to zero)
    "bn_rn50_train": ("RN50", "ViT-B/16", 16, 40, dict(lr=1e-4, selection_p=0.5, sample_k=6), -1),
    "bn_rn50_prior16": ("RN50", "ViT-B/16", 16, 40, dict(lr=1e-4, selection_p=0.5, sample_k=6), 16),
}
BASE_HP = dict(lr=7e-3, weight_decay=5e-4, sample_k=3, tta_steps=1, selection_p=0.5)

B16L14_SEED = int(os.environ.get("B16L14_SEED", "1000"))
RN_SEED = int(os.environ.get("RN_SEED", "1000"))
ENS_SEED = int(os.environ.get("ENS_SEED", "1000"))
TTA_CASES = {
    # name: (student, reward, N, C, hp overrides)
    "tta_tiny_s1": ("tiny", "tiny-r", 8, 16, {}),
    "tta_tiny_s3": ("tiny", "tiny-r", 8, 16, dict(tta_steps=3)),
    "tta_tiny_amplify": ("tiny", "tiny-r", 8, 16, dict(reward_amplify=True)),
    "tta_tiny_batchproc": ("tiny", "tiny-r", 8, 16, dict(process_batch=True)),
    "tta_tiny_minent": ("tiny", "tiny-r", 8, 16, dict(min_entropy_reg=1, min_entropy_w=0.2)),
    "tta_tiny_k1": ("tiny", "tiny-r", 8, 16, dict(sample_k=1, view_seed=1006)),
    "tta_small_s1": ("small", "small", 16, 40, dict(selection_p=0.25)),
    "tta_tiny_rres": ("tiny", "tiny-r64", 8, 16, dict(view_seed=1003)),     # reward resolution != view resolution: bicubic resample
    # reward ensemble: three reward CLIPs (one at another resolution), weighted sum / plain mean of the clamped scores
    "tta_tiny_ens": ("tiny", "tiny-r64+tiny-r+tiny-r", 8, 16, dict(reward_seeds="23+29+31", view_seed=ENS_SEED)),
    "tta_tiny_ensmean": ("tiny", "tiny-r64+tiny-r+tiny-r", 8, 16, dict(reward_seeds="23+29+31", weighted_scores=0, view_seed=ENS_SEED)),
    # the arch list get_reward_model really uses (clip_reward.py:31): L/14@336px, RN50x64, L/14 -> weights [0.56, 0.17, 0.28];
    # the middle member is a ModifiedResNet at twice the view resolution
    "tta_tiny_ensrn": ("tiny", "tiny-r64+tiny-rn+tiny-r", 8, 16, dict(reward_seeds="23+29+31", view_seed=ENS_SEED,
                                                                      reward_archs="ViT-L/14@336px+RN50x64+ViT-L/14")),
    # ModifiedResNet towers: as the reward model (with the bicubic resolution change) and as the frozen student image encoder
    "tta_tiny_rnreward": ("tiny", "tiny-rn", 8, 16, dict(view_seed=RN_SEED)),
    "tta_tiny_rnstudent": ("tiny-rn32", "tiny-r", 8, 16, dict(view_seed=RN_SEED)),
    # class tokens not at the end of the prompt (PromptLearner.forward 'front' / 'middle', '[CLS]' inside ctx_init: custom_clip.py:92-97,239-284)
    "tta_tiny_front": ("tiny", "tiny-r", 8, 16, dict(ctx_position="front")),
    "tta_tiny_middle": ("tiny", "tiny-r", 8, 16, dict(ctx_position="middle", tta_steps=2)),
    "tta_tiny_cls1": ("tiny", "tiny-r", 8, 16, dict(ctx_init="a_[CLS]_photo_of_a")),
    "tta_b16_n8": ("ViT-B/16", "ViT-B/16", 8, 1000, {}),
    # the setting of TPT/scripts/rlcf-prompt.sh: ViT-B/16 student, ViT-L/14 reward model, 3 tuning steps
    "tta_b16_rl14_s3": ("ViT-B/16", "ViT-L/14", 8, 1000, dict(tta_steps=3, view_seed=B16L14_SEED)),
    # view seed chosen (tools/find_seed.py) so that two views get non-zero CLIP rewards: a non-trivial gradient
    "tta_b16_n64": ("ViT-B/16", "ViT-B/16", 64, 1000, dict(selection_p=0.1, view_seed=1113)),
    # the paper's strongest reward setting at full size: the arch list get_reward_model really uses (clip_reward.py:31) — ViT-L/14@336px,
    # RN50x64 (448^2), ViT-L/14 — scoring the 224^2 views of a ViT-B/16 student through the bicubic align_corners=True upsample
    "tta_b16_ensfull_n64": ("ViT-B/16", "ViT-L/14@336px+RN50x64+ViT-L/14", 64, 1000,
                            dict(selection_p=0.1, reward_seeds="23+29+31", reward_archs="ViT-L/14@336px+RN50x64+ViT-L/14", view_seed=1113)),
    # BASELINE configs[4] at FULL geometry: RN50x64 student (448^2 views) + ViT-L/14 reward (bicubic 448 -> 224), N=32; 200 classes (the
    # reference's autograd tape over the 1024-wide text tower of 1000 x 77 tokens does not fit the build container's 62 GB)
    "tta_rn50x64_l14_n32": ("RN50x64", "ViT-L/14", 32, 200, dict(selection_p=0.1)),
}
GROUPS = {
    "tiny": [k for k in TTA_CASES if k.startswith("tta_tiny") and k != "tta_tiny_rres" and "ens" not in k and "_rn" not in k
             and k not in ("tta_tiny_front", "tta_tiny_middle", "tta_tiny_cls1")],
    "rn": ["tta_tiny_ensrn", "tta_tiny_rnreward", "tta_tiny_rnstudent"],
    "rres": ["tta_tiny_rres"],
    "ctxpos": ["tta_tiny_front", "tta_tiny_middle", "tta_tiny_cls1"],
    "ens": ["tta_tiny_ens", "tta_tiny_ensmean"],
    "small": ["tta_small_s1"],
    "b16n8": ["tta_b16_n8"],
    "b16l14": ["tta_b16_rl14_s3"],
    "b16n64": ["tta_b16_n64"],
    "cfg5": ["tta_rn50x64_l14_n32"],
    "ensfull": ["tta_b16_ensfull_n64"],
}


def gen_ops(ref):
    """G0: op-level vectors from the reference's own modules / functions."""
    torch.manual_seed(0)
    out = {}
    x = synth.normal(3, "ops.x", (5, 3, 128))
    ln = ref.refmodel.LayerNorm(128)
    ln.weight.data = synth.normal(3, "ops.lnw", (128,), 0.1, 1.0)
    ln.bias.data = synth.normal(3, "ops.lnb", (128,), 0.05)
    out["ln_y"] = ln(x).detach()
    out["gelu_y"] = ref.refmodel.QuickGELU()(x)
    geo = synth.GEOMETRIES["tiny"]
    sd = synth.make_state_dict(geo, seed=5)
    for masked in (False, True):
        L = 9
        mask = torch.full((L, L), float("-inf")).triu_(1) if masked else None
        blk = ref.refmodel.ResidualAttentionBlock(128, 2, mask)
        blk.load_state_dict({k[len("transformer.resblocks.0."):]: v for k, v in sd.items()
                             if k.startswith("transformer.resblocks.0.")})
        xb = synth.normal(4, "ops.blk", (L, 3, 128)).requires_grad_(True)      # LND
        y = blk(xb)
        gy = synth.normal(4, "ops.blk.g", (L, 3, 128))
        (gx,) = torch.autograd.grad((y * gy).sum(), xb)
        out[f"block_y_{int(masked)}"] = y.detach()
        out[f"block_gx_{int(masked)}"] = gx
    lg = synth.normal(6, "ops.logits", (16, 50), 3.0)
    for p in (0.1, 0.25, 0.5, 0.05):
        sel, idx = ref.tpt.select_confident_samples(lg, p)
        out[f"select_idx_{p}"] = idx
    out["avg_entropy"] = ref.tpt.avg_entropy(lg[:4])
    # AdamW vs torch.optim.AdamW, 3 steps with weight decay
    p = torch.nn.Parameter(synth.normal(8, "ops.p", (4, 64), 0.02))
    opt = torch.optim.AdamW([p], 7e-3, weight_decay=5e-4)
    for s in range(3):
        p.grad = synth.normal(8, f"ops.g{s}", (4, 64), 1e-3)
        opt.step()
        out[f"adamw_p{s + 1}"] = p.detach().clone()
    # reward post-processing, all switch combinations incl. the K==1 guard
    sc = synth.normal(9, "ops.score", (4, 3), 0.3, 0.5).clamp_min(0)
    for amp in (False, True):
        for pb in (False, True):
            rm = types.SimpleNamespace(reward_process=True, amplify_rewards=amp)
            out[f"rewards_amp{int(amp)}_pb{int(pb)}"] = ref.clip_reward.CLIPRewards.rewards_post_process(
                rm, sc.flatten() if pb else sc)
    rm = types.SimpleNamespace(reward_process=True, amplify_rewards=True)
    out["rewards_k1"] = ref.clip_reward.CLIPRewards.rewards_post_process(rm, sc[:, :1])
    out["rewards_in"] = sc
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


def gen_modules(ref):
    """G4: encode_image / encode_text of the reference CLIP class at full geometries."""
    out = {}
    for arch, tag in (("ViT-B/16", "b16"), ("ViT-L/14", "l14")):
        geo = synth.GEOMETRIES[arch]
        sd = synth.make_state_dict(geo, seed=11)
        m = ref.refmodel.CLIP(*geo.as_tuple())
        m.load_state_dict(sd)
        m = m.eval().float()
        views = synth.make_views(1000, 2, geo.image_resolution)
        toks = synth.make_token_bank(geo, 8, seed=7)
        with torch.no_grad():
            out[f"{tag}_image"] = m.encode_image(views)
            out[f"{tag}_text"] = m.encode_text(toks)
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


def gen_modules_rn(ref):
    """encode_image of the reference CLIP class with ModifiedResNet towers (TPT/clip/model.py:94-154): a reduced geometry, RN50 at
    224^2 and the reward model RN50x64 at 448^2."""
    out = {}
    for arch, tag, nv in (("tiny-rn", "tinyrn", 3), ("RN50", "rn50", 2), ("RN50x64", "rn50x64", 1)):
        geo = synth.GEOMETRIES[arch]
        sd = synth.make_state_dict(geo, seed=11)
        m = ref.refmodel.CLIP(*geo.as_tuple())
        m.load_state_dict(sd)
        m = m.eval().float()
        views = synth.make_views(1000, nv, geo.image_resolution)
        with torch.no_grad():
            out[f"{tag}_image"] = m.encode_image(views)
        print(arch, "done", flush=True)
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="tiny,small,ops")
    a = ap.parse_args()
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    ref = import_reference()
    for grp in a.only.split(","):
        if grp == "ops":
            save("ops", gen_ops(ref), {})
        elif grp == "modules":
            save("modules", gen_modules(ref), {})
        elif grp == "modules_rn":
            save("modules_rn", gen_modules_rn(ref), {})
        elif grp == "tokenizer":
            save("tokenizer", gen_tokenizer(ref), {})
        elif grp == "lnmom":
            hp = dict(BASE_HP, lr=1e-3, update_freq=2, update_w=0.5, momentum=0.9)
            arrays = run_reference_ln_momentum(ref, "tiny", "tiny-r", 8, 16, hp)
            save("ln_tiny_momentum", arrays, dict(student="tiny", reward="tiny-r", n_views=8, n_cls=16, student_seed=11, reward_seed=23,
                                                   bank_seed=7, n_ctx=4, n_samples=3, **hp))
        elif grp == "bnmom":
            # ModifiedResNet student with momentum_update: the EMA runs over the whole visual state dict (BatchNorm weights / biases AND buffers)
            hp = dict(BASE_HP, lr=1e-3, update_freq=2, update_w=0.5, momentum=0.9)
            arrays = run_reference_ln_momentum(ref, "tiny-rn", "tiny-r", 8, 16, hp)
            save("bn_tiny_momentum", arrays, dict(student="tiny-rn", reward="tiny-r", n_views=8, n_cls=16, student_seed=11, reward_seed=23,
                                                   bank_seed=7, n_ctx=4, n_samples=3, prior_strength=-1, **hp))
        elif grp == "vismom":
            hp = dict(BASE_HP, lr=1e-4, update_freq=2, update_w=0.5, momentum=0.9)
            arrays = run_reference_ln_momentum(ref, "tiny", "tiny-r", 8, 16, hp, only_norm=False)
            save("vis_tiny_momentum", arrays, dict(student="tiny", reward="tiny-r", n_views=8, n_cls=16, student_seed=11, reward_seed=23,
                                                    bank_seed=7, n_ctx=4, n_samples=3, only_norm=0, **hp))
        elif grp in ("vis", "visb16"):
            for name in ([k for k in VIS_CASES if "b16" not in k] if grp == "vis" else ["vis_b16_s3"]):
                student, reward, n, c, over, full = VIS_CASES[name]
                hp = dict(BASE_HP, **over)
                t0 = time.time()
                arrays = run_reference_ln(ref, student, reward, n, c, hp, only_norm=False, full_vectors=full)
                meta = dict(student=student, reward=reward, n_views=n, n_cls=c, student_seed=11, reward_seed=23, view_seed=1000,
                            bank_seed=7, n_ctx=4, only_norm=0, **hp)
                save(name, arrays, meta)
                print(f"  {name}: {time.time() - t0:.1f}s idx={arrays['selected_idx']} top5={arrays['top5']} "
                      f"|g|={np.linalg.norm(arrays['vis_grad_l2']):.3e} |d|={np.linalg.norm(arrays['vis_delta_l2']):.3e}")
        elif grp == "onlyvisual":
            # CHECK, no fixture: CLIPCLS_TTA(only_visual=False) is computationally the only_visual=True model (parameters() ignores the flag,
            # custom_clip.py:477-485; class features are cached under no_grad) — the reference's own runs must reproduce the committed
            # fixtures bit for bit.  rlcf_amd.custom_clip.CLIPCLS_TTA therefore serves only_visual=False on the same path.
            orig_cls = ref.custom.CLIPCLS_TTA

            class _TextToo(orig_cls):
                def __init__(self, *a, **k):
                    k["only_visual"] = False
                    super().__init__(*a, **k)
            ref.custom.CLIPCLS_TTA = _TextToo
            try:
                for name, kw, keys in (("ln_tiny_s1", dict(), ("logits", "ln_grad", "ln_after", "final_logits")),
                                       ("vis_tiny_s1", dict(only_norm=False), ("logits", "vis_grad_l2", "vis_delta_l2", "final_logits"))):
                    z = np.load(os.path.join(HERE, name + ".npz"))
                    meta = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
                    hp = dict(BASE_HP, lr=meta["lr"], tta_steps=meta["tta_steps"], selection_p=meta["selection_p"], sample_k=meta["sample_k"])
                    arrays = run_reference_ln(ref, meta["student"], meta["reward"], meta["n_views"], meta["n_cls"], hp, **kw)
                    for k in keys:
                        assert np.array_equal(arrays[k], z[k]), (name, k)
                    print(f"  only_visual=False == {name} (bit-identical: {', '.join(keys)})")
            finally:
                ref.custom.CLIPCLS_TTA = orig_cls
        elif grp in ("rnvis", "rnvisrn50"):
            for name in [k for k in RNVIS_CASES if ("rn50" in k) == (grp == "rnvisrn50")]:
                student, reward, n, c, over, full = RNVIS_CASES[name]
                hp = dict(BASE_HP, **over)
                t0 = time.time()
                arrays = run_reference_ln(ref, student, reward, n, c, hp, only_norm=False, full_vectors=full)
                meta = dict(student=student, reward=reward, n_views=n, n_cls=c, student_seed=11, reward_seed=23, view_seed=1000,
                            bank_seed=7, n_ctx=4, only_norm=0, prior_strength=-1, **hp)
                save(name, arrays, meta)
                print(f"  {name}: {time.time() - t0:.1f}s idx={arrays['selected_idx']} top5={arrays['top5']} "
                      f"|g|={np.linalg.norm(arrays['vis_grad_l2']):.3e} |d|={np.linalg.norm(arrays['vis_delta_l2']):.3e}")
        elif grp in ("bn", "bnrn50"):
            for name in [k for k in BN_CASES if ("rn50" in k) == (grp == "bnrn50")]:
                student, reward, n, c, over, ps = BN_CASES[name]
                hp = dict(BASE_HP, **over)
                t0 = time.time()
                arrays = run_reference_ln(ref, student, reward, n, c, hp, prior_strength=ps)
                meta = dict(student=student, reward=reward, n_views=n, n_cls=c, student_seed=11, reward_seed=23, view_seed=1000,
                            bank_seed=7, n_ctx=4, prior_strength=ps, **hp)
                save(name, arrays, meta)
                print(f"  {name}: {time.time() - t0:.1f}s idx={arrays['selected_idx']} top5={arrays['top5']} |g|={np.linalg.norm(arrays['ln_grad']):.3e}")
        elif grp in ("ln", "lnb16", "lnl14", "lnl14n64"):
            for name in ([k for k in LN_CASES if "b16" not in k and "l14" not in k] if grp == "ln" else ["ln_b16_n8"] if grp == "lnb16" else
                         ["ln_l14_n8"] if grp == "lnl14" else ["ln_l14_n64"]):
                student, reward, n, c, over = LN_CASES[name]
                hp = dict(BASE_HP, **over)
                arrays = run_reference_ln(ref, student, reward, n, c, hp)
                meta = dict(student=student, reward=reward, n_views=n, n_cls=c, student_seed=11, reward_seed=23, view_seed=1000,
                            bank_seed=7, n_ctx=4, **hp)
                save(name, arrays, meta)
                print(f"  {name}: idx={arrays['selected_idx']} top5={arrays['top5']} |g|={np.linalg.norm(arrays['ln_grad']):.3e}")
        elif grp == "lnl14stream":
            # BASELINE configs[2] at full size as a short STREAM: four consecutive test images (view seeds 1000..1003) through the harness
            # body of TPT/tune_cls_rl.py:206-227, one at a time (sample 0 is the ln_l14_n64 fixture itself); pins rlcf_tta_batch_ln on more
            # than one full-size sample.  ~15 min of CPU per sample here.
            hp = dict(BASE_HP, **LN_CASES["ln_l14_n64"][4])
            student, reward, n, c, _ = LN_CASES["ln_l14_n64"]
            arrays, n_s = {}, int(os.environ.get("STREAM_N", "4"))
            z0 = np.load(os.path.join(HERE, "ln_l14_n64.npz"))
            for i in range(n_s):
                t0 = time.time()
                if i == 0:
                    a_i = {k: z0[k] for k in z0.files if not k.startswith("meta_")}
                else:
                    a_i = run_reference_ln(ref, student, reward, n, c, dict(hp), view_seed=1000 + i)
                for k in ("selected_idx", "topk_idx", "clip_score", "rewards", "final_logits", "top5", "ln_grad"):
                    arrays[f"{k}_{i}"] = np.asarray(a_i[k])
                print(f"  ln stream sample {i}: {time.time() - t0:.1f}s idx={arrays[f'selected_idx_{i}']} top5={arrays[f'top5_{i}']}", flush=True)
            save("ln_l14_n64_stream", arrays, dict(student=student, reward=reward, n_views=n, n_cls=c, student_seed=11, reward_seed=23,
                                                   view_seed0=1000, n_samples=n_s, bank_seed=7, n_ctx=4, **hp))
        elif grp == "b16stream":
            # BASELINE configs[1] as a STREAM: eight consecutive test images (view seeds 1113..1120) through the harness body
            # TPT/tpt_cls_rl.py:251-262 one at a time (reset -> test_time_tuning -> clean-view logits); pins rlcf_tta_batch at the
            # images-per-pass counts bench.py runs (8 and 32)
            hp = dict(BASE_HP, selection_p=0.1)
            arrays, n_s = {}, int(os.environ.get("STREAM_N", "8"))
            for i in range(n_s):
                t0 = time.time()
                a_i = run_reference_tta(ref, "ViT-B/16", "ViT-B/16", 64, 1000, dict(hp), view_seed=1113 + i)
                for k in ("selected_idx", "topk_idx", "clip_score", "rewards", "ctx_after", "final_logits", "top5"):
                    arrays[f"{k}_{i}"] = a_i[k]
                print(f"  stream sample {i}: {time.time() - t0:.1f}s idx={a_i['selected_idx']} top5={a_i['top5']} "
                      f"rewards={a_i['rewards']}", flush=True)
            save("tta_b16_n64_stream", arrays, dict(student="ViT-B/16", reward="ViT-B/16", n_views=64, n_cls=1000, student_seed=11,
                                                    reward_seed=23, view_seed0=1113, n_samples=n_s, bank_seed=7, n_ctx=4, **hp))
        elif grp == "b16stream_fp16":
            # the b16stream case in the arithmetic of the reference's GPU path: the reference's own code under fp16 autocast with an enabled
            # GradScaler (see run_reference_tta, fp16_autocast).  Pins what RLCF_PREC_F16 is a performance mode OF: the test reports the
            # engine's distance to this run next to its distance to the float32 run (tests/test_gpu_round2.py).
            hp = dict(BASE_HP, selection_p=0.1)
            arrays, n_s = {}, int(os.environ.get("STREAM_N", "32"))
            for i in range(n_s):
                t0 = time.time()
                a_i = run_reference_tta(ref, "ViT-B/16", "ViT-B/16", 64, 1000, dict(hp, fp16_autocast=1), view_seed=1113 + i)
                for k in ("selected_idx", "topk_idx", "clip_score", "rewards", "ctx_after", "final_logits", "top5", "entropy"):
                    arrays[f"{k}_{i}"] = a_i[k]
                print(f"  fp16 stream sample {i}: {time.time() - t0:.1f}s idx={a_i['selected_idx']} top5={a_i['top5']} "
                      f"rewards={a_i['rewards']}", flush=True)
                save("tta_b16_n64_stream_fp16ref", arrays, dict(student="ViT-B/16", reward="ViT-B/16", n_views=64, n_cls=1000, student_seed=11,
                                                                reward_seed=23, view_seed0=1113, n_samples=i + 1, bank_seed=7, n_ctx=4,
                                                                arithmetic="torch.autocast(cpu, float16) + GradScaler(init_scale=1000)", **hp))
        elif grp == "b16gridstream":
            # the b16stream case on CHECKPOINT-GRID weights (every GEMM weight an fp16 value, synth.to_fp16_grid): the reference's own run on
            # the weights for which the engine drops the a_hi . w_lo pass — pins the two-pass products to the reference directly
            hp = dict(BASE_HP, selection_p=0.1)
            arrays, n_s = {}, int(os.environ.get("STREAM_N", "4"))
            for i in range(n_s):
                t0 = time.time()
                a_i = run_reference_tta(ref, "ViT-B/16", "ViT-B/16", 64, 1000, dict(hp, fp16_grid=1), view_seed=1113 + i)
                for k in ("selected_idx", "topk_idx", "clip_score", "rewards", "ctx_after", "final_logits", "top5"):
                    arrays[f"{k}_{i}"] = a_i[k]
                print(f"  grid stream sample {i}: {time.time() - t0:.1f}s idx={a_i['selected_idx']} top5={a_i['top5']} "
                      f"rewards={a_i['rewards']}", flush=True)
            save("tta_b16_n64_grid_stream", arrays, dict(student="ViT-B/16", reward="ViT-B/16", n_views=64, n_cls=1000, student_seed=11,
                                                         reward_seed=23, view_seed0=1113, n_samples=n_s, bank_seed=7, n_ctx=4, weights="fp16grid", **hp))
        else:
            for name in GROUPS[grp]:
                student, reward, n, c, over = TTA_CASES[name]
                hp = dict(BASE_HP, **over)
                vseed = hp.pop("view_seed", 1000)
                t0 = time.time()
                arrays = run_reference_tta(ref, student, reward, n, c, hp, view_seed=vseed)
                meta = dict(student=student, reward=reward, n_views=n, n_cls=c, student_seed=11, reward_seed=23,
                            view_seed=vseed, bank_seed=7, n_ctx=4, **hp)
                save(name, arrays, meta)
                print(f"  {name}: {time.time() - t0:.1f}s  idx={arrays['selected_idx']}  top5={arrays['top5']} "
                      f"score={arrays['clip_score']} |g|={np.linalg.norm(arrays['ctx_grad']):.3e}")


if __name__ == "__main__":
    main()
