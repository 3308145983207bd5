"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/clip_ref.py header; same rules,
same parity pin: fixtures generated from the imported reference by
tests/golden/make_golden.py).

CPU fp32 restatement of the RLCF per-sample test-time-adaptation step:
TPT/tpt_cls_rl.py:32-79 (select / entropy / tuning loop), :251-262 (reset, final
inference), TPT/clip_reward.py:111-165 (CLIPScore, reward post-processing) and
the torch.optim.AdamW update the harness applies (TPT/tpt_cls_rl.py:120).
This is the dense reference graph: 77-token text tower, autograd backward.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import clip_ref as C


@dataclass
class TTAHyper:
    """Flags read on the path (TPT/params.py:13-98; script values
    TPT/scripts/rlcf-prompt.sh:13-41)."""
    selection_p: float = 0.1
    tta_steps: int = 1
    sample_k: int = 3
    lr: float = 7e-3
    weight_decay: float = 5e-4
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    reward_process: bool = True
    process_batch: bool = False
    reward_amplify: bool = False
    clipscore_weight: float = 2.5
    min_entropy_reg: bool = False
    min_entropy_w: float = 0.2
    # PromptLearner class-token position (custom_clip.py:198-289): 'end' (every RLCF script), 'middle' (split_idx: where '[CLS]' stood in
    # ctx_init, None = n_ctx // 2), 'front'
    ctx_position: str = "end"
    split_idx: Optional[int] = None
    # reward ensemble (CLIPRewardsMultiple, TPT/clip_reward.py:180-257): per-model weights round(w/sum(w),2), or the plain mean
    reward_weights: Optional[tuple] = None
    weighted_scores: bool = True


def entropy_rows(logits: torch.Tensor) -> torch.Tensor:
    """-(softmax * log_softmax).sum(1), TPT/tpt_cls_rl.py:33."""
    lp = logits.log_softmax(dim=1)
    return -(lp.exp() * lp).sum(dim=1)


def select_confident_samples(logits: torch.Tensor, top: float):
    """TPT/tpt_cls_rl.py:32-35 — lowest-entropy int(N*top) rows, ascending."""
    ent = entropy_rows(logits)
    idx = torch.argsort(ent, descending=False)[: int(ent.shape[0] * top)]
    return logits[idx], idx


def avg_entropy(outputs: torch.Tensor) -> torch.Tensor:
    """TPT/tpt_cls_rl.py:38-44 — entropy of the view-averaged distribution."""
    lp = outputs - outputs.logsumexp(dim=-1, keepdim=True)
    avg = lp.logsumexp(dim=0) - math.log(lp.shape[0])
    avg = torch.clamp(avg, min=torch.finfo(avg.dtype).min)
    return -(avg * avg.exp()).sum(dim=-1)


def clip_score(class_features: torch.Tensor, image_features: torch.Tensor, class_index: torch.Tensor,
               sample_k: int, weight: float = 2.5) -> torch.Tensor:
    """CLIPRewards.CLIPScore(pairwise=False), TPT/clip_reward.py:111-128."""
    t = class_features[class_index]
    i = torch.repeat_interleave(image_features, sample_k, dim=0)
    s = weight * (t * i).sum(dim=-1)
    return torch.clamp_min(s, 0.0).squeeze()


def clip_score_any(reward_cls, rimg, class_index, hp: "TTAHyper") -> torch.Tensor:
    """CLIPScore of one reward model, or of the ensemble (CLIPRewardsMultiple.CLIPScore, TPT/clip_reward.py:227-257):
    per-model clamped scores stacked, then weighted sum over models (weighted_scores) or their mean."""
    if not isinstance(reward_cls, (list, tuple)):
        return clip_score(reward_cls, rimg, class_index, hp.sample_k, hp.clipscore_weight)
    scores = torch.stack([clip_score(c, i, class_index, hp.sample_k, hp.clipscore_weight) for c, i in zip(reward_cls, rimg)], dim=0)
    if hp.weighted_scores:
        w = torch.tensor(hp.reward_weights, dtype=scores.dtype).unsqueeze(1)
        return torch.sum(w * scores, dim=0)
    return torch.mean(scores, dim=0)


def rewards_post_process(score: torch.Tensor, reward_process: bool, amplify: bool) -> torch.Tensor:
    """CLIPRewards.rewards_post_process, TPT/clip_reward.py:152-165 (torch.std is
    the unbiased estimator)."""
    if score.shape[-1] > 1 and reward_process:
        mean = score.mean(dim=-1, keepdim=True)
        std = score.std(dim=-1, keepdim=True) + 1e-5 if amplify else 1.0
        score = (score - mean) / std
    return score.flatten()


def adamw_step(p, g, m, v, step: int, hp: TTAHyper):
    """torch.optim.AdamW (amsgrad=False, maximize=False) single-tensor update."""
    p = p * (1.0 - hp.lr * hp.weight_decay)
    m = hp.beta1 * m + (1.0 - hp.beta1) * g
    v = hp.beta2 * v + (1.0 - hp.beta2) * g * g
    bc1 = 1.0 - hp.beta1 ** step
    bc2 = 1.0 - hp.beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + hp.eps
    p = p - (hp.lr / bc1) * (m / denom)
    return p, m, v


def reward_image_features(reward_sd, images: torch.Tensor):
    if isinstance(reward_sd, (list, tuple)):                # one feature matrix per ensemble member (clip_reward.py:259-273)
        return [reward_image_features(sd, images) for sd in reward_sd]
    return _reward_image_features(reward_sd, images)


def _reward_image_features(reward_sd, images: torch.Tensor) -> torch.Tensor:
    """CLIPRewards.extract_image_features, TPT/clip_reward.py:130-137: bicubic (align_corners=True) resample to the reward
    model's input resolution when it differs, encode_image, float, L2 normalise."""
    if "visual.proj" in reward_sd:
        ps = reward_sd["visual.conv1.weight"].shape[-1]
        res = ps * round((reward_sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    else:                                                   # ModifiedResNet: build_model, TPT/clip/model.py:408-412
        res = 32 * round((reward_sd["visual.attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
    if images.shape[-1] != res:
        images = torch.nn.functional.interpolate(images, size=res, mode="bicubic", align_corners=True)
    return C.l2_normalize(C.encode_image(reward_sd, images).float())


def reward_class_features(reward_sd, tokens: torch.Tensor, truncate: bool = False):
    if isinstance(reward_sd, (list, tuple)):                # clip_reward.py:275-291
        return [reward_class_features(sd, tokens, truncate) for sd in reward_sd]
    return _reward_class_features(reward_sd, tokens, truncate)


def _reward_class_features(reward_sd, tokens: torch.Tensor, truncate: bool = False) -> torch.Tensor:
    """BaseRewards.set_class_features -> extract_text_features(tokenized_cap=...),
    TPT/clip_reward.py:55-57,139-150; called once per dataset (tpt_cls_rl.py:182-183)."""
    with torch.no_grad():
        return C.l2_normalize(C.encode_text(reward_sd, tokens, truncate).float())


def tta_sample(student_sd, reward_sd, views: torch.Tensor, tokens: torch.Tensor, ctx_init: torch.Tensor,
               hp: TTAHyper, reward_cls: Optional[torch.Tensor] = None, truncate: bool = False
               ) -> Dict[str, torch.Tensor]:
    """One iteration of the harness loop TPT/tpt_cls_rl.py:251-262: reset ->
    test_time_tuning (:47-79) -> final one-view inference on views[0].
    Returns every intermediate of the first tuning step plus the final outputs."""
    out: Dict[str, torch.Tensor] = {}
    if reward_cls is None:
        reward_cls = reward_class_features(reward_sd, tokens, truncate)
    ctx = ctx_init.clone()                                  # model.reset(), custom_clip.py:161-164
    m = torch.zeros_like(ctx)                               # optimizer.load_state_dict(optim_state), :255
    v = torch.zeros_like(ctx)
    selected = None
    for j in range(hp.tta_steps):
        ctx = ctx.detach().requires_grad_(True)
        if selected is None:                                # tpt_cls_rl.py:57-59
            logits_all = C.student_logits(student_sd, views, tokens, ctx, truncate, hp.ctx_position, hp.split_idx)
            output, selected = select_confident_samples(logits_all, hp.selection_p)
            with torch.no_grad():                           # clip_reward.py:130-137
                rimg = reward_image_features(reward_sd, views[selected])
        else:                                               # tpt_cls_rl.py:55
            logits_all = None
            output = C.student_logits(student_sd, views[selected], tokens, ctx, truncate, hp.ctx_position, hp.split_idx)
        bs = output.shape[0]
        _, index = torch.topk(output, hp.sample_k, dim=-1)  # :63
        flat = index.flatten()
        score = clip_score_any(reward_cls, rimg, flat, hp)
        rewards = rewards_post_process(score if hp.process_batch else score.reshape(bs, -1),
                                       hp.reward_process, hp.reward_amplify)
        rep = torch.repeat_interleave(output, hp.sample_k, dim=0)
        ce = torch.nn.functional.cross_entropy(rep, flat, reduction="none")
        loss = torch.mean(rewards * ce)                     # :69-71
        if hp.min_entropy_reg:                              # :73-74
            loss = loss + hp.min_entropy_w * avg_entropy(output)
        grad, dlogits = torch.autograd.grad(loss, [ctx, output])
        if j == 0:
            out.update(logits=logits_all.detach(), entropy=entropy_rows(logits_all.detach()),
                       selected_idx=selected.clone(), topk_idx=index.clone(), clip_score=score.detach().clone(),
                       rewards=rewards.detach().clone(), loss=loss.detach().clone(), ctx_grad=grad.clone(),
                       dlogits=dlogits.clone(),
                       reward_image_features=rimg[0].clone() if isinstance(rimg, list) else rimg.clone())
        with torch.no_grad():
            new_ctx, m, v = adamw_step(ctx.detach(), grad, m, v, j + 1, hp)
        ctx = new_ctx
        out[f"ctx_after_step{j + 1}"] = ctx.detach().clone()
    with torch.no_grad():                                   # tpt_cls_rl.py:260-262
        final = C.student_logits(student_sd, views[:1], tokens, ctx.detach(), truncate, hp.ctx_position, hp.split_idx)
    out["ctx_after"] = ctx.detach().clone()
    out["final_logits"] = final
    out["top5"] = torch.topk(final, min(5, final.shape[1]), dim=-1).indices[0]
    return out


# --------------------------------------------------------------------------------------------------
# LayerNorm / backbone tuning variant (BASELINE configs[2]): TPT/tune_cls_rl.py with CLIPCLS_TTA(only_norm=True)

def visual_ln_keys(sd):
    """CLIPCLS_TTA.parameters() with only_norm: visual parameters whose name contains 'ln'
    (TPT/clip/custom_clip.py:477-485), in named_parameters() order: ln_pre, per block ln_1, ln_2, ln_post."""
    n = C.n_blocks(sd, "visual.transformer")
    keys = ["visual.ln_pre.weight", "visual.ln_pre.bias"]
    for i in range(n):
        p = f"visual.transformer.resblocks.{i}."
        keys += [p + "ln_1.weight", p + "ln_1.bias", p + "ln_2.weight", p + "ln_2.bias"]
    return keys + ["visual.ln_post.weight", "visual.ln_post.bias"]


def is_resnet_sd(sd) -> bool:
    return "visual.layer1.0.conv1.weight" in sd


def visual_bn_keys(sd):
    """CLIPCLS_TTA.parameters() with only_norm for a ModifiedResNet (custom_clip.py:481-485): visual parameters whose name contains
    'bn', in named_parameters() order (stem bn1..3, then bn1, bn2, bn3 of every Bottleneck).  `downsample.1` — a BatchNorm too — is
    MISSED by the substring test and stays frozen, as in the reference."""
    keys = []
    for i in (1, 2, 3):
        keys += [f"visual.bn{i}.weight", f"visual.bn{i}.bias"]
    for li in (1, 2, 3, 4):
        nb = len({k.split(".")[2] for k in sd if k.startswith(f"visual.layer{li}.")})
        for b in range(nb):
            for i in (1, 2, 3):
                keys += [f"visual.layer{li}.{b}.bn{i}.weight", f"visual.layer{li}.{b}.bn{i}.bias"]
    return keys


def visual_bn_stat_keys(sd):
    """every BatchNorm2d of the image tower in execution order (stem bn1..3; per Bottleneck bn1, bn2, bn3 and, where the block has
    one, downsample.1): the layers whose running statistics a train-mode pass updates"""
    names = [f"visual.bn{i}" for i in (1, 2, 3)]
    for li in (1, 2, 3, 4):
        nb = len({k.split(".")[2] for k in sd if k.startswith(f"visual.layer{li}.")})
        for b in range(nb):
            p = f"visual.layer{li}.{b}."
            names += [p + "bn1", p + "bn2", p + "bn3"]
            if (p + "downsample.1.running_mean") in sd:
                names.append(p + "downsample.1")
    return names


def visual_param_keys(sd):
    """CLIPCLS_TTA.parameters() with only_norm=False (custom_clip.py:477-479): every parameter of clip_model.visual, in
    named_parameters() order of VisionTransformer (model.py:206-221: direct parameters first, then conv1, ln_pre, the
    resblocks, ln_post)."""
    if is_resnet_sd(sd):
        # ModifiedResNet (model.py:94-154): stem conv / bn 1..3, per Bottleneck conv1 bn1 conv2 bn2 conv3 bn3 [downsample.0 downsample.1],
        # then AttentionPool2d (model.py:58-66: positional_embedding, k_proj, q_proj, v_proj, c_proj)
        keys = []
        for i in (1, 2, 3):
            keys += [f"visual.conv{i}.weight", f"visual.bn{i}.weight", f"visual.bn{i}.bias"]
        for li in (1, 2, 3, 4):
            nb = len({k.split(".")[2] for k in sd if k.startswith(f"visual.layer{li}.")})
            for b in range(nb):
                p = f"visual.layer{li}.{b}."
                for i in (1, 2, 3):
                    keys += [p + f"conv{i}.weight", p + f"bn{i}.weight", p + f"bn{i}.bias"]
                if (p + "downsample.0.weight") in sd:
                    keys += [p + "downsample.0.weight", p + "downsample.1.weight", p + "downsample.1.bias"]
        keys.append("visual.attnpool.positional_embedding")
        for nm in ("k_proj", "q_proj", "v_proj", "c_proj"):
            keys += [f"visual.attnpool.{nm}.weight", f"visual.attnpool.{nm}.bias"]
        return keys
    n = C.n_blocks(sd, "visual.transformer")
    keys = ["visual.class_embedding", "visual.positional_embedding", "visual.proj", "visual.conv1.weight",
            "visual.ln_pre.weight", "visual.ln_pre.bias"]
    for i in range(n):
        p = f"visual.transformer.resblocks.{i}."
        keys += [p + "attn.in_proj_weight", p + "attn.in_proj_bias", p + "attn.out_proj.weight", p + "attn.out_proj.bias",
                 p + "ln_1.weight", p + "ln_1.bias", p + "mlp.c_fc.weight", p + "mlp.c_fc.bias", p + "mlp.c_proj.weight",
                 p + "mlp.c_proj.bias", p + "ln_2.weight", p + "ln_2.bias"]
    return keys + ["visual.ln_post.weight", "visual.ln_post.bias"]


def momentum_update(mom: torch.Tensor, cur: torch.Tensor, clip: torch.Tensor, momentum: float, update_w: float, apply: bool):
    """CLIPCLS_TTA.momentum_update_model on the tunable tensors (custom_clip.py:460-475): -> (new momentum state, new reset
    state or None).  float32 elementwise arithmetic as torch evaluates `m * a + (1.0 - m) * b`."""
    mom = momentum * mom + (1.0 - momentum) * cur
    return mom, ((1 - update_w) * clip + update_w * mom if apply else None)


def tta_sample_ln(student_sd, reward_sd, views: torch.Tensor, tokens: torch.Tensor, hp: TTAHyper,
                  reward_cls: Optional[torch.Tensor] = None, ln_init: Optional[torch.Tensor] = None,
                  only_norm: bool = True, prior_strength: int = -1) -> Dict[str, torch.Tensor]:
    """One iteration of the harness loop TPT/tune_cls_rl.py:183-256 with model = CLIPCLS_TTA(only_visual=True,
    only_norm=...): reset visual state -> test_time_tuning (tpt_cls_rl.py:47-79; the image encoder runs WITH grad,
    custom_clip.py:423-432; class text features are cached, :405-409) -> final clean-view inference.
    only_norm=False (the default of `--tune_norm`, params.py:73, what scripts/rlcf-tune.sh runs) tunes every visual parameter;
    `ln_grad` / `ln_after` / `ln_init` then hold all of them, concatenated in visual_param_keys order.
    A ModifiedResNet student with only_norm=False (the parser defaults `--arch RN50 --tune_norm 0`): every convolution, BatchNorm
    (downsample.1 included) and attention-pool tensor is tuned; CLIPCLS_TTA.train(mode) then is plain nn.Module.train(mode)
    (custom_clip.py:487-497 touches the norm layers only under only_norm), so the tuning passes run the BatchNorms in train mode and the
    final clean-view inference, after model.eval(), in EVAL mode on the running statistics those passes left behind.
    A ModifiedResNet student (only_norm): the tuned tensors are the BatchNorm weights / biases of visual_bn_keys; the tuning passes run
    the BatchNorm layers in train mode (prior_strength < 0, the parser default: batch statistics, running statistics updated in place
    and USED by the final clean-view inference) or through `_modified_bn_forward` (prior_strength >= 0, tune_cls_rl.py:35-44,73-76);
    `bn_stats_after` = the running statistics (mean | var per layer, visual_bn_stat_keys order) the final inference used."""
    out: Dict[str, torch.Tensor] = {}
    rn = is_resnet_sd(student_sd)
    if reward_cls is None:
        reward_cls = reward_class_features(reward_sd, tokens)
    with torch.no_grad():
        cls_feat = C.l2_normalize(C.encode_text(student_sd, tokens))            # get_class_features, custom_clip.py:405-409
    keys = (visual_bn_keys(student_sd) if rn else visual_ln_keys(student_sd)) if only_norm else visual_param_keys(student_sd)
    bn_train = C.BNMode("prior", prior_strength / (prior_strength + 1.0)) if prior_strength >= 0 else C.BNMode("train")
    params = {k: student_sd[k].clone() for k in keys}                           # model.reset(): pristine visual state
    if ln_init is not None:                                                     # ... or the momentum-updated initial_state_dict
        off = 0
        for k in keys:
            n = params[k].numel()
            params[k] = ln_init[off: off + n].reshape(params[k].shape).clone()
            off += n
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v2 = {k: torch.zeros_like(v) for k, v in params.items()}
    scale = student_sd["logit_scale"].exp()

    def logits_of(x, prm, train=True):
        sd = dict(student_sd)
        sd.update(prm)
        if not rn:
            return scale * C.l2_normalize(C.encode_image(sd, x)) @ cls_feat.t()
        # model.train() around test_time_tuning, model.eval() for the final inference (tune_cls_rl.py:216-218); the running statistics
        # a train-mode pass leaves behind carry over to the following passes
        mode = bn_train if train else C.BNMode("eval")
        mode.stats = bn_train.stats
        prev = C.set_bn_mode(mode)
        try:
            return scale * C.l2_normalize(C.encode_image(sd, x)) @ cls_feat.t()
        finally:
            C.set_bn_mode(prev)

    selected = None
    for j in range(hp.tta_steps):
        prm = {k: p.detach().requires_grad_(True) for k, p in params.items()}
        if selected is None:
            logits_all = logits_of(views, prm)
            output, selected = select_confident_samples(logits_all, hp.selection_p)
            with torch.no_grad():
                rimg = reward_image_features(reward_sd, views[selected])
        else:
            logits_all = None
            output = logits_of(views[selected], prm)
        bs = output.shape[0]
        _, index = torch.topk(output, hp.sample_k, dim=-1)
        flat = index.flatten()
        score = clip_score_any(reward_cls, rimg, flat, hp)
        rewards = rewards_post_process(score if hp.process_batch else score.reshape(bs, -1), hp.reward_process, hp.reward_amplify)
        rep = torch.repeat_interleave(output, hp.sample_k, dim=0)
        loss = torch.mean(rewards * torch.nn.functional.cross_entropy(rep, flat, reduction="none"))
        if hp.min_entropy_reg:
            loss = loss + hp.min_entropy_w * avg_entropy(output)
        grads = torch.autograd.grad(loss, [prm[k] for k in keys])
        if j == 0:
            out.update(logits=logits_all.detach(), selected_idx=selected.clone(), topk_idx=index.clone(),
                       clip_score=score.detach().clone(), rewards=rewards.detach().clone(), loss=loss.detach().clone(),
                       ln_grad=torch.cat([g.reshape(-1) for g in grads]))
        with torch.no_grad():
            for k, g in zip(keys, grads):
                params[k], m[k], v2[k] = adamw_step(prm[k].detach(), g, m[k], v2[k], j + 1, hp)
    with torch.no_grad():
        # QUIRK of the reference, reproduced: CLIPCLS_TTA.train(mode) with only_norm (custom_clip.py:487-497) calls m.train() on every
        # LayerNorm / BatchNorm2d whatever `mode` is, so model.eval() leaves the BatchNorm layers of a ResNet student in TRAINING
        # mode: the final clean-view inference normalises with the statistics of that ONE image (train mode; blended into the running
        # statistics under `--prior_strength`) and, in train mode, updates the running statistics once more.
        final = logits_of(views[:1], params, train=only_norm)
    if rn:
        st = [bn_train.stats.get(b, (student_sd[b + ".running_mean"], student_sd[b + ".running_var"])) for b in visual_bn_stat_keys(student_sd)]
        out["bn_stats_after"] = torch.cat([torch.cat([a.reshape(-1), b.reshape(-1)]) for a, b in st])
    out["ln_after"] = torch.cat([params[k].reshape(-1) for k in keys])
    out["final_logits"] = final
    out["top5"] = torch.topk(final, min(5, final.shape[1]), dim=-1).indices[0]
    return out
