"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path
(`rlcf_amd/`); only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may use it, and only as the checker / the timed CPU baseline.

CPU fp32 restatement (plain torch ops, functional style over an OpenAI-layout
state dict) of the CLIP modules the RLCF hot path drives.  Each function cites
the reference lines it restates (paths relative to /root/reference).

Parity pin: the reference ships no tests or golden vectors, so
this restatement is pinned against outputs of the reference itself, imported in
the build container by `tests/golden/make_golden.py`; the resulting fixtures
are committed under `tests/golden/` and checked by `tests/test_oracle_golden.py`.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
HEAD_DIM = 64          # every CLIP tower uses width // 64 heads (TPT/clip/model.py:272,421)
LN_EPS = 1e-5          # nn.LayerNorm default, used by TPT/clip/model.py:157-163


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """TPT/clip/model.py:157-163 — LayerNorm evaluated in fp32, biased variance."""
    x = x.float()
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + LN_EPS) * w + b


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """TPT/clip/model.py:166-168."""
    return x * torch.sigmoid(1.702 * x)


def causal_mask(length: int) -> torch.Tensor:
    """TPT/clip/model.py:328-334 — additive mask, -inf strictly above the diagonal."""
    m = torch.full((length, length), float("-inf"))
    return torch.triu(m, diagonal=1)


def multi_head_attention(x: torch.Tensor, sd: SD, p: str, mask: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.MultiheadAttention as used at TPT/clip/model.py:175,185-187 (packed
    in-proj rows [q;k;v], scale 1/sqrt(head_dim), additive mask, softmax over
    keys, out-proj).  x is batch-first here: [B, L, W]."""
    B, L, W = x.shape
    H = W // HEAD_DIM
    qkv = x @ sd[p + "attn.in_proj_weight"].t() + sd[p + "attn.in_proj_bias"]
    q, k, v = qkv.split(W, dim=-1)
    q = q.reshape(B, L, H, HEAD_DIM).transpose(1, 2)
    k = k.reshape(B, L, H, HEAD_DIM).transpose(1, 2)
    v = v.reshape(B, L, H, HEAD_DIM).transpose(1, 2)
    s = (q * (HEAD_DIM ** -0.5)) @ k.transpose(-1, -2)
    if mask is not None:
        s = s + mask
    a = torch.softmax(s, dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, L, W)
    return a @ sd[p + "attn.out_proj.weight"].t() + sd[p + "attn.out_proj.bias"]


def residual_block(x: torch.Tensor, sd: SD, p: str, mask: Optional[torch.Tensor]) -> torch.Tensor:
    """TPT/clip/model.py:189-192 — pre-LN block."""
    x = x + multi_head_attention(layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]), sd, p, mask)
    h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
    h = quick_gelu(h @ sd[p + "mlp.c_fc.weight"].t() + sd[p + "mlp.c_fc.bias"])
    return x + (h @ sd[p + "mlp.c_proj.weight"].t() + sd[p + "mlp.c_proj.bias"])


def n_blocks(sd: SD, prefix: str) -> int:
    return len([k for k in sd if k.startswith(prefix + ".resblocks.") and k.endswith("attn.in_proj_weight")])


def transformer(x: torch.Tensor, sd: SD, prefix: str, mask: Optional[torch.Tensor]) -> torch.Tensor:
    """TPT/clip/model.py:195-203."""
    for i in range(n_blocks(sd, prefix)):
        x = residual_block(x, sd, f"{prefix}.resblocks.{i}.", mask)
    return x


class BNMode:
    """How the BatchNorm2d layers of a ModifiedResNet normalise (a ResNet STUDENT whose norm layers are tuned,
    TPT/tune_cls_rl.py:35-44,73-76 + CLIPCLS_TTA.train, custom_clip.py:487-497):
      mode 'eval'   running statistics (inference; the only mode of reward models and of a frozen student);
      mode 'train'  torch's train-mode BatchNorm (`--prior_strength` < 0, the parser default): batch statistics over (N, H, W) with
                    the gradient flowing through them, running statistics updated in place with momentum 0.1 (unbiased variance);
      mode 'prior'  `_modified_bn_forward` (`--prior_strength` s >= 0, prior = s / (s + 1)): batch mean and UNBIASED batch variance
                    are blended into the running statistics, prior * running + (1 - prior) * batch, as detached constants; the
                    module's running statistics stay untouched.
    `stats` collects {bn name: (running_mean, running_var)} as they stand after the pass (mode 'train' only changes them)."""

    def __init__(self, mode: str = "eval", prior: float = 0.0):
        self.mode, self.prior, self.stats = mode, prior, {}


_BN = BNMode()


def set_bn_mode(m: Optional["BNMode"]) -> "BNMode":
    """install a BatchNorm mode for the following encode_image_resnet calls; returns the previous one"""
    global _BN
    prev, _BN = _BN, (m if m is not None else BNMode())
    return prev


def _batch_norm(sd: SD, x: torch.Tensor, bn: str) -> torch.Tensor:
    rm, rv, w, b = sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"], sd[bn + ".bias"]
    rm, rv = _BN.stats.get(bn, (rm, rv))
    if _BN.mode == "train":                                   # nn.BatchNorm2d.forward in training mode, momentum 0.1
        rm, rv = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)
        _BN.stats[bn] = (rm, rv)
        return y
    if _BN.mode == "prior":                                   # tune_cls_rl.py:35-44
        est_mean, est_var = torch.zeros_like(rm), torch.ones_like(rv)
        F.batch_norm(x, est_mean, est_var, None, None, True, 1.0, 1e-5)
        rm2 = _BN.prior * rm + (1 - _BN.prior) * est_mean
        rv2 = _BN.prior * rv + (1 - _BN.prior) * est_var
        return F.batch_norm(x, rm2, rv2, w, b, False, 0, 1e-5)
    return F.batch_norm(x, rm, rv, w, b, False, 0.0, 1e-5)


def _conv_bn(sd: SD, x: torch.Tensor, conv: str, bn: str, stride: int = 1, relu: bool = True) -> torch.Tensor:
    """bias-free Conv2d (kernel 1 or 3, padding k//2) followed by BatchNorm2d (eps 1e-5; eval mode = running statistics unless a
    BNMode is installed) and an optional ReLU: the conv/bn/relu triples of TPT/clip/model.py:18-31,108-116."""
    w = sd[conv + ".weight"]
    x = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2)
    x = _batch_norm(sd, x, bn)
    return F.relu(x) if relu else x


def _bottleneck(sd: SD, x: torch.Tensor, p: str, stride: int) -> torch.Tensor:
    """Bottleneck.forward, TPT/clip/model.py:42-55: 1x1 -> 3x3 -> (avgpool when stride > 1) -> 1x1 (x4), every conv at stride 1;
    the identity branch is avgpool + 1x1 conv + bn when the block changes resolution or width; ReLU after the sum."""
    out = _conv_bn(sd, x, p + "conv1", p + "bn1")
    out = _conv_bn(sd, out, p + "conv2", p + "bn2")
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = _conv_bn(sd, out, p + "conv3", p + "bn3", relu=False)
    if (p + "downsample.0.weight") in sd:
        idn = F.avg_pool2d(x, stride) if stride > 1 else x
        idn = _conv_bn(sd, idn, p + "downsample.0", p + "downsample.1", relu=False)
    else:
        idn = x
    return F.relu(out + idn)


def _attention_pool(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """AttentionPool2d.forward, TPT/clip/model.py:68-91: tokens = [mean over positions; positions] + positional embedding;
    one multi-head attention whose only query is the mean token (separate q/k/v projections with bias, head_dim 64,
    q scaled by 1/8 after its bias); output projection c_proj."""
    n, c, h, w = x.shape
    t = x.flatten(2).transpose(1, 2)                                    # [n, HW, C]
    t = torch.cat([t.mean(dim=1, keepdim=True), t], dim=1) + sd["visual.attnpool.positional_embedding"]
    heads = c // 64
    q = F.linear(t[:, :1], sd["visual.attnpool.q_proj.weight"], sd["visual.attnpool.q_proj.bias"]) * (64 ** -0.5)
    k = F.linear(t, sd["visual.attnpool.k_proj.weight"], sd["visual.attnpool.k_proj.bias"])
    v = F.linear(t, sd["visual.attnpool.v_proj.weight"], sd["visual.attnpool.v_proj.bias"])
    q = q.view(n, 1, heads, 64).transpose(1, 2)
    k = k.view(n, -1, heads, 64).transpose(1, 2)
    v = v.view(n, -1, heads, 64).transpose(1, 2)
    o = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(n, c)
    return F.linear(o, sd["visual.attnpool.c_proj.weight"], sd["visual.attnpool.c_proj.bias"])


def encode_image_resnet(sd: SD, images: torch.Tensor) -> torch.Tensor:
    """ModifiedResNet.forward, TPT/clip/model.py:138-154 (eval mode: BatchNorm uses its running statistics)."""
    x = _conv_bn(sd, images.float(), "visual.conv1", "visual.bn1", stride=2)
    x = _conv_bn(sd, x, "visual.conv2", "visual.bn2")
    x = _conv_bn(sd, x, "visual.conv3", "visual.bn3")
    x = F.avg_pool2d(x, 2)
    for li in (1, 2, 3, 4):
        nb = len({k.split(".")[2] for k in sd if k.startswith(f"visual.layer{li}.")})
        for b in range(nb):
            x = _bottleneck(sd, x, f"visual.layer{li}.{b}.", 2 if (b == 0 and li > 1) else 1)
    return _attention_pool(sd, x)


def encode_image(sd: SD, images: torch.Tensor) -> torch.Tensor:
    """VisionTransformer.forward, TPT/clip/model.py:223-240 (== CLIP.encode_image
    :340-341).  The stride==kernel convolution is written as patch gather + GEMM.
    A state dict without ``visual.proj`` holds a ModifiedResNet (build_model, :399-412)."""
    if "visual.proj" not in sd:
        return encode_image_resnet(sd, images)
    w = sd["visual.conv1.weight"]
    width, _, ps, _ = w.shape
    n = images.shape[0]
    patches = F.unfold(images.float(), kernel_size=ps, stride=ps)        # [N, 3*ps*ps, G*G], (c,i,j) major
    x = patches.transpose(1, 2) @ w.reshape(width, -1).t()               # [N, G*G, width]
    cls = sd["visual.class_embedding"].expand(n, 1, width)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    x = transformer(x, sd, "visual.transformer", None)
    x = layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    return x @ sd["visual.proj"]


def text_tower(sd: SD, x: torch.Tensor, eot: torch.Tensor) -> torch.Tensor:
    """TextEncoder.forward, TPT/clip/custom_clip.py:62-73 (== the tail of
    CLIP.encode_text, TPT/clip/model.py:346-354).  x: token/prompt embeddings
    [C, L, W] WITHOUT positional embedding; eot: int64 [C] row index of EOT."""
    L = x.shape[1]
    x = x + sd["positional_embedding"][:L]
    x = transformer(x, sd, "transformer", causal_mask(L))
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    return x[torch.arange(x.shape[0]), eot] @ sd["text_projection"]


def encode_text(sd: SD, tokens: torch.Tensor, truncate: bool = False) -> torch.Tensor:
    """CLIP.encode_text, TPT/clip/model.py:343-356.  EOT = argmax of the ids.
    ``truncate`` crops to max(EOT)+1 — exact under the causal mask (SURVEY.md §0
    fact 4); default is the dense 77-token reference graph."""
    eot = tokens.argmax(dim=-1)
    if truncate:
        tokens = tokens[:, : int(eot.max()) + 1]
    return text_tower(sd, sd["token_embedding.weight"][tokens], eot)


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """x / x.norm(dim=-1, keepdim=True) — TPT/clip/custom_clip.py:320,330; clip_reward.py:136,148."""
    return x / x.norm(dim=-1, keepdim=True)


def prompt_embeddings(sd: SD, tokens: torch.Tensor, ctx: torch.Tensor, position: str = "end", split_idx=None) -> torch.Tensor:
    """PromptLearner.forward with a 2-D ctx, TPT/clip/custom_clip.py:198-289.  'end' (:211-238): [SOS | ctx | class tokens, '.', EOS,
    pad]; 'middle' (:239-264): [SOS | ctx[:half] | class | ctx[half:] | '.', EOS, pad] with half = split_idx (the place of '[CLS]' in
    ctx_init, :92-97) or n_ctx // 2; 'front' (:266-284): [SOS | class | ctx | '.', EOS, pad].  name_len = the tokens between the
    context words and the final '.' of the tokenised prompt (:127)."""
    emb = sd["token_embedding.weight"][tokens]
    n_ctx = ctx.shape[0]
    c = tokens.shape[0]
    if position == "end":
        return torch.cat([emb[:, :1], ctx.unsqueeze(0).expand(c, -1, -1), emb[:, 1 + n_ctx:]], dim=1)
    half = (split_idx if split_idx is not None else n_ctx // 2) if position == "middle" else 0
    eot = tokens.argmax(dim=-1)
    rows = []
    for i in range(c):
        nl = int(eot[i]) - 1 - n_ctx - 1
        suffix = emb[i, 1 + n_ctx:]
        rows.append(torch.cat([emb[i, :1], ctx[:half], suffix[:nl], ctx[half:], suffix[nl:]], dim=0))
    return torch.stack(rows)


def ctx_from_tokens(sd: SD, ctx_token_ids) -> torch.Tensor:
    """ctx_init words -> token embeddings, TPT/clip/custom_clip.py:90-107."""
    ids = torch.as_tensor(list(ctx_token_ids), dtype=torch.int64)
    return sd["token_embedding.weight"][ids].clone()


def student_text_features(sd: SD, tokens: torch.Tensor, ctx: torch.Tensor, truncate: bool = False, position: str = "end",
                          split_idx=None) -> torch.Tensor:
    """ClipTestTimeTuning.get_text_features, TPT/clip/custom_clip.py:315-323
    (the stack+mean over a one-element list is the identity)."""
    eot = tokens.argmax(dim=-1)
    x = prompt_embeddings(sd, tokens, ctx, position, split_idx)
    if truncate:
        x = x[:, : int(eot.max()) + 1]
    return l2_normalize(text_tower(sd, x, eot))


def student_logits(sd: SD, images: torch.Tensor, tokens: torch.Tensor, ctx: torch.Tensor,
                   truncate: bool = False, position: str = "end", split_idx=None) -> torch.Tensor:
    """ClipTestTimeTuning.inference, TPT/clip/custom_clip.py:325-335: image tower
    under no_grad, text tower differentiable w.r.t. ctx."""
    with torch.no_grad():
        img = l2_normalize(encode_image(sd, images))
    txt = student_text_features(sd, tokens, ctx, truncate, position, split_idx)
    return sd["logit_scale"].exp() * img @ txt.t()
