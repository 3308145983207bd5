"""Seeded synthetic CLIP weights, views and class-token banks.

Everything here is produced by a counter-based integer hash evaluated with torch
int64 ops, so the same (seed, name, index) gives bit-identical fp32 values on the
CPU (this container, oracle + golden fixtures) and on the GPU box (HIP path) with
no dependence on torch's own generators.  Weight statistics follow
``CLIP.initialize_parameters`` (reference TPT/clip/model.py:299-326); the key
layout is the OpenAI state-dict layout accepted by ``build_model``
(TPT/clip/model.py:399-436, SURVEY.md §8 a-W).
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass
from typing import Dict, Optional, Tuple, Union

import torch

_M32 = 0xFFFFFFFF


@dataclass(frozen=True)
class ClipGeometry:
    """Constructor arguments of the reference ``CLIP`` class, same order
    (TPT/clip/model.py:244-257).  ``vision_layers`` is an int for a VisionTransformer image tower and a 4-tuple of
    block counts for a ModifiedResNet (then ``vision_patch_size`` is None), as in the reference (:262-270)."""
    embed_dim: int
    image_resolution: int
    vision_layers: Union[int, Tuple[int, int, int, int]]
    vision_width: int
    vision_patch_size: Optional[int]
    context_length: int
    vocab_size: int
    transformer_width: int
    transformer_heads: int
    transformer_layers: int

    @property
    def is_resnet(self) -> bool:
        return isinstance(self.vision_layers, (tuple, list))

    @property
    def vision_heads(self) -> int:
        return self.vision_width * 32 // 64 if self.is_resnet else self.vision_width // 64

    @property
    def grid(self) -> int:
        return self.image_resolution // (32 if self.is_resnet else self.vision_patch_size)

    @property
    def vision_tokens(self) -> int:
        return self.grid * self.grid + 1

    def as_tuple(self) -> Tuple[int, ...]:
        return (self.embed_dim, self.image_resolution, self.vision_layers, self.vision_width,
                self.vision_patch_size, self.context_length, self.vocab_size,
                self.transformer_width, self.transformer_heads, self.transformer_layers)


GEOMETRIES: Dict[str, ClipGeometry] = {
    # the OpenAI checkpoints the reference loads (TPT/clip/clip.py:30-40)
    "ViT-B/16": ClipGeometry(512, 224, 12, 768, 16, 77, 49408, 512, 8, 12),
    "ViT-B/32": ClipGeometry(512, 224, 12, 768, 32, 77, 49408, 512, 8, 12),
    "ViT-L/14": ClipGeometry(768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12),
    "ViT-L/14@336px": ClipGeometry(768, 336, 24, 1024, 14, 77, 49408, 768, 12, 12),
    "RN50": ClipGeometry(1024, 224, (3, 4, 6, 3), 64, None, 77, 49408, 512, 8, 12),
    "RN101": ClipGeometry(512, 224, (3, 4, 23, 3), 64, None, 77, 49408, 512, 8, 12),
    "RN50x4": ClipGeometry(640, 288, (4, 6, 10, 6), 80, None, 77, 49408, 640, 10, 12),
    "RN50x16": ClipGeometry(768, 384, (6, 8, 18, 8), 96, None, 77, 49408, 768, 12, 12),
    "RN50x64": ClipGeometry(1024, 448, (3, 15, 36, 10), 128, None, 77, 49408, 1024, 16, 12),
    # reduced geometries for tests (head_dim stays 64 as in every CLIP)
    "tiny": ClipGeometry(128, 32, 2, 128, 8, 77, 1024, 128, 2, 2),
    "tiny-r": ClipGeometry(64, 32, 2, 128, 8, 77, 1024, 64, 1, 2),
    "tiny-r64": ClipGeometry(64, 64, 2, 128, 16, 77, 1024, 64, 1, 2),     # reward model at twice the view resolution (bicubic path)
    "tiny-p6": ClipGeometry(128, 30, 2, 128, 6, 77, 1024, 128, 2, 2),     # 3*6*6 = 108 patch columns: zero-padded to the GEMM's K granule like ViT-L/14's 588
    "small": ClipGeometry(256, 64, 4, 256, 16, 77, 4096, 256, 4, 4),
    # ModifiedResNet image towers (attention-pool width 32*width, head_dim 64): 64x64 input -> 2x2 map -> 5 pooled tokens
    "tiny-rn": ClipGeometry(64, 64, (1, 2, 1, 1), 16, None, 77, 1024, 64, 1, 2),
    "tiny-rn32": ClipGeometry(128, 32, (2, 1, 1, 1), 16, None, 77, 1024, 128, 2, 2),      # as a student: 32x32 views, 1x1 map
}


def _hash32(x: torch.Tensor) -> torch.Tensor:
    """lowbias32-style avalanche on int64 tensors holding 32-bit values."""
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def _keys(seed: int, name: str) -> Tuple[int, int]:
    k1 = zlib.crc32(name.encode()) & _M32
    k2 = (zlib.crc32((name + "#").encode()) ^ (seed * 0x9E3779B1)) & _M32
    return k1, k2


def raw_u32(seed: int, name: str, n: int, lane: int, device="cpu") -> torch.Tensor:
    """n 32-bit words of stream (seed, name, lane) as int64."""
    k1, k2 = _keys(seed, name)
    c = torch.arange(n, dtype=torch.int64, device=device) * 4 + lane
    h = _hash32((c + k1 * 0x9E3779B1) & _M32)
    return _hash32(h ^ k2)


_IH_STD = math.sqrt(8.0 * (65536.0 ** 2 - 1.0) / 12.0)


def normal(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0,
           device="cpu") -> torch.Tensor:
    """Approximately normal fp32 tensor: Irwin-Hall sum of eight 16-bit uniforms,
    exact in integer arithmetic, then one fp64 scale and one fp32 rounding."""
    if str(device) == "meta":            # shapes only (loader tests of the 400-600 M parameter geometries)
        return torch.empty(*shape, dtype=torch.float32, device="meta")
    n = 1
    for s in shape:
        n *= int(s)
    acc = torch.zeros(n, dtype=torch.int64, device=device)
    for lane in range(4):
        w = raw_u32(seed, name, n, lane, device)
        acc += (w & 0xFFFF) + (w >> 16)
    z = (acc.to(torch.float64) - 8 * 32767.5) / _IH_STD
    return (z * std + mean).to(torch.float32).reshape(*shape)


def randint(seed: int, name: str, n: int, low: int, high: int, device="cpu") -> torch.Tensor:
    return low + raw_u32(seed, name, n, 0, device) % (high - low)


# --------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------

def _block(sd, seed, prefix, width, layers, attn_std, proj_std, fc_std, device):
    for i in range(layers):
        p = f"{prefix}.resblocks.{i}."
        sd[p + "attn.in_proj_weight"] = normal(seed, p + "in_w", (3 * width, width), attn_std, device=device)
        sd[p + "attn.in_proj_bias"] = normal(seed, p + "in_b", (3 * width,), 0.02, device=device)
        sd[p + "attn.out_proj.weight"] = normal(seed, p + "out_w", (width, width), proj_std, device=device)
        sd[p + "attn.out_proj.bias"] = normal(seed, p + "out_b", (width,), 0.02, device=device)
        sd[p + "ln_1.weight"] = normal(seed, p + "ln1_w", (width,), 0.1, 1.0, device=device)
        sd[p + "ln_1.bias"] = normal(seed, p + "ln1_b", (width,), 0.05, device=device)
        sd[p + "mlp.c_fc.weight"] = normal(seed, p + "fc_w", (4 * width, width), fc_std, device=device)
        sd[p + "mlp.c_fc.bias"] = normal(seed, p + "fc_b", (4 * width,), 0.02, device=device)
        sd[p + "mlp.c_proj.weight"] = normal(seed, p + "proj_w", (width, 4 * width), proj_std, device=device)
        sd[p + "mlp.c_proj.bias"] = normal(seed, p + "proj_b", (width,), 0.02, device=device)
        sd[p + "ln_2.weight"] = normal(seed, p + "ln2_w", (width,), 0.1, 1.0, device=device)
        sd[p + "ln_2.bias"] = normal(seed, p + "ln2_b", (width,), 0.05, device=device)


def make_state_dict(geo: ClipGeometry, seed: int, device="cpu",
                    logit_scale: float = math.log(100.0)) -> Dict[str, torch.Tensor]:
    """fp32 state dict in the OpenAI CLIP key layout.  LayerNorm gains/biases and
    linear biases are non-trivial on purpose so every epilogue is exercised.
    ``logit_scale`` defaults to ln(100), the value of the released checkpoints
    (a random-init CLIP has ln(1/0.07))."""
    sd: Dict[str, torch.Tensor] = {}
    vw, tw = geo.vision_width, geo.transformer_width
    if geo.is_resnet:
        _resnet(sd, seed, geo, device)
        _text(sd, seed, geo, device, logit_scale)
        return sd
    ps = geo.vision_patch_size
    sc = vw ** -0.5
    sd["visual.conv1.weight"] = normal(seed, "v.conv1", (vw, 3, ps, ps), (3 * ps * ps) ** -0.5, device=device)
    sd["visual.class_embedding"] = normal(seed, "v.cls", (vw,), sc, device=device)
    sd["visual.positional_embedding"] = normal(seed, "v.pos", (geo.vision_tokens, vw), sc, device=device)
    sd["visual.ln_pre.weight"] = normal(seed, "v.lnpre_w", (vw,), 0.1, 1.0, device=device)
    sd["visual.ln_pre.bias"] = normal(seed, "v.lnpre_b", (vw,), 0.05, device=device)
    v_proj_std = (vw ** -0.5) * ((2 * geo.vision_layers) ** -0.5)
    _block(sd, seed, "visual.transformer", vw, geo.vision_layers, vw ** -0.5, v_proj_std,
           (2 * vw) ** -0.5, device)
    sd["visual.ln_post.weight"] = normal(seed, "v.lnpost_w", (vw,), 0.1, 1.0, device=device)
    sd["visual.ln_post.bias"] = normal(seed, "v.lnpost_b", (vw,), 0.05, device=device)
    sd["visual.proj"] = normal(seed, "v.proj", (vw, geo.embed_dim), sc, device=device)
    _text(sd, seed, geo, device, logit_scale)
    return sd


def to_fp16_grid(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The state dict with the tensors a RELEASED CLIP checkpoint stores as fp16 rounded to fp16 values (still float32 tensors): what
    the reference actually holds after `clip.load` — the archives carry the weights of every Conv / Linear / MultiheadAttention and the
    two projections as fp16 (convert_weights, TPT/clip/model.py:375-396, applied before they were saved), and build_model copies them
    into float32 parameters (model.py:399-436, the re-conversion at :435 commented out).  LayerNorm parameters, embeddings and
    logit_scale stay as they are.  The engine recognises such weights at finalize (their split-f16 lo halves are zero) and drops the
    a_hi . w_lo pass of their products — bit for bit the same results in two MFMA passes instead of three."""
    out = {}
    for k, v in sd.items():
        grid = v.is_floating_point() and (
            k.endswith(("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias", "c_fc.weight", "c_fc.bias", "c_proj.weight",
                        "c_proj.bias", "conv1.weight", "conv2.weight", "conv3.weight", "q_proj.weight", "k_proj.weight", "v_proj.weight",
                        "q_proj.bias", "k_proj.bias", "v_proj.bias", "downsample.0.weight"))
            or k in ("visual.proj", "text_projection") or (k.startswith("visual.attnpool.") and k.endswith((".weight", ".bias")) and "positional" not in k))
        out[k] = v.half().float() if grid else v
    return out


def _bn(sd, seed, key, ch, device, gain=1.0):
    """BatchNorm2d buffers and affine in eval form: non-trivial running statistics so that the folding is exercised."""
    sd[key + ".weight"] = normal(seed, key + ".w", (ch,), 0.1, gain, device=device)
    sd[key + ".bias"] = normal(seed, key + ".b", (ch,), 0.05, device=device)
    sd[key + ".running_mean"] = normal(seed, key + ".rm", (ch,), 0.1, device=device)
    sd[key + ".running_var"] = 1.0 + 0.3 * torch.tanh(normal(seed, key + ".rv", (ch,), 1.0, device=device))
    sd[key + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64, device=device)


def _conv(sd, seed, key, cout, cin, k, device):
    sd[key + ".weight"] = normal(seed, key, (cout, cin, k, k), (2.0 / (cin * k * k)) ** 0.5, device=device)


def _resnet(sd, seed, geo, device):
    """ModifiedResNet key layout (TPT/clip/model.py:94-154): 3-conv stem, four stages of Bottlenecks (expansion 4, the first
    block of a stage carries the downsample branch), AttentionPool2d."""
    w = geo.vision_width
    _conv(sd, seed, "visual.conv1", w // 2, 3, 3, device);      _bn(sd, seed, "visual.bn1", w // 2, device)
    _conv(sd, seed, "visual.conv2", w // 2, w // 2, 3, device); _bn(sd, seed, "visual.bn2", w // 2, device)
    _conv(sd, seed, "visual.conv3", w, w // 2, 3, device);      _bn(sd, seed, "visual.bn3", w, device)
    inpl = w
    for li, (nb, mult) in enumerate(zip(geo.vision_layers, (1, 2, 4, 8)), start=1):
        planes = w * mult
        for b in range(nb):
            p = f"visual.layer{li}.{b}."
            stride = 2 if (b == 0 and li > 1) else 1
            _conv(sd, seed, p + "conv1", planes, inpl, 1, device);       _bn(sd, seed, p + "bn1", planes, device)
            _conv(sd, seed, p + "conv2", planes, planes, 3, device);     _bn(sd, seed, p + "bn2", planes, device)
            _conv(sd, seed, p + "conv3", planes * 4, planes, 1, device); _bn(sd, seed, p + "bn3", planes * 4, device, gain=0.5)
            if stride > 1 or inpl != planes * 4:
                _conv(sd, seed, p + "downsample.0", planes * 4, inpl, 1, device)
                _bn(sd, seed, p + "downsample.1", planes * 4, device)
            inpl = planes * 4
    e = w * 32
    hw = geo.image_resolution // 32
    sd["visual.attnpool.positional_embedding"] = normal(seed, "v.ap.pos", (hw * hw + 1, e), e ** -0.5, device=device)
    for nm, od in (("q_proj", e), ("k_proj", e), ("v_proj", e), ("c_proj", geo.embed_dim)):
        sd[f"visual.attnpool.{nm}.weight"] = normal(seed, "v.ap." + nm, (od, e), e ** -0.5, device=device)
        sd[f"visual.attnpool.{nm}.bias"] = normal(seed, "v.ap.b." + nm, (od,), 0.02, device=device)


def _text(sd, seed, geo, device, logit_scale):
    tw = geo.transformer_width
    sd["token_embedding.weight"] = normal(seed, "t.tok", (geo.vocab_size, tw), 0.02, device=device)
    sd["positional_embedding"] = normal(seed, "t.pos", (geo.context_length, tw), 0.01, device=device)
    t_proj_std = (tw ** -0.5) * ((2 * geo.transformer_layers) ** -0.5)
    _block(sd, seed, "transformer", tw, geo.transformer_layers, tw ** -0.5, t_proj_std,
           (2 * tw) ** -0.5, device)
    sd["ln_final.weight"] = normal(seed, "t.lnf_w", (tw,), 0.1, 1.0, device=device)
    sd["ln_final.bias"] = normal(seed, "t.lnf_b", (tw,), 0.05, device=device)
    sd["text_projection"] = normal(seed, "t.proj", (tw, geo.embed_dim), tw ** -0.5, device=device)
    sd["logit_scale"] = torch.tensor(logit_scale, dtype=torch.float32, device=device)
    return sd


# --------------------------------------------------------------------------
# inputs
# --------------------------------------------------------------------------

def make_views(seed: int, n_views: int, resolution: int, device="cpu") -> torch.Tensor:
    """[N,3,R,R] fp32 N(0,1): post-``Normalize`` statistics of the reference loader
    (TPT/tpt_cls_rl.py:132-133).  View 0 plays the clean centre crop."""
    return normal(seed, "views", (n_views, 3, resolution, resolution), device=device)


# name-length pmf chosen to reproduce the ImageNet prompt statistics measured in
# SURVEY.md §0 fact 4 (total length min 8 / mean 9.16 / max 18 with n_ctx = 4)
_NAME_LEN_PMF = ((1, 420), (2, 300), (3, 140), (4, 70), (5, 30), (6, 20), (7, 10), (8, 5), (9, 3), (11, 2))


def make_token_bank(geo: ClipGeometry, n_cls: int, seed: int = 7, n_ctx: int = 4,
                    ctx_token_ids: Optional[Tuple[int, ...]] = None) -> torch.Tensor:
    """int64 [C, context_length] rows ``[SOT, ctx.., name.., '.', EOT, 0..]`` in the
    format ``clip.tokenize`` produces (TPT/clip/clip.py:197-233).  EOT is the
    largest id of the vocabulary, as ``argmax`` EOT lookup requires
    (TPT/clip/custom_clip.py:71)."""
    sot, eot = geo.vocab_size - 2, geo.vocab_size - 1
    if ctx_token_ids is None:
        ctx_token_ids = ctx_token_ids_default(geo, n_ctx)
    assert len(ctx_token_ids) == n_ctx
    dot = 269 % (geo.vocab_size - 2)
    cum, tot = [], 0
    for ln, w in _NAME_LEN_PMF:
        tot += w
        cum.append((tot, ln))
    u = raw_u32(seed, "bank.len", n_cls, 0) % tot
    toks = torch.zeros(n_cls, geo.context_length, dtype=torch.int64)
    ids = raw_u32(seed, "bank.ids", n_cls * 16, 1).reshape(n_cls, 16) % (geo.vocab_size - 2)
    for c in range(n_cls):
        ln = next(l for t, l in cum if int(u[c]) < t)
        if c == n_cls - 1 and n_cls >= 100:
            ln = 11
        row = [sot, *ctx_token_ids, *ids[c, :ln].tolist(), dot, eot]
        toks[c, :len(row)] = torch.tensor(row, dtype=torch.int64)
    return toks


def ctx_token_ids_default(geo: ClipGeometry, n_ctx: int) -> Tuple[int, ...]:
    """ids of the hand-crafted context words; (320,1125,539,320) is
    'a photo of a' under the CLIP BPE."""
    base = (320, 1125, 539, 320)
    out = tuple(base[i % 4] % (geo.vocab_size - 2) for i in range(n_ctx))
    return out


def reward_members(spec: str, seeds, device=None) -> list:
    """'geoA+geoB+...' and '23+29+...' (or a list of ints) -> [(geometry, state dict), ...]: the members of a reward ensemble
    (device: where the weights are generated — the generator is bit-identical on CPU and GPU up to the last bit of BatchNorm statistics)."""
    names = spec.split("+")
    seeds = [int(x) for x in seeds.split("+")] if isinstance(seeds, str) else list(seeds)
    kw = {} if device is None else {"device": device}
    return [(GEOMETRIES[n], make_state_dict(GEOMETRIES[n], seed=s, **kw)) for n, s in zip(names, seeds)]
