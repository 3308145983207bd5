"""Python face of the HIP engine: torch tensors in, torch tensors out, every FLOP in
librlcf_hip.so.  torch is used for device memory and the stream only."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib as L
from .synth import ClipGeometry


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _cfg(g: ClipGeometry) -> L.ClipCfg:
    if g.is_resnet:          # the reference's vision_layers tuple selects ModifiedResNet (TPT/clip/model.py:262-270)
        return L.ClipCfg(g.embed_dim, g.image_resolution, 0, g.vision_width, 0, g.context_length, g.vocab_size,
                         g.transformer_width, g.transformer_heads, g.transformer_layers, (C.c_int * 4)(*g.vision_layers))
    return L.ClipCfg(g.embed_dim, g.image_resolution, g.vision_layers, g.vision_width, g.vision_patch_size,
                     g.context_length, g.vocab_size, g.transformer_width, g.transformer_heads, g.transformer_layers)


@dataclass
class TTAConfig:
    """Flags read on the path (reference TPT/params.py:13-98; script values TPT/scripts/rlcf-prompt.sh)."""
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
    min_entropy_w: float = 0.1           # TPT/params.py:66
    sparse_backward: bool = True

    def flags(self) -> int:
        return ((L.F_REWARD_PROCESS if self.reward_process else 0) | (L.F_AMPLIFY if self.reward_amplify else 0) |
                (L.F_PROCESS_BATCH if self.process_batch else 0) | (L.F_MIN_ENTROPY if self.min_entropy_reg else 0))

    def n_sel(self, n_views: int) -> int:
        """int(N * top) of select_confident_samples (TPT/tpt_cls_rl.py:34): Python's double product, truncated."""
        return int(n_views * self.selection_p)

    def c_args(self, n_views: int, skip_final: bool = False, ctx_in: Optional[torch.Tensor] = None) -> L.TTAArgs:
        # n_sel travels as an int: the float selection_p field alone can truncate the other way (N=10, p=0.7)
        return L.TTAArgs(self.selection_p, self.tta_steps, self.sample_k, self.lr, self.weight_decay, self.beta1,
                         self.beta2, self.eps, self.flags(), self.clipscore_weight, self.min_entropy_w,
                         1 if self.sparse_backward else 0, 1 if skip_final else 0,
                         ctx_in.data_ptr() if ctx_in is not None else None, self.n_sel(n_views))


class Engine:
    """One student CLIP (+ frozen reward CLIPs: one geometry, or a list of up to MAX_REWARDS for the ensemble of
    CLIPRewardsMultiple, slot m addressed as which = REWARD + m) resident on the current GPU."""

    def __init__(self, student: ClipGeometry, reward, max_views: int, max_classes: int, precision: int = L.PREC_F32):
        if not torch.cuda.is_available():
            raise L.RlcfError("rlcf_amd.Engine needs a GPU: the HIP path has no CPU fallback")
        self.lib = L.lib()
        self.rewards = [] if reward is None else (list(reward) if isinstance(reward, (list, tuple)) else [reward])
        if len(self.rewards) > L.MAX_REWARDS:
            raise L.RlcfError(f"at most {L.MAX_REWARDS} reward models")
        self.student, self.reward = student, (self.rewards[0] if self.rewards else None)
        self.max_views, self.max_classes, self.precision = max_views, max_classes, precision
        sc = _cfg(student)
        rcs = (L.ClipCfg * max(1, len(self.rewards)))(*[_cfg(r) for r in self.rewards])
        self.h = self.lib.rlcf_engine_create_ensemble(C.byref(sc), rcs if self.rewards else None, len(self.rewards), max_views,
                                                      max_classes, precision)
        if not self.h:
            raise L.RlcfError("rlcf_engine_create: " + self.lib.rlcf_last_error().decode())
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_cls = 0
        self.n_ctx = 0
        self.reset_moved = False     # an applied cross-sample EMA has moved the reset state away from the checkpoint's (momentum_update*)

    def close(self):
        if getattr(self, "h", None):
            self.lib.rlcf_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup -----------------------------------------------------------------
    def load_state_dict(self, which: int, sd: Dict[str, torch.Tensor]) -> None:
        """OpenAI-layout CLIP state dict (reference TPT/clip/model.py:399-436)."""
        for k, v in sd.items():
            t = v.detach().to(self.device, torch.float32).contiguous().reshape(-1)
            L.check(self.lib.rlcf_engine_load_weight(self.h, which, k.encode(), t.data_ptr(), t.numel()), f"load {k}")
        torch.cuda.synchronize()

    def finalize(self) -> None:
        L.check(self.lib.rlcf_engine_finalize(self.h, _stream()), "finalize")

    def set_reward_mix(self, weights=None, mean: bool = False) -> None:
        """Ensemble rule of CLIPRewardsMultiple.CLIPScore (clip_reward.py:248-255): weighted sum, or the mean over models."""
        ws = [1.0] * len(self.rewards) if weights is None else [float(x) for x in weights]
        w = (C.c_float * max(len(ws), 1))(*ws)
        L.check(self.lib.rlcf_engine_set_reward_mix(self.h, w, len(ws), 1 if mean else 0), "set_reward_mix")

    def set_class_bank(self, tokens: torch.Tensor, n_ctx: int, ctx_init: torch.Tensor,
                       text_mode: int = L.TEXT_SHARED, student_tokens: Optional[torch.Tensor] = None,
                       ctx_pos: Optional[torch.Tensor] = None) -> None:
        """student_tokens / ctx_pos: class tokens not at the end of the prompt (PromptLearner 'front' / 'middle'), see
        rlcf_engine_set_class_bank_ex."""
        tok = np.ascontiguousarray(tokens.detach().cpu().numpy().astype(np.int32))
        if ctx_pos is not None:
            ci = ctx_init.detach().to(self.device, torch.float32).contiguous()
            stok = np.ascontiguousarray(student_tokens.detach().cpu().numpy().astype(np.int32))
            cp = np.ascontiguousarray(ctx_pos.detach().cpu().numpy().astype(np.int32))
            assert stok.shape == tok.shape and cp.shape == (tok.shape[0], n_ctx)
            L.check(self.lib.rlcf_engine_set_class_bank_ex(self.h, tok.ctypes.data, tok.shape[0], n_ctx, ci.data_ptr(), text_mode,
                                                           stok.ctypes.data, cp.ctypes.data, _stream()), "set_class_bank_ex")
            self.n_cls, self.n_ctx = tok.shape[0], n_ctx
            return
        ci = ctx_init.detach().to(self.device, torch.float32).contiguous() if n_ctx > 0 else None      # n_ctx = 0: plain texts (caption bank)
        L.check(self.lib.rlcf_engine_set_class_bank(self.h, tok.ctypes.data, tok.shape[0], n_ctx, ci.data_ptr() if ci is not None else None,
                                                    text_mode, _stream()), "set_class_bank")
        self.n_cls, self.n_ctx = tok.shape[0], n_ctx

    # ---- tower passes ------------------------------------------------------------
    def encode_image(self, which: int, images: torch.Tensor) -> torch.Tensor:
        g = self.student if which == L.STUDENT else self.rewards[which - L.REWARD]
        images = images.to(self.device, torch.float32).contiguous()
        out = torch.empty(images.shape[0], g.embed_dim, device=self.device)
        if images.shape[-1] != g.image_resolution:          # bicubic resample inside the engine
            L.check(self.lib.rlcf_encode_image_resized(self.h, which, _ptr(images), images.shape[0], images.shape[-1], _ptr(out),
                                                       _stream()), "encode_image_resized")
        else:
            L.check(self.lib.rlcf_encode_image(self.h, which, _ptr(images), images.shape[0], _ptr(out), _stream()),
                    "encode_image")
        return out

    def text_features(self, ctx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """student text features of the class bank under prompt `ctx` (None for a bank without learnable rows, n_ctx = 0)"""
        ctx = ctx.detach().to(self.device, torch.float32).contiguous() if ctx is not None else None
        out = torch.empty(self.n_cls, self.student.embed_dim, device=self.device)
        L.check(self.lib.rlcf_text_features(self.h, _ptr(ctx), _ptr(out), _stream()), "text_features")
        return out

    def reward_class_features(self, slot: int = 0) -> torch.Tensor:
        out = torch.empty(self.n_cls, self.rewards[slot].embed_dim, device=self.device)
        L.check(self.lib.rlcf_reward_class_features(self.h, L.REWARD + slot, _ptr(out), _stream()), "reward_class_features")
        return out

    def logits(self, img: torch.Tensor, txt: torch.Tensor) -> torch.Tensor:
        out = torch.empty(img.shape[0], txt.shape[0], device=self.device)
        L.check(self.lib.rlcf_logits(self.h, _ptr(img.contiguous()), img.shape[0], _ptr(txt.contiguous()), txt.shape[0],
                                     _ptr(out), _stream()), "logits")
        return out

    def text_backward_dense(self, ctx: torch.Tensor, img: torch.Tensor, dlogits: torch.Tensor) -> torch.Tensor:
        out = torch.empty(self.n_ctx, self.student.transformer_width, device=self.device)
        L.check(self.lib.rlcf_text_backward_dense(self.h, _ptr(ctx.contiguous()), _ptr(img.contiguous()), img.shape[0],
                                                  _ptr(dlogits.contiguous()), _ptr(out), _stream()), "text_backward")
        return out

    # ---- the per-sample step -----------------------------------------------------
    def tta_sample(self, views: torch.Tensor, cfg: TTAConfig, want_intermediates: bool = True, skip_final: bool = False,
                   ctx_in: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        views = views.to(self.device, torch.float32).contiguous()
        N, Cn, K = views.shape[0], self.n_cls, cfg.sample_k
        n_sel = cfg.n_sel(N)
        Wt = self.student.transformer_width
        Dr = sum(r.embed_dim for r in self.rewards)     # per-model blocks [n_sel, Dr_m], one after another
        dev = self.device
        o: Dict[str, torch.Tensor] = {
            "final_logits": torch.empty(1, Cn, device=dev), "top5": torch.empty(5, dtype=torch.int32, device=dev),
            "ctx_after": torch.empty(self.n_ctx, Wt, device=dev),
            "step_skipped": torch.zeros(max(cfg.tta_steps, 1), dtype=torch.int32, device=dev)}
        if want_intermediates and cfg.tta_steps > 0:
            o.update(logits=torch.empty(N, Cn, device=dev), entropy=torch.empty(N, device=dev),
                     selected_idx=torch.empty(n_sel, dtype=torch.int32, device=dev),
                     topk_idx=torch.empty(n_sel, K, dtype=torch.int32, device=dev),
                     clip_score=torch.empty(n_sel * K, device=dev), rewards=torch.empty(n_sel * K, device=dev),
                     loss=torch.empty(1, device=dev), dlogits=torch.empty(n_sel, Cn, device=dev),
                     ctx_grad=torch.empty(self.n_ctx, Wt, device=dev),
                     reward_image_features=torch.empty(n_sel * Dr, device=dev))
        co = L.TTAOut(**{k: _ptr(o[k]) if k in o else None for k in L.TTA_OUT_FIELDS})
        if ctx_in is not None:
            ctx_in = ctx_in.detach().to(dev, torch.float32).contiguous()
        a = cfg.c_args(N, skip_final, ctx_in)
        L.check(self.lib.rlcf_tta_sample(self.h, _ptr(views), N, C.byref(a), C.byref(co), _stream()), "tta_sample")
        if "reward_image_features" in o:
            parts = torch.split(o["reward_image_features"], [n_sel * r.embed_dim for r in self.rewards])
            parts = [p.view(n_sel, r.embed_dim) for p, r in zip(parts, self.rewards)]
            o["reward_image_features"] = parts[0] if len(parts) == 1 else parts
        return o

    def ln_params(self, pristine: bool = False) -> torch.Tensor:
        out = torch.empty(int(self.lib.rlcf_engine_ln_param_count(self.h)), device=self.device)
        L.check(self.lib.rlcf_engine_get_ln_params(self.h, _ptr(out), 1 if pristine else 0, _stream()), "get_ln_params")
        return out

    def set_ln_params(self, p: torch.Tensor) -> None:
        p = p.detach().to(self.device, torch.float32).contiguous()
        L.check(self.lib.rlcf_engine_set_ln_params(self.h, _ptr(p), _stream()), "set_ln_params")

    def momentum_update(self, current: torch.Tensor, momentum: float, update_w: float, apply: bool) -> None:
        """CLIPCLS_TTA.momentum_update_model on the tunable LayerNorm set (custom_clip.py:460-475)."""
        cur = current.detach().to(self.device, torch.float32).contiguous()
        L.check(self.lib.rlcf_engine_momentum_update(self.h, _ptr(cur), float(momentum), float(update_w), 1 if apply else 0, _stream()),
                "momentum_update")
        self.reset_moved = self.reset_moved or bool(apply)

    def set_f16_lnfold(self, on: bool) -> None:
        """RLCF_PREC_F16: f16 residual stream with the LayerNorms folded into the products (opt-in; rlcf_engine_set_f16_lnfold)."""
        L.check(self.lib.rlcf_engine_set_f16_lnfold(self.h, 1 if on else 0), "set_f16_lnfold")

    def set_side_stream(self, on: bool) -> None:
        """False: one-image calls stay on the caller's stream (lane engines of samples in flight, rlcf_engine_set_side_stream)."""
        L.check(self.lib.rlcf_engine_set_side_stream(self.h, 1 if on else 0), "set_side_stream")

    def reset_visual_state(self) -> None:
        """State part of CLIPCLS_TTA.reset_classnames_and_state (custom_clip.py:449-454): reset state and EMA back to the checkpoint."""
        L.check(self.lib.rlcf_engine_reset_visual_state(self.h, _stream()), "reset_visual_state")
        self.reset_moved = False

    def tta_sample_ln(self, views: torch.Tensor, cfg: TTAConfig, skip_final: bool = False) -> Dict[str, torch.Tensor]:
        """LayerNorm-tuning step (reference TPT/tune_cls_rl.py with CLIPCLS_TTA(only_norm=True))."""
        views = views.to(self.device, torch.float32).contiguous()
        N, Cn, K = views.shape[0], self.n_cls, cfg.sample_k
        n_sel = cfg.n_sel(N)
        npar = int(self.lib.rlcf_engine_ln_param_count(self.h))
        dev = self.device
        o = dict(final_logits=torch.empty(1, Cn, device=dev), top5=torch.empty(5, dtype=torch.int32, device=dev),
                 ln_after=torch.empty(npar, device=dev), ln_grad=torch.empty(npar, device=dev),
                 step_skipped=torch.zeros(max(cfg.tta_steps, 1), dtype=torch.int32, device=dev),
                 logits=torch.empty(N, Cn, device=dev), selected_idx=torch.empty(n_sel, dtype=torch.int32, device=dev),
                 topk_idx=torch.empty(n_sel, K, dtype=torch.int32, device=dev), clip_score=torch.empty(n_sel * K, device=dev),
                 rewards=torch.empty(n_sel * K, device=dev), loss=torch.empty(1, device=dev))
        co = L.TTAOut(**{k: _ptr(o[k]) if k in o else None for k in L.TTA_OUT_FIELDS})
        a = cfg.c_args(N, skip_final)
        L.check(self.lib.rlcf_tta_sample_ln(self.h, _ptr(views), N, C.byref(a), C.byref(co), _stream()), "tta_sample_ln")
        return o

    # ---- ModifiedResNet student: BatchNorm tuning (tta_sample_ln / tta_batch_ln serve it; include/rlcf_hip.h)
    def set_bn_prior_strength(self, prior_strength: int):
        """`--prior_strength` (TPT/params.py:91): < 0 = nn.BatchNorm2d train mode, >= 0 = `_modified_bn_forward` (tune_cls_rl.py:35-44)."""
        L.check(self.lib.rlcf_engine_set_bn_prior_strength(self.h, int(prior_strength)), "set_bn_prior_strength")

    def encode_image_bn(self, images: torch.Tensor, eval_form: bool = False) -> torch.Tensor:
        """ResNet student, live tunable parameters; BatchNorms on batch statistics (what CLIPCLS_TTA's forward computes under norm-layer
        tuning) or, eval_form, on the running statistics as the last tuning pass left them (every-parameter tuning after model.eval():
        include/rlcf_hip.h)."""
        images = images.to(self.device, torch.float32).contiguous()
        out = torch.empty(images.shape[0], self.student.embed_dim, device=self.device)
        L.check(self.lib.rlcf_engine_encode_image_bn_form(self.h, _ptr(images), images.shape[0], 0 if eval_form else -1, _ptr(out), _stream()),
                "encode_image_bn")
        return out

    def bn_stats(self, pristine: bool = False) -> torch.Tensor:
        """[running_mean | running_var] per BatchNorm in execution order, as the last tta_sample_ln left them (or the checkpoint's)."""
        n = int(self.lib.rlcf_engine_bn_stats_count(self.h))
        out = torch.empty(n, device=self.device)
        L.check(self.lib.rlcf_engine_get_bn_stats(self.h, _ptr(out), 1 if pristine else 0, _stream()), "get_bn_stats")
        return out

    # ---- full image-encoder tuning (CLIPCLS_TTA only_norm=False, TPT/clip/custom_clip.py:477-479)
    def _resnet_visual_names(self):
        """names of the flat vector's tensors for a ModifiedResNet student: clip_model.visual.named_parameters() order without the
        tensors whose name contains 'bn' (those live in the norm-layer vector)"""
        names = ["visual.conv1.weight", "visual.conv2.weight", "visual.conv3.weight"]
        w = self.student.vision_width
        inpl = w
        for li, nb in enumerate(self.student.vision_layers):
            planes = w << li
            for b in range(nb):
                p = f"visual.layer{li + 1}.{b}."
                names += [p + "conv1.weight", p + "conv2.weight", p + "conv3.weight"]
                if (b == 0 and li > 0) or inpl != planes * 4:
                    names += [p + "downsample.0.weight", p + "downsample.1.weight", p + "downsample.1.bias"]
                inpl = planes * 4
        names.append("visual.attnpool.positional_embedding")
        for nm in ("k_proj", "q_proj", "v_proj", "c_proj"):
            names += [f"visual.attnpool.{nm}.weight", f"visual.attnpool.{nm}.bias"]
        return names

    def visual_layout(self):
        """[(state-dict key, offset, numel)] of the flat non-norm visual parameter vector (include/rlcf_hip.h)."""
        if not isinstance(self.student.vision_layers, int):          # ModifiedResNet student
            names = self._resnet_visual_names()
        else:
            layers = self.student.vision_layers
            names = ["visual.class_embedding", "visual.positional_embedding", "visual.proj", "visual.conv1.weight"]
            for i in range(layers):
                b = f"visual.transformer.resblocks.{i}."
                names += [b + "attn.in_proj_weight", b + "attn.in_proj_bias", b + "attn.out_proj.weight", b + "attn.out_proj.bias",
                          b + "mlp.c_fc.weight", b + "mlp.c_fc.bias", b + "mlp.c_proj.weight", b + "mlp.c_proj.bias"]
        n = len(names)
        off, num = (C.c_int64 * n)(), (C.c_int64 * n)()
        got = self.lib.rlcf_engine_visual_param_layout(self.h, off, num, n, _stream())
        if got < 0:
            L.check(got, "visual_param_layout")
        assert got == len(names)
        return [(k, int(off[i]), int(num[i])) for i, k in enumerate(names)]

    def merge_visual(self, ln_vec: torch.Tensor, vis_vec: torch.Tensor) -> torch.Tensor:
        """norm-layer vector + flat vector -> one vector in clip_model.visual.named_parameters() order (what the reference's
        optimizer sees with only_norm=False; oracle.rlcf_ref.visual_param_keys)."""
        t = {k: vis_vec[o: o + n] for k, o, n in self.visual_layout()}
        if not isinstance(self.student.vision_layers, int):
            # ModifiedResNet: the norm vector holds (weight | bias) per tuned BatchNorm in named order: stem bn1..3, then bn1..3 per block
            out, pos = [], 0

            def bn(c):
                nonlocal pos
                v = [ln_vec[pos: pos + c], ln_vec[pos + c: pos + 2 * c]]
                pos += 2 * c
                return v
            for i in (1, 2, 3):
                wt = t[f"visual.conv{i}.weight"]
                out += [wt] + bn(self._rn_cout(f"visual.conv{i}.weight"))
            for k, _, _ in self.visual_layout():
                if ".layer" not in k:
                    continue
                out.append(t[k])
                if k.endswith(("conv1.weight", "conv2.weight", "conv3.weight")):
                    out += bn(self._rn_cout(k))
            out.append(t["visual.attnpool.positional_embedding"])
            for nm in ("k_proj", "q_proj", "v_proj", "c_proj"):
                out += [t[f"visual.attnpool.{nm}.weight"], t[f"visual.attnpool.{nm}.bias"]]
            assert pos == ln_vec.numel()
            return torch.cat([x.reshape(-1) for x in out])
        Wv, Lv = self.student.vision_width, self.student.vision_layers
        ln = ln_vec.view(-1, Wv)                                   # rows: ln_pre.w, ln_pre.b, (ln_1.w, ln_1.b, ln_2.w, ln_2.b) x L, ln_post.w/.b
        out = [t["visual.class_embedding"], t["visual.positional_embedding"], t["visual.proj"], t["visual.conv1.weight"], ln[0], ln[1]]
        for i in range(Lv):
            b = f"visual.transformer.resblocks.{i}."
            out += [t[b + "attn.in_proj_weight"], t[b + "attn.in_proj_bias"], t[b + "attn.out_proj.weight"], t[b + "attn.out_proj.bias"],
                    ln[2 + 4 * i], ln[3 + 4 * i], t[b + "mlp.c_fc.weight"], t[b + "mlp.c_fc.bias"], t[b + "mlp.c_proj.weight"],
                    t[b + "mlp.c_proj.bias"], ln[4 + 4 * i], ln[5 + 4 * i]]
        out += [ln[2 + 4 * Lv], ln[3 + 4 * Lv]]
        return torch.cat([x.reshape(-1) for x in out])

    def _rn_cout(self, key: str) -> int:
        """output channels of a ModifiedResNet convolution, from the geometry (model.py:10-55,94-127)"""
        w = self.student.vision_width
        if ".layer" not in key:
            return {"visual.conv1.weight": w // 2, "visual.conv2.weight": w // 2, "visual.conv3.weight": w}[key]
        li = int(key.split(".layer")[1][0]) - 1
        planes = w << li
        return planes * 4 if key.endswith(("conv3.weight", "downsample.0.weight")) else planes

    def visual_params(self, which: int = 0) -> torch.Tensor:
        """flat vector: 0 live, 1 reset state, 2 checkpoint, 3 momentum state"""
        out = torch.empty(int(self.lib.rlcf_engine_visual_param_count(self.h, _stream())), device=self.device)
        L.check(self.lib.rlcf_engine_get_visual_params(self.h, _ptr(out), which, _stream()), "get_visual_params")
        return out

    def set_visual_params(self, p: torch.Tensor) -> None:
        p = p.detach().to(self.device, torch.float32).contiguous()
        L.check(self.lib.rlcf_engine_set_visual_params(self.h, _ptr(p), _stream()), "set_visual_params")

    def momentum_update_visual(self, current: torch.Tensor, momentum: float, update_w: float, apply: bool) -> None:
        cur = current.detach().to(self.device, torch.float32).contiguous()
        L.check(self.lib.rlcf_engine_momentum_update_visual(self.h, _ptr(cur), float(momentum), float(update_w), 1 if apply else 0,
                                                            _stream()), "momentum_update_visual")
        self.reset_moved = self.reset_moved or bool(apply)

    def tta_retrieval_image(self, images: torch.Tensor, cfg: TTAConfig, skip_final: bool = False) -> Dict[str, torch.Tensor]:
        """Image -> text retrieval step: tune_image + the evaluation of its loop (retrieval/clip_ret_policy.py:76-103,171-176) over the
        caption bank given to set_class_bank(n_ctx=0).  Every query image is 'selected'."""
        return self.tta_sample_visual(images, cfg, skip_final, retrieval=True)

    def tta_sample_visual(self, views: torch.Tensor, cfg: TTAConfig, skip_final: bool = False, retrieval: bool = False) -> Dict[str, torch.Tensor]:
        """Full image-encoder tuning step (reference TPT/tune_cls_rl.py with CLIPCLS_TTA(only_norm=False), scripts/rlcf-tune.sh)."""
        views = views.to(self.device, torch.float32).contiguous()
        N, Cn, K = views.shape[0], self.n_cls, cfg.sample_k
        n_sel = N if retrieval else cfg.n_sel(N)
        npar = int(self.lib.rlcf_engine_ln_param_count(self.h))
        nvis = int(self.lib.rlcf_engine_visual_param_count(self.h, _stream()))
        if nvis <= 0:
            L.check(-1, "tta_sample_visual (visual_param_count)")
        dev = self.device
        o = dict(final_logits=torch.empty(1, Cn, device=dev), top5=torch.empty(5, dtype=torch.int32, device=dev),
                 ln_after=torch.empty(npar, device=dev), ln_grad=torch.empty(npar, device=dev),
                 vis_after=torch.empty(nvis, device=dev), vis_grad=torch.empty(nvis, device=dev),
                 step_skipped=torch.zeros(max(cfg.tta_steps, 1), dtype=torch.int32, device=dev),
                 logits=torch.empty(N, Cn, device=dev), selected_idx=torch.empty(n_sel, dtype=torch.int32, device=dev),
                 topk_idx=torch.empty(n_sel, K, dtype=torch.int32, device=dev), clip_score=torch.empty(n_sel * K, device=dev),
                 rewards=torch.empty(n_sel * K, device=dev), loss=torch.empty(1, device=dev))
        co = L.TTAOut(**{k: _ptr(o[k]) if k in o else None for k in L.TTA_OUT_FIELDS})
        a = cfg.c_args(N, skip_final)
        fn = self.lib.rlcf_tta_retrieval_image if retrieval else self.lib.rlcf_tta_sample_visual
        L.check(fn(self.h, _ptr(views), N, C.byref(a), C.byref(co), _stream()), "tta_retrieval_image" if retrieval else "tta_sample_visual")
        return o

    # ---- text-encoder tuning (retrieval text -> image, CLIPRet_TTA only_visual=False, retrieval/custom_models.py:139-147)
    def text_layout(self):
        """[(state-dict key, offset, numel)] of the flat non-LayerNorm text parameter vector (include/rlcf_hip.h)."""
        layers = self.student.transformer_layers
        n = 3 + 8 * layers + 1
        off, num = (C.c_int64 * n)(), (C.c_int64 * n)()
        got = self.lib.rlcf_engine_text_param_layout(self.h, off, num, n, _stream())
        if got < 0:
            L.check(got, "text_param_layout")
        names = ["token_embedding.weight", "positional_embedding", "text_projection"]
        for i in range(layers):
            b = f"transformer.resblocks.{i}."
            names += [b + "attn.in_proj_weight", b + "attn.in_proj_bias", b + "attn.out_proj.weight", b + "attn.out_proj.bias",
                      b + "mlp.c_fc.weight", b + "mlp.c_fc.bias", b + "mlp.c_proj.weight", b + "mlp.c_proj.bias"]
        names.append("logit_scale")
        assert got == len(names)
        return [(k, int(off[i]), int(num[i])) for i, k in enumerate(names)]

    def merge_text(self, ln_vec: torch.Tensor, flat_vec: torch.Tensor) -> torch.Tensor:
        """text LayerNorm vector + flat vector -> one vector in the order the reference's optimizer sees them: the named_parameters()
        of the CLIP module without the 'visual' ones (oracle.retrieval_ref.text_param_keys)."""
        Wt, Lt = self.student.transformer_width, self.student.transformer_layers
        ln = ln_vec.view(-1, Wt)                                   # rows: ln_final.w, ln_final.b, (ln_1.w, ln_1.b, ln_2.w, ln_2.b) x L
        t = {k: flat_vec[o: o + n] for k, o, n in self.text_layout()}
        out = [t["positional_embedding"], t["text_projection"], t["logit_scale"]]
        for i in range(Lt):
            b = f"transformer.resblocks.{i}."
            out += [t[b + "attn.in_proj_weight"], t[b + "attn.in_proj_bias"], t[b + "attn.out_proj.weight"], t[b + "attn.out_proj.bias"],
                    ln[2 + 4 * i], ln[3 + 4 * i], t[b + "mlp.c_fc.weight"], t[b + "mlp.c_fc.bias"], t[b + "mlp.c_proj.weight"],
                    t[b + "mlp.c_proj.bias"], ln[4 + 4 * i], ln[5 + 4 * i]]
        out += [t["token_embedding.weight"], ln[0], ln[1]]
        return torch.cat([x.reshape(-1) for x in out])

    def text_params(self, which: int = 1):
        """(flat vector, LayerNorm vector) of the tunable text side: which = 0 live, 1 reset state"""
        ln_count = C.c_int(0)
        nflat = int(self.lib.rlcf_engine_text_param_count(self.h, C.byref(ln_count), _stream()))
        if nflat <= 0:
            L.check(-1, "text_params (text_param_count)")
        flat, ln = torch.empty(nflat, device=self.device), torch.empty(ln_count.value, device=self.device)
        L.check(self.lib.rlcf_engine_get_text_params(self.h, _ptr(flat), _ptr(ln), which, _stream()), "get_text_params")
        return flat, ln

    def momentum_update_text(self, cur_flat: torch.Tensor, cur_ln: torch.Tensor, momentum: float, update_w: float, apply: bool) -> None:
        f, l = cur_flat.detach().to(self.device, torch.float32).contiguous(), cur_ln.detach().to(self.device, torch.float32).contiguous()
        L.check(self.lib.rlcf_engine_momentum_update_text(self.h, _ptr(f), _ptr(l), float(momentum), float(update_w), 1 if apply else 0,
                                                          _stream()), "momentum_update_text")

    def set_image_bank(self, student_feats: torch.Tensor, reward_feats) -> None:
        """The bank of the text -> image direction: L2-normalised image features under the student [n, D] and under every reward model
        (a tensor [n, Dr], or a list of them).  Replaces the class / caption bank."""
        sf = student_feats.detach().to(self.device, torch.float32).contiguous()
        rf = reward_feats if isinstance(reward_feats, (list, tuple)) else [reward_feats]
        rf = torch.cat([r.detach().to(self.device, torch.float32).contiguous().reshape(-1) for r in rf])
        L.check(self.lib.rlcf_engine_set_image_bank(self.h, _ptr(sf), _ptr(rf), sf.shape[0], _stream()), "set_image_bank")
        self.n_cls = sf.shape[0]

    def tta_retrieval_text(self, tokens: torch.Tensor, cfg: TTAConfig, skip_final: bool = False) -> Dict[str, torch.Tensor]:
        """Text -> image retrieval step: tune_text + the evaluation of its loop (retrieval/clip_ret_policy.py:106-137,193-196) for ONE
        query caption (tokens [context_length]) over the image bank given to set_image_bank."""
        tok = tokens.reshape(-1).to("cpu", torch.int32).contiguous()
        assert tok.numel() == self.student.context_length
        ln_count = C.c_int(0)
        nflat = int(self.lib.rlcf_engine_text_param_count(self.h, C.byref(ln_count), _stream()))
        if nflat <= 0:
            L.check(-1, "tta_retrieval_text (text_param_count)")
        dev, n, K = self.device, self.n_cls, cfg.sample_k
        Dr = self.rewards[0].embed_dim
        o = dict(final_logits=torch.empty(1, n, device=dev), logits=torch.empty(1, n, device=dev), dlogits=torch.empty(1, n, device=dev),
                 ln_after=torch.empty(ln_count.value, device=dev), ln_grad=torch.empty(ln_count.value, device=dev),
                 vis_after=torch.empty(nflat, device=dev), vis_grad=torch.empty(nflat, device=dev),
                 step_skipped=torch.zeros(max(cfg.tta_steps, 1), dtype=torch.int32, device=dev),
                 reward_image_features=torch.empty(1, Dr, device=dev),
                 topk_idx=torch.empty(1, K, dtype=torch.int32, device=dev), clip_score=torch.empty(K, device=dev),
                 rewards=torch.empty(K, device=dev), loss=torch.empty(1, device=dev))
        co = L.TTAOut(**{k: _ptr(o[k]) if k in o else None for k in L.TTA_OUT_FIELDS})
        a = cfg.c_args(1, skip_final)
        L.check(self.lib.rlcf_tta_retrieval_text(self.h, tok.data_ptr(), C.byref(a), C.byref(co), _stream()), "tta_retrieval_text")
        o["text_after"], o["text_grad"], o["reward_text_features"] = o.pop("vis_after"), o.pop("vis_grad"), o.pop("reward_image_features")
        return o

    def tta_batch(self, views: torch.Tensor, cfg: TTAConfig, want_logits: bool = False):
        """views [count,N,3,R,R] -> top5 [count,5] (and final logits [count,C])."""
        views = views.to(self.device, torch.float32).contiguous()
        count, N = views.shape[0], views.shape[1]
        top5 = torch.empty(count, 5, dtype=torch.int32, device=self.device)
        fl = torch.empty(count, self.n_cls, device=self.device) if want_logits else None
        a = cfg.c_args(N)
        L.check(self.lib.rlcf_tta_batch(self.h, _ptr(views), count, N, C.byref(a), _ptr(fl), _ptr(top5), _stream()),
                "tta_batch")
        return (top5, fl) if want_logits else top5

    def tta_batch_ln(self, views: torch.Tensor, cfg: TTAConfig, want_logits: bool = False):
        """LayerNorm tuning of views [count,N,3,R,R] -> top5 [count,5] (and final logits [count,C])."""
        views = views.to(self.device, torch.float32).contiguous()
        count, N = views.shape[0], views.shape[1]
        top5 = torch.empty(count, 5, dtype=torch.int32, device=self.device)
        fl = torch.empty(count, self.n_cls, device=self.device) if want_logits else None
        a = cfg.c_args(N)
        L.check(self.lib.rlcf_tta_batch_ln(self.h, _ptr(views), count, N, C.byref(a), _ptr(fl), _ptr(top5), _stream()), "tta_batch_ln")
        return (top5, fl) if want_logits else top5

    def last_flops(self) -> float:
        return float(self.lib.rlcf_engine_last_flops(self.h))

    def text_rows(self) -> int:
        return int(self.lib.rlcf_engine_text_rows(self.h))


_LANE_STREAMS: dict = {}                  # device index -> the lanes' streams (torch pool streams, drawn once)


class Lanes:
    """K engines with one non-blocking stream each for test images IN FLIGHT (include/rlcf_hip.h, rlcf_lanes_*): `submit` enqueues one
    sample on the next lane from the caller's thread and returns at once; nothing waits on the host."""

    def __init__(self, engines):
        self.lib = L.lib()
        self.engines = list(engines)
        arr = (C.c_void_p * len(self.engines))(*[e.h for e in self.engines])
        # torch's own (pool, non-blocking) streams: the caching allocator keeps a recorded stream's handle for as long as the block lives,
        # so the lanes must not run on streams that are destroyed with the lanes object
        # ... and they are a process-wide resource, drawn ONCE per device and shared by every Lanes object after that: torch hands pool streams
        # out round robin from 32, so lanes that drew fresh ones each time ended up, ten objects later, on streams other parts of the process
        # own (a loader's upload stream) and on triples that were not created together — which share hardware queues (round 6: every third
        # leg of the bench's three-lane setting, the one whose draw wrapped the pool, read 85 instead of 97 images/s).  Two Lanes objects alive
        # at once on one device share streams: correct, serialised.
        dev = torch.device(self.engines[0].device)
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        pool = _LANE_STREAMS.setdefault(key, [])
        while len(pool) < len(self.engines):
            pool.append(torch.cuda.Stream(device=dev))
        self.streams = pool[:len(self.engines)]
        sarr = (C.c_void_p * len(self.engines))(*[s.cuda_stream for s in self.streams])
        self.h = self.lib.rlcf_lanes_create_on(arr, len(self.engines), sarr)
        if not self.h:
            raise L.RlcfError("rlcf_lanes_create_on: " + self.lib.rlcf_last_error().decode())
        self.next_lane = 0                                  # the lane the next `submit` takes (round robin, rlcf_lanes_submit)

    def submit(self, views: torch.Tensor, cfg: TTAConfig, top5_row: torch.Tensor, norm_layers: bool = False,
               final_logits: Optional[torch.Tensor] = None) -> int:
        """views [N,3,R,R] (one test image) or [count,N,3,R,R]; top5_row int32 [count*5]; -> the lane that took it."""
        assert views.is_cuda and views.dtype == torch.float32 and views.is_contiguous()
        count = 1 if views.dim() == 4 else views.shape[0]
        N = views.shape[-4]
        a = cfg.c_args(N)
        k = self.lib.rlcf_lanes_submit(self.h, _ptr(views), count, N, C.byref(a), _ptr(final_logits), top5_row.data_ptr(), 1 if norm_layers else 0,
                                       _stream())
        if k < 0:
            L.check(k, "rlcf_lanes_submit")
        self.next_lane = (k + 1) % len(self.engines)
        for t in (views, top5_row, final_logits):
            if t is not None:
                t.record_stream(self.streams[k])       # the caching allocator must not hand the block out again before the lane has run
        return k

    def join(self) -> None:
        """the current stream waits (on the device) for everything submitted so far"""
        L.check(self.lib.rlcf_lanes_join(self.h, _stream()), "rlcf_lanes_join")

    def close(self):
        if getattr(self, "h", None):
            self.lib.rlcf_lanes_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
