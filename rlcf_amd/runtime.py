"""Process-wide binding between the reference-shaped Python objects (ClipTestTimeTuning,
CLIPRewards) and the one HIP engine that serves them.  The reference keeps the student and the
reward model as two independent torch modules (TPT/tpt_cls_rl.py:94,124); here both live inside
one `rlcf_engine` so that the fused per-sample step needs a single C call."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib as L
from .engine import Engine

PRECISIONS = {"f32": L.PREC_F32, "f16x3": L.PREC_F16X3, "f16": L.PREC_F16}
TEXT_MODES = {"dense": L.TEXT_DENSE, "packed": L.TEXT_PACKED, "shared": L.TEXT_SHARED}


class Session:
    def __init__(self):
        self.student = None          # ClipCheckpoint
        self.rewards = []            # ClipCheckpoints: one (CLIPRewards) or several (CLIPRewardsMultiple)
        self.reward_mix = None       # ensemble weights (None: single model)
        self.reward_mean = False
        self.tokens: Optional[torch.Tensor] = None
        self.student_tokens = self.ctx_pos = None
        self.n_ctx = 0
        self.ctx_init: Optional[torch.Tensor] = None
        self.max_views = int(os.environ.get("RLCF_MAX_VIEWS", "64"))
        self.precision = PRECISIONS[os.environ.get("RLCF_PRECISION", "f16x3")]
        self.text_mode = TEXT_MODES[os.environ.get("RLCF_TEXT_MODE", "shared")]
        self._engine: Optional[Engine] = None
        self._lanes = []             # [[Engine, bank key]]: further engines for test images in flight side by side (lane_engines)
        self._key = None
        self.image_bank = None       # (student features [n, D], [reward features [n, Dr]]): the bank of the text -> image retrieval direction
        self._bank_version = 0       # bumped by set_bank: the engine re-reads the class bank when its copy is older
        self._bank_applied = -1
        self.bn_prior_strength = -1  # `--prior_strength` (ResNet students, tune_cls_rl.py:73-76): re-applied whenever the engine is rebuilt

    def set_student(self, ckpt):
        self.student = ckpt

    def set_bn_prior_strength(self, prior_strength: int):
        self.bn_prior_strength = int(prior_strength)
        if self._engine is not None and self.student is not None and self.student.geometry.is_resnet:
            self._engine.set_bn_prior_strength(self.bn_prior_strength)

    @property
    def reward(self):
        return self.rewards[0] if self.rewards else None

    def set_reward(self, ckpt):
        self.rewards, self.reward_mix, self.reward_mean = [ckpt], None, False

    def set_rewards(self, ckpts, weights, mean: bool):
        self.rewards, self.reward_mix, self.reward_mean = list(ckpts), [float(w) for w in weights], bool(mean)

    def set_bank(self, tokens: torch.Tensor, n_ctx: int, ctx_init: torch.Tensor, student_tokens=None, ctx_pos=None):
        self.tokens, self.n_ctx, self.ctx_init = tokens.detach().cpu(), n_ctx, ctx_init.detach().clone()
        self.student_tokens, self.ctx_pos = student_tokens, ctx_pos       # class tokens not at the end of the prompt ('front' / 'middle')
        self.image_bank = None
        self._bank_version += 1

    def set_image_bank(self, student_feats: torch.Tensor, reward_feats):
        """text -> image retrieval: a bank of image features replaces the class / caption bank"""
        rf = list(reward_feats) if isinstance(reward_feats, (list, tuple)) else [reward_feats]
        self.image_bank = (student_feats.detach().clone(), [r.detach().clone() for r in rf])
        self.tokens = None
        self._bank_version += 1

    def engine(self, n_views: int = 1) -> Engine:
        if self.student is None:
            raise L.RlcfError("no student CLIP bound: construct ClipTestTimeTuning / get_coop first")
        if n_views > self.max_views:
            self.max_views = n_views
        n_cls = int(self.tokens.shape[0]) if self.tokens is not None else (int(self.image_bank[0].shape[0]) if self.image_bank is not None else 1)
        key = (id(self.student), tuple(id(r) for r in self.rewards), self.max_views, self.precision)
        if self._engine is None or key != self._key or n_cls > self._engine.max_classes:
            if self._engine is not None:
                self._engine.close()
            self._close_lanes()
            self._engine, self._key, self._bank_applied = self._build(n_cls), key, None
        bkey = self._bank_key()
        if bkey is not None and bkey != self._bank_applied:
            self._apply_bank(self._engine)
            self._bank_applied = bkey
        return self._engine

    def _build(self, n_cls: int) -> Engine:
        eng = Engine(self.student.geometry, [r.geometry for r in self.rewards] or None, self.max_views, max(n_cls, 1), self.precision)
        eng.load_state_dict(L.STUDENT, self.student.state_dict)
        for m, r in enumerate(self.rewards):
            eng.load_state_dict(L.REWARD + m, r.state_dict)
        eng.finalize()
        if self.reward_mix is not None:
            eng.set_reward_mix(self.reward_mix, self.reward_mean)
        if self.student.geometry.is_resnet:
            eng.set_bn_prior_strength(self.bn_prior_strength)
        return eng

    def _bank_key(self):
        if self.tokens is not None:
            return (self._bank_version, self.text_mode)       # a counter, not id()/checksums: no stale bank, no device sync per call
        if self.image_bank is not None:
            return (self._bank_version, "images")
        return None

    def _apply_bank(self, eng: Engine):
        if self.tokens is not None:
            eng.set_class_bank(self.tokens, self.n_ctx, self.ctx_init, self.text_mode, getattr(self, "student_tokens", None), getattr(self, "ctx_pos", None))
        elif self.image_bank is not None:
            eng.set_image_bank(*self.image_bank)

    def lane_engines(self, lanes: int, n_views: int = 1):
        """`lanes` engines for test images IN FLIGHT side by side (tpt_cls_rl.test_time_adapt_eval(in_flight=...)): lane 0 is the session's
        engine, the others are further engines over the same checkpoints and class bank with their own weights copies, scratch and state
        (an engine serves one call at a time).  Built on first use, kept until the session's engine is rebuilt or closed."""
        first = self.engine(n_views)
        n_cls = first.max_classes
        bkey = self._bank_key()
        while len(self._lanes) < lanes - 1:
            self._lanes.append([self._build(n_cls), None])
        for lane in self._lanes[: lanes - 1]:
            if bkey is not None and lane[1] != bkey:
                self._apply_bank(lane[0])
                lane[1] = bkey
        return [first] + [lane[0] for lane in self._lanes[: lanes - 1]]

    def reset_state_moved(self) -> bool:
        """True when an applied cross-sample EMA (momentum_update_model) has moved the session engine's reset state away from the
        checkpoint's: further lane engines are built from the checkpoint and would tune from another state than lane 0."""
        return self._engine is not None and self._engine.reset_moved

    def _close_lanes(self):
        for lane in self._lanes:
            lane[0].close()
        self._lanes = []

    def close(self):
        if self._engine is not None:
            self._engine.close()
        self._close_lanes()
        self._engine = None
        self._key = self._bank_applied = None


SESSION = Session()


def reset_session() -> Session:
    global SESSION
    SESSION.close()
    SESSION = Session()
    return SESSION
