"""Host-side mirror of the RLCF retrieval policy: `tune_image` / `tune_text` (retrieval/clip_ret_policy.py:76-137), `CLIPRet_TTA`
(retrieval/custom_models.py:29-163) and the retrieval `CLIPRewards` surface (retrieval/clip_reward.py:107-222: text_features /
image_features, CLIPScore(text_index=..., images_index=..., pairwise=...)).  Same signatures; the arithmetic is one call into the HIP
engine per query:
  image -> text  (only_visual=True):  rlcf_tta_retrieval_image over a caption bank = a class bank without learnable rows
                                       (rlcf_engine_set_class_bank, n_ctx = 0); the image encoder is tuned;
  text -> image  (only_visual=False): rlcf_tta_retrieval_text over an image bank (rlcf_engine_set_image_bank); every non-visual
                                       parameter (embeddings, text transformer, ln_final, text_projection, logit_scale) is tuned.
`text2image_loss` is the loss section of tune_text alone (rlcf_reward_loss with the two banks exchanged)."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import clip_reward as _cr
from . import clip_store, runtime
from .engine import TTAConfig

_VERSION = [0]       # bumped whenever a model or a reward model takes new image features (tune_text re-sends the bank to the engine then)


def _next_version() -> int:
    _VERSION[0] += 1
    return _VERSION[0]


class CLIPRet_TTA(nn.Module):
    """retrieval/custom_models.py:29-163.  only_visual=True: `parameters()` = [LayerNorm vector, flat vector of every other visual
    tensor] (Engine.visual_layout()), as rlcf_amd.custom_clip.CLIPCLS_TTA(only_norm=False).  only_visual=False: [text LayerNorm vector,
    flat vector of every other non-visual tensor] (Engine.text_layout())."""

    def __init__(self, device, arch="ViT-B-16", only_visual=True, momentum_update=False, update_freq=256, update_w=1.0, momentum=0.9999):
        super().__init__()
        self.clip_model, _, _ = clip_store.load(arch, device=device)
        runtime.SESSION.set_student(self.clip_model)
        self.device, self.only_visual, self.momentum_update = device, only_visual, momentum_update
        self.update_freq, self.update_w, self.momentum, self.update_counter = update_freq, update_w, momentum, 0
        self.text_features = None
        self.image_features = None
        self._ln = self._vis = None
        self._tuned_query = None     # (tokens, logits_per_text of the tuned text encoder) of the last tune_text call, until reset_initial()

    # the caption bank: tokens go to the engine once (model.set_text_features + reward_model.set_many_text_features, :150-156)
    def set_text_bank(self, texts: Optional[List[str]] = None, tokenized_prompts: Optional[torch.Tensor] = None):
        tok = clip_store.tokenize(texts) if tokenized_prompts is None else tokenized_prompts
        self.tokenized_prompts = tok
        runtime.SESSION.set_bank(tok, 0, torch.zeros(0))
        self.text_features = runtime.SESSION.engine().text_features(None)
        return self.text_features

    def set_text_features(self, text=None, tokenized_prompts=None, text_features=None):
        if text is not None or tokenized_prompts is not None:
            self.set_text_bank(text, tokenized_prompts)
        else:
            self.text_features = text_features

    def get_image_features(self, images):
        return runtime.SESSION.engine(images.shape[0]).encode_image(L.STUDENT, images)

    def set_image_features(self, images=None, image_features=None):              # custom_models.py:91-95
        if images is not None:
            step = runtime.SESSION.max_views
            image_features = torch.cat([self.get_image_features(images[i: i + step]) for i in range(0, images.shape[0], step)])
        self.image_features, self._img_version = image_features, _next_version()

    def _fetch(self):
        eng = runtime.SESSION.engine()
        if self.only_visual:
            self._ln_init, self._vis_init = eng.ln_params(pristine=True), eng.visual_params(1)
        else:
            self._vis_init, self._ln_init = eng.text_params(1)
        self._ln, self._vis = nn.Parameter(self._ln_init.clone()), nn.Parameter(self._vis_init.clone())

    @property
    def ln(self):                # LayerNorm vector of the tuned side
        if self._ln is None:
            self._fetch()
        return self._ln

    @property
    def vis(self):               # flat vector of every other tuned tensor (visual tensors, or the non-visual ones with only_visual=False)
        if self._vis is None:
            self._fetch()
        return self._vis

    def parameters(self, recurse: bool = True):
        return [self.ln, self.vis]

    @torch.no_grad()
    def reset_initial(self):                                # custom_models.py:124-126
        self.ln.data.copy_(self._ln_init)
        self.vis.data.copy_(self._vis_init)
        self._tuned_query = None

    @torch.no_grad()
    def momentum_update_model(self):                        # custom_models.py:128-143
        if not self.momentum_update:
            return
        self.update_counter += 1
        apply = self.update_counter >= self.update_freq
        if apply:
            self.update_counter = 0
        eng = runtime.SESSION.engine()
        if self.only_visual:
            eng.momentum_update(self.ln.data, self.momentum, self.update_w, apply)
            eng.momentum_update_visual(self.vis.data, self.momentum, self.update_w, apply)
            if apply:
                self._ln_init, self._vis_init = eng.ln_params(pristine=True), eng.visual_params(1)
        else:
            eng.momentum_update_text(self.vis.data, self.ln.data, self.momentum, self.update_w, apply)
            if apply:
                self._vis_init, self._ln_init = eng.text_params(1)

    @torch.no_grad()
    def forward(self, images=None, text=None, tokenized_prompts=None):
        """(logits_per_image, logits_per_text) with the current (possibly tuned) encoder, custom_models.py:66-75."""
        if not self.only_visual:
            return self._forward_text(text, tokenized_prompts)
        if text is not None or tokenized_prompts is not None:
            self.set_text_bank(text, tokenized_prompts)
        eng = runtime.SESSION.engine(images.shape[0])
        adapted = not (torch.equal(self.ln.data, self._ln_init) and torch.equal(self.vis.data, self._vis_init))
        if adapted:
            eng.set_ln_params(self.ln.data)
            eng.set_visual_params(self.vis.data)
        per_image = eng.logits(eng.encode_image(L.STUDENT, images), self.text_features)
        if adapted:
            eng.set_ln_params(self._ln_init)
            eng.set_visual_params(self._vis_init)
        return per_image, per_image.t()

    def _forward_text(self, text, tokenized_prompts):
        """text -> image: logits of ONE query caption against the image bank given to set_image_features (the reference's call shape
        `model(images=None, text=text)`, clip_ret_policy.py:122,194).  With tuned parameters only the query they were tuned on can
        be scored (the engine holds tuned text weights only inside rlcf_tta_retrieval_text)."""
        tok = (clip_store.tokenize(text) if tokenized_prompts is None else tokenized_prompts).reshape(-1, self.clip_model.geometry.context_length)
        if tok.shape[0] != 1:
            raise NotImplementedError("text -> image scoring takes one query caption per call (bs = 1, clip_ret_policy.py:112)")
        adapted = not (torch.equal(self.ln.data, self._ln_init) and torch.equal(self.vis.data, self._vis_init))
        if adapted:
            if self._tuned_query is None or not torch.equal(self._tuned_query[0], tok.cpu()):
                raise NotImplementedError("tuned text parameters score the caption they were tuned on; call reset_initial() first")
            per_text = self._tuned_query[1]
        else:
            per_text = runtime.SESSION.engine().tta_retrieval_text(tok[0], TTAConfig(tta_steps=0, sample_k=1))["final_logits"]
        return per_text.t(), per_text


class CLIPRewards(_cr.CLIPRewards):
    """retrieval/clip_reward.py:107-222: the classification reward model with the bank called `text_features` and both index
    directions in CLIPScore."""

    @property
    def text_features(self):
        return self.class_features

    @text_features.setter
    def text_features(self, v):
        self.class_features = v

    @torch.no_grad()
    def set_many_text_features(self, texts, text_bs=128):
        self.class_features = self.extract_text_features(captions=texts)

    @torch.no_grad()
    def set_text_features(self, captions=None, tokenized_cap=None, text_features=None):           # retrieval/clip_reward.py:64-68
        self.class_features = self.extract_text_features(captions=captions, tokenized_cap=tokenized_cap) if text_features is None else text_features

    @torch.no_grad()
    def set_image_features(self, images=None, image_features=None):                               # retrieval/clip_reward.py:71-76
        if image_features is None:
            step = runtime.SESSION.max_views
            image_features = torch.cat([self.extract_image_features(images[i: i + step]) for i in range(0, images.shape[0], step)])
        self.image_features, self._img_version = image_features, _next_version()

    @torch.no_grad()
    def set_image_features_with_dataloder(self, data_loader):                                     # retrieval/clip_reward.py:208-215
        self.image_features = torch.cat([self.extract_image_features(s["image"].to(self.device)) for s in data_loader], dim=0)
        self._img_version = _next_version()

    @torch.no_grad()
    def CLIPScore(self, text_index=None, images_index=None, pairwise=True):
        t = self.class_features[text_index.long()] if text_index is not None else self.class_features.repeat_interleave(self.sample_k, dim=0)
        i = self.image_features[images_index.long()] if images_index is not None else self.image_features.repeat_interleave(self.sample_k, dim=0)
        if not pairwise:               # row-wise scores (retrieval/clip_reward.py:124-126): n*K dot products, not an [nK, nK] matrix and its diagonal
            return (self.clipscore_weight * (t * i).sum(-1)).clamp_min(0).squeeze()
        return _cr._gemm_nt(t, i, self.clipscore_weight).clamp_min(0).squeeze()


class CLIPRewardsMultiple(_cr.CLIPRewardsMultiple):
    """retrieval/clip_reward.py:230-400: the reward ensemble with the banks called `text_features` / `image_features` (one matrix per
    model) and both index directions in CLIPScore.  tune_image / tune_text hand the per-model banks to the engine, whose loss kernel
    mixes the clamped scores with the same weights."""

    @property
    def text_features(self):
        return self.class_features

    @text_features.setter
    def text_features(self, v):
        self.class_features = v

    @torch.no_grad()
    def set_many_text_features(self, texts, text_bs=128):
        self.class_features = self.extract_text_features(captions=texts)

    @torch.no_grad()
    def set_text_features(self, captions=None, tokenized_cap=None, text_features=None):
        self.class_features = self.extract_text_features(captions=captions, tokenized_cap=tokenized_cap) if text_features is None else text_features

    @torch.no_grad()
    def set_image_features(self, images=None, image_features=None):
        if image_features is None:
            step = runtime.SESSION.max_views
            chunks = [self.extract_image_features(images[i: i + step]) for i in range(0, images.shape[0], step)]
            image_features = [torch.cat([c[m] for c in chunks]) for m in range(self.n_model)]
        self.image_features, self._img_version = image_features, _next_version()

    @torch.no_grad()
    def set_image_features_with_dataloder(self, data_loader):
        chunks = [self.extract_image_features(s["image"].to(self.device)) for s in data_loader]
        self.image_features = [torch.cat([c[m] for c in chunks]) for m in range(self.n_model)]
        self._img_version = _next_version()

    @torch.no_grad()
    def CLIPScore(self, text_index=None, images_index=None, pairwise=True):
        per_model = []
        for t_all, i_all in zip(self.class_features, self.image_features):
            t = t_all[text_index.long()] if text_index is not None else t_all.repeat_interleave(self.sample_k, dim=0)
            i = i_all[images_index.long()] if images_index is not None else i_all.repeat_interleave(self.sample_k, dim=0)
            sim = _cr._gemm_nt(t, i, self.clipscore_weight)
            per_model.append((sim if pairwise else torch.diagonal(sim)).clamp_min(0).squeeze())
        per_model = torch.stack(per_model)
        if not self.weighted_scores:
            return per_model.mean(dim=0)
        return (per_model.new_tensor(self.weights).reshape(-1, *([1] * (per_model.dim() - 1))) * per_model).sum(dim=0)


def tune_image(image, model, reward_model, optimizer, scaler, args=None):
    """retrieval/clip_ret_policy.py:76-103.  `optimizer` supplies the AdamW hyper-parameters; `scaler` is accepted and unused."""
    g = optimizer.param_groups[0]
    b1, b2 = g.get("betas", (0.9, 0.999))
    cfg = TTAConfig(selection_p=1.0, tta_steps=args.tta_steps, sample_k=reward_model.sample_k, lr=g["lr"], weight_decay=g.get("weight_decay", 0.0),
                    beta1=b1, beta2=b2, eps=g.get("eps", 1e-8), reward_process=bool(reward_model.reward_process),
                    process_batch=bool(reward_model.process_batch), reward_amplify=bool(reward_model.amplify_rewards),
                    clipscore_weight=reward_model.clipscore_weight)
    if not (torch.equal(model.ln.data, model._ln_init) and torch.equal(model.vis.data, model._vis_init)):
        raise NotImplementedError("tune_image starts from the reset state (model.reset_initial(), clip_ret_policy.py:180)")
    out = runtime.SESSION.engine(image.shape[0]).tta_retrieval_image(image, cfg, skip_final=True)
    with torch.no_grad():
        model.ln.data.copy_(out["ln_after"])
        model.vis.data.copy_(out["vis_after"])
    return out


def tune_text(text, model, reward_model, optimizer, scaler, args=None):
    """retrieval/clip_ret_policy.py:106-137: `text` = one caption (bs = 1).  The image bank is model.image_features (student) and
    reward_model.image_features (set by the loop before the queries, :184-185); `optimizer` supplies the AdamW hyper-parameters;
    `scaler` is accepted and unused (its inf / NaN step skip is built into the engine)."""
    g = optimizer.param_groups[0]
    b1, b2 = g.get("betas", (0.9, 0.999))
    cfg = TTAConfig(selection_p=1.0, tta_steps=args.tta_steps, sample_k=reward_model.sample_k, lr=g["lr"], weight_decay=g.get("weight_decay", 0.0),
                    beta1=b1, beta2=b2, eps=g.get("eps", 1e-8), reward_process=bool(reward_model.reward_process),
                    process_batch=bool(reward_model.process_batch), reward_amplify=bool(reward_model.amplify_rewards),
                    clipscore_weight=reward_model.clipscore_weight)
    if model.only_visual:
        raise ValueError("tune_text needs CLIPRet_TTA(only_visual=False)")
    if model.image_features is None or reward_model.image_features is None:
        raise L.RlcfError("tune_text: set_image_features of the model and of the reward model first (clip_ret_policy.py:184-185)")
    if not (torch.equal(model.ln.data, model._ln_init) and torch.equal(model.vis.data, model._vis_init)):
        raise NotImplementedError("tune_text starts from the reset state (model.reset_initial(), clip_ret_policy.py:196)")
    tok = clip_store.tokenize(text).reshape(1, -1)
    # the engine's bank follows the two feature tensors the loop set (:184-185): version counters bumped by the setters, not id()s
    rf = reward_model.image_features
    src = (getattr(model, "_img_version", None), getattr(reward_model, "_img_version", None), model.image_features.data_ptr(),
           tuple(r.data_ptr() for r in rf) if isinstance(rf, (list, tuple)) else rf.data_ptr(),
           tuple(model.image_features.shape))                                                 # (a tensor assigned past the setters has no version: re-sent)
    if runtime.SESSION.image_bank is None or None in src or getattr(runtime.SESSION, "_image_bank_src", None) != src:
        runtime.SESSION.set_image_bank(model.image_features, reward_model.image_features)
        runtime.SESSION._image_bank_src = src
    out = runtime.SESSION.engine().tta_retrieval_text(tok[0], cfg)
    if not isinstance(reward_model.image_features, (list, tuple)):
        reward_model.class_features = out["reward_text_features"]         # reward_model.set_text_features(captions=text), :117
    with torch.no_grad():
        model.ln.data.copy_(out["ln_after"])
        model.vis.data.copy_(out["text_after"])
    model._tuned_query = (tok.cpu(), out["final_logits"])
    return out


def text2image_loss(logits_per_text, reward_model, sample_k=None):
    """Loss section of tune_text (clip_ret_policy.py:123-133) for one query caption on the HIP loss kernel: logits_per_text [1, n_images];
    reward_model.text_features [1, Dr] (the query) and .image_features [n_images, Dr] (the bank).
    -> dict(topk_idx, clip_score, rewards, loss, dlogits)."""
    K = sample_k or reward_model.sample_k
    lg = logits_per_text.float().contiguous()
    n, C = lg.shape
    bank, q = reward_model.image_features.float().contiguous(), reward_model.class_features.float().contiguous()
    dev = lg.device
    o = dict(topk_idx=torch.empty(n, K, dtype=torch.int32, device=dev), clip_score=torch.empty(n * K, device=dev),
             rewards=torch.empty(n * K, device=dev), loss=torch.empty(1, device=dev), dlogits=torch.empty(n, C, device=dev))
    flags = (L.F_REWARD_PROCESS if reward_model.reward_process else 0) | (L.F_AMPLIFY if reward_model.amplify_rewards else 0) | \
            (L.F_PROCESS_BATCH if reward_model.process_batch else 0)
    L.check(L.lib().rlcf_reward_loss(lg.data_ptr(), C, None, n, C, K, bank.data_ptr(), q.data_ptr(), bank.shape[1], float(reward_model.clipscore_weight),
                                     flags, 0.0, o["topk_idx"].data_ptr(), o["clip_score"].data_ptr(), o["rewards"].data_ptr(),
                                     o["loss"].data_ptr(), o["dlogits"].data_ptr(), torch.cuda.current_stream().cuda_stream), "reward_loss")
    return o
