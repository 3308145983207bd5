"""Host-side mirror of the reference reward surface (TPT/clip_reward.py): `get_reward_model` (:29-40),
`BaseRewards` (:43-73), `CLIPRewards` (:76-177), `CLIPRewardsMultiple` (:180-307).  The frozen reward CLIP lives in the shared HIP
engine (rlcf_amd.runtime); this class carries the flags the tuning loop reads and the cached features."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from . import clip_store, runtime


# relative confidence of each reward arch in the ensemble (TPT/clip_reward.py:21-26); add entries for other checkpoints
CONFIDECES = {"ViT-L/14@336px": 10, "ViT-L/14": 5, "RN50x64": 3, "ViT-B/16": 1}
ENSEMBLE_ARCHS = ["ViT-L/14@336px", "RN50x64", "ViT-L/14"]     # the fixed list of get_reward_model (:31)


def get_reward_model(device, args):
    """TPT/clip_reward.py:29-40."""
    if getattr(args, "multiple_reward_models", 0):
        return CLIPRewardsMultiple(device, arch=list(ENSEMBLE_ARCHS), classification=True, amplify_rewards=args.reward_amplify,
                                   sample_k=args.sample_k, reward_process=args.reward_process, process_batch=args.process_batch,
                                   weighted_scores=args.weighted_scores)
    return CLIPRewards(device, arch=args.reward_arch, classification=True, amplify_rewards=args.reward_amplify,
                       sample_k=args.sample_k, reward_process=args.reward_process, process_batch=args.process_batch)


def _gemm_nt(a: torch.Tensor, w: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """alpha * a @ w^T on the HIP GEMM (rlcf_gemm_nt, f32-MFMA kernel); K is zero-padded to the kernel's granule of 16."""
    a, w = a.float().contiguous(), w.float().contiguous()
    k = a.shape[1]
    if k % 16:
        a, w = nn.functional.pad(a, (0, 16 - k % 16)), nn.functional.pad(w, (0, 16 - k % 16))
        k = a.shape[1]
    out = torch.empty(a.shape[0], w.shape[0], device=a.device)
    L.check(L.lib().rlcf_gemm_nt(a.data_ptr(), k, w.data_ptr(), k, None, None, 0, None, 0, out.data_ptr(), w.shape[0], a.shape[0],
                                 w.shape[0], k, float(alpha), L.EPI_NONE, L.PREC_F32, torch.cuda.current_stream().cuda_stream), "gemm_nt")
    return out


def _clamped_scores(bank: torch.Tensor, feats: torch.Tensor, index: torch.Tensor, k: int, weight: float, pairwise: bool) -> torch.Tensor:
    """CLIPScore of the reference (clip_reward.py:118-128): text rows bank[index] [n*K, D] against the image features repeated K
    times [n*K, D]; pairwise: the full [n*K, n*K] matrix weight * text @ image^T, else entry e against its own view e // K.
    max(., 0), squeezed.  The products run on the HIP GEMM; torch only indexes."""
    rows = bank[index.long()]
    sim = _gemm_nt(rows, feats, weight)                                   # [n*K, n]: entry e against every view
    view_of = torch.arange(rows.shape[0], device=rows.device) // k
    sim = sim[:, view_of] if pairwise else sim[torch.arange(rows.shape[0], device=rows.device), view_of]
    return sim.clamp_min(0).squeeze()


def _baseline(score: torch.Tensor, enabled: bool, amplify: bool) -> torch.Tensor:
    """Subtract the mean over the last axis (the REINFORCE baseline), optionally divide by std + 1e-5; flat result."""
    if enabled and score.shape[-1] > 1:
        score = score - score.mean(dim=-1, keepdim=True)
        if amplify:
            score = score / (score.std(dim=-1, keepdim=True) + 1e-5)
    return score.reshape(-1)


class BaseRewards(nn.Module):
    """TPT/clip_reward.py:43-73."""

    def __init__(self) -> None:
        super().__init__()

    @torch.no_grad()
    def extract_image_features(self, images):
        pass

    @torch.no_grad()
    def extract_text_features(self, captions=None, tokenized_cap=None):
        pass

    @torch.no_grad()
    def set_class_features(self, classnames=None, tokenized_classes=None):
        self.class_features = self.extract_text_features(captions=classnames, tokenized_cap=tokenized_classes)

    @torch.no_grad()
    def set_image_features(self, images):
        self.image_features = self.extract_image_features(images)

    @torch.no_grad()
    def confidence_gap(self, predictions):
        value, index = torch.topk(predictions, 2, dim=-1)
        gap = value[:, 0] - value[:, 1]
        return gap - torch.mean(gap)


class CLIPRewards(BaseRewards):
    """TPT/clip_reward.py:76-177."""

    def __init__(self, device, arch="ViT-B/16", clipscore_weight=2.5, classification=True, amplify_rewards=False,
                 sample_k=5, reward_process=True, process_batch=False, default_resolutions=224) -> None:
        super().__init__()
        self.default_resolutions = default_resolutions
        self.clip_model, self.embed_dim, self.preprocess = clip_store.load(arch, device=device)
        runtime.SESSION.set_reward(self.clip_model)
        self.resolutions = self.clip_model.geometry.image_resolution
        self.clipscore_weight, self.device, self.classification = clipscore_weight, device, classification
        self.class_features = None
        self.image_features = None
        self.amplify_rewards, self.sample_k = amplify_rewards, sample_k
        self.reward_process, self.process_batch = reward_process, process_batch

    @torch.no_grad()
    def extract_image_features(self, images):
        """clip_reward.py:130-137: encode_image, float, L2 normalise."""
        # a resolution change (bicubic, align_corners=True, clip_reward.py:133-134) happens inside the engine
        return runtime.SESSION.engine(images.shape[0]).encode_image(L.REWARD, images)

    @torch.no_grad()
    def extract_text_features(self, captions=None, tokenized_cap=None):
        """clip_reward.py:139-150.  The class bank of the reward model is the tokenised bank of the student
        (tpt_cls_rl.py:183); the engine caches its features in rlcf_engine_set_class_bank."""
        if tokenized_cap is None:
            tokenized_cap = clip_store.tokenize(captions, truncate=True)          # clip_reward.py:142
        bank = runtime.SESSION.tokens
        if bank is None or bank.shape != tokenized_cap.shape or not torch.equal(bank, tokenized_cap.detach().cpu()):
            raise NotImplementedError("reward class bank must be the student's tokenized_prompts (tpt_cls_rl.py:183)")
        return runtime.SESSION.engine().reward_class_features()

    @torch.no_grad()
    def CLIPScore(self, class_index, images=None, image_features=None, captions=None, tokenized_cap=None, text_features=None,
                  pairwise=True):
        """clip_reward.py:111-128 (classification branch).  Stand-alone convenience on device tensors; the
        tuning loop gets the same numbers from the fused rlcf_reward_loss kernel."""
        feats = self.image_features if image_features is None else image_features
        return _clamped_scores(self.class_features, feats, class_index, self.sample_k, self.clipscore_weight, pairwise)

    @torch.no_grad()
    def rewards_post_process(self, clip_score):
        """clip_reward.py:152-165."""
        return _baseline(clip_score, bool(self.reward_process), bool(self.amplify_rewards))

    @torch.no_grad()
    def calulate_similarity(self):
        """clip_reward.py:167-177: (logits_per_image, logits_per_text) = (exp(logit_scale) * image @ class^T, its transpose), the
        pair tune_cls_kd.py:46 unpacks; logit_scale is the reward checkpoint's."""
        scale = float(self.clip_model.state_dict["logit_scale"].float().exp())
        logits_per_image = _gemm_nt(self.image_features, self.class_features, scale)
        return logits_per_image, logits_per_image.t()


class CLIPRewardsMultiple(BaseRewards):
    """TPT/clip_reward.py:180-307: CLIP reward from an ensemble of frozen CLIP models.  Every model lives in a reward slot of
    the shared HIP engine; the fused step mixes the per-model clamped scores in the reward kernel (rlcf_reward_loss_ensemble)."""

    def __init__(self, device, arch=("ViT-B/16", "RN50x64", "ViT-L/14"), clipscore_weight=2.5, classification=True,
                 amplify_rewards=False, sample_k=5, reward_process=True, process_batch=True, weighted_scores=True,
                 default_resolutions=224) -> None:
        super().__init__()
        self.default_resolutions = default_resolutions
        self.clip_models, self.preprocess, self.resolutions, weights = [], [], [], []
        for ar in arch:
            ckpt, _, pre = clip_store.load(ar, device=device)
            self.clip_models.append(ckpt)
            self.preprocess.append(pre)
            self.resolutions.append(ckpt.geometry.image_resolution)
            weights.append(CONFIDECES[ar])
        self.n_model = len(self.clip_models)
        self.weights = [round(x / sum(weights), 2) for x in weights]          # clip_reward.py:206
        self.clipscore_weight, self.device, self.classification = clipscore_weight, device, classification
        self.class_features = None
        self.image_features = None
        self.amplify_rewards, self.sample_k = amplify_rewards, sample_k
        self.reward_process, self.process_batch, self.weighted_scores = reward_process, process_batch, weighted_scores
        runtime.SESSION.set_rewards(self.clip_models, self.weights, mean=not weighted_scores)

    @torch.no_grad()
    def extract_image_features(self, images):
        """clip_reward.py:259-273: one normalised feature matrix per model (bicubic resize inside the engine)."""
        eng = runtime.SESSION.engine(images.shape[0])
        return [eng.encode_image(L.REWARD + i, images) for i in range(self.n_model)]

    @torch.no_grad()
    def extract_text_features(self, captions=None, tokenized_cap=None):
        """clip_reward.py:275-291."""
        if tokenized_cap is None:
            tokenized_cap = clip_store.tokenize(captions, truncate=True)
        bank = runtime.SESSION.tokens
        if bank is None or bank.shape != tokenized_cap.shape or not torch.equal(bank, tokenized_cap.detach().cpu()):
            raise NotImplementedError("reward class bank must be the student's tokenized_prompts (tpt_cls_rl.py:183)")
        eng = runtime.SESSION.engine()
        return [eng.reward_class_features(i) for i in range(self.n_model)]

    @torch.no_grad()
    def CLIPScore(self, class_index, images=None, image_features=None, captions=None, tokenized_cap=None, text_features=None,
                  pairwise=True):
        """clip_reward.py:227-257: per-model clamped similarity, then the weighted sum (or the mean) over models."""
        if pairwise:
            raise NotImplementedError               # as the reference (:240)
        per_model = torch.stack([_clamped_scores(c, f, class_index, self.sample_k, self.clipscore_weight, False)
                                 for c, f in zip(self.class_features, self.image_features)])
        if not self.weighted_scores:
            return per_model.mean(dim=0)
        w = per_model.new_tensor(self.weights)[:, None]
        return (w * per_model).sum(dim=0)

    rewards_post_process = CLIPRewards.rewards_post_process
