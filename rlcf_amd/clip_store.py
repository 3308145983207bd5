"""CLIP checkpoints and tokenisation for the host-side mirror of the reference API.

The reference `clip.load(name, device, download_root)` (TPT/clip/clip.py:94-194) downloads /
opens OpenAI JIT archives and returns ``(model, embed_dim, preprocess)``.  There is no network
here: a checkpoint is an OpenAI-layout ``state_dict`` (the key layout ``build_model`` accepts,
TPT/clip/model.py:399-436) that is either registered in-process (`register_checkpoint`) or read from
``<clip_root>/<name with / -> ->.pt`` — a TorchScript archive as OpenAI publishes them (``torch.jit.load``,
clip.py:119-131), a ``torch.save`` of the dict, or of ``{"state_dict": dict}``; the geometry is inferred from
the tensor shapes exactly as ``build_model`` does (:400-422).  `tokenize` is pluggable (`set_tokenizer`): the BPE vocabulary file is data the user owns.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Tuple, Union

import torch

from .synth import GEOMETRIES, ClipGeometry

_CHECKPOINTS: Dict[str, Tuple[ClipGeometry, Dict[str, torch.Tensor]]] = {}
_TOKENIZER: Optional[Callable] = None
_TOKENIZER_TRUNCATES = True
CLIP_ROOT = os.environ.get("RLCF_CLIP_ROOT", "")


def register_checkpoint(name: str, geometry: ClipGeometry, state_dict: Dict[str, torch.Tensor]) -> None:
    _CHECKPOINTS[name] = (geometry, state_dict)


def geometry_from_state_dict(sd: Dict[str, torch.Tensor]) -> ClipGeometry:
    """Shape inference of build_model (TPT/clip/model.py:400-422)."""
    tw = sd["ln_final.weight"].shape[0]
    tl = len(set(k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")))
    if "visual.proj" not in sd:            # ModifiedResNet branch (:408-415)
        counts = tuple(len(set(k.split(".")[2] for k in sd if k.startswith(f"visual.layer{b}"))) for b in (1, 2, 3, 4))
        out_w = round((sd["visual.attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
        assert out_w ** 2 + 1 == sd["visual.attnpool.positional_embedding"].shape[0]
        return ClipGeometry(sd["text_projection"].shape[1], out_w * 32, counts, sd["visual.layer1.0.conv1.weight"].shape[0], None,
                            sd["positional_embedding"].shape[0], sd["token_embedding.weight"].shape[0], tw, tw // 64, tl)
    vw = sd["visual.conv1.weight"].shape[0]
    vl = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    ps = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    return ClipGeometry(sd["text_projection"].shape[1], ps * grid, vl, vw, ps, sd["positional_embedding"].shape[0],
                        sd["token_embedding.weight"].shape[0], tw, tw // 64, tl)


def load(name: str, device: Union[str, torch.device] = "cuda", jit: bool = False, download_root: Optional[str] = None):
    """-> (checkpoint handle, embed_dim, preprocess=None), the 3-tuple of the reference (clip.py:137-142)."""
    if name in _CHECKPOINTS:
        geo, sd = _CHECKPOINTS[name]
    else:
        root = download_root or CLIP_ROOT
        path = os.path.join(root, name.replace("/", "-") + ".pt") if root else ""
        if not path or not os.path.isfile(path):
            raise RuntimeError(f"Model {name} not found; registered = {sorted(_CHECKPOINTS)}; "
                               f"looked for {path or '<RLCF_CLIP_ROOT unset>'}")
        try:                                           # OpenAI releases are TorchScript archives (clip.py:119-131)
            sd = torch.jit.load(path, map_location="cpu").state_dict()
        except RuntimeError:
            sd = torch.load(path, map_location="cpu")
            sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd.state_dict()
        sd = {k: v.float() for k, v in sd.items() if k not in ("input_resolution", "context_length", "vocab_size")}
        geo = geometry_from_state_dict(sd)
    return ClipCheckpoint(name, geo, sd), geo.embed_dim, None


class ClipCheckpoint:
    def __init__(self, name: str, geometry: ClipGeometry, state_dict: Dict[str, torch.Tensor]):
        self.name, self.geometry, self.state_dict = name, geometry, state_dict


def set_tokenizer(fn: Callable) -> None:
    """fn(texts: str | list[str], context_length=77[, truncate=False]) -> int64 [n, context_length] (clip.tokenize contract).  Whether
    the tokenizer takes `truncate` is decided HERE, once, from its signature — not by catching TypeError round the call, which would
    also swallow a TypeError raised inside a three-argument tokenizer and silently repeat the call without truncation."""
    global _TOKENIZER, _TOKENIZER_TRUNCATES
    import inspect
    try:
        ps = inspect.signature(fn).parameters
        _TOKENIZER_TRUNCATES = ("truncate" in ps or len([p for p in ps.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]) >= 3
                                or any(p.kind == p.VAR_POSITIONAL for p in ps.values()))
    except (TypeError, ValueError):        # builtins without a signature: the clip.tokenize contract has three arguments
        _TOKENIZER_TRUNCATES = True
    _TOKENIZER = fn


def tokenize(texts: Union[str, List[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
    if _TOKENIZER is None:
        raise RuntimeError("no tokenizer installed: call rlcf_amd.clip_store.set_tokenizer(fn) with a CLIP BPE tokenizer "
                           "(or a SyntheticBank.tokenize for seeded runs)")
    if _TOKENIZER_TRUNCATES:
        return _TOKENIZER(texts, context_length, truncate)
    return _TOKENIZER(texts, context_length)       # a two-argument tokenizer: over-long texts raise in it, as with truncate=False


class SyntheticBank:
    """Seeded stand-in for the BPE tokenizer: class name 'c<i>' -> row i of rlcf_amd.synth's token bank."""

    def __init__(self, geometry: ClipGeometry, n_cls: int, n_ctx: int = 4, seed: int = 7):
        from . import synth
        self.geometry, self.n_ctx = geometry, n_ctx
        self.tokens = synth.make_token_bank(geometry, n_cls, seed=seed, n_ctx=n_ctx)
        self.ctx_ids = synth.ctx_token_ids_default(geometry, n_ctx)
        self.classnames = [f"c{i}" for i in range(n_cls)]

    def tokenize(self, texts, context_length: int = 77, truncate: bool = False) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        rows = []
        for t in texts:
            t = t.strip()
            if t.endswith("."):
                rows.append(self.tokens[int(t.rstrip(".").split("c")[-1])])
            else:                                   # the ctx_init words alone
                r = torch.zeros(self.geometry.context_length, dtype=torch.int64)
                ids = [self.geometry.vocab_size - 2, *self.ctx_ids, self.geometry.vocab_size - 1]
                r[: len(ids)] = torch.tensor(ids)
                rows.append(r)
        return torch.stack(rows)


__all__ = ["register_checkpoint", "load", "tokenize", "set_tokenizer", "SyntheticBank", "ClipCheckpoint", "GEOMETRIES",
           "geometry_from_state_dict"]
