"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of how the reference builds the N views of one test image
(TPT/data/datautils.py:76-128 `get_preaugment` / `augmix` with an empty aug_list / `AugMixAugmenter.__call__`, transforms set up
at TPT/tpt_cls_rl.py:132-150): view 0 = Resize(224, bicubic) + CenterCrop(224); views 1..N-1 = RandomResizedCrop(224) (bilinear)
+ RandomHorizontalFlip; every view then ToTensor + Normalize(CLIP mean/std).

The arithmetic lives in third-party dependencies that are NOT under /root/reference: torchvision==0.14.1 (requirements.txt:37;
transform semantics restated from its published source: `_compute_resized_output_size`, `center_crop`,
`RandomResizedCrop.get_params`, `resized_crop` = crop then resize) and Pillow (requirements.txt:5, unpinned; 12.2.0 in this image)
whose `Image.resize` is the 8-bit two-pass separable resampler of libImaging/Resample.c (`precompute_coeffs`,
`normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc` / `Vertical_8bpc`, PRECISION_BITS = 32 - 8 - 2).
Pinned by tests/golden/views_*.npz, generated with Pillow itself (tests/golden/make_views_golden.py): bit-exact.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BILINEAR, BICUBIC = 0, 1
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # TPT/tpt_cls_rl.py:132-133
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _filter(kind: int, x: float) -> float:
    if kind == BILINEAR:                     # support 1
        x = abs(x)
        return 1.0 - x if x < 1.0 else 0.0
    a = -0.5                                 # bicubic, support 2 (Pillow's a = -0.5)
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int, kind: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size): -> xmin[out], count[out], int32 coeffs[out, ksize]."""
    support0 = 1.0 if kind == BILINEAR else 2.0
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = [_filter(kind, (x + lo - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(n):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        xmin[xx], cnt[xx] = lo, n
    return xmin, cnt, kk


def _pass(img: np.ndarray, out_size: int, kind: int) -> np.ndarray:
    """One separable pass along axis 1 of uint8 img [rows, in_size, C] -> [rows, out_size, C] (ImagingResampleHorizontal_8bpc)."""
    xmin, cnt, kk = resample_coeffs(img.shape[1], out_size, kind)
    out = np.empty((img.shape[0], out_size, img.shape[2]), np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        acc = np.full((img.shape[0], img.shape[2]), 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(cnt[xx]):
            acc += src[:, xmin[xx] + x, :] * int(kk[xx, x])
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_u8(img: np.ndarray, out_w: int, out_h: int, kind: int) -> np.ndarray:
    """Image.resize((out_w, out_h), kind) of a uint8 HWC image: horizontal pass, then vertical pass, each rounded to 8 bits;
    a pass whose size does not change is skipped (ImagingResample)."""
    h, w = img.shape[:2]
    x = img
    if out_w != w:
        x = _pass(x, out_w, kind)
    if out_h != h:
        x = _pass(x.transpose(1, 0, 2), out_h, kind).transpose(1, 0, 2)
    return np.ascontiguousarray(x)


def resized_output_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision `_compute_resized_output_size` for an int size: shorter side -> size, longer side int(size * long / short)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)       # (new_h, new_w)


def center_view_u8(img: np.ndarray, res: int) -> np.ndarray:
    """Resize(res, BICUBIC) + CenterCrop(res) (tpt_cls_rl.py:144-146)."""
    h, w = img.shape[:2]
    nh, nw = resized_output_size(h, w, res)
    x = resize_u8(img, nw, nh, BICUBIC)
    top, left = int(round((nh - res) / 2.0)), int(round((nw - res) / 2.0))
    return x[top: top + res, left: left + res]


def crop_view_u8(img: np.ndarray, top: int, left: int, h: int, w: int, flip: bool, res: int) -> np.ndarray:
    """resized_crop (crop, then bilinear resize to res x res) + optional horizontal flip (datautils.py:87-90)."""
    x = resize_u8(img[top: top + h, left: left + w], res, res, BILINEAR)
    return x[:, ::-1] if flip else x


def to_tensor_normalize(u8: np.ndarray) -> np.ndarray:
    """ToTensor (uint8 HWC -> float32 CHW / 255) + Normalize (tpt_cls_rl.py:147-149), float32 arithmetic."""
    x = u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    mean = np.asarray(CLIP_MEAN, np.float32)[:, None, None]
    std = np.asarray(CLIP_STD, np.float32)[:, None, None]
    return (x - mean) / std


def random_resized_crop_params(height: int, width: int, rng, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision RandomResizedCrop.get_params with `rng` standing in for torch's generator: rng.uniform(a, b) -> float,
    rng.randint(n) -> int in [0, n)."""
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * rng.uniform(scale[0], scale[1])
        aspect_ratio = math.exp(rng.uniform(log_ratio[0], log_ratio[1]))
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            return rng.randint(height - h + 1), rng.randint(width - w + 1), h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def make_views(img: np.ndarray, crops: List[Tuple[int, int, int, int, bool]], res: int = 224) -> np.ndarray:
    """AugMixAugmenter.__call__ with an empty aug_list (datautils.py:125-128): [image] + views -> float32 [1+len(crops), 3, res, res]."""
    out = [to_tensor_normalize(center_view_u8(img, res))]
    for top, left, h, w, flip in crops:
        out.append(to_tensor_normalize(crop_view_u8(img, top, left, h, w, flip, res)))
    return np.stack(out)
