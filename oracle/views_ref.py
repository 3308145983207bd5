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


# ------------------------------------------------------------------------------------------------------------------------------
# AugMix op chains (TPT/data/datautils.py:94-110 `augmix` with aug_list = TPT/data/augmix_ops.py:144-147 `augmentations`; switched
# on for the fine-grained sets, tpt_cls_rl.py:149-150, i.e. by scripts/rlcf-prompt-fine.sh).  The ops are thin wrappers around
# Pillow (absent from /root/reference): ImageOps.autocontrast / equalize / posterize / solarize (per-band 256-entry look-up tables,
# ImageOps.py) and Image.rotate / Image.transform(AFFINE, BILINEAR) (libImaging/Geometry.c: affine_transform + bilinear_filter32RGB,
# double arithmetic, result truncated to 8 bits, pixels whose source point falls outside the image filled with 0).
# Pinned by tests/golden/augmix_*.npz, generated with Pillow through the reference's own op functions.
AUG_OPS = ("autocontrast", "equalize", "posterize", "rotate", "solarize", "shear_x", "shear_y", "translate_x", "translate_y")
AUG_IMAGE_SIZE = 224                     # augmix_ops.py:21


def _apply_lut(u8: np.ndarray, lut: np.ndarray) -> np.ndarray:
    """Image.point with a 3 x 256 table: band b of the output = lut[b][band b of the input]."""
    out = np.empty_like(u8)
    for b in range(3):
        out[..., b] = lut[b][u8[..., b]]
    return out


def lut_autocontrast(u8: np.ndarray) -> np.ndarray:
    """ImageOps.autocontrast(image, cutoff=0): per band, stretch [lowest, highest occupied level] to [0, 255]."""
    lut = np.empty((3, 256), np.uint8)
    for b in range(3):
        h = np.bincount(u8[..., b].reshape(-1), minlength=256)
        nz = np.nonzero(h)[0]
        lo, hi = int(nz[0]), int(nz[-1])
        if hi <= lo:
            lut[b] = np.arange(256)
        else:
            scale = 255.0 / (hi - lo)
            offset = -lo * scale
            for ix in range(256):
                v = int(ix * scale + offset)
                lut[b, ix] = 0 if v < 0 else 255 if v > 255 else v
    return lut


def lut_equalize(u8: np.ndarray) -> np.ndarray:
    """ImageOps.equalize(image): per band histogram equalisation with integer arithmetic."""
    lut = np.empty((3, 256), np.uint8)
    for b in range(3):
        h = np.bincount(u8[..., b].reshape(-1), minlength=256)
        histo = h[h > 0]
        step = 0 if len(histo) <= 1 else (int(histo.sum()) - int(histo[-1])) // 255
        if not step:
            lut[b] = np.arange(256)
        else:
            n = step // 2
            for i in range(256):
                lut[b, i] = min(n // step, 255)          # (n // step <= 255 by construction)
                n += int(h[i])
    return lut


def lut_posterize(bits: int) -> np.ndarray:
    """ImageOps.posterize: keep the `bits` most significant bits."""
    mask = ~(2 ** (8 - bits) - 1)
    return np.tile((np.arange(256) & mask).astype(np.uint8), (3, 1))


def lut_solarize(threshold: int) -> np.ndarray:
    """ImageOps.solarize: invert every level >= threshold."""
    i = np.arange(256)
    return np.tile(np.where(i < threshold, i, 255 - i).astype(np.uint8), (3, 1))


def rotate_coeffs(w: int, h: int, degrees: float):
    """The affine matrix Image.rotate(degrees) hands to transform() (Image.py: rotation about the centre, no expand); None for
    the angles it serves by transposition (0 / 180, and 90 / 270 of a square image)."""
    angle = degrees % 360.0
    if angle == 0:
        return None
    cx, cy = w / 2.0, h / 2.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return tuple(m)


def affine_bilinear_u8(u8: np.ndarray, c) -> np.ndarray:
    """Image.transform(size, AFFINE, c, BILINEAR) for an output of the input's size (Geometry.c: ImagingGenericTransform with
    affine_transform and bilinear_filter32RGB)."""
    H, W = u8.shape[:2]
    a0, a1, a2, a3, a4, a5 = [float(v) for v in c]
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64) + 0.5, np.arange(W, dtype=np.float64) + 0.5, indexing="ij")
    xin = a0 * xx + a1 * yy + a2
    yin = a3 * xx + a4 * yy + a5
    inside = (xin >= 0.0) & (xin < W) & (yin >= 0.0) & (yin < H)
    xs, ys = xin - 0.5, yin - 0.5
    x, y = np.floor(xs), np.floor(ys)
    dx, dy = xs - x, ys - y
    x, y = x.astype(np.int64), y.astype(np.int64)
    x0, x1 = np.clip(x, 0, W - 1), np.clip(x + 1, 0, W - 1)
    yc = np.clip(y, 0, H - 1)
    src = u8.astype(np.float64)
    out = np.zeros_like(u8)
    for b in range(3):
        p = src[..., b]
        v1 = p[yc, x0] + (p[yc, x1] - p[yc, x0]) * dx
        has2 = (y + 1 >= 0) & (y + 1 < H)
        y2 = np.clip(y + 1, 0, H - 1)
        v2 = np.where(has2, p[y2, x0] + (p[y2, x1] - p[y2, x0]) * dx, v1)
        v = v1 + (v2 - v1) * dy
        out[..., b] = np.where(inside, v, 0.0).astype(np.uint8)              # (UINT8) truncation
    return out


def draw_augmix_op(rng, severity):
    """One `np.random.choice(aug_list)(x_aug, severity)` call of datautils.py:106 reduced to its random draws (same calls, same
    order, on numpy's legacy global stream when rng is np.random): -> (op name, integer parameter, affine coefficients or None)."""
    op = AUG_OPS[int(rng.choice(len(AUG_OPS)))]
    if op in ("autocontrast", "equalize"):
        return op, 0, None
    level = rng.uniform(low=0.1, high=severity)                               # sample_level, augmix_ops.py:52-53
    S = AUG_IMAGE_SIZE
    if op == "posterize":
        return op, 4 - int(level * 4 / 10), None
    if op == "solarize":
        return op, 256 - int(level * 256 / 10), None
    if op == "rotate":
        deg = int(level * 30 / 10)
        if rng.uniform() > 0.5:
            deg = -deg
        return op, deg, rotate_coeffs(S, S, deg)
    if op in ("shear_x", "shear_y"):
        lv = float(level) * 0.3 / 10.
        if rng.uniform() > 0.5:
            lv = -lv
        return op, 0, ((1, lv, 0, 0, 1, 0) if op == "shear_x" else (1, 0, 0, lv, 1, 0))
    lv = int(level * (S / 3) / 10)                                            # translate_x / translate_y
    if rng.random_sample() > 0.5:
        lv = -lv
    return op, lv, ((1, 0, lv, 0, 1, 0) if op == "translate_x" else (1, 0, 0, 0, 1, lv))


def apply_augmix_op(u8: np.ndarray, op: str, ip: int, coeffs) -> np.ndarray:
    if op == "autocontrast":
        return _apply_lut(u8, lut_autocontrast(u8))
    if op == "equalize":
        return _apply_lut(u8, lut_equalize(u8))
    if op == "posterize":
        return _apply_lut(u8, lut_posterize(ip))
    if op == "solarize":
        return _apply_lut(u8, lut_solarize(ip))
    if coeffs is None:                                                         # rotate by 0 degrees: Image.rotate returns a copy
        return u8.copy()
    return affine_bilinear_u8(u8, coeffs)


def draw_augmix_plan(rng, severity=1):
    """The random draws of one `augmix` call (datautils.py:100-107) -> (w float32[3], m float32, 3 chains of 1-3 ops)."""
    w = np.float32(rng.dirichlet([1.0, 1.0, 1.0]))
    m = np.float32(rng.beta(1.0, 1.0))
    chains = []
    for _ in range(3):
        chains.append([draw_augmix_op(rng, severity) for _ in range(rng.randint(1, 4))])
    return w, m, chains


def augmix_view(x_orig_u8: np.ndarray, plan) -> np.ndarray:
    """datautils.py:94-110 after the pre-augmentation: float32 [3, R, R] = m * pre(x) + (1 - m) * sum_i w_i * pre(chain_i(x))."""
    w, m, chains = plan
    xp = to_tensor_normalize(x_orig_u8)
    mix = np.zeros_like(xp)
    for i, chain in enumerate(chains):
        x = x_orig_u8
        for op, ip, coeffs in chain:
            x = apply_augmix_op(x, op, ip, coeffs)
        mix = mix + w[i] * to_tensor_normalize(x)
    return m * xp + (np.float32(1) - m) * mix


# ------------------------------------------------------------------------------------------------------------------------------
# hard_aug: the BYOL-style pre-augmentation (TPT/data/datautils.py:77-87, get_preaugment(hard_aug=True), reached through
# --hard_aug 1 of tune_cls_tpt.py / tune_cls_kd.py): RandomResizedCrop(224, scale=(0.2, 1)) -> RandomApply([ColorJitter(0.4, 0.4,
# 0.2, 0.1)], p=0.5) -> RandomGrayscale(p=0.2) -> RandomApply([GaussianBlur(3, sigma=(0.1, 2))], p=0.1) -> RandomHorizontalFlip.
# On PIL images torchvision 0.14.1 (absent from /root/reference; restated from its published source) maps these to Pillow:
#   adjust_brightness / contrast / saturation = ImageEnhance.Brightness / Contrast / Color = Image.blend(degenerate, image, factor)
#     (libImaging/Blend.c: float arithmetic, truncation, clipping only when the factor leaves [0, 1]); degenerate = black / the solid
#     grey int(mean(L) + 0.5) / the image's own L; L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16 (libImaging/Convert.c rgb2l);
#   adjust_hue = convert("HSV"), h += uint8(hue_factor * 255) (wraps), convert back (Convert.c rgb2hsv_row / hsv2rgb);
#   rgb_to_grayscale(3 channels) = L stacked three times;
#   gaussian_blur = torchvision's TENSOR kernel on pil_to_tensor(img): float32 3x3 kernel (outer product of the normalised 1-d
#     pdf), reflect padding, conv2d, torch.round (half to even) back to uint8.
# Pinned by tests/golden/hardaug_*.npz (generated with Pillow + torch, tests/golden/make_views_golden.py) and, for the colour
# conversions, exhaustively against the installed Pillow (tests/test_views.py).  Blur: nine fused multiply-adds in row-major tap order
# (= torch's CPU conv2d here); another summation order can move a value that lies within ~1e-5 of x.5 by one level (~3 pixels per
# million), which is what a reference run on another conv2d backend would also see.
def rgb_to_l(u8: np.ndarray) -> np.ndarray:
    x = u8.astype(np.int64)
    return ((x[..., 0] * 19595 + x[..., 1] * 38470 + x[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend_u8(in1: np.ndarray, in2: np.ndarray, alpha: float) -> np.ndarray:
    """Image.blend(im1, im2, alpha) (libImaging/Blend.c): out = (UINT8)(in1 + alpha * (in2 - in1)) in C float arithmetic."""
    a = np.float32(alpha)
    if a == 0.0:
        return in1.copy()
    if a == 1.0:
        return in2.copy()
    i1, i2 = in1.astype(np.int32), in2.astype(np.int32)
    t = i1.astype(np.float32) + a * (i2 - i1).astype(np.float32)              # float32 product, float32 sum (no contraction)
    if 0.0 <= a <= 1.0:
        return t.astype(np.int32).astype(np.uint8)                             # (truncation; the value lies in [0, 255])
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def adjust_brightness_u8(u8: np.ndarray, factor: float) -> np.ndarray:
    return blend_u8(np.zeros_like(u8), u8, factor)


def adjust_contrast_u8(u8: np.ndarray, factor: float) -> np.ndarray:
    lum = rgb_to_l(u8)
    mean = int(float(lum.astype(np.int64).sum()) / lum.size + 0.5)            # int(ImageStat.Stat(L).mean[0] + 0.5)
    return blend_u8(np.full_like(u8, mean), u8, factor)


def adjust_saturation_u8(u8: np.ndarray, factor: float) -> np.ndarray:
    return blend_u8(np.repeat(rgb_to_l(u8)[..., None], 3, axis=-1), u8, factor)


def rgb_to_hsv_u8(u8: np.ndarray) -> np.ndarray:
    """Convert.c rgb2hsv_row: float (C float) intermediates, h through double (fmod(h / 6.0 + 1.0, 1.0)), truncation to 8 bits."""
    r, g, b = (u8[..., i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    safe = np.where(cr == 0, np.float32(1), cr)
    s = cr / np.where(maxc == 0, 1, maxc).astype(np.float32)
    rc, gc, bc = ((maxc - c).astype(np.float32) / safe for c in (r, g, b))
    rd, gd, bd = rc.astype(np.float64), gc.astype(np.float64), bc.astype(np.float64)       # (`2.0 + rc - bc`: the literal makes it double)
    h = np.where(r == maxc, (bc - gc).astype(np.float64), np.where(g == maxc, 2.0 + rd - bd, 4.0 + gd - rd)).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    grey = minc == maxc
    return np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], axis=-1).astype(np.uint8)


def hsv_to_rgb_u8(hsv: np.ndarray) -> np.ndarray:
    """Convert.c hsv2rgb: i = floor(h * 6 / 255), f = the remainder (C float), p / q / t = round(v * (1 - ...)) (half away from zero)."""
    h, s, v = (hsv[..., i].astype(np.float32) for i in range(3))
    h6 = h.astype(np.float64) * 6.0 / 255.0
    i = np.floor(h6).astype(np.int32)
    f = (h6 - i).astype(np.float32)
    fs = (s.astype(np.float64) / 255.0).astype(np.float32)
    vd = v.astype(np.float64)

    def rnd(x):                                                                # C round(): half away from zero (x >= 0 here)
        return np.clip(np.floor(x + 0.5).astype(np.int32), 0, 255)
    p = rnd(vd * (1.0 - fs.astype(np.float64)))
    q = rnd(vd * (1.0 - (fs * f).astype(np.float64)))
    t = rnd(vd * (1.0 - (fs * (np.float32(1.0) - f)).astype(np.float64)))
    vi = hsv[..., 2].astype(np.int32)
    k = i % 6
    r = np.choose(k, [vi, q, p, p, t, vi])
    g = np.choose(k, [t, vi, vi, q, p, p])
    b = np.choose(k, [p, p, t, vi, vi, q])
    grey = hsv[..., 1] == 0
    return np.stack([np.where(grey, vi, r), np.where(grey, vi, g), np.where(grey, vi, b)], axis=-1).astype(np.uint8)


def hue_shift_u8(hue_factor: float) -> int:
    """np.uint8(hue_factor * 255) of torchvision's F_pil.adjust_hue: truncation toward zero, then modulo 256."""
    return int(hue_factor * 255) % 256


def adjust_hue_u8(u8: np.ndarray, hue_factor: float) -> np.ndarray:
    hsv = rgb_to_hsv_u8(u8)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_u8(hue_factor)).astype(np.uint8)
    return hsv_to_rgb_u8(hsv)


def to_grayscale3_u8(u8: np.ndarray) -> np.ndarray:
    return np.repeat(rgb_to_l(u8)[..., None], 3, axis=-1)


def gaussian_kernel3(sigma: float) -> np.ndarray:
    """torchvision _get_gaussian_kernel2d for kernel_size 3: float32 [3, 3], with torch's own float32 ops (numpy's exp differs from
    torch's in the last bit for some sigma)."""
    import torch
    x = torch.linspace(-1.0, 1.0, steps=3)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    k1 = pdf / pdf.sum()
    return torch.mm(k1[:, None], k1[None, :]).numpy()


def gaussian_blur3_u8(u8: np.ndarray, kernel: np.ndarray) -> np.ndarray:
    """reflect padding; acc = fma(k[i][j], x, acc) over the nine taps in row-major order (what torch's CPU conv2d computes, bit for
    bit, in this image's build: checked on the float outputs); round half to even, back to uint8.  The fused multiply-add is
    emulated in float64: the product of two float32 values is exact there, and so is its sum with acc whenever they overlap."""
    x = np.pad(u8.astype(np.float32), ((1, 1), (1, 1), (0, 0)), mode="reflect").astype(np.float64)
    H, W = u8.shape[:2]
    acc = np.zeros(u8.shape, np.float32)
    for i in range(3):
        for j in range(3):
            acc = (acc.astype(np.float64) + np.float64(kernel[i, j]) * x[i: i + H, j: j + W]).astype(np.float32)
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


JITTER_FNS = ("brightness", "contrast", "saturation", "hue")


def hard_aug_u8(u8: np.ndarray, plan) -> np.ndarray:
    """The colour / blur part of the hard recipe on the 8-bit resized crop.  plan = (order, b, c, s, h, gray, kernel): order = the
    randperm(4) of ColorJitter.get_params or None (RandomApply skipped it), gray = RandomGrayscale's coin, kernel = the float32
    [3, 3] blur kernel or None.  The flip that follows commutes with all of it."""
    order, b, c, s, h, gray, kernel = plan
    x = u8
    if order is not None:
        for fn in order:
            x = (adjust_brightness_u8(x, b) if fn == 0 else adjust_contrast_u8(x, c) if fn == 1 else
                 adjust_saturation_u8(x, s) if fn == 2 else adjust_hue_u8(x, h))
    if gray:
        x = to_grayscale3_u8(x)
    if kernel is not None:
        x = gaussian_blur3_u8(x, kernel)
    return x


def draw_hard_plan(rng):
    """The draws between the crop box and the flip coin (torchvision 0.14.1 transforms.py: RandomApply.forward, ColorJitter.get_params,
    RandomGrayscale.forward, GaussianBlur.get_params), `rng` standing in for torch's global generator: rng.rand() -> float in [0, 1),
    rng.randperm(n) -> list, rng.uniform(a, b) -> float."""
    order = b = c = s = h = None
    if not (0.5 < rng.rand()):
        order = list(rng.randperm(4))
        b, c, s, h = rng.uniform(0.6, 1.4), rng.uniform(0.6, 1.4), rng.uniform(0.8, 1.2), rng.uniform(-0.1, 0.1)
    gray = rng.rand() < 0.2
    kernel = None
    if not (0.1 < rng.rand()):
        kernel = gaussian_kernel3(rng.uniform(0.1, 2.0))
    return order, b, c, s, h, gray, kernel
