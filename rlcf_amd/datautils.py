"""Device-side mirror of the reference's per-sample view pipeline (TPT/data/datautils.py:76-128): `get_preaugment`,
`AugMixAugmenter`.  The reference runs RandomResizedCrop + RandomHorizontalFlip through torchvision/PIL on CPU workers, 63 times
per test image, and ships N float views (38.5 MB at N=64) to the GPU (TPT/tpt_cls_rl.py:236-248); here the decoded uint8 image
goes up once and `rlcf_make_views` (rlcf_amd/csrc/views.hip) writes the N normalised views in HBM, bit-exact with Pillow's
resampler.  The random boxes and flips are drawn on the host with the same torch generator calls torchvision makes, in the same
order, so a seeded run reproduces the reference's crops.

Not built: the AugMix op chains (`augmix=True`: only used for the fine-grained sets, tpt_cls_rl.py:149-150) and the BYOL-style
`hard_aug` recipe — both raise.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # TPT/tpt_cls_rl.py:132-133
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class RandomResizedCropParams:
    """torchvision.transforms.RandomResizedCrop.get_params + RandomHorizontalFlip's coin (torchvision 0.14.1, pinned at
    requirements.txt:37), drawing from torch's global generator exactly like the originals."""

    def __init__(self, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), flip_p: float = 0.5):
        self.scale, self.ratio, self.flip_p = scale, ratio, flip_p

    def __call__(self, height: int, width: int) -> Tuple[int, int, int, int, bool]:
        area = height * width
        log_ratio = torch.log(torch.tensor(self.ratio))
        box = None
        for _ in range(10):
            target_area = area * torch.empty(1).uniform_(self.scale[0], self.scale[1]).item()
            aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
            w = int(round(math.sqrt(target_area * aspect_ratio)))
            h = int(round(math.sqrt(target_area / aspect_ratio)))
            if 0 < w <= width and 0 < h <= height:
                i = torch.randint(0, height - h + 1, size=(1,)).item()
                j = torch.randint(0, width - w + 1, size=(1,)).item()
                box = (i, j, h, w)
                break
        if box is None:                       # fallback to a central crop
            in_ratio = float(width) / float(height)
            if in_ratio < min(self.ratio):
                w = width
                h = int(round(w / min(self.ratio)))
            elif in_ratio > max(self.ratio):
                h = height
                w = int(round(h * max(self.ratio)))
            else:
                w, h = width, height
            box = ((height - h) // 2, (width - w) // 2, h, w)
        flip = bool(torch.rand(1) < self.flip_p)
        return (*box, flip)


def get_preaugment(hard_aug=False, resolution=224, crop_min=0.2):
    """datautils.py:76-91: the parameter sampler of RandomResizedCrop(resolution) + RandomHorizontalFlip()."""
    if hard_aug:
        raise NotImplementedError("hard_aug (ColorJitter / Grayscale / GaussianBlur recipe, datautils.py:77-87) is not built")
    return RandomResizedCropParams()


def _as_u8_hwc(x) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        t = x
    else:                                     # PIL.Image or ndarray
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x.convert("RGB") if hasattr(x, "convert") else x)))
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("expected a decoded RGB image: uint8 [H, W, 3]")
    return t.contiguous()


def make_views(image, crops: Sequence[Tuple[int, int, int, int, bool]], resolution: int = 224, mean=CLIP_MEAN, std=CLIP_STD,
               device=None) -> torch.Tensor:
    """[1 + len(crops), 3, R, R] float32 on the GPU: view 0 = Resize(R, bicubic) + CenterCrop(R) of the image, the others its
    resized crops (bilinear) with optional flip; ToTensor + Normalize.  No CPU fallback."""
    if not torch.cuda.is_available():
        raise L.RlcfError("rlcf_amd.datautils.make_views needs a GPU: the HIP path has no CPU fallback")
    dev = torch.device(device or "cuda")
    img = _as_u8_hwc(image).to(dev)
    H, W = int(img.shape[0]), int(img.shape[1])
    n = len(crops)
    arr = (L.Crop * max(n, 1))(*[L.Crop(int(t), int(l), int(h), int(w), int(bool(f))) for t, l, h, w, f in crops])
    lib = L.lib()
    nbytes = int(lib.rlcf_make_views_scratch_bytes(H, n, resolution))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.empty(1 + n, 3, resolution, resolution, device=dev)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    L.check(lib.rlcf_make_views(img.data_ptr(), H, W, arr, n, resolution, m3, s3, out.data_ptr(), scratch.data_ptr(), nbytes,
                                torch.cuda.current_stream().cuda_stream), "make_views")
    return out


class AugMixAugmenter:
    """datautils.py:113-128.  `__call__(x)` returns `[image] + views` like the reference (a list of N tensors [3, R, R]) —
    here they are views of one device tensor; `views(x)` returns that [N, 3, R, R] tensor itself.  base_transform / preprocess
    are accepted for signature compatibility: their fixed content (Resize+CenterCrop, ToTensor+Normalize with the CLIP statistics,
    tpt_cls_rl.py:132-149) is what the kernel implements."""

    def __init__(self, base_transform=None, preprocess=None, n_views=2, augmix=False, severity=1, hard_aug=False, resolution=224,
                 device=None):
        if augmix:
            raise NotImplementedError("AugMix op chains (fine-grained sets only, tpt_cls_rl.py:149-150) are not built")
        self.n_views, self.resolution, self.device = n_views, resolution, device
        self.aug_list: List = []
        self.severity = severity
        self.preaugment = get_preaugment(hard_aug=hard_aug, resolution=resolution, crop_min=0.2)

    def views(self, x) -> torch.Tensor:
        img = _as_u8_hwc(x)
        H, W = int(img.shape[0]), int(img.shape[1])
        crops = [self.preaugment(H, W) for _ in range(self.n_views)]
        return make_views(img, crops, self.resolution, device=self.device)

    def __call__(self, x):
        return list(self.views(x).unbind(0))
