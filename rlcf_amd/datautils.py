"""Device-side mirror of the reference's per-sample view pipeline (TPT/data/datautils.py:76-128): `get_preaugment`,
`AugMixAugmenter`.  The reference runs RandomResizedCrop + RandomHorizontalFlip through torchvision/PIL on CPU workers, 63 times
per test image, and ships N float views (38.5 MB at N=64) to the GPU (TPT/tpt_cls_rl.py:236-248); here the decoded uint8 image
goes up once and `rlcf_make_views` (rlcf_amd/csrc/views.hip) writes the N normalised views in HBM, bit-exact with Pillow's
resampler.  The random boxes and flips are drawn on the host with the same torch generator calls torchvision makes, in the same
order, so a seeded run reproduces the reference's crops.

`augmix=True` (the fine-grained sets, tpt_cls_rl.py:149-150, scripts/rlcf-prompt-fine.sh) adds the AugMix op chains of
datautils.py:94-110 / augmix_ops.py: the ops, levels, signs and the Dirichlet / Beta mixing weights are drawn here with the numpy
calls the reference makes (same order, numpy's global legacy stream, so `np.random.seed` reproduces the reference's chains) and
`rlcf_make_views_augmix` applies them on the device, bit-exact with Pillow.

`hard_aug=True` (get_preaugment(hard_aug=True), datautils.py:77-87; --hard_aug 1 of tune_cls_tpt.py / tune_cls_kd.py) is the BYOL-style
recipe: RandomResizedCrop(scale=(0.2, 1)), then ColorJitter (p 0.5), RandomGrayscale (p 0.2), GaussianBlur (p 0.1), then the flip.
`HardAugParams` makes torchvision's generator calls in torchvision's order; `rlcf_make_views_hard` applies the draws on the device,
bit-exact with Pillow's ImageEnhance / HSV / L arithmetic (the blur: torchvision's float32 tensor kernel).
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
        lr = torch.log(torch.tensor(self.ratio))           # float32, as torchvision computes it; the bounds reach uniform_ as Python floats
        self._log_ratio = (lr[0].item(), lr[1].item())
        # The draws run IN PLACE on two one-element scratch tensors: `torch.empty(1).uniform_(a, b)`, `torch.rand(1)`, `torch.randint(0, n, (1,))`
        # and `torch.exp(t)` are `t.uniform_(a, b)`, `t.uniform_(0, 1)`, `t.random_(0, n)` and `t.exp_()` on a fresh tensor — the same kernels, the
        # same words of the generator in the same order (tests/test_views.py compares the boxes AND the generator state afterwards) — at a
        # third of the dispatches: 63 views cost 0.4 instead of 0.9 ms of host time.  (One instance = one drawing thread.)
        self._f = torch.empty(1)
        self._i = torch.empty(1, dtype=torch.int64)
        self._fv, self._iv = self._f.numpy(), self._i.numpy()      # (views of the same memory: reading the drawn value without a tensor dispatch)
        self._p32 = float(torch.tensor(float(flip_p), dtype=torch.float32))      # (`tensor < p` compares in float32)

    def __call__(self, height: int, width: int) -> Tuple[int, int, int, int, bool]:
        box = self.draw_box(height, width)
        self._f.uniform_(0.0, 1.0)
        flip = self._fv.item() < self._p32
        return (*box, flip)

    def draw_box(self, height: int, width: int) -> Tuple[int, int, int, int]:
        area = height * width
        log_ratio = self._log_ratio
        box = None
        f, it, fv, iv = self._f, self._i, self._fv, self._iv
        s0, s1 = self.scale
        l0, l1 = log_ratio
        for _ in range(10):
            f.uniform_(s0, s1)
            target_area = area * fv.item()
            f.uniform_(l0, l1).exp_()
            aspect_ratio = fv.item()
            w = int(round(math.sqrt(target_area * aspect_ratio)))
            h = int(round(math.sqrt(target_area / aspect_ratio)))
            if 0 < w <= width and 0 < h <= height:
                it.random_(0, height - h + 1)
                i = iv.item()
                it.random_(0, width - w + 1)
                j = iv.item()
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
        return box


class HardAugParams:
    """The draws of get_preaugment(hard_aug=True) (datautils.py:77-87) for one view, from torch's global generator in the order
    torchvision 0.14.1 makes them: RandomResizedCrop.get_params (scale (crop_min, 1)); RandomApply.forward's `p < torch.rand(1)` and,
    if ColorJitter runs, ColorJitter.get_params (randperm(4), then brightness / contrast / saturation / hue uniforms);
    RandomGrayscale's `torch.rand(1) < p`; RandomApply's coin and GaussianBlur.get_params (sigma uniform); RandomHorizontalFlip's
    coin.  -> ((top, left, h, w, flip), plan) with plan = (order | None, b, c, s, hue_factor, gray, float32 [3, 3] kernel | None)."""

    def __init__(self, crop_min=0.2):
        self.box = RandomResizedCropParams(scale=(crop_min, 1.0), flip_p=0.0)

    @staticmethod
    def gaussian_kernel3(sigma: float) -> torch.Tensor:
        """torchvision _get_gaussian_kernel2d(kernel_size=3, sigma): float32 [3, 3]"""
        x = torch.linspace(-1.0, 1.0, steps=3)
        pdf = torch.exp(-0.5 * (x / sigma).pow(2))
        k1 = pdf / pdf.sum()
        return torch.mm(k1[:, None], k1[None, :])

    def __call__(self, height: int, width: int):
        box = self.box.draw_box(height, width)
        order = b = c = s = h = None
        if not (0.5 < torch.rand(1)):                                  # RandomApply([ColorJitter(0.4, 0.4, 0.2, 0.1)], p=0.5)
            order = torch.randperm(4).tolist()
            b = float(torch.empty(1).uniform_(0.6, 1.4))
            c = float(torch.empty(1).uniform_(0.6, 1.4))
            s = float(torch.empty(1).uniform_(0.8, 1.2))
            h = float(torch.empty(1).uniform_(-0.1, 0.1))
        gray = bool(torch.rand(1) < 0.2)                               # RandomGrayscale(p=0.2)
        kernel = None
        if not (0.1 < torch.rand(1)):                                  # RandomApply([GaussianBlur(3, sigma=(0.1, 2.0))], p=0.1)
            kernel = self.gaussian_kernel3(torch.empty(1).uniform_(0.1, 2.0).item())
        flip = bool(torch.rand(1) < 0.5)                               # RandomHorizontalFlip()
        return (*box, flip), (order, b, c, s, h, gray, kernel)


def get_preaugment(hard_aug=False, resolution=224, crop_min=0.2):
    """datautils.py:76-91: the parameter sampler of the pre-augmentation — RandomResizedCrop(resolution) + RandomHorizontalFlip(), or
    the hard_aug recipe (HardAugParams)."""
    if hard_aug:
        return HardAugParams(crop_min=crop_min)
    return RandomResizedCropParams()


AUG_OPS = ("autocontrast", "equalize", "posterize", "rotate", "solarize", "shear_x", "shear_y", "translate_x", "translate_y")
AUG_IMAGE_SIZE = 224                     # augmix_ops.py:21 (transform size and the scale of the translations)


def _rotate_coeffs(w: int, h: int, degrees: float):
    """the matrix PIL's Image.rotate(degrees) passes to transform(AFFINE) (rotation about the centre); None: no resampling"""
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


def draw_augmix_op(severity=1, rng=np.random):
    """The random draws of one `np.random.choice(aug_list)(x_aug, severity)` (datautils.py:106 with the ops of
    augmix_ops.py:56-115): -> (op id, integer parameter, six affine coefficients or None)."""
    op = int(rng.choice(len(AUG_OPS)))
    name = AUG_OPS[op]
    if name in ("autocontrast", "equalize"):
        return op, 0, None
    level = rng.uniform(low=0.1, high=severity)                 # sample_level
    S = AUG_IMAGE_SIZE
    if name == "posterize":
        return op, 4 - int(level * 4 / 10), None
    if name == "solarize":
        return op, 256 - int(level * 256 / 10), None
    if name == "rotate":
        deg = int(level * 30 / 10)
        if rng.uniform() > 0.5:
            deg = -deg
        co = _rotate_coeffs(S, S, deg)
        return (op, deg, co) if co is not None else (-1, 0, None)
    if name in ("shear_x", "shear_y"):
        lv = float(level) * 0.3 / 10.
        if rng.uniform() > 0.5:
            lv = -lv
        return op, 0, ((1, lv, 0, 0, 1, 0) if name == "shear_x" else (1, 0, 0, lv, 1, 0))
    lv = int(level * (S / 3) / 10)
    if rng.random_sample() > 0.5:
        lv = -lv
    return op, lv, ((1, 0, lv, 0, 1, 0) if name == "translate_x" else (1, 0, 0, 0, 1, lv))


def draw_augmix_plan(severity=1, rng=np.random):
    """The random draws of one `augmix` call (datautils.py:100-107): (w float32[3], m float32, three chains of 1-3 ops)."""
    w = np.float32(rng.dirichlet([1.0, 1.0, 1.0]))
    m = np.float32(rng.beta(1.0, 1.0))
    chains = [[draw_augmix_op(severity, rng) for _ in range(rng.randint(1, 4))] for _ in range(3)]
    return w, m, chains


def _as_u8_hwc(x) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        t = x
    else:                                     # PIL.Image or ndarray
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x.convert("RGB") if hasattr(x, "convert") else x)))
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("expected a decoded RGB image: uint8 [H, W, 3]")
    return t.contiguous()


_COPY_STREAMS = {}


def _upload(t: torch.Tensor, dev: torch.device) -> torch.Tensor:
    """Host -> device copy of the decoded image on a side stream: a copy issued on the compute stream would make the host wait for
    every launch already queued there (the previous test image's whole step) before it could draw the next image's views."""
    if t.is_cuda:
        return t.to(dev)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    cs = _COPY_STREAMS.get(key)
    if cs is None:
        cs = _COPY_STREAMS[key] = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)
    with torch.cuda.stream(cs):
        d = t.to(dev)                                      # (blocks the host only until THIS copy is done: the side stream is idle)
    cur.wait_stream(cs)
    d.record_stream(cur)
    return d


def hue_shift_u8(hue_factor: float) -> int:
    """np.uint8(hue_factor * 255) of torchvision's F_pil.adjust_hue under the numpy 1.x the reference pins: truncation toward zero,
    then modulo 256"""
    return int(hue_factor * 255) % 256


def _hard_structs(hard_plans):
    arr = (L.HardAug * len(hard_plans))()
    for v, (order, b, c, s, h, gray, kernel) in enumerate(hard_plans):
        o = arr[v]
        if order is None:
            o.order[0] = -1
        else:
            for q in range(4):
                o.order[q] = int(order[q])
            o.b, o.c, o.s, o.hue = float(b), float(c), float(s), hue_shift_u8(h)
        o.gray = int(bool(gray))
        o.blur = int(kernel is not None)
        if kernel is not None:
            k = torch.as_tensor(kernel, dtype=torch.float32).reshape(9).tolist()
            for q in range(9):
                o.k[q] = k[q]
    return arr


def make_views(image, crops: Sequence[Tuple[int, int, int, int, bool]], resolution: int = 224, mean=CLIP_MEAN, std=CLIP_STD,
               device=None, augmix_plans=None, hard_plans=None) -> torch.Tensor:
    """[1 + len(crops), 3, R, R] float32 on the GPU: view 0 = Resize(R, bicubic) + CenterCrop(R) of the image, the others its
    resized crops (bilinear) with optional flip; ToTensor + Normalize.  augmix_plans: one draw_augmix_plan() result per crop — the
    crop views then go through the AugMix loop.  No CPU fallback."""
    if not torch.cuda.is_available():
        raise L.RlcfError("rlcf_amd.datautils.make_views needs a GPU: the HIP path has no CPU fallback")
    dev = torch.device(device or "cuda")
    img = _upload(_as_u8_hwc(image), dev)
    H, W = int(img.shape[0]), int(img.shape[1])
    n = len(crops)
    arr = (L.Crop * max(n, 1))(*[L.Crop(int(t), int(l), int(h), int(w), int(bool(f))) for t, l, h, w, f in crops])
    lib = L.lib()
    nbytes = int(lib.rlcf_make_views_scratch_bytes(H, n, resolution))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.empty(1 + n, 3, resolution, resolution, device=dev)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    if hard_plans is not None and (len(hard_plans) != n or n == 0):
        raise ValueError("hard_plans: one plan per crop")
    if augmix_plans is not None:
        if len(augmix_plans) != n or n == 0:
            raise ValueError("augmix_plans: one plan per crop")
        if resolution != AUG_IMAGE_SIZE:
            raise ValueError("the AugMix ops work on 224 x 224 views (augmix_ops.py:21)")
        ops = (L.AugmixOp * (n * 9))()
        wv, mv = (C.c_float * (n * 3))(), (C.c_float * n)()
        for v, (w, m, chains) in enumerate(augmix_plans):
            mv[v] = float(m)
            for i in range(3):
                wv[v * 3 + i] = float(w[i])
                for j in range(3):
                    o = ops[(v * 3 + i) * 3 + j]
                    if j < len(chains[i]):
                        o.op, o.ip = int(chains[i][j][0]), int(chains[i][j][1])
                        if chains[i][j][2] is not None:
                            for q in range(6):
                                o.c[q] = float(chains[i][j][2][q])
                    else:
                        o.op = -1
        if hard_plans is not None:
            nbytes = int(lib.rlcf_make_views_hard_scratch_bytes(H, n, resolution))
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            L.check(lib.rlcf_make_views_hard(img.data_ptr(), H, W, arr, n, resolution, m3, s3, _hard_structs(hard_plans), ops, wv, mv,
                                             out.data_ptr(), scratch.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream),
                    "make_views_hard")
            return out
        nbytes = int(lib.rlcf_make_views_augmix_scratch_bytes(H, n, resolution))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        L.check(lib.rlcf_make_views_augmix(img.data_ptr(), H, W, arr, n, resolution, m3, s3, ops, wv, mv, out.data_ptr(),
                                           scratch.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream), "make_views_augmix")
        return out
    if hard_plans is not None:
        nbytes = int(lib.rlcf_make_views_hard_scratch_bytes(H, n, resolution))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        L.check(lib.rlcf_make_views_hard(img.data_ptr(), H, W, arr, n, resolution, m3, s3, _hard_structs(hard_plans), None, None, None,
                                         out.data_ptr(), scratch.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream),
                "make_views_hard")
        return out
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
        self.n_views, self.resolution, self.device = n_views, resolution, device
        self.aug_list: List = list(AUG_OPS) if augmix else []          # (names: the ops themselves run on the device)
        self.severity = severity
        self.hard_aug = bool(hard_aug)
        self.preaugment = get_preaugment(hard_aug=hard_aug, resolution=resolution, crop_min=0.2)

    def draw(self, H: int, W: int):
        """The HOST half of one call: the random draws of the n_views views of an H x W image — per view the pre-augmentation draws (torch's
        global generator, torchvision's order), then the AugMix draws (numpy's global stream).  -> (crops, augmix plans | None, hard plans |
        None).  Depends on the image's size only, so a loader may run it ahead of the loop (ViewPrefetcher)."""
        crops, plans, hard = [], [] if self.aug_list else None, [] if self.hard_aug else None
        for _ in range(self.n_views):
            if self.hard_aug:
                box, hp = self.preaugment(H, W)
                crops.append(box)
                hard.append(hp)
            else:
                crops.append(self.preaugment(H, W))
            if self.aug_list:
                plans.append(draw_augmix_plan(self.severity))
        return crops, plans, hard

    def apply(self, img: torch.Tensor, params) -> torch.Tensor:
        """The DEVICE half: the decoded uint8 image + the draws of `draw` -> [1 + n_views, 3, R, R] on the GPU (rlcf_make_views*)."""
        crops, plans, hard = params
        return make_views(img, crops, self.resolution, device=self.device, augmix_plans=plans, hard_plans=hard)

    def views(self, x) -> torch.Tensor:
        img = _as_u8_hwc(x)
        return self.apply(img, self.draw(int(img.shape[0]), int(img.shape[1])))

    def __call__(self, x):
        return list(self.views(x).unbind(0))



_UPLOAD_STREAMS: dict = {}


class ViewPrefetcher:
    """The reference draws a test image's 63 crop boxes (and AugMix plans) inside DataLoader workers — `DataLoader(val_dataset, ...,
    num_workers=args.workers)`, TPT/tpt_cls_rl.py:187-188, with the transform `AugMixAugmenter` running in `__getitem__`
    (TPT/data/datautils.py:113-128) — so the draws of image i + 1 happen while image i is being tuned.  The device-side augmenter
    cannot run in forked workers; this is its equivalent: ONE loader thread walks `dataset` (any iterable / indexable of (image,
    target)), decodes the image to uint8 [H, W, 3] and makes the host-side draws (`AugMixAugmenter.draw`: ~1 ms of generator calls per
    image) up to `depth` images ahead (a queued item is the uint8 image + its draws, ~0.5 MB: a pass of 32 images is drawn while the previous one runs); the consuming loop's thread only launches the device half (`apply`, ~0.1 ms of host time).
    One thread draws, in dataset order: a seeded run makes the same draws as the plain loop.  Yields what the reference's loader
    yields with batch size 1: ([view [1, 3, R, R]] * N, target)."""

    on_device = True            # the views are made on the device inside next(), on the CALLER's current stream (tpt_cls_rl._eval_in_flight)

    def __init__(self, dataset, augmenter: "AugMixAugmenter", depth: int = 32, as_list: bool = True):
        self.dataset, self.aug, self.depth, self.as_list = dataset, augmenter, max(1, int(depth)), as_list
        self.stats = {"items": 0, "wait_s": 0.0, "apply_s": 0.0}      # consuming thread: seconds blocked on the loader thread / in `apply`

    def __len__(self):
        return len(self.dataset)

    def __iter__(self):
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        dev = torch.device(getattr(self.aug, "device", None) or "cuda") if torch.cuda.is_available() else None
        up = None
        if dev is not None:                               # (one upload stream per device for the process: torch's pool of 32 is shared with everybody)
            key = dev.index if dev.index is not None else torch.cuda.current_device()
            up = _UPLOAD_STREAMS.get(key)
            if up is None:
                up = _UPLOAD_STREAMS[key] = torch.cuda.Stream(device=dev)

        def work():
            try:
                for image, target in self.dataset:
                    if stop.is_set():
                        return
                    img = _as_u8_hwc(image)
                    ev = None
                    if dev is not None and not img.is_cuda:
                        # the decoded image goes to the device HERE, images ahead of the loop: a copy from pageable memory blocks its caller
                        # until the device has run it, behind whatever shares the copy stream's hardware queue — with samples in flight that
                        # was a lane's whole step in one leg out of two (~10 ms per image inside `apply` on the loop's thread: 82 instead
                        # of 97 images/s, round 6).  Blocked here it costs the loader's lead, not the loop.  (Pinned staging is no way out
                        # on this platform: CPU writes into pinned memory ran at tens of MB/s — measured, three variants.)
                        with torch.cuda.stream(up):
                            img = img.to(dev)
                            ev = torch.cuda.Event()
                            ev.record()
                    item = (img, self.aug.draw(int(img.shape[0]), int(img.shape[1])), target, ev)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                q.put(None)
            except BaseException as exc:                  # noqa: BLE001 — re-raised in the consuming thread
                q.put(exc)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        try:
            import time
            while True:
                t0 = time.perf_counter()
                item = q.get()
                t1 = time.perf_counter()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                img, params, target, ev = item
                if ev is not None:                        # (device-side: the copy ran images ago)
                    cur = torch.cuda.current_stream(dev)
                    cur.wait_event(ev)
                    img.record_stream(cur)
                v = self.aug.apply(img, params)
                self.stats["items"] += 1
                self.stats["wait_s"] += t1 - t0
                self.stats["apply_s"] += time.perf_counter() - t1
                t = target if isinstance(target, torch.Tensor) else torch.tensor([int(target)])
                yield ([x.unsqueeze(0) for x in v.unbind(0)] if self.as_list else v), t
        finally:
            stop.set()
            th.join(timeout=5)
