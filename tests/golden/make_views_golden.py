"""Golden vectors for device-side view generation, produced with Pillow itself (the resampler the reference's torchvision
transforms call; TPT/data/datautils.py:76-128, tpt_cls_rl.py:132-150) on seeded synthetic uint8 images.

    python tests/golden/make_views_golden.py

Each case: image size (H, W), output resolution, crop boxes (top, left, h, w, flip).  The fixture stores the crop list and, per
view, the uint8 result of the PIL pipeline (Resize(bicubic)+CenterCrop for view 0; crop + resize(bilinear) + FLIP_LEFT_RIGHT for the
others) — whole for small outputs, SHA-1 + first 16x16 block for 224x224 outputs.  ToTensor/Normalize are exact float32 formulas.
"""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from rlcf_amd import synth  # noqa: E402

CASES = {
    # name: (H, W, res, [(top, left, h, w, flip), ...])
    "views_small": (48, 64, 32, [(0, 0, 48, 64, False), (5, 7, 20, 31, True), (10, 3, 9, 12, False), (0, 30, 48, 17, True), (40, 60, 8, 4, False)]),
    "views_up": (20, 23, 32, [(0, 0, 20, 23, False), (3, 2, 11, 7, True), (19, 22, 1, 1, False)]),
    "views_imagenet": (375, 500, 224, [(0, 0, 375, 500, False), (37, 101, 240, 313, True), (300, 10, 60, 45, False), (1, 2, 373, 300, True)]),
    "views_tall": (640, 427, 224, [(100, 50, 333, 250, False), (0, 0, 640, 427, True)]),
}


def synth_image(name, h, w):
    """Smooth-ish seeded uint8 image: low-frequency ramps plus hash noise, so that resampling errors are visible."""
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    noise = (synth.raw_u32(5, "img." + name, h * w * 3, 0).numpy().reshape(h, w, 3) % 97).astype(np.int64)
    base = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], axis=-1)
    return ((base + noise) % 256).astype(np.uint8)


def resized_output_size(h, w, size):
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


# hard_aug recipe (datautils.py:77-87): name -> (H, W, res, [(top, left, h, w, flip, order | None, b, c, s, h, gray, sigma | None)])
HARD_CASES = {
    "hardaug_small": (48, 64, 32, [
        (0, 0, 48, 64, False, (0, 1, 2, 3), 1.3, 0.7, 1.1, 0.05, False, None),
        (5, 7, 20, 31, True, (3, 2, 1, 0), 0.65, 1.39, 0.81, -0.1, False, 1.0),
        (10, 3, 9, 12, False, None, None, None, None, None, True, None),
        (0, 30, 48, 17, True, None, None, None, None, None, False, 0.1),
        (40, 60, 8, 4, False, (1, 3, 0, 2), 1.0, 1.0, 1.0, 0.0, True, 2.0),
        (2, 2, 40, 50, False, (2, 0, 3, 1), 1.4, 0.6, 1.2, -0.02, False, 0.55)]),
    "hardaug_imagenet": (375, 500, 224, [
        (37, 101, 240, 313, True, (1, 0, 3, 2), 0.8241, 1.2203, 0.9417, 0.0831, False, None),
        (300, 10, 60, 45, False, (3, 1, 2, 0), 1.3711, 0.6102, 1.1903, -0.0612, True, 1.7312),
        (1, 2, 373, 300, True, None, None, None, None, None, False, 0.3149),
        (0, 0, 375, 500, False, (0, 2, 1, 3), 0.6, 1.4, 0.8, 0.1, False, 0.9)]),
}


def hard_plan_for_oracle(entry):
    """(order, b, c, s, h, gray, kernel) of oracle.views_ref.hard_aug_u8 from a HARD_CASES entry (factors as float32 values: the
    factors torchvision draws are float32 uniforms)."""
    from oracle import views_ref as V
    order, b, c, s, h, gray, sigma = entry[5:]
    f32 = lambda v: None if v is None else float(np.float32(v))
    return (order, f32(b), f32(c), f32(s), f32(h), gray, None if sigma is None else V.gaussian_kernel3(f32(sigma)))


def pil_hard_view(img, entry, res):
    """The reference's pipeline on a PIL image, written with Pillow + torch as torchvision 0.14.1 does (F_pil.adjust_* =
    ImageEnhance / HSV round trip; rgb_to_grayscale; F_t.gaussian_blur on the uint8 tensor), flip LAST as in the recipe."""
    import torch
    import torch.nn.functional as F
    from PIL import ImageEnhance
    t, l, ch, cw, flip, order, b, c, s, h, gray, sigma = entry
    f32 = lambda v: float(np.float32(v))
    v = img.crop((l, t, l + cw, t + ch)).resize((res, res), Image.BILINEAR)
    if order is not None:
        for fn in order:
            if fn == 0:
                v = ImageEnhance.Brightness(v).enhance(f32(b))
            elif fn == 1:
                v = ImageEnhance.Contrast(v).enhance(f32(c))
            elif fn == 2:
                v = ImageEnhance.Color(v).enhance(f32(s))
            else:
                hh, ss, vv = v.convert("HSV").split()
                np_h = np.array(hh, dtype=np.uint8)
                np_h += np.uint8(int(f32(h) * 255) % 256)          # numpy 1.x: np.uint8(negative float) wraps
                v = Image.merge("HSV", (Image.fromarray(np_h, "L"), ss, vv)).convert("RGB")
    if gray:
        g = np.array(v.convert("L"), dtype=np.uint8)
        v = Image.fromarray(np.dstack([g, g, g]), "RGB")
    if sigma is not None:
        x = torch.linspace(-1.0, 1.0, steps=3)
        pdf = torch.exp(-0.5 * (x / f32(sigma)).pow(2))
        k1 = pdf / pdf.sum()
        k = torch.mm(k1[:, None], k1[None, :]).expand(3, 1, 3, 3)
        timg = torch.from_numpy(np.asarray(v).copy()).permute(2, 0, 1).unsqueeze(0).to(torch.float32)
        timg = F.conv2d(F.pad(timg, [1, 1, 1, 1], mode="reflect"), k, groups=3)
        v = Image.fromarray(torch.round(timg).squeeze(0).to(torch.uint8).permute(1, 2, 0).numpy(), "RGB")
    if flip:
        v = v.transpose(Image.FLIP_LEFT_RIGHT)
    return np.asarray(v)


def main_hard():
    for name, (h, w, res, entries) in HARD_CASES.items():
        arr = synth_image(name, h, w)
        img = Image.fromarray(arr, "RGB")
        outs = [pil_hard_view(img, e, res) for e in entries]
        out = {"hw_res": np.asarray([h, w, res], np.int32), "n": np.asarray(len(entries), np.int32)}
        if res <= 32:
            out["views_u8"] = np.stack(outs)
        else:
            out["sha1"] = np.asarray([hashlib.sha1(np.ascontiguousarray(o).tobytes()).hexdigest() for o in outs])
            out["corner_u8"] = np.stack([o[:16, :16] for o in outs])
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes")


def main():
    main_hard()
    for name, (h, w, res, crops) in CASES.items():
        arr = synth_image(name, h, w)
        img = Image.fromarray(arr, "RGB")
        outs = []
        nh, nw = resized_output_size(h, w, res)
        v0 = img.resize((nw, nh), Image.BICUBIC)
        top, left = int(round((nh - res) / 2.0)), int(round((nw - res) / 2.0))
        outs.append(np.asarray(v0.crop((left, top, left + res, top + res))))
        for (t, l, ch, cw, flip) in crops:
            v = img.crop((l, t, l + cw, t + ch)).resize((res, res), Image.BILINEAR)
            if flip:
                v = v.transpose(Image.FLIP_LEFT_RIGHT)
            outs.append(np.asarray(v))
        out = {"crops": np.asarray([[t, l, ch, cw, int(f)] for t, l, ch, cw, f in crops], np.int32), "hw_res": np.asarray([h, w, res], np.int32)}
        if res <= 32:
            out["views_u8"] = np.stack(outs)
        else:
            out["sha1"] = np.asarray([hashlib.sha1(np.ascontiguousarray(o).tobytes()).hexdigest() for o in outs])
            out["corner_u8"] = np.stack([o[:16, :16] for o in outs])
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
