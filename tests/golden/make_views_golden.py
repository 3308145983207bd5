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


def main():
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
