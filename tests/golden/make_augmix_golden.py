"""Golden vectors for the AugMix op chains of the view pipeline (TPT/data/datautils.py:94-110, TPT/data/augmix_ops.py), produced
with Pillow through the reference's own op functions (imported from /root/reference in this container only).

    python tests/golden/make_augmix_golden.py

augmix_ops.npz   every op of `augmentations` applied to a seeded 224x224 uint8 image at severity 1, 5 and 10 (numpy seeded per
                 case, so the oracle re-draws the same level): SHA-1 of the uint8 result + its top-left 16x16 block.
augmix_mix.npz   the whole `augmix` loop (Dirichlet / Beta weights, three chains of 1-3 random ops, float32 mix) on the same
                 image for a few numpy seeds and severities: SHA-1 of the float32 result + a 3x8x8 block.
"""
import hashlib
import os
import sys

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference/TPT/data")
import augmix_ops as ref_ops  # noqa: E402  (the reference's ops: PIL + numpy only)
from make_views_golden import synth_image  # noqa: E402

MEAN = np.asarray([0.48145466, 0.4578275, 0.40821073], np.float32)
STD = np.asarray([0.26862954, 0.26130258, 0.27577711], np.float32)


def preprocess(pil):          # ToTensor + Normalize (tpt_cls_rl.py:147-149) as torch evaluates them
    x = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).float().div(255)
    return (x - torch.from_numpy(MEAN)[:, None, None]) / torch.from_numpy(STD)[:, None, None]


def augmix_loop(x_orig, aug_list, severity):
    """the body of datautils.py:94-110 after `x_orig = preaugment(image)`"""
    x_processed = preprocess(x_orig)
    w = np.float32(np.random.dirichlet([1.0, 1.0, 1.0]))
    m = np.float32(np.random.beta(1.0, 1.0))
    mix = torch.zeros_like(x_processed)
    for i in range(3):
        x_aug = x_orig.copy()
        for _ in range(np.random.randint(1, 4)):
            x_aug = np.random.choice(aug_list)(x_aug, severity)
        mix += w[i] * preprocess(x_aug)
    return m * x_processed + (1 - m) * mix


def main():
    arr = synth_image("augmix", 224, 224)
    img = Image.fromarray(arr, "RGB")
    out = {}
    for i, op in enumerate(ref_ops.augmentations):
        for sev in (1, 5, 10):
            np.random.seed(100 + 10 * i + sev)
            r = np.asarray(op(img.copy(), sev))
            out[f"{op.__name__}_s{sev}_sha1"] = np.frombuffer(hashlib.sha1(np.ascontiguousarray(r).tobytes()).digest(), np.uint8)
            out[f"{op.__name__}_s{sev}_block"] = r[:16, :16].copy()
    np.savez_compressed(os.path.join(HERE, "augmix_ops.npz"), **out)
    out = {}
    for seed, sev in ((1, 1), (2, 1), (3, 1), (4, 3), (5, 10)):
        np.random.seed(seed)
        r = augmix_loop(img, ref_ops.augmentations, sev).numpy()
        out[f"seed{seed}_s{sev}_sha1"] = np.frombuffer(hashlib.sha1(np.ascontiguousarray(r).tobytes()).digest(), np.uint8)
        out[f"seed{seed}_s{sev}_block"] = r[:, 100:108, 100:108].copy()
    np.savez_compressed(os.path.join(HERE, "augmix_mix.npz"), **out)
    print("wrote augmix_ops.npz, augmix_mix.npz")


if __name__ == "__main__":
    main()
