"""Device-side view generation: oracle (numpy restatement of Pillow's resampler + the torchvision transform
semantics) against Pillow-generated golden vectors on CPU; the HIP kernels against both on the GPU (bit-exact)."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

from oracle import views_ref as V

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)
import make_views_golden as G  # noqa: E402  (CASES + the seeded synthetic images; needs PIL only in its main())

NAMES = sorted(G.CASES)


def _oracle_u8(name):
    h, w, res, crops = G.CASES[name]
    img = G.synth_image(name, h, w)
    return img, [V.center_view_u8(img, res)] + [V.crop_view_u8(img, t, l, ch, cw, bool(f), res) for t, l, ch, cw, f in crops]


def _check_u8(outs, z):
    if "views_u8" in z.files:
        for o, g in zip(outs, z["views_u8"]):
            assert np.array_equal(o, g)
    else:
        for o, s, c in zip(outs, z["sha1"], z["corner_u8"]):
            assert np.array_equal(o[:16, :16], c)
            assert hashlib.sha1(np.ascontiguousarray(o).tobytes()).hexdigest() == s


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_pillow_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    _, outs = _oracle_u8(name)
    assert z["crops"].tolist() == [[t, l, h, w, int(f)] for t, l, h, w, f in G.CASES[name][3]]
    _check_u8(outs, z)


def test_random_resized_crop_params_host_mirror():
    """rlcf_amd.datautils draws boxes with torchvision's generator calls; oracle.views_ref restates the same algorithm over an
    abstract rng: fed the same draws they agree, the boxes lie inside the image, and the fallback is the central crop."""
    from rlcf_amd import datautils as D

    class Replay:
        def uniform(self, a, b):
            return torch.empty(1).uniform_(a, b).item()

        def randint(self, n):
            return torch.randint(0, n, size=(1,)).item()

    for (h, w) in ((375, 500), (64, 48), (10, 400)):
        torch.manual_seed(h * 1000 + w)
        mine = [D.RandomResizedCropParams(flip_p=0.0)(h, w) for _ in range(20)]
        torch.manual_seed(h * 1000 + w)
        # log-ratio bounds pass through float32 tensors in torchvision: replay with the same values
        lr = torch.log(torch.tensor((3.0 / 4.0, 4.0 / 3.0)))
        ref = []
        for _ in range(20):
            ref.append(V.random_resized_crop_params(h, w, Replay(), ratio=(float(torch.exp(lr[0])), float(torch.exp(lr[1])))))
            torch.rand(1)                    # the flip coin datautils draws after every box
        for (t, l, ch, cw, f), r in zip(mine, ref):
            assert 0 <= t and 0 <= l and t + ch <= h and l + cw <= w and ch > 0 and cw > 0
        assert [m[:4] for m in mine] == [tuple(r) for r in ref]
    # a 1:40 strip never fits the ratio range -> central-crop fallback
    t, l, ch, cw, _ = D.RandomResizedCropParams()(10, 400)
    assert (ch, cw) == (10, 13) and (t, l) == (0, (400 - 13) // 2)


def test_make_views_refuses_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rlcf_amd import _lib, datautils as D
    with pytest.raises(_lib.RlcfError):
        D.make_views(torch.zeros(8, 8, 3, dtype=torch.uint8), [])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_views_bit_exact(name):
    from rlcf_amd import datautils as D
    h, w, res, crops = G.CASES[name]
    img, outs = _oracle_u8(name)
    views = D.make_views(torch.from_numpy(img), crops, res).cpu().numpy()
    ref = np.stack([V.to_tensor_normalize(o) for o in outs])
    assert views.shape == ref.shape
    assert np.array_equal(views, ref)                      # bytes, rounding, flip, /255 and Normalize all identical
    # and through the bytes: undo Normalize/ToTensor and compare with the Pillow fixture
    mean = np.asarray(V.CLIP_MEAN, np.float32)[:, None, None]
    std = np.asarray(V.CLIP_STD, np.float32)[:, None, None]
    u8 = np.rint((views * std + mean) * 255.0).astype(np.uint8).transpose(0, 2, 3, 1)
    _check_u8(list(u8), np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.mark.gpu
def test_augmenter_surface_and_errors():
    from rlcf_amd import _lib, datautils as D
    img = torch.from_numpy(G.synth_image("views_imagenet", 375, 500))
    torch.manual_seed(3)
    aug = D.AugMixAugmenter(None, None, n_views=63)
    out = aug(img)
    assert len(out) == 64 and out[0].shape == (3, 224, 224) and out[0].is_cuda
    torch.manual_seed(3)
    again = aug.views(img)
    assert torch.equal(torch.stack(out), again)            # same generator state -> same crops -> same bytes
    ref0 = V.to_tensor_normalize(V.center_view_u8(img.numpy(), 224))
    assert np.array_equal(out[0].cpu().numpy(), ref0)
    with pytest.raises(_lib.RlcfError):                     # crop outside the image
        D.make_views(img, [(0, 0, 376, 10, False)])
    with pytest.raises(_lib.RlcfError):                     # 40x downscale needs more taps than the kernel carries
        D.make_views(torch.zeros(9000, 300, 3, dtype=torch.uint8), [(0, 0, 9000, 300, False)])
    with pytest.raises(NotImplementedError):
        D.AugMixAugmenter(None, None, n_views=3, augmix=True)
