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


def test_in_place_draws_consume_the_generator_like_the_plain_calls():
    """datautils draws on one-element scratch tensors in place (t.uniform_, t.exp_, t.random_); torchvision 0.14.1 makes the same draws as
    fresh tensors (RandomResizedCrop.get_params: torch.empty(1).uniform_, torch.exp, torch.randint; RandomHorizontalFlip.forward:
    torch.rand(1) < p).  Written out literally here: 63 views per image, many seeds, ordinary and degenerate image shapes (the ten-attempt
    fallback), two parameter sets — same boxes, same flips, and the SAME generator state afterwards."""
    import math

    from rlcf_amd import datautils as D

    def plain(height, width, scale, ratio, p):
        area = height * width
        log_ratio = torch.log(torch.tensor(ratio))
        box = None
        for _ in range(10):
            target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
            aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
            w = int(round(math.sqrt(target_area * aspect_ratio)))
            h = int(round(math.sqrt(target_area / aspect_ratio)))
            if 0 < w <= width and 0 < h <= height:
                i = torch.randint(0, height - h + 1, size=(1,)).item()
                j = torch.randint(0, width - w + 1, size=(1,)).item()
                box = (i, j, h, w)
                break
        if box is None:
            in_ratio = float(width) / float(height)
            if in_ratio < min(ratio):
                w = width
                h = int(round(w / min(ratio)))
            elif in_ratio > max(ratio):
                h = height
                w = int(round(h * max(ratio)))
            else:
                w, h = width, height
            box = ((height - h) // 2, (width - w) // 2, h, w)
        return (*box, bool(torch.rand(1) < p))

    shapes = [(375, 500), (500, 333), (64, 2000), (3000, 40), (224, 224), (17, 9000)]
    for scale, p in (((0.08, 1.0), 0.5), ((0.3, 1.0), 0.3)):
        ratio = (3.0 / 4.0, 4.0 / 3.0)
        mine = D.RandomResizedCropParams(scale=scale, ratio=ratio, flip_p=p)
        for seed in range(60):
            H, W = shapes[seed % len(shapes)]
            torch.manual_seed(seed)
            a = [plain(H, W, scale, ratio, p) for _ in range(63)]
            sa = torch.get_rng_state()
            torch.manual_seed(seed)
            b = [mine(H, W) for _ in range(63)]
            assert a == b, (seed, H, W)
            assert torch.equal(sa, torch.get_rng_state())


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
    with pytest.raises(_lib.RlcfError):                     # a ColorJitter order that is not a permutation
        D.make_views(img, [(0, 0, 100, 100, False)], hard_plans=[([0, 0, 1, 2], 1.0, 1.0, 1.0, 0.0, False, None)])


# ------------------------------------------------------------------------------ AugMix op chains (fine-grained sets)
def _sha(a):
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


class _ForcedOp:
    """numpy.random with `choice` pinned to one op: what calling that op function directly draws"""

    def __init__(self, i):
        self.i = i
        self.uniform, self.random_sample = np.random.uniform, np.random.random_sample

    def choice(self, n):
        return self.i


AUG_CASES = [(i, sev) for i in range(9) for sev in (1, 5, 10)]


@pytest.mark.parametrize("i,sev", AUG_CASES)
def test_oracle_augmix_ops_match_pillow(i, sev):
    """Every op of augmix_ops.augmentations at three severities, against Pillow's output through the reference's own functions."""
    g = np.load(os.path.join(GOLDEN, "augmix_ops.npz"))
    arr = G.synth_image("augmix", 224, 224)
    np.random.seed(100 + 10 * i + sev)
    name, ip, co = V.draw_augmix_op(_ForcedOp(i), sev)
    r = V.apply_augmix_op(arr, name, ip, co)
    assert np.array_equal(r[:16, :16], g[f"{name}_s{sev}_block"])
    assert np.array_equal(_sha(r), g[f"{name}_s{sev}_sha1"])


MIX_CASES = [(1, 1), (2, 1), (3, 1), (4, 3), (5, 10)]


@pytest.mark.parametrize("seed,sev", MIX_CASES)
def test_oracle_augmix_loop_matches_reference(seed, sev):
    """The whole augmix loop (numpy draw order, chains, float32 mixing) against the reference's ops driven by its loop."""
    g = np.load(os.path.join(GOLDEN, "augmix_mix.npz"))
    arr = G.synth_image("augmix", 224, 224)
    np.random.seed(seed)
    r = V.augmix_view(arr, V.draw_augmix_plan(np.random, sev))
    assert r.dtype == np.float32
    assert np.array_equal(r[:, 100:108, 100:108], g[f"seed{seed}_s{sev}_block"])
    assert np.array_equal(_sha(r), g[f"seed{seed}_s{sev}_sha1"])


def test_host_plan_draws_like_the_oracle():
    from rlcf_amd import datautils as D
    for seed in range(8):
        np.random.seed(seed)
        w, m, chains = D.draw_augmix_plan(1 + seed)
        np.random.seed(seed)
        w2, m2, chains2 = V.draw_augmix_plan(np.random, 1 + seed)
        assert np.array_equal(w, w2) and m == m2 and [len(c) for c in chains] == [len(c) for c in chains2]
        for c, c2 in zip(chains, chains2):
            for (op, ip, co), (name, ip2, co2) in zip(c, c2):
                assert (op == -1 and name == "rotate" and co2 is None) or (V.AUG_OPS[op] == name and ip == ip2 and co == co2)


def _to_oracle_plan(plan):
    w, m, chains = plan
    return w, m, [[("rotate" if op < 0 else V.AUG_OPS[op], ip, co) for op, ip, co in c] for c in chains]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,sev", MIX_CASES)
def test_hip_augmix_views_bit_exact(seed, sev):
    """rlcf_make_views_augmix: the golden image as a full-size 'crop' (resize 224 -> 224 is the identity), so view 1 is the
    reference's augmix output for that numpy seed, bit for bit; two more crops against the numpy oracle."""
    from rlcf_amd import datautils as D
    g = np.load(os.path.join(GOLDEN, "augmix_mix.npz"))
    arr = G.synth_image("augmix", 224, 224)
    crops = [(0, 0, 224, 224, False), (10, 20, 150, 120, True), (100, 3, 60, 200, False)]
    np.random.seed(seed)
    plans = [D.draw_augmix_plan(sev) for _ in crops]
    out = D.make_views(torch.from_numpy(arr), crops, 224, augmix_plans=plans).cpu().numpy()
    assert np.array_equal(out[1][:, 100:108, 100:108], g[f"seed{seed}_s{sev}_block"])
    assert np.array_equal(_sha(out[1]), g[f"seed{seed}_s{sev}_sha1"])
    assert np.array_equal(out[0], V.to_tensor_normalize(V.center_view_u8(arr, 224)))
    for v, (t, l, h, w, f) in enumerate(crops):
        ref = V.augmix_view(V.crop_view_u8(arr, t, l, h, w, f, 224), _to_oracle_plan(plans[v]))
        assert np.array_equal(out[1 + v], ref), v


@pytest.mark.gpu
def test_hip_augmix_every_op_and_augmenter():
    """each op alone (severity 10: the largest rotations / shears / shifts) against the oracle, then the augmenter surface"""
    from rlcf_amd import _lib, datautils as D
    arr = G.synth_image("augmix", 224, 224)
    crops, plans = [], []
    for i in range(9):
        np.random.seed(100 + 10 * i + 10)
        name, ip, co = V.draw_augmix_op(_ForcedOp(i), 10)
        op = -1 if (name == "rotate" and co is None) else i
        crops.append((0, 0, 224, 224, False))
        plans.append((np.float32([1.0, 0.0, 0.0]), np.float32(0.25), [[(op, ip, co)], [(1, 0, None), (0, 0, None)], [(4, 100, None), (2, 3, None), (1, 0, None)]]))
    out = D.make_views(torch.from_numpy(arr), crops, 224, augmix_plans=plans).cpu().numpy()
    for v in range(9):
        assert np.array_equal(out[1 + v], V.augmix_view(arr, _to_oracle_plan(plans[v]))), V.AUG_OPS[v]
    img = torch.from_numpy(G.synth_image("views_imagenet", 375, 500))
    torch.manual_seed(3); np.random.seed(3)
    aug = D.AugMixAugmenter(None, None, n_views=7, augmix=True, severity=3)
    a = aug.views(img)
    torch.manual_seed(3); np.random.seed(3)
    assert torch.equal(a, aug.views(img)) and a.shape == (8, 3, 224, 224) and bool(torch.isfinite(a).all())
    torch.manual_seed(3)
    plain = D.AugMixAugmenter(None, None, n_views=7).views(img)
    assert torch.equal(plain[0], a[0]) and not torch.equal(plain[1:], a[1:])
    bad = [(np.float32([1, 0, 0]), np.float32(0.5), [[(11, 0, None)], [(0, 0, None)], [(0, 0, None)]])]
    with pytest.raises(_lib.RlcfError, match="AugMix"):
        D.make_views(torch.from_numpy(arr), [(0, 0, 224, 224, False)], 224, augmix_plans=bad)


# ------------------------------------------------------------------------------ hard_aug recipe (datautils.py:77-87)
HARD_NAMES = sorted(G.HARD_CASES)


def _oracle_hard_u8(name):
    h, w, res, entries = G.HARD_CASES[name]
    img = G.synth_image(name, h, w)
    outs = []
    for e in entries:
        t, l, ch, cw, flip = e[:5]
        x = V.hard_aug_u8(V.crop_view_u8(img, t, l, ch, cw, False, res), G.hard_plan_for_oracle(e))
        outs.append(x[:, ::-1] if flip else x)             # the recipe flips LAST
    return img, outs


@pytest.mark.parametrize("name", HARD_NAMES)
def test_oracle_hard_aug_matches_pillow_golden(name):
    """fixtures: the recipe run on PIL images with Pillow's ImageEnhance / HSV / L conversions and torch's conv2d, as torchvision 0.14.1
    does (tests/golden/make_views_golden.py:pil_hard_view)"""
    _, outs = _oracle_hard_u8(name)
    _check_u8(outs, np.load(os.path.join(GOLDEN, name + ".npz")))


def test_oracle_colour_arithmetic_exhaustive_against_pillow():
    """Every RGB colour through convert('L') and convert('HSV'), every HSV triple through convert('RGB'), and Image.blend on all
    65536 (in1, in2) byte pairs for factors inside and outside [0, 1]: the restatement equals the installed Pillow everywhere."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    g, b = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    for r in range(256):
        rgb = np.stack([np.full_like(g, r), g, b], -1).astype(np.uint8)
        im = Image.fromarray(rgb, "RGB")
        assert np.array_equal(V.rgb_to_l(rgb), np.asarray(im.convert("L")))
        assert np.array_equal(V.rgb_to_hsv_u8(rgb), np.asarray(im.convert("HSV")))
        assert np.array_equal(V.hsv_to_rgb_u8(rgb), np.asarray(Image.fromarray(rgb, "HSV").convert("RGB")))
    a = np.repeat(g[..., None], 3, axis=-1).astype(np.uint8)
    c = np.repeat(b[..., None], 3, axis=-1).astype(np.uint8)
    ia, ic = Image.fromarray(a, "RGB"), Image.fromarray(c, "RGB")
    for f in (0.0, 1.0, 0.6, 0.8, 0.7531, 0.999999, 1.000001, 1.2, 1.4, 1.39999, 0.5, 1.0 / 3.0):
        assert np.array_equal(V.blend_u8(a, c, f), np.asarray(Image.blend(ia, ic, f))), f


def test_hard_aug_draw_order_host_mirror():
    """rlcf_amd.datautils.HardAugParams makes torch's generator calls; oracle.views_ref.draw_hard_plan restates the order over an
    abstract rng: replaying the same generator through both gives the same boxes, jitter orders, factors, coins and kernels."""
    from rlcf_amd import datautils as D

    class Replay:
        def uniform(self, a, b):
            return torch.empty(1).uniform_(a, b).item()

        def randint(self, n):
            return torch.randint(0, n, size=(1,)).item()

        def rand(self):
            return float(torch.rand(1))

        def randperm(self, n):
            return torch.randperm(n).tolist()

    lr = torch.log(torch.tensor((3.0 / 4.0, 4.0 / 3.0)))
    ratio = (float(torch.exp(lr[0])), float(torch.exp(lr[1])))
    torch.manual_seed(77)
    mine = [D.HardAugParams()(375, 500) for _ in range(200)]
    torch.manual_seed(77)
    seen = {"jitter": 0, "gray": 0, "blur": 0}
    for (box, plan) in mine:
        rb = V.random_resized_crop_params(375, 500, Replay(), scale=(0.2, 1.0), ratio=ratio)
        rp = V.draw_hard_plan(Replay())
        flip = float(torch.rand(1)) < 0.5
        assert tuple(box[:4]) == tuple(rb) and box[4] == flip
        assert plan[0] == rp[0] and plan[5] == rp[5] and (plan[6] is None) == (rp[6] is None)
        if plan[0] is not None:
            assert [float(np.float32(v)) for v in plan[1:5]] == [float(np.float32(v)) for v in rp[1:5]]
            assert 0.6 <= plan[1] <= 1.4 and 0.6 <= plan[2] <= 1.4 and 0.8 <= plan[3] <= 1.2 and -0.1 <= plan[4] <= 0.1
        if plan[6] is not None:
            assert np.array_equal(plan[6].numpy(), rp[6])
        seen["jitter"] += plan[0] is not None; seen["gray"] += bool(plan[5]); seen["blur"] += plan[6] is not None
    assert 70 <= seen["jitter"] <= 130 and 20 <= seen["gray"] <= 65 and 5 <= seen["blur"] <= 40          # p = 0.5 / 0.2 / 0.1 of 200
    assert D.hue_shift_u8(-12.7 / 255) == 244 and D.hue_shift_u8(0.1) == 25 and V.hue_shift_u8(-0.05) == D.hue_shift_u8(-0.05)


@pytest.mark.gpu
@pytest.mark.parametrize("name", HARD_NAMES)
def test_hip_hard_aug_views_bit_exact(name):
    from rlcf_amd import datautils as D
    h, w, res, entries = G.HARD_CASES[name]
    img, outs = _oracle_hard_u8(name)
    plans = [G.hard_plan_for_oracle(e) for e in entries]
    views = D.make_views(torch.from_numpy(img), [e[:5] for e in entries], res, hard_plans=plans).cpu().numpy()
    ref = np.stack([V.to_tensor_normalize(V.center_view_u8(img, res))] + [V.to_tensor_normalize(o) for o in outs])
    assert views.shape == ref.shape
    for v in range(ref.shape[0]):
        assert np.array_equal(views[v], ref[v]), v
    mean = np.asarray(V.CLIP_MEAN, np.float32)[:, None, None]
    std = np.asarray(V.CLIP_STD, np.float32)[:, None, None]
    u8 = np.rint((views[1:] * std + mean) * 255.0).astype(np.uint8).transpose(0, 2, 3, 1)
    _check_u8(list(u8), np.load(os.path.join(GOLDEN, name + ".npz")))          # and the Pillow + torch fixture itself


@pytest.mark.gpu
def test_hip_hard_aug_with_augmix_and_augmenter_surface():
    """hard_aug feeds the AugMix chains (x_orig = preaugment(image), datautils.py:96): device views == oracle on the same plans; the
    augmenter with hard_aug=True is reproducible under a seed and equals the oracle replay of its own draws."""
    from rlcf_amd import datautils as D
    name = "hardaug_imagenet"
    h, w, res, entries = G.HARD_CASES[name]
    img, outs = _oracle_hard_u8(name)
    hplans = [G.hard_plan_for_oracle(e) for e in entries]
    np.random.seed(11)
    aplans = [D.draw_augmix_plan(3) for _ in entries]
    views = D.make_views(torch.from_numpy(img), [e[:5] for e in entries], res, hard_plans=hplans, augmix_plans=aplans).cpu().numpy()
    for v, o in enumerate(outs):
        assert np.array_equal(views[1 + v], V.augmix_view(o, _to_oracle_plan(aplans[v]))), v
    torch.manual_seed(5)
    aug = D.AugMixAugmenter(None, None, n_views=15, hard_aug=True)
    a = aug.views(torch.from_numpy(img))
    torch.manual_seed(5)
    drawn = [D.HardAugParams()(h, w) for _ in range(15)]
    torch.manual_seed(5)
    b = aug.views(torch.from_numpy(img))
    assert a.shape == (16, 3, 224, 224) and torch.equal(a, b)
    for v, (box, plan) in enumerate(drawn):
        t, l, ch, cw, flip = box
        kern = None if plan[6] is None else plan[6].numpy()
        x = V.hard_aug_u8(V.crop_view_u8(img, t, l, ch, cw, False, 224), (plan[0], plan[1], plan[2], plan[3], plan[4], plan[5], kern))
        assert np.array_equal(a[1 + v].cpu().numpy(), V.to_tensor_normalize(x[:, ::-1] if flip else x)), v
