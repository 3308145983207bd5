"""CPU tests of host-side logic added in round 5 (no GPU, no compute through the library): the checkpoint-grid rounding of synthetic
state dicts and the mirror's write tracker."""
import pytest
import torch

from rlcf_amd import synth


def test_to_fp16_grid_rounds_what_a_released_checkpoint_stores_as_fp16_and_nothing_else():
    """convert_weights (TPT/clip/model.py:375-396) turns Conv / Linear / MultiheadAttention weights and biases, text_projection and
    visual.proj into fp16 before the archives are saved; LayerNorm / BatchNorm parameters, embeddings and logit_scale stay float32."""
    for geo in ("tiny", "tiny-rn"):
        sd = synth.make_state_dict(synth.GEOMETRIES[geo], 11)
        g = synth.to_fp16_grid(sd)
        assert set(g) == set(sd)
        changed = {k for k in sd if sd[k].is_floating_point() and not torch.equal(sd[k], g[k])}
        for k in changed:
            assert g[k].dtype == torch.float32 and torch.equal(g[k], g[k].half().float()), k           # on the grid, still float32
            assert not any(t in k for t in ("ln_", "bn", "embedding", "logit_scale", "downsample.1")), k
        assert any(k.endswith("in_proj_weight") or k.endswith("conv1.weight") for k in changed)
        assert any(k in changed for k in ("text_projection",)) and ("visual.proj" in changed or geo == "tiny-rn")
        for k in sd:
            if any(t in k for t in ("ln_", "embedding", "logit_scale")) or ".bn" in k:
                assert torch.equal(sd[k], g[k]), k
        g2 = synth.to_fp16_grid(g)
        assert all(torch.equal(g[k], g2[k]) for k in g)                                               # idempotent


def test_reset_tracker_generation_and_lazy_guard_on_cpu_tensors():
    """custom_clip._ResetTracker: mirror-side writes move the generation; a guard whose comparison is false raises at the next check and is
    consumed by it; a passing guard never raises (CPU tensors: no event, read at once)."""
    from rlcf_amd.custom_clip import _ResetTracker
    t = _ResetTracker()
    p = torch.nn.Parameter(torch.zeros(4))
    t.mark_reset(p)
    assert t.at_reset(p)
    t.guard(p.data, torch.zeros(4), "unchanged")
    t.check()                                         # equal: silent
    with torch.no_grad():
        p.add_(1.0)                                   # an in-place edit bumps Parameter._version
    assert not t.at_reset(p)
    t.mark_reset(p)
    p.data.copy_(torch.full((4,), 2.0))               # `.data` edit: invisible to the version counter ...
    assert t.at_reset(p)
    t.guard(p.data, torch.ones(4), "edited through .data")          # ... but not to the guard
    try:
        t.check()
        raise AssertionError("the guard did not fire")
    except RuntimeError as e:
        assert "behind the mirror's back" in str(e)
    t.check(wait=True)                                # consumed: nothing left
    t.wrote()
    assert not t.at_reset(p)


def test_loop_options_come_from_keyword_then_args_then_environment(monkeypatch):
    """The reference's main_worker calls test_time_adapt_eval(val_loader, model, optimizer, optim_state, scaler, args) with nothing else
    (TPT/tpt_cls_rl.py:187-188): `images_per_pass` / `in_flight` then come from args (rlcf_amd.params --images_per_pass / --in_flight) or
    from RLCF_IMAGES_PER_PASS / RLCF_IN_FLIGHT, and default to the reference's one-image loop."""
    import types
    from rlcf_amd import params
    from rlcf_amd.tpt_cls_rl import _loop_option
    monkeypatch.delenv("RLCF_IN_FLIGHT", raising=False)
    monkeypatch.delenv("RLCF_IMAGES_PER_PASS", raising=False)
    bare = types.SimpleNamespace()                                  # the reference's own argparse namespace has neither attribute
    assert _loop_option(None, bare, "in_flight") == 1 and _loop_option(None, bare, "images_per_pass") == 1
    monkeypatch.setenv("RLCF_IN_FLIGHT", "2")
    assert _loop_option(None, bare, "in_flight") == 2 and _loop_option(None, bare, "images_per_pass") == 1
    a = params.get_args(["--in_flight", "3"])
    assert a.images_per_pass is None and _loop_option(None, a, "in_flight") == 3          # args before the environment
    assert _loop_option(1, a, "in_flight") == 1                                            # the keyword before both
    assert _loop_option(None, params.get_args([]), "in_flight") == 2                       # unset flag -> environment
    monkeypatch.setenv("RLCF_IMAGES_PER_PASS", "0")
    with pytest.raises(ValueError):
        _loop_option(None, bare, "images_per_pass")


def test_view_prefetcher_makes_the_draws_of_the_plain_loop():
    """datautils.ViewPrefetcher runs the host half of the augmenter (crop boxes, AugMix plans) on ONE loader thread ahead of the loop:
    for a seeded run the draws, their order and the targets are those of the plain loop (the device half is stubbed: no GPU here)."""
    import numpy as np
    import torch
    from rlcf_amd import datautils

    class Aug(datautils.AugMixAugmenter):
        def apply(self, img, params):
            self.seen.append((tuple(img.shape), params))
            return torch.zeros(1 + self.n_views, 3, 2, 2)

    imgs = [(torch.zeros(30 + 3 * i, 40 + i, 3, dtype=torch.uint8), i) for i in range(6)]
    for augmix in (False, True):
        a = Aug(n_views=7, augmix=augmix)
        torch.manual_seed(3)
        np.random.seed(3)
        plain = [(tuple(im.shape), a.draw(im.shape[0], im.shape[1])) for im, _ in imgs]
        a.seen = []
        torch.manual_seed(3)
        np.random.seed(3)
        got = list(datautils.ViewPrefetcher(imgs, a, depth=2))
        assert [int(t) for _, t in got] == list(range(6)) and all(len(v) == 8 and v[0].shape == (1, 3, 2, 2) for v, _ in got)
        assert len(a.seen) == 6
        for (s0, p0), (s1, p1) in zip(plain, a.seen):
            assert s0 == s1 and p0[0] == p1[0]
            if augmix:
                for (w0, m0, c0), (w1, m1, c1) in zip(p0[1], p1[1]):
                    assert np.array_equal(w0, w1) and m0 == m1 and c0 == c1
    # an exception of the dataset reaches the consuming loop
    def bad():
        yield imgs[0]
        raise ValueError("decode failed")
    a = Aug(n_views=2)
    a.seen = []
    try:
        list(datautils.ViewPrefetcher(bad(), a))
        raise AssertionError("the loader's exception was swallowed")
    except ValueError as exc:
        assert "decode failed" in str(exc)


def test_local_device_refuses_more_nccl_ranks_than_gpus(monkeypatch):
    """shard.local_device: under nccl (= RCCL) a local rank beyond the visible devices is an error with a message (bench.py / eval.py used
    to wrap it modulo the device count and die inside RCCL); gloo may share a device; no device at all is an error either way."""
    import pytest
    import torch
    from rlcf_amd import shard
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    assert shard.local_device(1, "nccl") == 1 and shard.local_device(3, "gloo") == 1
    with pytest.raises(RuntimeError, match="visible"):
        shard.local_device(2, "nccl")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    with pytest.raises(RuntimeError, match="no visible GPU"):
        shard.local_device(0, "gloo")
