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
