"""CPU tests of host-side logic added in round 5 (no GPU, no compute through the library): the checkpoint-grid rounding of synthetic
state dicts and the mirror's write tracker."""
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
