"""CPU: the checkpoint loader of the host-side mirror (rlcf_amd/clip_store.py) against the contract of the reference's
`clip.load` (TPT/clip/clip.py:94-194) and `build_model` (TPT/clip/model.py:399-439): geometry inferred from tensor shapes for
every OpenAI arch the RLCF scripts name, and the three on-disk forms a maintainer can point RLCF_CLIP_ROOT at."""
import dataclasses
import os

import pytest
import torch
import torch.nn as nn

from rlcf_amd import clip_store, synth


@pytest.mark.parametrize("arch", ["ViT-B/16", "ViT-B/32", "ViT-L/14", "ViT-L/14@336px", "RN50", "RN101", "RN50x4", "RN50x16", "RN50x64"])
def test_geometry_from_state_dict_matches_build_model(arch):
    """Shape inference (model.py:400-422) on a shapes-only (meta) state dict of the published geometry: vision layers counted
    from `.attn.in_proj_weight` keys / `visual.layer{b}` blocks, grid from the positional embedding, heads = width // 64."""
    geo = synth.GEOMETRIES[arch]
    sd = synth.make_state_dict(geo, 0, device="meta")
    got = clip_store.geometry_from_state_dict(sd)
    assert dataclasses.astuple(got) == dataclasses.astuple(geo), (got, geo)
    if arch == "ViT-L/14@336px":
        assert got.image_resolution == 336 and sd["visual.positional_embedding"].shape[0] == 24 * 24 + 1
    if arch == "RN50x64":
        assert got.vision_layers == (3, 15, 36, 10) and got.image_resolution == 448
        assert sd["visual.attnpool.positional_embedding"].shape[0] == 14 * 14 + 1


def _module_tree(sd):
    """nn.Module hierarchy whose state_dict() has exactly the (dotted) keys of `sd` — what an OpenAI TorchScript archive holds."""

    class Node(nn.Module):
        def forward(self, x: torch.Tensor) -> torch.Tensor:
            return x

    root = Node()
    for key, val in sd.items():
        parts, node = key.split("."), root
        for p in parts[:-1]:
            if not hasattr(node, p):
                node.add_module(p, Node())
            node = getattr(node, p)
        if val.is_floating_point() and val.dim() > 0:
            node.register_parameter(parts[-1], nn.Parameter(val.clone(), requires_grad=False))
        else:
            node.register_buffer(parts[-1], val.clone())
    return root


@pytest.mark.parametrize("arch", ["tiny", "tiny-rn"])
@pytest.mark.parametrize("form", ["dict", "wrapped", "jit"])
def test_load_reads_the_three_on_disk_forms(tmp_path, arch, form):
    """load(name, download_root=dir) -> (checkpoint, embed_dim, None) from `<dir>/<name with / -> ->.pt` stored as a plain
    state dict, as {"state_dict": ...} (the CoOp convention the scripts use for --load) and as a TorchScript archive
    (clip.py:119-131: jit.load first, plain load on failure); the archive's extra scalar buffers are dropped (model.py:431-433)."""
    geo = synth.GEOMETRIES[arch]
    sd = synth.make_state_dict(geo, 3)
    name = f"unit/{arch}@test"
    path = os.path.join(tmp_path, name.replace("/", "-") + ".pt")
    if form == "dict":
        torch.save(sd, path)
    elif form == "wrapped":
        torch.save({"state_dict": sd, "epoch": 1}, path)
    else:
        extra = dict(sd, input_resolution=torch.tensor(geo.image_resolution), context_length=torch.tensor(geo.context_length),
                     vocab_size=torch.tensor(geo.vocab_size))
        torch.jit.script(_module_tree(extra)).save(path)
    ckpt, embed_dim, preprocess = clip_store.load(name, device="cpu", download_root=str(tmp_path))
    assert preprocess is None and embed_dim == geo.embed_dim
    assert dataclasses.astuple(ckpt.geometry) == dataclasses.astuple(geo)
    assert set(ckpt.state_dict) == set(sd)
    for k, v in sd.items():
        assert ckpt.state_dict[k].dtype == torch.float32
        assert torch.equal(ckpt.state_dict[k], v.float()), k


def test_load_unknown_model_raises(tmp_path):
    with pytest.raises(RuntimeError, match="not found"):
        clip_store.load("no/such-model", download_root=str(tmp_path))


def test_registered_checkpoint_wins_over_disk(tmp_path):
    geo = synth.GEOMETRIES["tiny"]
    sd = synth.make_state_dict(geo, 5)
    clip_store.register_checkpoint("unit/registered", geo, sd)
    ckpt, d, _ = clip_store.load("unit/registered", download_root=str(tmp_path))
    assert ckpt.state_dict is sd and d == geo.embed_dim


def test_tokenize_forwards_truncate():
    """clip.tokenize(texts, context_length, truncate) (clip.py:197-233): the flag reaches the installed tokenizer
    (CLIPRewards.extract_text_features passes truncate=True, clip_reward.py:142)."""
    seen = {}

    def tok3(texts, context_length=77, truncate=False):
        seen["truncate"] = truncate
        return torch.zeros(1, context_length, dtype=torch.int64)

    old = clip_store._TOKENIZER
    try:
        clip_store.set_tokenizer(tok3)
        clip_store.tokenize("x", truncate=True)
        assert seen["truncate"] is True
        clip_store.set_tokenizer(lambda texts, context_length=77: torch.ones(1, context_length, dtype=torch.int64))
        assert int(clip_store.tokenize("x", truncate=True).sum()) == 77       # two-argument tokenizers still work
    finally:
        clip_store._TOKENIZER = old


def test_params_mirror_has_every_reference_flag_with_its_default():
    """rlcf_amd.params.get_args against the reference's parser (TPT/params.py:13-98), read in the build container only: every option
    string of the reference parses here, with the same default (so the command lines of TPT/scripts/*.sh parse identically)."""
    import argparse
    import importlib.util
    import os
    import sys
    import pytest
    ref = "/root/reference/TPT/params.py"
    if not os.path.exists(ref):
        pytest.skip("reference checkout not present")
    captured = {}
    real_parse = argparse.ArgumentParser.parse_args

    def fake_parse(self, args=None, namespace=None):
        captured["parser"] = self
        return real_parse(self, ["/data", "--output", "/tmp/rlcf_params_test"], namespace)
    spec = importlib.util.spec_from_file_location("ref_params", ref)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    argparse.ArgumentParser.parse_args = fake_parse
    try:
        spec.loader.exec_module(mod)
        mod.save_hp_to_json = lambda *a, **k: None
        ref_ns = mod.get_args()
    finally:
        argparse.ArgumentParser.parse_args = real_parse
    from rlcf_amd.params import get_args
    mine = get_args(["/data", "--output", "/tmp/rlcf_params_test"])
    for act in captured["parser"]._actions:
        if act.dest in ("help",):
            continue
        assert hasattr(mine, act.dest), f"flag {act.option_strings or act.dest} of the reference is missing"
        assert getattr(mine, act.dest) == getattr(ref_ns, act.dest), (act.dest, getattr(mine, act.dest), getattr(ref_ns, act.dest))
    # the command line of scripts/rlcf-prompt.sh
    a = get_args(["/data", "--test_sets", "A", "-a", "ViT-B/16", "-b", "64", "--gpu", "0", "--tpt", "--ctx_init", "a_photo_of_a", "--lr", "7e-3",
                  "--tta_steps", "3", "--sample_k", "3", "--reward_arch", "ViT-L/14", "--reward_amplify", "0", "--reward_process", "1",
                  "--process_batch", "0", "--weight_decay", "5e-4", "--selection_p", "0.1", "--output", "/tmp/x", "--multiple_reward_models", "0"])
    assert (a.arch, a.batch_size, a.tta_steps, a.sample_k, a.augmix, a.hard_aug) == ("ViT-B/16", 64, 3, 3, 1, 0)
