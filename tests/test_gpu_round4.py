"""GPU parity, round-4 additions: BASELINE configs[3] run AS A WORKLOAD at full size (256 ViT-B/16 samples, N = 64, C = 1000) through
the sharded driver on one rank and on two ranks, its first eight samples pinned to the reference-generated stream; bench.py's
multi-rank timing record."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import rlcf_ref as RR
from rlcf_amd import synth
from test_gpu_parity import _cfg_from_meta, _tensor_norms, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ENV = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")


def _torchrun(n, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port)]


def test_config3_full_size_one_rank_equals_two_ranks_and_reference_stream(tmp_path):
    """BASELINE configs[3]: `python -m rlcf_amd.eval --total-images 256` — ViT-B/16 student + ViT-B/16 reward, N = 64 views, 1000
    classes, 256 independent test images — on ONE rank and on TWO ranks (gloo, both on this GPU, launched as the driver launches
    bench.py).  (1) No data-path collective: the gathered predictions of the two shards are the one-rank predictions, digest and hit
    counts equal.  (2) With --first-seed 1113 stream samples 0..7 are the samples of tests/golden/tta_b16_n64_stream.npz, produced by
    the reference's own harness body (TPT/tpt_cls_rl.py:251-262) one image at a time: their top-5 must be identical and their final
    logits within 1e-3, while they run inside 32-image passes of a 256-image stream."""
    g, meta = load_golden("tta_b16_n64_stream")
    common = ["--total-images", "256", "--first-seed", str(meta["view_seed0"]), "--keep-logits", str(meta["n_samples"]), "--images-per-pass", "32"]
    one, two = os.path.join(tmp_path, "one.json"), os.path.join(tmp_path, "two.json")
    r = subprocess.run([sys.executable, "-m", "rlcf_amd.eval", "--gpus", "1", "--out", one] + common, cwd=ROOT, env=ENV, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run(_torchrun(2, 29527) + ["-m", "rlcf_amd.eval", "--gpus", "2", "--dist-backend", "gloo", "--out", two] + common,
                       cwd=ROOT, env=ENV, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = json.load(open(one)), json.load(open(two))
    assert a["images"] == b["images"] == 256 and b["n_gpus"] == 2 and len(b["rank_seconds"]) == 2
    assert a["top5"] == b["top5"] and a["predictions_sha256"] == b["predictions_sha256"]
    assert (a["acc1"], a["acc5"]) == (b["acc1"], b["acc5"])
    worst = 0.0
    for rec in (a, b):
        fl = torch.tensor(rec["final_logits_first"])
        for i in range(meta["n_samples"]):
            assert rec["top5"][i] == g[f"top5_{i}"].tolist(), f"stream sample {i}"
            err = (fl[i] - g[f"final_logits_{i}"][0]).abs().max().item()
            worst = max(worst, err)
            assert err < 1e-3, f"stream sample {i}: max|dlogit| {err:.2e}"
    print(f"[configs[3]] 256 images: one rank {a['images_per_s']:.1f} images/s (incl. view synthesis), two ranks on one GPU "
          f"{b['images_per_s']:.1f}; first 8 = reference stream, worst max|dlogit| {worst:.2e}")


def test_bench_multi_rank_timing_record():
    """bench.py with two ranks (gloo on one GPU): the JSON line carries each rank's own seconds of the timed region, the
    slowest / fastest ratio, the settle passes, and a sustained leg that ran on EVERY rank."""
    r = subprocess.run(_torchrun(2, 29529) + ["bench.py", "--gpus", "2", "--dist-backend", "gloo", "--steps", "4", "--warmup", "2", "--views", "16",
                                              "--classes", "64", "--batch", "2", "--no-cpu-baseline", "--sustain-seconds", "0.5",
                                              "--settle-seconds", "0.3"], cwd=ROOT, env=ENV, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    d = rec["distributed"]
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["steps"] == 4
    assert len(d["timed_region_seconds_per_rank"]) == 2 and d["timed_region_slowest_over_fastest_rank"] >= 1.0
    assert d["settle_passes_before_timed_region"] >= 1
    s = d["sustained"]
    assert len(s["images_per_s_per_rank"]) == 2 and all(p >= 5 for p in s["passes_per_rank"])
    assert abs(s["images_per_s_aggregate"] - sum(s["images_per_s_per_rank"])) < 1e-6
    # the timed region still times exactly --steps images per rank
    assert rec["config"]["timed_images_per_rank"] == 4


# ------------------------------------------------------------------------------ every-parameter tuning of a ModifiedResNet student
@pytest.fixture(scope="module")
def L():
    from rlcf_amd import _lib
    _lib.lib()
    return _lib


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _rn_engine(L, dev, meta, prec):
    from rlcf_amd.engine import Engine
    sg, rg = synth.GEOMETRIES[meta["student"]], synth.GEOMETRIES[meta["reward"]]
    ssd = synth.make_state_dict(sg, meta["student_seed"], device=dev)
    rsd = synth.make_state_dict(rg, meta["reward_seed"], device=dev)
    eng = Engine(sg, rg, meta["n_views"], meta["n_cls"], prec)
    eng.load_state_dict(L.STUDENT, ssd)
    eng.load_state_dict(L.REWARD, rsd)
    eng.finalize()
    tokens = synth.make_token_bank(sg, meta["n_cls"], seed=meta["bank_seed"], n_ctx=meta["n_ctx"])
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(sg, meta["n_ctx"]), device=dev)].clone()
    eng.set_class_bank(tokens, meta["n_ctx"], ctx0, L.TEXT_SHARED)
    views = synth.make_views(meta["view_seed"], meta["n_views"], sg.image_resolution, device=dev)
    return eng, ssd, rsd, tokens, views


@pytest.mark.parametrize("prec", [0, 2])
@pytest.mark.parametrize("name", ["rnvis_tiny_s1", "rnvis_tiny_s3", "rnvis_rn50"])
def test_resnet_every_parameter_tuning_matches_reference_fixture(L, dev, name, prec):
    """The parser-default path of TPT/tune_cls_rl.py — `--arch RN50` (params.py:23) with `--tune_norm 0` (:73): CLIPCLS_TTA(only_norm=False)
    on a ModifiedResNet, parameters() = every tensor of clip_model.visual (custom_clip.py:477-479).  rlcf_tta_sample_visual against the
    reference's own run (tests/golden/make_golden.py --only rnvis,rnvisrn50): selection, sampled classes, rewards, the gradient of every
    convolution / BatchNorm / attention-pool tensor (per-tensor norms; every 7th element where the fixture has them), the AdamW updates,
    the running statistics, and the final logits — computed, as the reference computes them after model.eval(), with the BatchNorms in
    EVAL form on the statistics the tuning passes left behind.  RN50 geometry at N = 16 is the full-size case."""
    g, meta = load_golden(name)
    eng, ssd, rsd, tokens, views = _rn_engine(L, dev, meta, prec)
    cfg = _cfg_from_meta(meta)
    base = eng.tta_sample_ln(views, cfg)["final_logits"].clone()               # norm-layer path before: must be unaffected after
    o = eng.tta_sample_visual(views, cfg)
    torch.cuda.synchronize()
    c = lambda k: o[k].cpu()
    big = meta["student"] == "RN50"
    assert c("selected_idx").tolist() == g["selected_idx"].tolist()
    assert c("topk_idx").reshape(-1).tolist() == g["topk_idx"].reshape(-1).tolist()
    assert c("top5").tolist()[: g["top5"].numel()] == g["top5"].tolist()
    # first-pass logits are forward only: the bar of the BatchNorm-tuning test (test_gpu_parity.py: 3e-4 against the float32 fixture, and
    # pinned to the float64 value of the reference-pinned oracle, tests/golden/make_bn_f64.py rnvis_*) — the forward is the same kernels
    torch.testing.assert_close(c("logits"), g["logits"], atol=3e-4, rtol=0)
    torch.testing.assert_close(c("rewards"), g["rewards"].reshape(-1), atol=5e-5, rtol=1e-3)
    z64 = np.load(os.path.join(GOLDEN, name + "_f64.npz")) if os.path.exists(os.path.join(GOLDEN, name + "_f64.npz")) else None
    if z64 is not None:
        e_l = (c("logits").double() - torch.from_numpy(z64["logits"])).abs().max().item()
        assert e_l < max(2e-4, 4 * float(z64["ref_logit_err"])), f"first-pass logits vs f64: {e_l:.2e} (reference: {float(z64['ref_logit_err']):.2e})"
        # final logits ride on one AdamW step of ~lr sign(g) per element: against float64 no further than 1e-3 / three times the reference
        e_f = (c("final_logits").double() - torch.from_numpy(z64["final_logits"])).abs().max().item()
        assert e_f < max(1e-3, 3 * float(z64["ref_final_err"])), f"final logits vs f64: {e_f:.2e} (reference: {float(z64['ref_final_err']):.2e})"
    torch.testing.assert_close(c("final_logits"), g["final_logits"], atol=2e-3 if big else 1e-3, rtol=0)
    torch.testing.assert_close(eng.bn_stats().cpu(), g["bn_stats_after"], atol=1e-4, rtol=1e-3)
    keys = RR.visual_param_keys(ssd)
    grad, after = eng.merge_visual(o["ln_grad"], o["vis_grad"]), eng.merge_visual(o["ln_after"], o["vis_after"])
    gn, rn_ = _tensor_norms(ssd, keys, grad), g["vis_grad_l2"]
    # attnpool.k_proj.bias: exactly zero gradient in exact arithmetic (softmax is blind to a common shift of the keys): noise on both sides
    kb = keys.index("visual.attnpool.k_proj.bias")
    keep = torch.tensor([i for i in range(len(keys)) if i != kb])
    if meta["tta_steps"] == 1:
        # train-mode BatchNorm amplifies rounding noise through the backward (BatchNorm tuning: at RN50 size the
        # reference's own f32 gradient is up to 7e-3 from the f64 value): tensors are compared at that width at full size
        torch.testing.assert_close(gn[keep], rn_[keep], rtol=1e-2 if big else 3e-3, atol=1e-8)
        assert gn[kb] < 1e-6 * rn_.max()
        if z64 is not None:
            # ... and every tensor's gradient norm against the float64 value, next to the reference's own float32 distance from it: the TYPICAL
            # tensor no further than twice the reference's median (measured at RN50: 1.09e-4 f32 mode / 1.52e-4 split-f16 against 1.05e-4 —
            # the same noise band), the WORST tensor within four times the reference's worst (5.3e-3 / 3.4e-3 against 1.6e-3: which tensor
            # draws the tail of the train-form BatchNorm noise differs between two float32 summation orders)
            n64, r64 = torch.from_numpy(z64["vis_grad_l2"])[keep], torch.from_numpy(z64["ref_grad_l2_relerr"])[keep]
            err = ((gn[keep].double() - n64).abs() / n64)
            print(f"[{name} prec {prec}] per-tensor |g| vs f64: worst {err.max():.2e} (reference {r64.max():.2e}), median {err.median():.2e} "
                  f"(reference {r64.median():.2e}); first-pass logits {e_l:.2e}, final logits {e_f:.2e}")
            assert err.max() < max(1e-3, 4 * r64.max().item()) and err.median() < max(3e-4, 2 * r64.median().item())
    torch.testing.assert_close(_tensor_norms(ssd, keys, after, ssd)[keep], g["vis_delta_l2"][keep], rtol=0.05 if big else 0.01, atol=1e-7)
    if "vis_grad_sample" in g:
        gr, og = g["vis_grad_sample"], grad[::7].cpu()
        assert (og - gr).norm() / gr.norm() < 2e-3
        d = (after[::7].cpu() - g["vis_after_sample"]).abs()
        assert (d > 0.1 * meta["lr"]).float().mean() < 0.01
    if not big and meta["tta_steps"] == 1:                                      # the whole gradient vector against the oracle on the same inputs
        ref = RR.tta_sample_ln({k: v.cpu() for k, v in ssd.items()}, {k: v.cpu() for k, v in rsd.items()}, views.cpu(), tokens,
                               RR.TTAHyper(selection_p=meta["selection_p"], tta_steps=1, sample_k=meta["sample_k"], lr=meta["lr"],
                                           weight_decay=meta["weight_decay"]), only_norm=False)
        assert (grad.cpu() - ref["ln_grad"]).norm() / ref["ln_grad"].norm() < 2e-3
    # the engine is back in its pristine state: the call repeats bit for bit, and the norm-layer path gives what it gave before
    o2 = eng.tta_sample_visual(views, cfg)
    if prec == 2:          # split-f16 mode: fixed-order reductions throughout (the f32 mode's small-GEMM / column-sum kernels still use float atomics)
        assert torch.equal(o2["final_logits"], o["final_logits"]) and torch.equal(o2["vis_grad"], o["vis_grad"])
    else:
        torch.testing.assert_close(o2["final_logits"], o["final_logits"], atol=1e-4, rtol=0)
        assert (o2["vis_grad"] - o["vis_grad"]).norm() / o["vis_grad"].norm() < 1e-4
    torch.testing.assert_close(eng.tta_sample_ln(views, cfg)["final_logits"], base, atol=1e-5, rtol=0)
    assert torch.equal(eng.visual_params(0), eng.visual_params(1))
    eng.close()


def test_resnet_every_parameter_tuning_through_the_mirror(L, dev):
    """The reference's call sequence (tune_cls_rl.py:206-221: reset, model.train(), test_time_tuning, model.eval(), model(image)) with
    rlcf_amd.custom_clip.CLIPCLS_TTA(arch = a ModifiedResNet, only_norm=False) — the constructor the parser defaults build — in place of
    the reference's class: the final logits of the reference's own run."""
    import copy
    import types
    from rlcf_amd import clip_reward, clip_store, custom_clip, runtime, tpt_cls_rl
    g, meta = load_golden("rnvis_tiny_s1")
    runtime.reset_session()
    sg, rg = synth.GEOMETRIES[meta["student"]], synth.GEOMETRIES[meta["reward"]]
    clip_store.register_checkpoint("tiny-rn", sg, synth.make_state_dict(sg, meta["student_seed"]))
    clip_store.register_checkpoint("tiny-r", rg, synth.make_state_dict(rg, meta["reward_seed"]))
    bank = clip_store.SyntheticBank(sg, meta["n_cls"], meta["n_ctx"], meta["bank_seed"])
    clip_store.set_tokenizer(bank.tokenize)
    args = types.SimpleNamespace(tta_steps=meta["tta_steps"], selection_p=meta["selection_p"], gpu=0, tpt=True, print_freq=1000, min_entropy_reg=0,
                                 min_entropy_w=0.2, reward_arch="tiny-r", multiple_reward_models=0, weighted_scores=1, sample_k=meta["sample_k"],
                                 reward_amplify=False, reward_process=True, process_batch=False)
    model = custom_clip.CLIPCLS_TTA(dev, bank.classnames, arch="tiny-rn", prompt_prefix="a_photo_of_a", only_visual=True, only_norm=False)
    optimizer = torch.optim.AdamW(model.parameters(), meta["lr"], weight_decay=meta["weight_decay"])
    optim_state = copy.deepcopy(optimizer.state_dict())
    reward_model = clip_reward.get_reward_model(dev, args)
    reward_model.set_class_features(tokenized_classes=model.tokenized_prompts)
    views = synth.make_views(meta["view_seed"], meta["n_views"], sg.image_resolution).to(dev)
    for _ in range(2):                                                           # twice: the second sample starts from the reset state again
        model.reset()
        optimizer.load_state_dict(optim_state)
        model.train()
        tpt_cls_rl.test_time_tuning(model, views, optimizer, None, args, reward_model=reward_model)
        model.eval()
        with torch.no_grad():
            out = model(views[:1])
        torch.testing.assert_close(out.cpu(), g["final_logits"], atol=1e-3, rtol=0)
    runtime.reset_session()


# ------------------------------------------------------------------------------ BASELINE configs[4] at the class count bench.py runs
def test_config4_full_geometry_1000_classes_properties(L, dev):
    """BASELINE configs[4] exactly as `bench.py --config 4` runs it — RN50x64 student @448^2, ViT-L/14 reward behind the bicubic resample,
    N = 32 views, 1000 classes (the reference-generated fixture of this geometry stops at 200 classes: the reference's autograd tape over
    1000 does not fit the build container) — through size-independent properties:
    (1) a permutation of views 1..31 permutes the selection and leaves the adapted prompt, the final logits and the top-5 unchanged;
    (2) the K rewards of every selected view sum to zero and so does the loss gradient of its logits;
    (3) lr = 0 is plain inference: final logits = the first-pass logits of view 0.
    (The 200-class reference fixture itself: tests/test_gpu_round2.py::test_config5_full_geometry_matches_reference_fixture.)"""
    from rlcf_amd.engine import Engine, TTAConfig
    sg, rg = synth.GEOMETRIES["RN50x64"], synth.GEOMETRIES["ViT-L/14"]
    ssd, rsd = synth.make_state_dict(sg, 11, device=dev), synth.make_state_dict(rg, 23, device=dev)
    N, C = 32, 1000
    tokens = synth.make_token_bank(sg, C, seed=7, n_ctx=4)
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(sg, 4), device=dev)].clone()
    eng = Engine(sg, rg, N, C, L.PREC_F16X3)
    eng.load_state_dict(L.STUDENT, ssd); eng.load_state_dict(L.REWARD, rsd); eng.finalize()
    eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
    cfg = TTAConfig(selection_p=0.1, sample_k=3, lr=7e-3, weight_decay=5e-4)
    views = synth.make_views(1000, N, 448, device=dev)
    o = eng.tta_sample(views, cfg)
    gperm = torch.Generator().manual_seed(5)
    perm = torch.cat([torch.zeros(1, dtype=torch.long), 1 + torch.randperm(N - 1, generator=gperm)]).to(dev)
    o2 = eng.tta_sample(views[perm], cfg)
    sel, sel2 = o["selected_idx"].long(), o2["selected_idx"].long()
    assert perm[sel2].tolist() == sel.tolist()                                                            # (1)
    torch.testing.assert_close(o2["ctx_after"], o["ctx_after"], atol=1e-6, rtol=0)
    torch.testing.assert_close(o2["final_logits"], o["final_logits"], atol=2e-4, rtol=0)
    assert o2["top5"].tolist() == o["top5"].tolist()
    n_sel = sel.numel()
    assert n_sel == 3
    torch.testing.assert_close(o["rewards"].view(n_sel, 3).sum(1), torch.zeros(n_sel, device=dev), atol=2e-5, rtol=0)      # (2)
    torch.testing.assert_close(o["dlogits"].sum(1), torch.zeros(n_sel, device=dev), atol=1e-6, rtol=0)
    o0 = eng.tta_sample(views, TTAConfig(selection_p=0.1, sample_k=3, lr=0.0, weight_decay=0.0), want_intermediates=True)  # (3)
    torch.testing.assert_close(o0["final_logits"][0], o0["logits"][0], atol=2e-4, rtol=0)
    eng.close()


# ------------------------------------------------------------------------------ attention backward, two-kernel form (attention_bwd_x3b.hip)
@pytest.mark.parametrize("tokens,n_seq,W", [(257, 3, 128), (197, 2, 192), (50, 4, 64), (65, 2, 64), (577, 1, 64)])
def test_attention_backward_two_kernel_form(L, dev, tokens, n_seq, W):
    """dQ per 64 queries, dK / dV per 64 keys with the accumulators kept in registers over all query blocks (no parking, no atomics): the
    plain (no prefix, no mask) sequences of the image towers.  Against the float64 gradient, bit-reproducible run to run, and equal to
    the one-kernel form (RLCF_ATTN_BWD_OLD) within rounding — incl. ragged tails (257 = 4 x 64 + 1, 65, 577 tokens)."""
    from test_gpu_parity import _attn_ref
    st = lambda: torch.cuda.current_stream().cuda_stream
    T = tokens * n_seq
    seqs = [(i * tokens, tokens, 0, 0) for i in range(n_seq)]
    qkv = synth.normal(5, f"attb.{tokens}.{W}", (T, 3 * W), 1.5)
    do = synth.normal(5, f"attb.do.{tokens}.{W}", (T, W)) * 1e-4            # the magnitude the tuning paths see
    sq = (L.Seq * len(seqs))(*[L.Seq(*s) for s in seqs])
    sbuf = torch.frombuffer(bytearray(bytes(sq)), dtype=torch.int32).to(dev)
    qd, dod = qkv.to(dev), do.to(dev)
    out, lse = torch.zeros(T, W, device=dev), torch.zeros(T, W // 64, device=dev)
    L.check(L.lib().rlcf_attention_fwd(qd.data_ptr(), sbuf.data_ptr(), n_seq, tokens, W, 0, out.data_ptr(), lse.data_ptr(), L.PREC_F32, st()))
    q64 = qkv.double().requires_grad_(True)
    (_attn_ref(q64, seqs, W, 0) * do.double()).sum().backward()
    runs = []
    for _ in range(2):
        dq = torch.full((T, 3 * W), float("nan"), device=dev)             # (every element must be WRITTEN: there is no zero fill any more)
        L.check(L.lib().rlcf_attention_bwd_flash_prec(qd.data_ptr(), out.data_ptr(), lse.data_ptr(), dod.data_ptr(), sbuf.data_ptr(), n_seq,
                                                      tokens, W, 0, dq.data_ptr(), L.PREC_F16X3, st()))
        runs.append(dq.cpu())
    assert torch.isfinite(runs[0]).all() and torch.equal(runs[0], runs[1])
    err, gref = (runs[0].double() - q64.grad).abs(), q64.grad.abs()
    assert float((err > 3e-9 + 2e-4 * gref).double().mean()) < 1e-3
    assert float(err.max()) <= 2e-4 * float(gref.max())


# ------------------------------------------------------------------------------ 192 x 256 tile form of the split-f16 GEMM (gemm_f16x3.hip, MT = 3)
@pytest.mark.parametrize("N,K,epi,res,pair", [(768, 768, 0, True, False), (768, 3072, 0, True, False), (768, 768, 1, False, True), (768, 768, 0, False, False)])
def test_gemm_192_row_tiles_bit_identical_to_the_other_tile_shapes(L, dev, N, K, epi, res, pair):
    """One image's token matrix (M = 12608 = 64 views x 197 tokens) against a W x W / W x 4W weight: 150 tiles of 256 x 256 fill 59 %
    of one round of workgroups, so the launcher takes 192 x 256 tiles there.  Every output element is the same sum of the same
    products in the same K order with the same epilogue arithmetic whatever the tile shape: the product computed in ONE call
    (192-row tiles) must be BIT-IDENTICAL to the same product computed as two calls of 6304 rows (too few tiles for that form: they
    run on the 256 x 128 / 128 x 128 kernels), and within 3e-4 of the float64 product."""
    st = lambda: torch.cuda.current_stream().cuda_stream
    lib = L.lib()
    M = 12608
    a = synth.normal(9, f"mt3.a.{K}", (M, K)).to(dev)
    w = (synth.normal(9, f"mt3.w.{N}.{K}", (N, K)) * K ** -0.5).to(dev)
    b = (synth.normal(9, f"mt3.b.{N}", (N,)) * 0.1).to(dev)
    x = synth.normal(9, f"mt3.x.{N}", (M, N)).to(dev) if res else None

    def pairs(t):
        R, Kk = t.shape
        h, l = torch.empty(R, Kk, dtype=torch.float16, device=dev), torch.empty(R, Kk, dtype=torch.float16, device=dev)
        L.check(lib.rlcf_split_f16x2(t.data_ptr(), h.data_ptr(), l.data_ptr(), R * Kk, st()))
        return torch.stack([h.view(R, Kk // 32, 32), l.view(R, Kk // 32, 32)], dim=2).reshape(R, 2 * Kk).contiguous()
    a2, w2 = pairs(a), pairs(w)

    def run(r0, r1):
        rows = r1 - r0
        c = None if pair else torch.empty(rows, N, device=dev)
        ch = torch.empty(rows, N, dtype=torch.float16, device=dev) if pair else None
        cl = torch.empty_like(ch) if pair else None
        ap = a2.data_ptr() + r0 * 2 * K * 2
        L.check(lib.rlcf_gemm_f16x3(ap, ap + 64, 2 * K, w2.data_ptr(), w2.data_ptr() + 64, 2 * K, b.data_ptr(),
                                    x[r0:r1].data_ptr() if res else None, N, None, 0, c.data_ptr() if c is not None else None, N,
                                    ch.data_ptr() if pair else None, cl.data_ptr() if pair else None, N, rows, N, K, 1.0, epi, st()))
        torch.cuda.synchronize()
        return (ch, cl) if pair else (c,)
    whole = run(0, M)
    parts = [run(0, M // 2), run(M // 2, M)]
    for i, t in enumerate(whole):
        assert torch.equal(t, torch.cat([parts[0][i], parts[1][i]]))
    got = (whole[0].float() + whole[1].float()) if pair else whole[0]
    rows = torch.arange(0, M, 53, device=dev)
    ref = a[rows].double() @ w.double().t() + b.double()
    if epi == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    if res:
        ref = ref + x[rows].double()
    assert float((got[rows].double() - ref).abs().max()) < 3e-4


def test_config4_images_per_pass_equals_one_at_a_time(L, dev):
    """`bench.py --config 4` runs EIGHT test images per tower pass (the convolutions' GEMMs then see 256 views): at the full geometry —
    RN50x64 student @448^2, ViT-L/14 reward, N = 32 — every image of a fused pass must come out as if it had been processed alone
    (independent units): same top-5, final logits within 2e-4.  Three images (one pass of 3) against three
    single-image calls; 200 classes."""
    from rlcf_amd.engine import Engine, TTAConfig
    sg, rg = synth.GEOMETRIES["RN50x64"], synth.GEOMETRIES["ViT-L/14"]
    ssd, rsd = synth.make_state_dict(sg, 11, device=dev), synth.make_state_dict(rg, 23, device=dev)
    N, C, B = 32, 200, 3
    tokens = synth.make_token_bank(sg, C, seed=7, n_ctx=4)
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(sg, 4), device=dev)].clone()
    cfg = TTAConfig(selection_p=0.1, sample_k=3, lr=7e-3, weight_decay=5e-4)
    vs = torch.stack([synth.make_views(3000 + i, N, 448, device=dev) for i in range(B)])
    outs = []
    for n_img in (1, B):
        eng = Engine(sg, rg, N * n_img, C, L.PREC_F16X3)
        eng.load_state_dict(L.STUDENT, ssd); eng.load_state_dict(L.REWARD, rsd); eng.finalize()
        eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
        if n_img == 1:
            ref = [eng.tta_sample(vs[i], cfg, want_intermediates=False) for i in range(B)]
            outs.append(([r["top5"].tolist() for r in ref], torch.stack([r["final_logits"][0] for r in ref])))
        else:
            top5, fl = eng.tta_batch(vs, cfg, want_logits=True)
            outs.append(([t.tolist() for t in top5], fl))
        eng.close()
    assert outs[0][0] == outs[1][0]
    torch.testing.assert_close(outs[1][1], outs[0][1], atol=2e-4, rtol=0)


def test_mirror_reset_state_fast_path_still_sees_edits(L, dev):
    """rlcf_amd.tpt_cls_rl.test_time_tuning starts from the cached reset state without comparing tensors when PromptLearner.reset() just
    ran (no host-device round trip per test image).  It must NOT take that shortcut after the prompt was edited: (1) reset -> tune gives
    the reference fixture's result; (2) reset -> an in-place edit of the Parameter (what an optimizer step is) -> tune gives the oracle's
    result FROM THE EDITED PROMPT; (3) a new ctx_init_state -> reset -> tune starts from the new state.  And model(image) on the clean
    view returns the fused step's logits for the very tensor the loop named, and recomputes for any other tensor."""
    from test_gpu_parity import _harness_objects
    from rlcf_amd import runtime, tpt_cls_rl
    g, meta = load_golden("tta_tiny_s1")
    model, optimizer, optim_state, reward_model, args = _harness_objects(dev, meta)
    pl = model.prompt_learner
    views = synth.make_views(meta["view_seed"], meta["n_views"], 32).to(dev)
    sg, rg = synth.GEOMETRIES[meta["student"]], synth.GEOMETRIES[meta["reward"]]
    ssd, rsd = synth.make_state_dict(sg, meta["student_seed"]), synth.make_state_dict(rg, meta["reward_seed"])
    tokens = synth.make_token_bank(sg, meta["n_cls"], seed=meta["bank_seed"], n_ctx=meta["n_ctx"])
    hyper = RR.TTAHyper(selection_p=meta["selection_p"], tta_steps=meta["tta_steps"], sample_k=meta["sample_k"], lr=meta["lr"],
                        weight_decay=meta["weight_decay"])

    def tune():
        optimizer.load_state_dict(optim_state)
        tpt_cls_rl.test_time_tuning(model, views, optimizer, None, args, reward_model=reward_model)
        with torch.no_grad():
            return model(views[:1]).cpu()
    with torch.no_grad():
        model.reset()
    assert pl._track.at_reset(pl.ctx)
    out1 = tune()
    assert not pl._track.at_reset(pl.ctx)
    torch.testing.assert_close(out1, g["final_logits"], atol=1e-3, rtol=0)                                   # (1)
    with torch.no_grad():
        model.reset()
        pl.ctx.add_(synth.normal(41, "edit.ctx", tuple(pl.ctx.shape), 0.05).to(dev))       # (2) in-place edit after reset() (not a uniform shift: LayerNorm removes those)
    edited = pl.ctx.detach().cpu().clone()
    out2 = tune()
    ref2 = RR.tta_sample(ssd, rsd, views.cpu(), tokens, edited, hyper)
    torch.testing.assert_close(out2, ref2["final_logits"], atol=1e-3, rtol=0)
    assert (out2 - out1).abs().max() > 1e-3
    pre = synth.normal(31, "coop.ctx", tuple(pl.ctx.shape), 0.02).to(dev)                                  # (3)
    pl.ctx_init_state = pre
    assert not pl._track.at_reset(pl.ctx)
    with torch.no_grad():
        model.reset()
    out3 = tune()
    ref3 = RR.tta_sample(ssd, rsd, views.cpu(), tokens, pre.cpu(), hyper)
    torch.testing.assert_close(out3, ref3["final_logits"], atol=1e-3, rtol=0)
    # the clean-view cache: the loop's own tensor is answered from the fused step, another tensor with other contents is computed afresh
    other = synth.make_views(meta["view_seed"] + 1, 1, 32).to(dev)
    with torch.no_grad():
        o_other = model(other).cpu()
    assert (o_other - out3).abs().max() > 1e-3
    # (round 5) writes that do NOT bump Parameter._version — the `.data` idiom of the reference and of this mirror:
    # (4) PromptLearner.reset() called DIRECTLY after a tuning call: model(image) on the clean view is the reset prompt's answer, not the
    #     cached logits of the tuned prompt
    with torch.no_grad():
        model.reset()
    out4 = tune()
    with torch.no_grad():
        pl.reset()                                            # (not ClipTestTimeTuning.reset: the cache is invalidated by the generation)
        o_reset = model(views[:1]).cpu()
    zero_shot = RR.tta_sample(ssd, rsd, views.cpu(), tokens, pl.ctx_init_state.cpu(), hyper)["logits"][:1]
    torch.testing.assert_close(o_reset, zero_shot, atol=1e-3, rtol=0)
    assert (o_reset - out4).abs().max() > 1e-4
    # (5) a `.data` edit between reset() and test_time_tuning is invisible from the host: the call runs from the reset state, and the
    #     device-side guard makes the NEXT entry point raise instead of leaving a silently wrong result behind
    with torch.no_grad():
        model.reset()
        pl.ctx.data.copy_(edited.to(dev))
    optimizer.load_state_dict(optim_state)
    tpt_cls_rl.test_time_tuning(model, views, optimizer, None, args, reward_model=reward_model)
    torch.cuda.synchronize()              # (the guard is read without waiting: an entry point reports it once its comparison has finished)
    with pytest.raises(RuntimeError, match="behind the mirror's back"):
        with torch.no_grad():
            model(views[:1])
    # (6) the hint of the harness loop is consumed by the call it was set for, also when that call returns early
    model._clean_view_hint = views[:1]
    args0 = type(args)(**{**vars(args), "tta_steps": 0}) if hasattr(args, "__dict__") else args
    tpt_cls_rl.test_time_tuning(model, views, optimizer, None, args0, reward_model=reward_model)
    assert getattr(model, "_clean_view_hint", None) is None
    runtime.reset_session()


@pytest.mark.parametrize("M,N,K", [(239, 512, 512), (200, 100, 96), (33, 36, 32), (256, 2048, 1536), (129, 64, 2080)])
@pytest.mark.parametrize("mode", ["scale1", "amax_in", "local"])
def test_gemm_skinny_matches_f64(L, dev, M, N, K, mode):
    """rlcf_gemm_skinny (the few-row split-f16 product of the one-image path: A split in the kernel, coalesced loads staged through LDS,
    K chunks of 64 with a half-empty last chunk when K % 64 == 32, K slices + ordered reduce from K = 1024): against the float64
    product for the three operand scales — 1, from a producer's max|A|, per workgroup found in the kernel — on gradients-sized data
    (1e-5) for the scaled modes; with bias, residual, the QuickGELU epilogue and its backward form; ragged M, N not a multiple of 32."""
    st = lambda: torch.cuda.current_stream().cuda_stream
    lib = L.lib()
    scale = 1.0 if mode == "scale1" else 1e-5
    a = (synth.normal(13, f"sk.a.{M}.{K}", (M, K)) * scale).to(dev)
    w = (synth.normal(13, f"sk.w.{N}.{K}", (N, K)) * K ** -0.5).to(dev)
    b = (synth.normal(13, f"sk.b.{N}", (N,)) * 0.1 * scale).to(dev)
    res = (synth.normal(13, f"sk.r.{M}.{N}", (M, N)) * scale).to(dev)
    aux = synth.normal(13, f"sk.x.{M}.{N}", (M, N)).to(dev)
    wp = torch.empty(N, K, device=dev)
    L.check(lib.rlcf_split_pairs(w.data_ptr(), wp.data_ptr(), N * K, L.PREC_F16X3, st()))
    amax = a.abs().max().reshape(1).contiguous()
    ref0 = a.double() @ w.double().t() + b.double()
    for epi, use_res in ((0, False), (0, True), (1, False), (2, True)):
        if epi == 1 and mode != "scale1":
            continue                                           # (QuickGELU of 1e-5-sized values is linear: nothing to see)
        c = torch.full((M, N), float("nan"), device=dev)
        L.check(lib.rlcf_gemm_skinny(a.data_ptr(), K, wp.data_ptr(), b.data_ptr(), res.data_ptr() if use_res else None, N,
                                     aux.data_ptr() if epi == 2 else None, N if epi == 2 else 0, c.data_ptr(), N, M, N, K, 1.0, epi,
                                     amax.data_ptr() if mode == "amax_in" else None, 1 if mode == "local" else 0, st()))
        ref = ref0.clone()
        if epi == 1:
            ref = ref * torch.sigmoid(1.702 * ref)
        if epi == 2:
            x = aux.double(); sg = torch.sigmoid(1.702 * x)
            ref = ref * (sg * (1 + 1.702 * x * (1 - sg)))
        if use_res:
            ref = ref + res.double()
        assert torch.isfinite(c).all()
        err = float((c.double() - ref).abs().max() / ref.abs().max())
        assert err < (3e-6 if epi != 2 else 2e-5), (epi, use_res, err)


@pytest.mark.parametrize("M,N,K,epi,res,pair", [(1536, 1024, 512, 0, True, False), (1500, 1000, 768, 0, False, False), (1400, 1024, 512, 1, False, True),
                                                (1536, 1024, 2048, 0, True, False)])
def test_gemm_half_tile_and_eight_wave_forms_bit_identical(L, dev, M, N, K, epi, res, pair):
    """The 128x128 split-f16 kernel's two new forms (gemm_f16x3.hip, RT = 1): grids of 65-128 tiles run on 64x128 tiles, the others on eight
    waves.  One call over M rows (96 / 94 / 88 tiles: the half-tile form) against two calls over its halves (<= 64 tiles each: the
    eight-wave form): every element is the same sum of the same products in the same order, so the outputs must be bit-identical — ragged
    last row tile, N not a multiple of 128, residual, QuickGELU -> operand pairs, and a K loop long enough for K slices in the engine —
    and within 3e-4 of the float64 product."""
    st = lambda: torch.cuda.current_stream().cuda_stream
    lib = L.lib()
    a = synth.normal(17, f"h.a.{M}.{K}", (M, K)).to(dev)
    w = (synth.normal(17, f"h.w.{N}.{K}", (N, K)) * K ** -0.5).to(dev)
    b = (synth.normal(17, f"h.b.{N}", (N,)) * 0.1).to(dev)
    x = synth.normal(17, f"h.x.{M}.{N}", (M, N)).to(dev) if res else None

    def pairs(t):
        p = torch.empty(t.shape[0], t.shape[1], device=dev)
        L.check(lib.rlcf_split_pairs(t.contiguous().data_ptr(), p.data_ptr(), t.numel(), L.PREC_F16X3, st()))
        return p
    a2, w2 = pairs(a), pairs(w)

    def run(r0, r1):
        rows = r1 - r0
        c = None if pair else torch.empty(rows, N, device=dev)
        ch = torch.empty(rows, N, dtype=torch.float16, device=dev) if pair else None
        cl = torch.empty_like(ch) if pair else None
        ap = a2.data_ptr() + r0 * K * 4
        L.check(lib.rlcf_gemm_f16x3(ap, ap + 64, 2 * K, w2.data_ptr(), w2.data_ptr() + 64, 2 * K, b.data_ptr(), x[r0:r1].data_ptr() if res else None, N,
                                    None, 0, c.data_ptr() if c is not None else None, N, ch.data_ptr() if pair else None,
                                    cl.data_ptr() if pair else None, N, rows, N, K, 1.0, epi, st()))
        torch.cuda.synchronize()
        return torch.cat([ch, cl], dim=1).view(torch.int16) if pair else c.view(torch.int32)
    whole = run(0, M)
    h = (M // 2 + 63) // 64 * 64
    parts = torch.cat([run(0, h), run(h, M)])
    assert torch.equal(whole, parts)
    whole = whole.view(torch.float32) if not pair else whole
    if not pair:
        rows = torch.arange(0, M, 37, device=dev)
        ref = a[rows].double() @ w.double().t() + b.double()
        if res:
            ref = ref + x[rows].double()
        assert float((whole[rows].double() - ref).abs().max()) < 3e-4
