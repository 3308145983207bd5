"""GPU tests, round-5 additions: the dedicated single-pass f16 GEMM of the performance mode (rlcf_gemm_f16 -> gemm_nt_f16_p8_kernel,
rlcf_amd/csrc/gemm_f16.hip) against float64 products of the SAME f16 operands — what is checked is the kernel (tiling, DMA ring
ordering, phase schedule, epilogues), not the f16 rounding of the inputs, which is the mode's labelled arithmetic
(TPT/tpt_cls_rl.py:52: torch.cuda.amp.autocast)."""
import ctypes
import os

import pytest
import torch

from rlcf_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _st():
    return torch.cuda.current_stream().cuda_stream


def _gemm_f16(a16, w16, bias, res, epi, want_f32, want_f16, alpha=1.0):
    M, K = a16.shape
    N = w16.shape[0]
    c = res.clone() if res is not None else (torch.empty(M, N, device=DEV) if want_f32 else None)
    c16 = torch.empty(M, N, dtype=torch.float16, device=DEV) if want_f16 else None
    L.check(L.lib().rlcf_gemm_f16(a16.data_ptr(), K, w16.data_ptr(), K, bias.data_ptr() if bias is not None else None,
                                  c.data_ptr() if res is not None else None, N, c.data_ptr() if c is not None else None, N,
                                  c16.data_ptr() if c16 is not None else None, N, M, N, K, alpha, epi, _st()))
    torch.cuda.synchronize()
    return c, c16


def _ref(a16, w16, bias, res, epi, alpha=1.0):
    r = alpha * (a16.double() @ w16.double().t())
    if bias is not None:
        r = r + bias.double()
    if epi == L.EPI_QUICKGELU:
        r = r * torch.sigmoid(1.702 * r)
    if res is not None:
        r = r + res.double()
    return r


# (M, N, K): every shape runs >= 256 tiles of 256x128 (the regime the engine hands to the eight-phase kernel); ragged M and N, one K
# tile (prologue only), two, an odd count (the unrolled pair of K tiles ends on its first half), the layer shapes of ViT-B/16
# f16 outputs with K % 128 == 0 take the PERSISTENT kernel (one workgroup per CU walks several tiles: > 256 tiles = the cross-tile DMA
# hand-over; K = 128 = a tile that is nothing but the hand-over); everything else the one-workgroup-per-tile kernel
SHAPES = [(256 * 16 + 37, 256 * 8, 64), (256 * 16, 256 * 8 + 132, 128), (256 * 20 + 250, 1024 + 4, 192), (256 * 40 + 100, 2048 + 132, 128),
          (256 * 40 + 100, 2048 + 132, 256), (12608 * 2, 768, 768), (12608 * 2, 2304, 768), (12608, 3072, 768), (12608 * 2, 768, 3072)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("kind", ["f32", "f32_res", "f16", "gelu_f16"])
def test_gemm_f16_eight_phase_kernel_matches_f64(M, N, K, kind):
    torch.manual_seed(M + N + K)
    a16 = torch.randn(M, K, device=DEV).half()
    w16 = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    bias = torch.randn(N, device=DEV) * 0.1
    res = torch.randn(M, N, device=DEV) if kind == "f32_res" else None
    epi = L.EPI_QUICKGELU if kind == "gelu_f16" else L.EPI_NONE
    c, c16 = _gemm_f16(a16, w16, bias, res, epi, kind in ("f32", "f32_res"), kind in ("f16", "gelu_f16"), alpha=0.5)
    ref = _ref(a16, w16, bias, res, epi, alpha=0.5)
    if c is not None:
        err = (c.double() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, K / 768), f"f32 out: {err:.3e}"            # f32 accumulation of exact f16 products
    if c16 is not None:
        # one f16 rounding of the f32 result
        err = ((c16.double() - ref).abs() / (ref.abs() + 1.0)).max().item()
        assert err < 6e-4, f"f16 out: {err:.3e}"


def test_gemm_f16_eight_phase_kernel_is_bit_reproducible():
    """No atomics, no split K: two runs on the same operands are identical to the last bit (the mode's top-1 report depends on it)."""
    import hashlib
    M, N, K = 12608 * 2, 2304, 768
    torch.manual_seed(5)
    a = torch.randn(M, K, device=DEV).half()
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    digests = []
    for _ in range(3):
        c, _ = _gemm_f16(a, w, None, None, L.EPI_NONE, True, False)
        digests.append(hashlib.sha256(c.cpu().numpy().tobytes()).hexdigest())
    assert digests[0] == digests[1] == digests[2]


# ------------------------------------------------------------------ LayerNorm folded into the single-pass f16 products (MODE 1 / 2)
def _ln_fold_operands(W, gamma, beta, b):
    """What engine.hip's make_lnfold prepares: the f16 copy of W diag(gamma) 2^s, alpha = 2^-s, the row sums of that copy times alpha,
    and W beta + b."""
    wg = W * gamma[None, :]
    sh = 9 - int(torch.floor(torch.log2(wg.abs().max())).item())
    scale = 2.0 ** max(-8, min(12, sh))
    w16 = (wg * scale).half()
    s = (w16.double().sum(1) / scale).float()
    bprime = (b.double() + W.double() @ beta.double()).float()
    return w16, 1.0 / scale, s, bprime


@pytest.mark.parametrize("M,N,K,epi", [(12608 * 2, 2304, 768, "none"), (12608 * 2 + 77, 3072, 768, "gelu"), (256 * 9 + 5, 1024, 1024, "none"),
                                       (100, 256, 256, "gelu")])
def test_gemm_f16_layernorm_folded_consumer_matches_f64(M, N, K, epi):
    """MODE 1: LN(x16) W^T + b computed as rstd (x16 (W gamma)^T - mu s) + (W beta + b) on the f16 residual rows themselves, against the
    float64 LayerNorm of the SAME f16 rows followed by the float64 product (nn.LayerNorm: biased variance, eps 1e-5; TPT/clip/model.py:157-163).
    The statistics come from rlcf_resid16_init (a residual stream with a large common offset and outlier channels, as CLIP's is)."""
    torch.manual_seed(5)
    x = torch.randn(M, K, device=DEV) * 1.5 + 0.7
    x[:, 7] += 40.0
    x[:, 300 % K] -= 25.0
    W = torch.randn(N, K, device=DEV) * K ** -0.5
    gamma, beta, b = 1.0 + 0.3 * torch.randn(K, device=DEV), 0.2 * torch.randn(K, device=DEV), 0.1 * torch.randn(N, device=DEV)
    w16, alpha, s, bprime = _ln_fold_operands(W, gamma, beta, b)
    x16 = torch.empty(M, K, dtype=torch.float16, device=DEV)
    mr = torch.empty(M, 2, device=DEV)
    lib = L.lib()
    L.check(lib.rlcf_resid16_init(x.data_ptr(), x16.data_ptr(), mr.data_ptr(), M, K, _st()))
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    e = L.EPI_QUICKGELU if epi == "gelu" else L.EPI_NONE
    L.check(lib.rlcf_gemm_f16_ln(x16.data_ptr(), K, w16.data_ptr(), K, bprime.data_ptr(), out.data_ptr(), N, M, N, K, alpha, e, 1, mr.data_ptr(),
                                 s.data_ptr(), None, _st()))
    torch.cuda.synchronize()
    assert torch.equal(x16, x.half())
    xd = x16.double()
    mu, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    torch.testing.assert_close(mr[:, 0].double(), mu[:, 0], atol=1e-5, rtol=1e-6)
    torch.testing.assert_close(mr[:, 1].double(), (var[:, 0] + 1e-5).rsqrt(), atol=0, rtol=2e-5)
    rows = torch.randint(0, M, (256,), device=DEV)
    ln = (xd[rows] - mu[rows]) / (var[rows] + 1e-5).sqrt() * gamma.double() + beta.double()
    ref = ln @ W.double().t() + b.double()
    if epi == "gelu":
        ref = ref * torch.sigmoid(1.702 * ref)
    got = out[rows].double()
    # f16 weight rounding (2^-11 relative per element, ~K^0.5 terms) + the f16 rounding of the output: the plain kernel's bar
    err = ((got - ref).abs() / (ref.abs() + 1.0)).max().item()
    assert err < 4e-3, err


@pytest.mark.parametrize("M,N,K", [(12608 * 2, 768, 768), (12608 * 2 + 77, 768, 3072), (256 * 9 + 5, 1024, 1024), (100, 256, 256)])
def test_gemm_f16_residual_in_place_with_row_statistics(M, N, K):
    """MODE 2: x16 <- f16(x16 + A W^T + b) in place, and the (mean, rstd) of the STORED rows via the partial sums + rlcf_ln_stats_final.  The
    update is exact up to the one f16 rounding (float64 product of the same f16 operands); the statistics must describe the stored row."""
    torch.manual_seed(6)
    a16 = torch.randn(M, K, device=DEV).half()
    w16 = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    b = 0.1 * torch.randn(N, device=DEV)
    x16 = (torch.randn(M, N, device=DEV) * 2 + 0.5).half()
    x16[:, 11] += 30.0
    x0 = x16.clone()
    P = N // 64
    part = torch.full((P, M, 2), float("nan"), device=DEV)
    mr = torch.empty(M, 2, device=DEV)
    lib = L.lib()
    L.check(lib.rlcf_gemm_f16_ln(a16.data_ptr(), K, w16.data_ptr(), K, b.data_ptr(), x16.data_ptr(), N, M, N, K, 1.0, L.EPI_NONE, 2, None, None,
                                 part.data_ptr(), _st()))
    L.check(lib.rlcf_ln_stats_final(part.data_ptr(), P, M, N, mr.data_ptr(), _st()))
    torch.cuda.synchronize()
    rows = torch.randint(0, M, (256,), device=DEV)
    ref = x0[rows].double() + a16[rows].double() @ w16.double().t() + b.double()
    err = (x16[rows].double() - ref).abs() / (ref.abs() + 1.0)
    assert err.max().item() < 1.5e-3, err.max().item()          # one f16 rounding of the sum (2^-11) + f32 accumulation
    xd = x16.double()
    mu, var = xd.mean(1), xd.var(1, unbiased=False)
    assert torch.isfinite(part).all()
    torch.testing.assert_close(mr[:, 0].double(), mu, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(mr[:, 1].double(), (var + 1e-5).rsqrt(), atol=0, rtol=1e-4)
    # bit-reproducible: same launch again from the same state
    x16b = x0.clone()
    part2 = torch.empty_like(part)
    L.check(lib.rlcf_gemm_f16_ln(a16.data_ptr(), K, w16.data_ptr(), K, b.data_ptr(), x16b.data_ptr(), N, M, N, K, 1.0, L.EPI_NONE, 2, None, None,
                                 part2.data_ptr(), _st()))
    torch.cuda.synchronize()
    assert torch.equal(x16b, x16) and torch.equal(part2, part)


# ------------------------------------------------------------------ two-pass products for weights on the fp16 grid (parity mode)
def test_weights_on_the_fp16_grid_run_two_exact_passes(monkeypatch):
    """Every Conv / Linear / attention / projection weight of a released CLIP checkpoint is an fp16 number (the archives store them as
    fp16; TPT/clip/model.py:375-436 copies them into float32 parameters), so its split-f16 lo half is identically zero and the
    a_hi . w_lo MFMA pass of a product with it adds exact zeros.  The engine recognises such weights at finalize and the 256x256
    split-f16 kernel drops that pass.  Checked at ViT-B/16 size (64 views x 197 tokens: the 256x256 / 192x256 kernels run): with the
    three-pass form forced (RLCF_X3_WLO0=0 at finalize) the image features and the whole prompt-tuning step give the SAME numbers
    (torch.equal: an exact zero can only differ in sign); weights off the grid are not mistaken for grid weights."""
    from rlcf_amd import synth
    from rlcf_amd.engine import Engine, TTAConfig
    g = synth.GEOMETRIES["ViT-B/16"]
    sd32 = synth.make_state_dict(g, 11, device=DEV)
    sd16 = synth.to_fp16_grid(sd32)
    tokens = synth.make_token_bank(g, 16, seed=7, n_ctx=4)
    ctx0 = sd32["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(g, 4), device=DEV)].clone()
    views = synth.make_views(1113, 64, g.image_resolution, device=DEV)
    outs = []
    for env, sd in (("0", sd16), (None, sd16), (None, sd32)):
        if env is None:
            monkeypatch.delenv("RLCF_X3_WLO0", raising=False)
        else:
            monkeypatch.setenv("RLCF_X3_WLO0", env)
        eng = Engine(g, g, 64, 16, L.PREC_F16X3)
        eng.load_state_dict(L.STUDENT, sd)
        eng.load_state_dict(L.REWARD, sd)
        eng.finalize()
        import ctypes
        others = ctypes.c_int(0)
        on = int(L.lib().rlcf_engine_f16_grid_weights(eng.h, L.STUDENT, ctypes.byref(others)))
        eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
        f = eng.encode_image(L.STUDENT, views).clone()
        o = eng.tta_sample(views, TTAConfig(selection_p=0.1))
        outs.append((on, others.value, f, o["final_logits"].clone(), o["ctx_after"].clone()))
        eng.close()
    (on0, off0, f0, l0, c0), (on1, off1, f1, l1, c1), (on2, off2, f2, l2, c2) = outs
    assert on0 == 0 and on1 > 90 and on2 == 0, (on0, on1, on2)        # 12 + 12 blocks x 4 (x 2 with the transposed copies of the text tower) + projections
    assert off1 == 0 or off1 < on1
    assert torch.equal(f0, f1) and torch.equal(l0, l1) and torch.equal(c0, c1)
    assert not torch.equal(f1, f2)                                    # (other weights: other numbers)


@pytest.mark.parametrize("path", ["ln", "visual"])
def test_fp16_grid_weights_through_the_tuning_paths(monkeypatch, path):
    """The same exactness through the image-encoder tuning paths at small geometry (the 128x128 kernels): LayerNorm tuning keeps every GEMM
    weight frozen (two passes throughout); every-parameter tuning moves the weights off the grid at its first optimizer step (three passes
    for the second step and the final inference) and every sample's reset puts the checkpoint's values back (two passes for the 64-view
    forward, the selected views' forward and the backward's transposed products) — both must give the numbers of the forced three-pass build."""
    from rlcf_amd import synth
    from rlcf_amd.engine import Engine, TTAConfig
    sg, rg = synth.GEOMETRIES["small"], synth.GEOMETRIES["small"]
    ssd, rsd = synth.to_fp16_grid(synth.make_state_dict(sg, 11, device=DEV)), synth.to_fp16_grid(synth.make_state_dict(rg, 23, device=DEV))
    tokens = synth.make_token_bank(sg, 40, seed=7, n_ctx=4)
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(sg, 4), device=DEV)].clone()
    views = synth.make_views(2002, 64, sg.image_resolution, device=DEV)
    outs = []
    for env in ("0", None):
        if env is None:
            monkeypatch.delenv("RLCF_X3_WLO0", raising=False)
        else:
            monkeypatch.setenv("RLCF_X3_WLO0", env)
        eng = Engine(sg, rg, 64, 40, L.PREC_F16X3)
        eng.load_state_dict(L.STUDENT, ssd)
        eng.load_state_dict(L.REWARD, rsd)
        eng.finalize()
        eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
        cfg = TTAConfig(selection_p=0.25, tta_steps=2, lr=1e-4)
        o = eng.tta_sample_ln(views, cfg) if path == "ln" else eng.tta_sample_visual(views, cfg)
        keys = ("final_logits", "ln_grad", "ln_after") + (("vis_grad", "vis_after") if path == "visual" else ())
        rec = {k: o[k].clone() for k in keys}
        if path == "visual":
            # a second sample starts from the reset: same numbers again, and the weights are back on the grid afterwards
            o2 = eng.tta_sample_visual(views, cfg)
            for k in keys:
                assert torch.equal(o2[k], rec[k]), k
            others = ctypes.c_int(0)
            rec["on_grid_after_reset"] = int(L.lib().rlcf_engine_f16_grid_weights(eng.h, L.STUDENT, ctypes.byref(others)))
            # an averaged reset state (custom_clip.py:460-475) is off the grid: its products run three passes from then on
            eng.momentum_update_visual(o["vis_after"], 0.5, 1.0, True)
            on_ema = int(L.lib().rlcf_engine_f16_grid_weights(eng.h, L.STUDENT, ctypes.byref(others)))
            o3 = eng.tta_sample_visual(views, cfg)
            on_ema2 = int(L.lib().rlcf_engine_f16_grid_weights(eng.h, L.STUDENT, ctypes.byref(others)))
            rec["ema_final_logits"], rec["ema_vis_grad"] = o3["final_logits"].clone(), o3["vis_grad"].clone()
            assert on_ema == on_ema2 and on_ema < max(rec["on_grid_after_reset"], 1) or env == "0", (on_ema, on_ema2, rec["on_grid_after_reset"])
            eng.reset_visual_state()
            assert int(L.lib().rlcf_engine_f16_grid_weights(eng.h, L.STUDENT, ctypes.byref(others))) == rec["on_grid_after_reset"]
        outs.append(rec)
        eng.close()
    for k in outs[0]:
        if k == "on_grid_after_reset":
            assert outs[0][k] == 0 and outs[1][k] > 20, (outs[0][k], outs[1][k])      # 2 blocks x 4 weights + their transposes + text tower
        else:
            assert torch.equal(outs[0][k], outs[1][k]), k


def test_checkpoint_grid_weights_against_the_reference_run_on_the_same_weights():
    """The two-pass products against the REFERENCE itself: tests/golden/tta_b16_n64_grid_stream.npz is the reference's own harness body
    (TPT/tpt_cls_rl.py:251-262, one image at a time) on ViT-B/16 + ViT-B/16, N = 64, 1000 classes with every GEMM weight rounded to an
    fp16 value (tests/golden/make_golden.py --only b16gridstream; synth.to_fp16_grid = what a released checkpoint holds).  One image at a
    time through rlcf_tta_sample (selection, sampled classes, scores, rewards, adapted prompt, final logits) and the whole stream in one
    pass through rlcf_tta_batch — with every split weight of the student on the grid, i.e. on the two-pass kernels."""
    from test_gpu_parity import _cfg_from_meta, load_golden
    from rlcf_amd import synth
    from rlcf_amd.engine import Engine
    g, meta = load_golden("tta_b16_n64_grid_stream")
    assert meta["weights"] == "fp16grid"
    n, N = meta["n_samples"], meta["n_views"]
    sg = synth.GEOMETRIES[meta["student"]]
    ssd = synth.to_fp16_grid(synth.make_state_dict(sg, meta["student_seed"], device=DEV))
    rsd = synth.to_fp16_grid(synth.make_state_dict(synth.GEOMETRIES[meta["reward"]], meta["reward_seed"], device=DEV))
    eng = Engine(sg, synth.GEOMETRIES[meta["reward"]], N * n, meta["n_cls"], L.PREC_F16X3)
    eng.load_state_dict(L.STUDENT, ssd)
    eng.load_state_dict(L.REWARD, rsd)
    eng.finalize()
    others = ctypes.c_int(0)
    on = int(L.lib().rlcf_engine_f16_grid_weights(eng.h, L.STUDENT, ctypes.byref(others)))
    assert on > 90 and others.value == 0, (on, others.value)
    tokens = synth.make_token_bank(sg, meta["n_cls"], seed=meta["bank_seed"], n_ctx=meta["n_ctx"])
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(sg, meta["n_ctx"]), device=DEV)].clone()
    eng.set_class_bank(tokens, meta["n_ctx"], ctx0, L.TEXT_SHARED)
    cfg = _cfg_from_meta(meta, sparse=True)
    views = torch.stack([synth.make_views(meta["view_seed0"] + i, N, sg.image_resolution, device=DEV) for i in range(n)])
    worst, flipped = 0.0, []
    for i in range(n):
        o = eng.tta_sample(views[i], cfg)
        assert o["selected_idx"].cpu().tolist() == g[f"selected_idx_{i}"].tolist()
        assert o["topk_idx"].cpu().reshape(-1).tolist() == g[f"topk_idx_{i}"].reshape(-1).tolist()
        assert o["top5"].cpu().tolist() == g[f"top5_{i}"].tolist()
        torch.testing.assert_close(o["clip_score"].cpu(), g[f"clip_score_{i}"].reshape(-1), atol=1e-5, rtol=1e-4)
        torch.testing.assert_close(o["rewards"].cpu(), g[f"rewards_{i}"].reshape(-1), atol=5e-5, rtol=1e-3)
        torch.testing.assert_close(o["final_logits"].cpu(), g[f"final_logits_{i}"], atol=1e-3, rtol=0)
        d = (o["ctx_after"].cpu() - g[f"ctx_after_{i}"]).abs()
        flipped.append(int((d > 0.1 * meta["lr"]).sum()))
        assert flipped[-1] <= 0.01 * d.numel(), f"sample {i}: {flipped[-1]} prompt elements off by more than 0.1 lr"
        worst = max(worst, (o["final_logits"].cpu() - g[f"final_logits_{i}"]).abs().max().item())
    top5, fl = eng.tta_batch(views, cfg, want_logits=True)
    torch.cuda.synchronize()
    for i in range(n):
        assert top5[i].cpu().tolist() == g[f"top5_{i}"].tolist(), f"sample {i} in the batched pass"
        err = (fl[i].cpu() - g[f"final_logits_{i}"][0]).abs().max().item()
        worst = max(worst, err)
        assert err < 1e-3, f"sample {i} in the batched pass: max|dlogit| {err:.2e}"
    print(f"[grid-weight stream] {n} samples, {on} split weights on the fp16 grid: worst max|dlogit| vs the reference = {worst:.2e}; "
          f"sign-fragile prompt elements per sample {flipped}")
    eng.close()


@pytest.mark.parametrize("student,reward,n_views,n_cls", [("tiny-rn", "tiny-r64", 32, 40), ("RN50", "ViT-B/32", 16, 64)])
def test_fp16_grid_weights_through_resnet_every_parameter_tuning(monkeypatch, student, reward, n_views, n_cls):
    """Every-parameter tuning of a ModifiedResNet student (the parser defaults of TPT/tune_cls_rl.py) on checkpoint-grid weights: the train-form
    convolutions read the stored weights (on the grid), each sample's reset puts them back, so everything up to the optimizer step runs two
    passes and the eval-form final inference on the tuned weights three.  Same bits as the forced three-pass build: first sample, a second
    sample after the reset, and a sample after an applied momentum update of the reset state (three passes throughout from then on)."""
    from rlcf_amd import synth
    from rlcf_amd.engine import Engine, TTAConfig
    sg, rg = synth.GEOMETRIES[student], synth.GEOMETRIES[reward]
    ssd, rsd = synth.to_fp16_grid(synth.make_state_dict(sg, 11, device=DEV)), synth.to_fp16_grid(synth.make_state_dict(rg, 23, device=DEV))
    tokens = synth.make_token_bank(sg, n_cls, seed=7, n_ctx=4)
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(sg, 4), device=DEV)].clone()
    views = synth.make_views(2003, n_views, sg.image_resolution, device=DEV)
    keys = ("logits", "final_logits", "ln_grad", "ln_after", "vis_grad", "vis_after")
    outs = []
    for env in ("0", None):
        if env is None:
            monkeypatch.delenv("RLCF_X3_WLO0", raising=False)
        else:
            monkeypatch.setenv("RLCF_X3_WLO0", env)
        eng = Engine(sg, rg, n_views, n_cls, L.PREC_F16X3)
        eng.load_state_dict(L.STUDENT, ssd)
        eng.load_state_dict(L.REWARD, rsd)
        eng.finalize()
        eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
        cfg = TTAConfig(selection_p=0.25, tta_steps=2, lr=1e-4)
        o = eng.tta_sample_visual(views, cfg)
        rec = {k: o[k].clone() for k in keys}
        o2 = eng.tta_sample_visual(views, cfg)
        for k in keys:
            assert torch.equal(o2[k], rec[k]), k
        others = ctypes.c_int(0)
        rec["on_grid_after_reset"] = int(L.lib().rlcf_engine_f16_grid_weights(eng.h, L.STUDENT, ctypes.byref(others)))
        eng.momentum_update_visual(o["vis_after"], 0.5, 1.0, True)
        rec["on_grid_after_ema"] = int(L.lib().rlcf_engine_f16_grid_weights(eng.h, L.STUDENT, ctypes.byref(others)))
        o3 = eng.tta_sample_visual(views, cfg)
        rec["ema_final_logits"], rec["ema_vis_grad"] = o3["final_logits"].clone(), o3["vis_grad"].clone()
        assert int(L.lib().rlcf_engine_f16_grid_weights(eng.h, L.STUDENT, ctypes.byref(others))) == rec["on_grid_after_ema"]
        outs.append(rec)
        eng.close()
    assert outs[0]["on_grid_after_reset"] == 0 and outs[1]["on_grid_after_reset"] > outs[1]["on_grid_after_ema"] > 0, \
        (outs[0]["on_grid_after_reset"], outs[1]["on_grid_after_reset"], outs[1]["on_grid_after_ema"])     # (the text tower stays on the grid)
    assert not torch.equal(outs[1]["ema_final_logits"], outs[1]["final_logits"])
    for k in outs[0]:
        if not k.startswith("on_grid"):
            assert torch.equal(outs[0][k], outs[1][k]), k


def test_resnet_convolutions_on_the_fp16_grid_keep_the_batchnorm_scale_in_the_epilogue(monkeypatch):
    """A ModifiedResNet tower folds its BatchNorms into the convolutions (model.py:18-31 in eval mode), which takes the weights off the fp16
    grid.  For checkpoint weights the engine keeps an UNFOLDED copy (two MFMA passes) and applies gamma / sqrt(var + eps) per output column
    in the GEMM epilogue instead (ConvW::wg / cs, GemmX3Args::col_scale; the fused pair-emitting blocks).  Same function, another rounding
    order: the features must agree with the folded form (RLCF_CONV_GRID=0) to float32 accuracy, and with float32 random weights nothing
    changes at all."""
    from rlcf_amd import synth
    from rlcf_amd.engine import Engine
    g, rg = synth.GEOMETRIES["RN50"], synth.GEOMETRIES["tiny-r"]
    sd32 = synth.make_state_dict(g, 11, device=DEV)
    sd16 = synth.to_fp16_grid(sd32)
    rsd = synth.make_state_dict(rg, 23, device=DEV)
    views = synth.make_views(3000, 32, g.image_resolution, device=DEV)       # (32 views: the fused pair-emitting blocks are taken)
    feats = {}
    for name, env, sd in (("grid", None, sd16), ("folded", "0", sd16), ("f32", None, sd32), ("f32_folded", "0", sd32)):
        if env is None:
            monkeypatch.delenv("RLCF_CONV_GRID", raising=False)
        else:
            monkeypatch.setenv("RLCF_CONV_GRID", env)
        eng = Engine(g, rg, 32, 16, L.PREC_F16X3)
        eng.load_state_dict(L.STUDENT, sd)
        eng.load_state_dict(L.REWARD, rsd)
        eng.finalize()
        feats[name] = eng.encode_image(L.STUDENT, views).clone()
        eng.close()
    assert torch.equal(feats["f32"], feats["f32_folded"])              # weights off the grid: the folded form, bit for bit
    d = (feats["grid"] - feats["folded"]).abs().max().item()
    assert 0 < d < 2e-5, d                                             # (unit-norm features; the two forms round differently)


def test_two_pass_packed_weight_rows_at_vit_l14_shapes(monkeypatch):
    """The same bit-equality at the other tower geometry the BASELINE configs use (ViT-L/14: K = 1024 / 4096, N = 3072 / 1024 / 4096, 257
    tokens per view — ragged row tiles), image features of 48 views: forced three passes against two passes on packed hi-only W rows."""
    from rlcf_amd import synth
    from rlcf_amd.engine import Engine
    g, rg = synth.GEOMETRIES["ViT-L/14"], synth.GEOMETRIES["tiny-r"]
    sd = synth.to_fp16_grid(synth.make_state_dict(g, 11, device=DEV))
    rsd = synth.make_state_dict(rg, 23, device=DEV)
    views = synth.make_views(4000, 48, g.image_resolution, device=DEV)
    feats = []
    for env in ("0", None):
        if env is None:
            monkeypatch.delenv("RLCF_X3_WLO0", raising=False)
        else:
            monkeypatch.setenv("RLCF_X3_WLO0", env)
        eng = Engine(g, rg, 48, 16, L.PREC_F16X3)
        eng.load_state_dict(L.STUDENT, sd)
        eng.load_state_dict(L.REWARD, rsd)
        eng.finalize()
        feats.append(eng.encode_image(L.STUDENT, views).clone())
        eng.close()
    assert torch.equal(feats[0], feats[1])


@pytest.mark.parametrize("lanes", [2, 3])
def test_harness_with_samples_in_flight_counts_what_the_one_by_one_loop_counts(lanes):
    """test_time_adapt_eval(in_flight=K): still one test image per engine call (the reference's operating point, tpt_cls_rl.py:233-262), K
    samples side by side on K engines / streams driven by K host threads.  Samples are independent (per-sample reset), so the hit counts
    must be those of the serial loop — seven samples over 2 and 3 lanes (ragged last round), labels chosen so that top-1 and top-5 differ —
    and the per-sample predictions of a lane engine must be bit-identical to the session engine's."""
    from test_gpu_parity import _harness_objects, load_golden
    from rlcf_amd import runtime, synth, tpt_cls_rl
    dev = torch.device(DEV)
    g, meta = load_golden("tta_small_s1")
    R = synth.GEOMETRIES[meta["student"]].image_resolution
    samples = [synth.make_views(1000 + i, meta["n_views"], R) for i in range(7)]
    res = {}
    for k in (1, lanes):
        model, optimizer, optim_state, reward_model, args = _harness_objects(dev, meta)
        loader = [([v.unsqueeze(0) for v in s], torch.tensor([int(g["top5"][0]) if i == 0 else (int(g["top5"][3]) if i % 2 else 0)]))
                  for i, s in enumerate(samples)]
        res[k] = tpt_cls_rl.test_time_adapt_eval(loader, model, optimizer, optim_state, None, args, reward_model=reward_model, in_flight=k)
        if k > 1:
            engs = runtime.SESSION.lane_engines(k, meta["n_views"])
            assert len(engs) == k and len({id(e) for e in engs}) == k
            cfg = tpt_cls_rl._config(args, optimizer, reward_model)
            v = torch.stack([s.to(dev) for s in samples[:2]])
            outs = [e.tta_batch(v, cfg, want_logits=True) for e in engs]
            for t5, fl in outs[1:]:
                assert torch.equal(t5, outs[0][0]) and torch.equal(fl, outs[0][1])
        runtime.reset_session()
    assert res[1] == res[lanes] and res[1][0] >= 100.0 / 7 - 1e-3, res


def test_harness_takes_the_lanes_from_args_or_the_environment(monkeypatch):
    """The reference's main_worker passes no keyword to test_time_adapt_eval (TPT/tpt_cls_rl.py:187-188): after the import swap the lanes are
    asked for through RLCF_IN_FLIGHT or args.in_flight (rlcf_amd.params --in_flight), and the counts are the serial loop's."""
    from test_gpu_parity import _harness_objects, load_golden
    from rlcf_amd import runtime, synth, tpt_cls_rl
    dev = torch.device(DEV)
    g, meta = load_golden("tta_small_s1")
    R = synth.GEOMETRIES[meta["student"]].image_resolution
    samples = [synth.make_views(1000 + i, meta["n_views"], R) for i in range(5)]
    calls, real = [], tpt_cls_rl._eval_in_flight
    monkeypatch.setattr(tpt_cls_rl, "_eval_in_flight", lambda *a_: (calls.append(a_[-1]), real(*a_))[1])
    res = []
    for env, attr in ((None, None), ("2", None), ("2", 3)):
        monkeypatch.delenv("RLCF_IN_FLIGHT", raising=False)
        if env:
            monkeypatch.setenv("RLCF_IN_FLIGHT", env)
        model, optimizer, optim_state, reward_model, args = _harness_objects(dev, meta)
        if attr:
            args.in_flight = attr
        loader = [([v.unsqueeze(0) for v in s], torch.tensor([int(g["top5"][0]) if i == 0 else (int(g["top5"][3]) if i % 2 else 0)]))
                  for i, s in enumerate(samples)]
        res.append(tpt_cls_rl.test_time_adapt_eval(loader, model, optimizer, optim_state, None, args, reward_model=reward_model))
        runtime.reset_session()
    assert calls == [2, 3] and res[0] == res[1] == res[2], (calls, res)


def test_norm_layer_tuning_harness_with_samples_in_flight():
    """The tune_cls_rl.py form of the loop (CLIPCLS_TTA(only_norm=True), no cross-sample EMA) with two samples in flight: lanes call
    rlcf_tta_batch_ln with one image each; same hit counts as the serial loop, and a model with momentum_update=True (samples are NOT
    independent there) must fall back to the serial loop whatever in_flight says."""
    import copy
    import types
    from test_gpu_parity import load_golden
    from rlcf_amd import clip_reward, clip_store, custom_clip, runtime, synth, tpt_cls_rl
    dev = torch.device(DEV)
    g, meta = load_golden("ln_tiny_s1")
    sg, rg = synth.GEOMETRIES[meta["student"]], synth.GEOMETRIES[meta["reward"]]
    samples = [synth.make_views(1000 + i, meta["n_views"], sg.image_resolution) for i in range(5)]
    res = {}
    for tag, lanes, ema in (("serial", 1, False), ("lanes", 2, False), ("ema", 2, True)):
        runtime.reset_session()
        clip_store.register_checkpoint(meta["student"], sg, synth.make_state_dict(sg, meta["student_seed"]))
        clip_store.register_checkpoint("tiny-r", rg, synth.make_state_dict(rg, meta["reward_seed"]))
        bank = clip_store.SyntheticBank(sg, meta["n_cls"], meta["n_ctx"], meta["bank_seed"])
        clip_store.set_tokenizer(bank.tokenize)
        args = types.SimpleNamespace(tta_steps=meta["tta_steps"], selection_p=meta["selection_p"], gpu=0, tpt=True, print_freq=1000, min_entropy_reg=0,
                                     min_entropy_w=0.1, reward_arch="tiny-r", multiple_reward_models=0, sample_k=meta["sample_k"],
                                     reward_amplify=False, reward_process=True, process_batch=False)
        model = custom_clip.CLIPCLS_TTA(dev, bank.classnames, arch=meta["student"], prompt_prefix="a_photo_of_a", only_visual=True, only_norm=True,
                                        momentum_update=ema, update_freq=2, update_w=1.0, momentum=0.9)
        reward_model = clip_reward.get_reward_model(dev, args)
        reward_model.set_class_features(tokenized_classes=model.tokenized_prompts)
        optimizer = torch.optim.AdamW(model.parameters(), meta["lr"], weight_decay=meta["weight_decay"])
        optim_state = copy.deepcopy(optimizer.state_dict())
        loader = [([v.unsqueeze(0) for v in s], torch.tensor([int(g["top5"][0]) if i == 0 else (int(g["top5"][2]) if i % 2 else 0)])) for i, s in enumerate(samples)]
        res[tag] = tpt_cls_rl.test_time_adapt_eval(loader, model, optimizer, optim_state, None, args, reward_model=reward_model, in_flight=lanes)
        if tag == "ema":
            assert not runtime.SESSION._lanes                      # the serial loop ran: no lane engine was built
    runtime.reset_session()
    assert res["serial"] == res["lanes"] and res["serial"][1] >= 20.0, res


def test_harness_conveniences_run_as_kernels_and_match_the_reference_expressions():
    """rlcf_amd.tpt_cls_rl.avg_entropy / accuracy on device tensors are single launches (rlcf_avg_entropy, rlcf_accuracy) since round 5:
    against the reference's torch expressions (TPT/tpt_cls_rl.py:38-44, TPT/utils/tools.py:84-98), on logits at CLIP's scale."""
    import math
    from rlcf_amd import tpt_cls_rl
    torch.manual_seed(3)
    x = (torch.randn(64, 1000, device=DEV) * 3.0 + torch.randn(1, 1000, device=DEV) * 5.0)
    logp = x.double() - x.double().logsumexp(-1, keepdim=True)
    avg = logp.logsumexp(0) - math.log(64)
    ref = -(avg * avg.exp()).sum()
    got = tpt_cls_rl.avg_entropy(x)
    assert got.shape == () and abs(got.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    one = tpt_cls_rl.avg_entropy(x[:1])                                   # one view: the row's own entropy
    p1 = logp[:1].exp()
    assert abs(one.item() - float(-(p1 * logp[:1]).sum())) < 1e-5
    # accuracy: batch of rows, topk (1, 5), (1,), (5,)
    tgt = torch.randint(0, 1000, (64,), device=DEV)
    tgt[:10] = x[:10].argmax(-1)                                          # some top-1 hits
    tgt[10:20] = x[10:20].topk(5, -1).indices[:, 3]                       # some top-5-only hits
    _, pred = x.topk(5, 1, True, True)
    corr = pred.t().eq(tgt.view(1, -1).expand(5, -1))
    ref1, ref5 = corr[:1].float().sum() * (100.0 / 64), corr[:5].float().sum() * (100.0 / 64)
    a1, a5 = tpt_cls_rl.accuracy(x, tgt, topk=(1, 5))
    assert a1.shape == (1,) and abs(a1.item() - ref1.item()) < 1e-4 and abs(a5.item() - ref5.item()) < 1e-4
    assert abs(tpt_cls_rl.accuracy(x, tgt, topk=(5,))[0].item() - ref5.item()) < 1e-4
    assert abs(tpt_cls_rl.accuracy(x[:1], tgt[:1])[0].item() - 100.0) < 1e-4
