"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the committed
golden fixtures generated from the imported reference.  Run with `-m gpu` on an MI355X."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import clip_ref as CR
from oracle import rlcf_ref as RR
from rlcf_amd import synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def L():
    from rlcf_amd import _lib
    _lib.lib()
    return _lib


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def st():
    return torch.cuda.current_stream().cuda_stream


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("meta_")}
    meta = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
    return arrays, meta


# ------------------------------------------------------------------------------ op level
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (197, 300, 128), (1000, 64, 512), (37, 1000, 64), (2049, 768, 768)])
@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("prec", [0, 2])
def test_gemm_nt(L, dev, M, N, K, epi, prec):
    a = synth.normal(1, "g.a", (M, K)).to(dev)
    w = synth.normal(1, "g.w", (N, K), K ** -0.5).to(dev)
    b = synth.normal(1, "g.b", (N,), 0.1).to(dev)
    r = synth.normal(1, "g.r", (M, N)).to(dev)
    aux = synth.normal(1, "g.aux", (M, N)).to(dev)
    c = torch.empty(M, N, device=dev)
    L.check(L.lib().rlcf_gemm_nt(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), r.data_ptr(), N, aux.data_ptr(), N,
                                 c.data_ptr(), N, M, N, K, 0.5, epi, prec, st()))
    v = 0.5 * (a.double().cpu() @ w.double().cpu().t()) + b.double().cpu()
    if epi == 1:
        v = v * torch.sigmoid(1.702 * v)
    elif epi == 2:
        f = aux.double().cpu()
        s = torch.sigmoid(1.702 * f)
        v = v * (s * (1 + 1.702 * f * (1 - s)))
    v = v + r.double().cpu()
    torch.testing.assert_close(c.cpu().double(), v, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("rows,width", [(5, 128), (1000, 512), (197 * 3, 768), (7, 1024), (9, 64)])
def test_layernorm(L, dev, rows, width):
    x = synth.normal(2, "ln.x", (rows, width), 2.0, 0.3).to(dev)
    g = synth.normal(2, "ln.g", (width,), 0.1, 1.0).to(dev)
    b = synth.normal(2, "ln.b", (width,), 0.05).to(dev)
    dy = synth.normal(2, "ln.dy", (rows, width)).to(dev)
    y = torch.empty_like(x)
    L.check(L.lib().rlcf_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), rows, width, st()))
    xc = x.cpu().double().requires_grad_(True)
    gc = g.cpu().double().requires_grad_(True)
    bc = b.cpu().double().requires_grad_(True)
    yc = CR.layer_norm(xc.float(), gc.float(), bc.float()) if False else torch.nn.functional.layer_norm(xc, (width,), gc, bc, 1e-5)
    torch.testing.assert_close(y.cpu().double(), yc.detach(), atol=5e-6, rtol=1e-5)
    torch.testing.assert_close(y.cpu(), CR.layer_norm(x.cpu(), g.cpu(), b.cpu()), atol=5e-6, rtol=1e-5)
    (yc * dy.cpu().double()).sum().backward()
    dx = torch.empty_like(x)
    dg = torch.zeros_like(g)
    db = torch.zeros_like(b)
    L.check(L.lib().rlcf_layernorm_bwd(x.data_ptr(), g.data_ptr(), dy.data_ptr(), dx.data_ptr(), dg.data_ptr(),
                                       db.data_ptr(), rows, width, st()))
    torch.testing.assert_close(dx.cpu().double(), xc.grad, atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(dg.cpu().double(), gc.grad, atol=1e-3, rtol=1e-4)
    torch.testing.assert_close(db.cpu().double(), bc.grad, atol=1e-3, rtol=1e-4)


def _attn_ref(qkv, seqs, W, causal):
    """float64 reference of the attention core over packed sequences (differentiable)."""
    H = W // 64
    out = torch.zeros(qkv.shape[0], W, dtype=qkv.dtype)
    for (qs, ql, ps, pl) in seqs:
        rows = list(range(ps, ps + pl)) + list(range(qs, qs + ql))
        q = qkv[qs:qs + ql, :W].reshape(ql, H, 64).transpose(0, 1)
        k = qkv[rows, W:2 * W].reshape(len(rows), H, 64).transpose(0, 1)
        v = qkv[rows, 2 * W:].reshape(len(rows), H, 64).transpose(0, 1)
        s = (q * 0.125) @ k.transpose(-1, -2)
        if causal:
            i = torch.arange(ql)[:, None]
            j = torch.arange(len(rows))[None, :]
            s = s.masked_fill(j > pl + i, float("-inf"))
        o = torch.softmax(s, -1) @ v
        out[qs:qs + ql] = o.transpose(0, 1).reshape(ql, W)
    return out


ATT_CASES = [
    ("vit", [(0, 50, 0, 0), (50, 50, 0, 0), (100, 50, 0, 0)], 150, 128, 0),
    ("vit197", [(i * 197, 197, 0, 0) for i in range(2)], 394, 128, 0),
    ("dense_causal", [(i * 77, 77, 0, 0) for i in range(3)], 231, 128, 1),
    ("shared", [(5, 4, 0, 5), (9, 13, 0, 5), (22, 1, 0, 5), (23, 7, 0, 5), (0, 5, 0, 0)], 30, 192, 1),
    ("ragged", [(0, 33, 0, 0), (33, 1, 0, 0), (34, 64, 0, 0), (98, 65, 0, 0)], 163, 64, 1),
    ("vit257", [(i * 257, 257, 0, 0) for i in range(2)], 514, 128, 0),
    ("long_prefix_causal", [(40, 130, 0, 40), (170, 37, 0, 40), (0, 40, 0, 0)], 207, 64, 1),
]


@pytest.mark.parametrize("name,seqs,T,W,causal", ATT_CASES)
def test_attention_fwd_split_f16(L, dev, name, seqs, T, W, causal):
    """The split-f16 attention kernel (3 f16 MFMAs per product) against the f64 reference: f32-grade."""
    qkv = synth.normal(3, "att." + name, (T, 3 * W), 1.5)
    sq = (L.Seq * len(seqs))(*[L.Seq(*s) for s in seqs])
    sbuf = torch.frombuffer(bytearray(bytes(sq)), dtype=torch.int32).to(dev)
    out = torch.zeros(T, W, device=dev)
    L.check(L.lib().rlcf_attention_fwd(qkv.to(dev).data_ptr(), sbuf.data_ptr(), len(seqs), max(s[1] for s in seqs), W, causal,
                                       out.data_ptr(), None, L.PREC_F16X3, st()))
    ref = _attn_ref(qkv.double(), seqs, W, causal)
    torch.testing.assert_close(out.cpu().double(), ref, atol=5e-6, rtol=1e-5)


@pytest.mark.parametrize("name,seqs,T,W,causal", ATT_CASES)
def test_attention_fwd_bwd(L, dev, name, seqs, T, W, causal):
    qkv = synth.normal(3, "att." + name, (T, 3 * W), 1.5)
    sq = (L.Seq * len(seqs))(*[L.Seq(*s) for s in seqs])
    sbuf = torch.frombuffer(bytearray(bytes(sq)), dtype=torch.int32).to(dev)
    qd = qkv.to(dev)
    out = torch.zeros(T, W, device=dev)
    lse = torch.zeros(T, W // 64, device=dev)
    mq = max(s[1] for s in seqs)
    L.check(L.lib().rlcf_attention_fwd(qd.data_ptr(), sbuf.data_ptr(), len(seqs), mq, W, causal, out.data_ptr(),
                                       lse.data_ptr(), L.PREC_F32, st()))
    q64 = qkv.double().requires_grad_(True)
    ref = _attn_ref(q64, seqs, W, causal)
    torch.testing.assert_close(out.cpu().double(), ref.detach(), atol=6e-6, rtol=1e-5)      # f32 accumulation over up to 257 keys of N(0, 1.5) data
    mk = max(s[1] + s[3] for s in seqs)
    if mk <= 320:
        do = synth.normal(3, "att.do." + name, (T, W))
        (ref * do.double()).sum().backward()
        dq = torch.zeros(T, 3 * W, device=dev)
        L.check(L.lib().rlcf_attention_bwd(qd.data_ptr(), do.to(dev).data_ptr(), sbuf.data_ptr(), len(seqs), mk, W, causal,
                                           dq.data_ptr(), st()))
        torch.testing.assert_close(dq.cpu().double(), q64.grad, atol=2e-5, rtol=1e-4)
    else:
        do = synth.normal(3, "att.do." + name, (T, W))
        (ref * do.double()).sum().backward()
    # the MFMA (flash-style) backward from the forward's output and log-sum-exp: any length, prefix and causal masks
    dq2 = torch.zeros(T, 3 * W, device=dev)
    L.check(L.lib().rlcf_attention_bwd_flash(qd.data_ptr(), out.data_ptr(), lse.data_ptr(), do.to(dev).data_ptr(), sbuf.data_ptr(), len(seqs),
                                             mq, W, causal, dq2.data_ptr(), st()))
    torch.testing.assert_close(dq2.cpu().double(), q64.grad, atol=2e-5, rtol=1e-4)
    # ... and its split-f16 form (attention_bwd_x3.hip: what the engine's backward passes run in RLCF_PREC_F16X3), on a gradient of
    # the magnitude the tuning paths see (1e-4: far below f16's normal range without the device-side power-of-two lift)
    for gs in (1.0, 1e-4):
        dq3 = torch.zeros(T, 3 * W, device=dev)
        dos = (do * gs).to(dev)
        L.check(L.lib().rlcf_attention_bwd_flash_prec(qd.data_ptr(), out.data_ptr(), lse.data_ptr(), dos.data_ptr(), sbuf.data_ptr(), len(seqs),
                                                      mq, W, causal, dq3.data_ptr(), L.PREC_F16X3, st()))
        # accuracy of the split-f16 products: every operand carries 22 bits, so a sum of products is off by ~2.4e-7 x the sum of the
        # |terms| — invisible in dK / dV here, visible (1e-4 of the largest gradient) in the rare dQ row whose dS entries are large
        # and cancel.  Asserted: 99.9 % of the elements as tight as the f32-MFMA kernel's check, none beyond 2e-4 of the largest gradient.
        err, gref = (dq3.cpu().double() / gs - q64.grad).abs(), q64.grad.abs()
        assert float((err > 3e-5 + 2e-4 * gref).double().mean()) < 1e-3
        assert float(err.max()) <= 2e-4 * float(gref.max())


@pytest.mark.parametrize("n,Cn,p", [(16, 50, 0.25), (64, 1000, 0.1), (8, 16, 0.5), (16, 50, 0.05)])
def test_entropy_select(L, dev, n, Cn, p):
    lg = synth.normal(6, "ops.logits", (16, 50), 3.0) if (n, Cn) == (16, 50) else synth.normal(6, "es", (n, Cn), 3.0)
    ent = torch.empty(n, device=dev)
    n_sel = int(n * p)
    idx = torch.full((max(n_sel, 1),), -1, dtype=torch.int32, device=dev)
    L.check(L.lib().rlcf_entropy_select(lg.to(dev).data_ptr(), n, Cn, n_sel, ent.data_ptr(), idx.data_ptr(), st()))
    torch.testing.assert_close(ent.cpu(), RR.entropy_rows(lg), atol=2e-6, rtol=1e-5)
    _, ref_idx = RR.select_confident_samples(lg, p)
    assert idx.cpu()[:n_sel].tolist() == ref_idx.tolist()
    if (n, Cn) == (16, 50):            # the reference's own output (tests/golden/ops.npz)
        g, _ = load_golden("ops")
        assert idx.cpu()[:n_sel].tolist() == g[f"select_idx_{p}"].tolist()


@pytest.mark.parametrize("flags_kw", [dict(), dict(reward_amplify=True), dict(process_batch=True),
                                      dict(min_entropy_reg=True), dict(reward_process=False),
                                      dict(process_batch=True, reward_amplify=True, min_entropy_reg=True)])
@pytest.mark.parametrize("n_sel,Cn,K,Dr", [(4, 16, 3, 64), (6, 1000, 3, 512), (3, 40, 1, 128), (5, 200, 5, 768),
                                           # retrieval-shaped banks (retrieval/clip_ret_policy.py:76-137 runs the same
                                           # top-K -> CLIPScore -> baseline -> weighted-CE loss over 5k images / 25k captions)
                                           (1, 5000, 5, 512), (2, 25000, 5, 512),
                                           # ... with the sample counts of retrieval/scripts/tta_coco_ret.sh:19-20 (12 text->image, 20 image->text)
                                           (1, 5000, 12, 512), (2, 25000, 20, 512)])
def test_reward_loss(L, dev, flags_kw, n_sel, Cn, K, Dr):
    from rlcf_amd.engine import TTAConfig
    cfg = TTAConfig(sample_k=K, **flags_kw)
    logits = synth.normal(7, "rl.logits", (n_sel, Cn), 2.0)
    cf = CR.l2_normalize(synth.normal(7, "rl.cf", (Cn, Dr)))
    ri = CR.l2_normalize(synth.normal(7, "rl.ri", (n_sel, Dr)) + 0.5)
    d = lambda t: t.to(dev).contiguous()
    topk = torch.empty(n_sel, K, dtype=torch.int32, device=dev)
    score = torch.empty(n_sel * K, device=dev)
    rew = torch.empty(n_sel * K, device=dev)
    loss = torch.empty(1, device=dev)
    dl = torch.empty(n_sel, Cn, device=dev)
    lg, cfd, rid = d(logits), d(cf), d(ri)
    L.check(L.lib().rlcf_reward_loss(lg.data_ptr(), Cn, None, n_sel, Cn, K, cfd.data_ptr(), rid.data_ptr(), Dr,
                                     cfg.clipscore_weight, cfg.flags(), cfg.min_entropy_w, topk.data_ptr(), score.data_ptr(),
                                     rew.data_ptr(), loss.data_ptr(), dl.data_ptr(), st()))
    x = logits.clone().requires_grad_(True)
    _, index = torch.topk(x, K, dim=-1)
    flat = index.flatten()
    sc = RR.clip_score(cf, ri, flat, K, cfg.clipscore_weight).reshape(-1)
    r = RR.rewards_post_process(sc if cfg.process_batch else sc.reshape(n_sel, -1), cfg.reward_process, cfg.reward_amplify)
    ce = torch.nn.functional.cross_entropy(torch.repeat_interleave(x, K, dim=0), flat, reduction="none")
    ref_loss = torch.mean(r * ce)
    if cfg.min_entropy_reg:
        ref_loss = ref_loss + cfg.min_entropy_w * RR.avg_entropy(x)
    ref_loss.backward()
    assert topk.cpu().tolist() == index.tolist()
    torch.testing.assert_close(score.cpu(), sc, atol=2e-6, rtol=1e-5)
    torch.testing.assert_close(rew.cpu(), r, atol=5e-5, rtol=2e-4)
    torch.testing.assert_close(loss.cpu()[0], ref_loss.detach(), atol=1e-5, rtol=2e-4)
    torch.testing.assert_close(dl.cpu(), x.grad, atol=2e-6, rtol=5e-4)


def test_adamw_matches_reference_fixture(L, dev):
    g, _ = load_golden("ops")
    p = synth.normal(8, "ops.p", (4, 64), 0.02).to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for s in range(3):
        gr = synth.normal(8, f"ops.g{s}", (4, 64), 1e-3).to(dev)
        L.check(L.lib().rlcf_adamw_step(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), s + 1, 7e-3,
                                        0.9, 0.999, 1e-8, 5e-4, st()))
        torch.testing.assert_close(p.cpu(), g[f"adamw_p{s + 1}"], atol=1e-7, rtol=1e-6)


def test_block_fixture_through_ops(L, dev):
    """One ResidualAttentionBlock (reference output in ops.npz) assembled from the op-level ABI."""
    g, _ = load_golden("ops")
    sd = {k: v.to(dev) for k, v in synth.make_state_dict(synth.GEOMETRIES["tiny"], seed=5).items()}
    p = "transformer.resblocks.0."
    W, Lq, B = 128, 9, 3
    for masked in (0, 1):
        x = synth.normal(4, "ops.blk", (Lq, B, W)).transpose(0, 1).contiguous().reshape(B * Lq, W).to(dev)
        T = B * Lq
        lib = L.lib()
        h = torch.empty_like(x)
        L.check(lib.rlcf_layernorm_fwd(x.data_ptr(), sd[p + "ln_1.weight"].data_ptr(), sd[p + "ln_1.bias"].data_ptr(),
                                       h.data_ptr(), T, W, st()))
        qkv = torch.empty(T, 3 * W, device=dev)
        L.check(lib.rlcf_gemm_nt(h.data_ptr(), W, sd[p + "attn.in_proj_weight"].data_ptr(), W,
                                 sd[p + "attn.in_proj_bias"].data_ptr(), None, 0, None, 0, qkv.data_ptr(), 3 * W, T, 3 * W, W,
                                 1.0, 0, 0, st()))
        seqs = [(i * Lq, Lq, 0, 0) for i in range(B)]
        sq = (L.Seq * B)(*[L.Seq(*s) for s in seqs])
        sbuf = torch.frombuffer(bytearray(bytes(sq)), dtype=torch.int32).to(dev)
        a = torch.empty(T, W, device=dev)
        L.check(lib.rlcf_attention_fwd(qkv.data_ptr(), sbuf.data_ptr(), B, Lq, W, masked, a.data_ptr(), None, 0, st()))
        x1 = torch.empty_like(x)
        L.check(lib.rlcf_gemm_nt(a.data_ptr(), W, sd[p + "attn.out_proj.weight"].data_ptr(), W,
                                 sd[p + "attn.out_proj.bias"].data_ptr(), x.data_ptr(), W, None, 0, x1.data_ptr(), W, T, W, W,
                                 1.0, 0, 0, st()))
        L.check(lib.rlcf_layernorm_fwd(x1.data_ptr(), sd[p + "ln_2.weight"].data_ptr(), sd[p + "ln_2.bias"].data_ptr(),
                                       h.data_ptr(), T, W, st()))
        f = torch.empty(T, 4 * W, device=dev)
        L.check(lib.rlcf_gemm_nt(h.data_ptr(), W, sd[p + "mlp.c_fc.weight"].data_ptr(), W, sd[p + "mlp.c_fc.bias"].data_ptr(),
                                 None, 0, None, 0, f.data_ptr(), 4 * W, T, 4 * W, W, 1.0, 1, 0, st()))
        y = torch.empty_like(x)
        L.check(lib.rlcf_gemm_nt(f.data_ptr(), 4 * W, sd[p + "mlp.c_proj.weight"].data_ptr(), 4 * W,
                                 sd[p + "mlp.c_proj.bias"].data_ptr(), x1.data_ptr(), W, None, 0, y.data_ptr(), W, T, W, 4 * W,
                                 1.0, 0, 0, st()))
        ref = g[f"block_y_{masked}"].transpose(0, 1).reshape(T, W)
        torch.testing.assert_close(y.cpu(), ref, atol=2e-5, rtol=1e-5)


# ------------------------------------------------------------------------------ engine level
def make_engine(meta_or_names, n_views, n_cls, text_mode, student_seed=11, reward_seed=23, bank_seed=7, n_ctx=4, prec=0):
    from rlcf_amd import _lib
    from rlcf_amd.engine import Engine
    s_name, r_name = meta_or_names
    sg, rg = synth.GEOMETRIES[s_name], synth.GEOMETRIES[r_name]
    # weights are generated on the device (the counter-hash generator is bit-identical on CPU and GPU: test_synth_generator_bit_identical_on_gpu) and
    # handed back as CPU copies for the oracle
    gdev = torch.device("cuda", torch.cuda.current_device())
    ssd_d, rsd_d = synth.make_state_dict(sg, student_seed, device=gdev), synth.make_state_dict(rg, reward_seed, device=gdev)
    eng = Engine(sg, rg, n_views, n_cls, prec)
    eng.load_state_dict(_lib.STUDENT, ssd_d)
    eng.load_state_dict(_lib.REWARD, rsd_d)
    eng.finalize()
    ssd, rsd = {k: v.cpu() for k, v in ssd_d.items()}, {k: v.cpu() for k, v in rsd_d.items()}
    tokens = synth.make_token_bank(sg, n_cls, seed=bank_seed, n_ctx=n_ctx)
    ctx0 = CR.ctx_from_tokens(ssd, synth.ctx_token_ids_default(sg, n_ctx))
    eng.set_class_bank(tokens, n_ctx, ctx0, text_mode)
    return eng, ssd, rsd, tokens, ctx0


@pytest.mark.parametrize("geo", ["tiny", "small", "tiny-rn"])
def test_synth_generator_bit_identical_on_gpu(L, dev, geo):
    """The seeded weight / view generator gives the same bits on the CPU (oracle, fixture generation) and on the GPU (tests, bench);
    ModifiedResNet BatchNorm statistics to within one ulp."""
    g = synth.GEOMETRIES[geo]
    a, b = synth.make_state_dict(g, 11), synth.make_state_dict(g, 11, device=dev)
    assert a.keys() == b.keys()
    for k in a:
        if geo.endswith("rn") and not torch.equal(a[k], b[k].cpu()):     # BatchNorm statistics go through exp(): last-bit differences
            torch.testing.assert_close(a[k], b[k].cpu(), rtol=1e-6, atol=0)
        else:
            assert torch.equal(a[k], b[k].cpu()), k
    assert torch.equal(synth.make_views(1000, 5, g.image_resolution), synth.make_views(1000, 5, g.image_resolution, device=dev).cpu())


@pytest.mark.parametrize("geo", ["tiny", "small"])
def test_encode_image_vs_oracle(L, dev, geo):
    eng, ssd, rsd, tokens, ctx0 = make_engine((geo, geo), 8, 16, L.TEXT_SHARED)
    views = synth.make_views(1000, 5, synth.GEOMETRIES[geo].image_resolution)
    for which, sd in ((L.STUDENT, ssd), (L.REWARD, rsd)):
        f = eng.encode_image(which, views.to(dev)).cpu()
        ref = CR.l2_normalize(CR.encode_image(sd, views))
        torch.testing.assert_close(f, ref, atol=5e-6, rtol=1e-4)
    eng.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("geo,n_cls", [("tiny", 16), ("small", 40)])
def test_text_features_vs_oracle(L, dev, mode, geo, n_cls):
    eng, ssd, rsd, tokens, ctx0 = make_engine((geo, geo), 8, n_cls, mode)
    ctx = ctx0 + synth.normal(5, "ctx.delta", tuple(ctx0.shape), 0.01)
    t = eng.text_features(ctx.to(dev)).cpu()
    ref = CR.student_text_features(ssd, tokens, ctx)
    torch.testing.assert_close(t, ref, atol=5e-6, rtol=1e-4)
    rc = eng.reward_class_features().cpu()
    torch.testing.assert_close(rc, RR.reward_class_features(rsd, tokens), atol=5e-6, rtol=1e-4)
    # dense backward of an arbitrary dlogits against autograd on the oracle
    img = CR.l2_normalize(synth.normal(5, "img", (3, ssd["text_projection"].shape[1])))
    dl = synth.normal(5, "dl", (3, n_cls), 0.01)
    cg = ctx.clone().requires_grad_(True)
    logits = ssd["logit_scale"].exp() * img @ CR.student_text_features(ssd, tokens, cg).t()
    (logits * dl).sum().backward()
    g = eng.text_backward_dense(ctx.to(dev), img.to(dev), dl.to(dev)).cpu()
    assert (g - cg.grad).norm() / cg.grad.norm() < 2e-4
    eng.close()


TTA_FIXTURES = ["tta_tiny_s1", "tta_tiny_s3", "tta_tiny_amplify", "tta_tiny_batchproc", "tta_tiny_minent", "tta_tiny_k1",
                "tta_small_s1", "tta_tiny_rres", "tta_tiny_rnreward", "tta_tiny_rnstudent", "tta_tiny_front", "tta_tiny_middle", "tta_tiny_cls1"]


def _ctx_arrangement(meta, tokens):
    """(student_tokens, ctx_pos) of a fixture whose class tokens are not at the end of the prompt (None, None otherwise): what
    rlcf_amd.custom_clip.PromptLearner._arrangement hands to rlcf_engine_set_class_bank_ex."""
    ci = str(meta.get("ctx_init", "a_photo_of_a")).replace("_", " ")
    position, split = ("middle", ci.split(" ").index("[CLS]")) if "[CLS]" in ci else (str(meta.get("ctx_position", "end")), None)
    if position == "end":
        return None, None
    n = meta["n_ctx"]
    half = (split if split is not None else n // 2) if position == "middle" else 0
    stok, pos = tokens.clone(), torch.zeros(tokens.shape[0], n, dtype=torch.int64)
    for i in range(tokens.shape[0]):
        nl = int(tokens[i].argmax()) - 1 - n - 1
        stok[i, 1 + half: 1 + half + nl] = tokens[i, 1 + n: 1 + n + nl]
        stok[i, 1: 1 + half] = 0
        stok[i, 1 + half + nl: 1 + n + nl] = 0
        pos[i, :half] = torch.arange(1, 1 + half)
        pos[i, half:] = torch.arange(1 + half + nl, 1 + n + nl)
    return stok, pos


def _cfg_from_meta(meta, sparse=True):
    from rlcf_amd.engine import TTAConfig
    return TTAConfig(selection_p=meta["selection_p"], tta_steps=meta["tta_steps"], sample_k=meta["sample_k"], lr=meta["lr"],
                     weight_decay=meta["weight_decay"], reward_amplify=bool(meta.get("reward_amplify", False)),
                     process_batch=bool(meta.get("process_batch", False)), min_entropy_reg=bool(meta.get("min_entropy_reg", 0)),
                     min_entropy_w=float(meta.get("min_entropy_w", 0.2)), sparse_backward=sparse)


def _check_against(o, g, meta, final_atol=1e-3):
    c = lambda k: o[k].cpu()
    assert c("selected_idx").tolist() == g["selected_idx"].tolist()
    assert c("topk_idx").reshape(-1).tolist() == g["topk_idx"].reshape(-1).tolist()
    assert c("top5").tolist()[: g["top5"].numel()] == g["top5"].tolist()
    torch.testing.assert_close(c("logits"), g["logits"], atol=1e-3, rtol=0)
    torch.testing.assert_close(c("entropy"), g["entropy"], atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(c("clip_score"), g["clip_score"].reshape(-1), atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(c("rewards"), g["rewards"].reshape(-1), atol=5e-5, rtol=1e-3)
    torch.testing.assert_close(c("final_logits"), g["final_logits"], atol=final_atol, rtol=0)
    if meta["tta_steps"] == 1:
        gr, og = g["ctx_grad"], c("ctx_grad")
        if gr.norm() == 0:                 # all K CLIP scores of every view clamp to 0 -> zero rewards -> zero gradient
            assert og.abs().max() < 1e-9
            return
        assert (og - gr).norm() / gr.norm() < 1e-3
        big = gr.abs() > 1e-3 * gr.abs().max()
        assert torch.equal(torch.sign(og[big]), torch.sign(gr[big]))
    # post-AdamW prompt: Adam's first step is -lr*sign(g), so an element may differ from the reference only where
    # the reference gradient is ~0 (its sign is noise).  Single-step fixtures carry that gradient: every differing element must be
    # such an element; the count is reported, not rate-limited.
    d = (c("ctx_after") - g["ctx_after"]).abs()
    differing = d > 1e-4
    if meta["tta_steps"] == 1:
        gr = g["ctx_grad"]
        fragile = gr.abs() <= 1e-3 * gr.abs().max()
        assert not (differing & ~fragile).any(), f"{int((differing & ~fragile).sum())} prompt elements with a solid gradient differ"
    else:                                  # multi-step fixtures hold no per-step gradients: a handful of sign-fragile elements at most
        assert int(differing.sum()) <= max(2, differing.numel() // 100)
    if differing.any():
        print(f"[ctx_after] {int(differing.sum())} of {differing.numel()} elements differ by > 1e-4 (all at ~zero reference gradient)")


@pytest.mark.parametrize("prec", [0, 2])           # 2 = split-f16, the product default (runtime.Session, bench.py)
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("sparse", [True, False])
@pytest.mark.parametrize("name", TTA_FIXTURES)
def test_tta_sample_matches_reference_fixture(L, dev, name, sparse, mode, prec):
    g, meta = load_golden(name)
    eng, ssd, rsd, tokens, ctx0 = make_engine((meta["student"], meta["reward"]), meta["n_views"], meta["n_cls"], mode,
                                              meta["student_seed"], meta["reward_seed"], meta["bank_seed"], meta["n_ctx"], prec=prec)
    stok, pos = _ctx_arrangement(meta, tokens)
    if pos is not None:                 # class tokens in front of / inside the context: rlcf_engine_set_class_bank_ex
        eng.set_class_bank(tokens, meta["n_ctx"], ctx0, mode, stok, pos)
    views = synth.make_views(meta["view_seed"], meta["n_views"], synth.GEOMETRIES[meta["student"]].image_resolution)
    o = eng.tta_sample(views.to(dev), _cfg_from_meta(meta, sparse))
    torch.cuda.synchronize()
    _check_against(o, g, meta)
    if pos is not None:                 # ... and through the sample-batched call (per-sample prompt copies, scanned ctx gradient)
        big, *_ = make_engine((meta["student"], meta["reward"]), meta["n_views"] * 2, meta["n_cls"], mode, meta["student_seed"],
                              meta["reward_seed"], meta["bank_seed"], meta["n_ctx"], prec=prec)
        big.set_class_bank(tokens, meta["n_ctx"], ctx0, mode, stok, pos)
        top5, fl = big.tta_batch(torch.stack([views, views]).to(dev), _cfg_from_meta(meta, True), want_logits=True)
        for b in range(2):
            assert top5[b].cpu().tolist() == g["top5"].tolist()
            torch.testing.assert_close(fl[b].cpu(), g["final_logits"][0], atol=1e-3, rtol=0)
        big.close()
    eng.close()


def make_ensemble_engine(meta, mode, n_views=None, prec=0, device=None):
    from rlcf_amd import _lib
    from rlcf_amd.engine import Engine
    sg = synth.GEOMETRIES[meta["student"]]
    ssd = synth.make_state_dict(sg, meta["student_seed"], device=device) if device is not None else synth.make_state_dict(sg, meta["student_seed"])
    members = synth.reward_members(meta["reward"], meta["reward_seeds"], device=device)     # (full-size members: generated on the GPU)
    eng = Engine(sg, [g for g, _ in members], n_views or meta["n_views"], meta["n_cls"], prec)
    eng.load_state_dict(_lib.STUDENT, ssd)
    for m, (_, sd) in enumerate(members):
        eng.load_state_dict(_lib.REWARD + m, sd)
    eng.finalize()
    tokens = synth.make_token_bank(sg, meta["n_cls"], seed=meta["bank_seed"], n_ctx=meta["n_ctx"])
    eng.set_class_bank(tokens, meta["n_ctx"], CR.ctx_from_tokens(ssd, synth.ctx_token_ids_default(sg, meta["n_ctx"])), mode)
    return eng, members, tokens


@pytest.mark.parametrize("prec", [0, 2])
@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("sparse", [True, False])
@pytest.mark.parametrize("name", ["tta_tiny_ens", "tta_tiny_ensmean", "tta_tiny_ensrn"])
def test_reward_ensemble_matches_reference_fixture(L, dev, name, sparse, mode, prec):
    """CLIPRewardsMultiple (clip_reward.py:180-307): three reward CLIPs (one at another input resolution), weighted / mean."""
    g, meta = load_golden(name)
    eng, members, tokens = make_ensemble_engine(meta, mode, prec=prec)
    eng.set_reward_mix(g["reward_weights"].tolist(), mean=not meta.get("weighted_scores", 1))
    for m in range(len(members)):
        torch.testing.assert_close(eng.reward_class_features(m).cpu(), g[f"reward_class_features_{m}"], atol=2e-5, rtol=1e-4)
    views = synth.make_views(meta["view_seed"], meta["n_views"], synth.GEOMETRIES[meta["student"]].image_resolution)
    o = eng.tta_sample(views.to(dev), _cfg_from_meta(meta, sparse))
    torch.cuda.synchronize()
    for m in range(len(members)):
        torch.testing.assert_close(o["reward_image_features"][m].cpu(), g[f"reward_image_features_{m}"], atol=2e-5, rtol=1e-4)
    _check_against(o, g, meta)
    # the fused sample batch mixes the same way
    top5 = eng.tta_batch(views.to(dev)[None], _cfg_from_meta(meta, True))
    assert top5[0].cpu().tolist() == g["top5"].tolist()
    eng.close()


def test_reward_loss_ensemble_abi(L, dev):
    """rlcf_reward_loss_ensemble with one member == rlcf_reward_loss; with mix (a, b) on the same member == (a+b) * score."""
    import ctypes as C
    n_sel, Cn, K, Dr = 4, 50, 3, 64
    logits = synth.normal(31, "ens.logits", (n_sel, Cn), 2.0).to(dev)
    cf = CR.l2_normalize(synth.normal(31, "ens.cf", (Cn, Dr))).to(dev)
    im = CR.l2_normalize(synth.normal(31, "ens.im", (n_sel, Dr))).to(dev)

    def run(n, mix, mean):
        topk = torch.empty(n_sel, K, dtype=torch.int32, device=dev)
        sc, rw, dl = torch.empty(n_sel * K, device=dev), torch.empty(n_sel * K, device=dev), torch.empty(n_sel, Cn, device=dev)
        loss = torch.empty(1, device=dev)
        if n == 0:
            rc = L.lib().rlcf_reward_loss(logits.data_ptr(), Cn, None, n_sel, Cn, K, cf.data_ptr(), im.data_ptr(), Dr, 2.5, 1, 0.0,
                                          topk.data_ptr(), sc.data_ptr(), rw.data_ptr(), loss.data_ptr(), dl.data_ptr(), None)
        else:
            cfs = (C.c_void_p * n)(*[cf.data_ptr()] * n)
            ims = (C.c_void_p * n)(*[im.data_ptr()] * n)
            drs = (C.c_int * n)(*[Dr] * n)
            mx = (C.c_float * n)(*mix)
            rc = L.lib().rlcf_reward_loss_ensemble(logits.data_ptr(), Cn, None, n_sel, Cn, K, n, cfs, ims, drs, mx, mean, 2.5, 1, 0.0,
                                                   topk.data_ptr(), sc.data_ptr(), rw.data_ptr(), loss.data_ptr(), dl.data_ptr(), None)
        L.check(rc, "reward_loss")
        torch.cuda.synchronize()
        return sc.cpu(), rw.cpu(), dl.cpu()

    base = run(0, None, 0)
    one = run(1, [1.0], 0)
    assert torch.equal(base[0], one[0]) and torch.equal(base[2], one[2])
    two = run(2, [0.25, 0.5], 0)
    torch.testing.assert_close(two[0], 0.75 * base[0], atol=1e-7, rtol=1e-6)
    mean3 = run(3, [1.0, 1.0, 1.0], 1)
    torch.testing.assert_close(mean3[0], base[0], atol=1e-7, rtol=1e-6)


def test_tta_batch_and_reset(L, dev):
    """Per-sample reset (tpt_cls_rl.py:251-255): a batch gives the same predictions as one-by-one calls."""
    g, meta = load_golden("tta_tiny_s1")
    eng, *_ = make_engine((meta["student"], meta["reward"]), 8, 16, L.TEXT_SHARED)
    cfg = _cfg_from_meta(meta)
    R = synth.GEOMETRIES["tiny"].image_resolution
    vs = torch.stack([synth.make_views(1000 + i, 8, R) for i in range(3)]).to(dev)
    top5, fl = eng.tta_batch(vs, cfg, want_logits=True)
    for i in (2, 0, 1):
        o = eng.tta_sample(vs[i], cfg, want_intermediates=False)
        assert o["top5"].tolist() == top5[i].tolist()
        # dK/dV of shared prefix keys accumulate with float atomics: order-dependent in the last bits
        torch.testing.assert_close(o["final_logits"][0], fl[i], atol=2e-4, rtol=0)
    assert top5[0].cpu().tolist() == g["top5"].tolist()
    eng.close()


@pytest.mark.parametrize("steps", [1, 3])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("geo,reward,n_cls,p", [("tiny", "tiny-r", 16, 0.5), ("small", "small", 40, 0.25), ("tiny-rn", "tiny-r", 16, 0.5)])
def test_fused_sample_batch_equals_per_sample(L, dev, geo, reward, n_cls, p, mode, steps):
    """rlcf_tta_batch runs B samples per tower pass when the engine has room for B*N views; every sample must come
    out as if it had been processed alone (independent units) — also with several tuning steps
    (rlcf-prompt.sh runs --tta_steps 3), where every sample carries its own prompt through the later steps."""
    from rlcf_amd.engine import TTAConfig
    N, B = 8, 3
    R = synth.GEOMETRIES[geo].image_resolution
    cfg = TTAConfig(selection_p=p, tta_steps=steps)
    vs = torch.stack([synth.make_views(2000 + i, N, R) for i in range(B + 2)]).to(dev)     # 5 samples: 3 fused + 2 fused
    one, *_ = make_engine((geo, reward), N, n_cls, mode)
    ref = [one.tta_sample(vs[i], cfg, want_intermediates=False) for i in range(B + 2)]
    one.close()
    big, *_ = make_engine((geo, reward), N * B, n_cls, mode)
    top5, fl = big.tta_batch(vs, cfg, want_logits=True)
    for i in range(B + 2):
        if steps == 1:
            assert top5[i].tolist() == ref[i]["top5"].tolist()
            torch.testing.assert_close(fl[i], ref[i]["final_logits"][0], atol=2e-4, rtol=0)
        else:       # Adam's first steps are ~ -lr*sign(g): float-atomic order can flip near-zero gradient elements
            assert top5[i][0].item() == ref[i]["top5"][0].item()
            torch.testing.assert_close(fl[i], ref[i]["final_logits"][0], atol=1e-3, rtol=0)
    big.close()


def test_fused_multi_step_matches_reference_fixture(L, dev):
    """The fused sample-batch path with --tta_steps 3 against the reference's own 3-step run (tta_tiny_s3)."""
    g, meta = load_golden("tta_tiny_s3")
    eng, *_ = make_engine((meta["student"], meta["reward"]), 8 * 2, 16, L.TEXT_SHARED)
    R = synth.GEOMETRIES["tiny"].image_resolution
    vs = torch.stack([synth.make_views(meta["view_seed"], 8, R), synth.make_views(2001, 8, R)]).to(dev)
    top5, fl = eng.tta_batch(vs, _cfg_from_meta(meta), want_logits=True)
    assert top5[0].cpu().tolist() == g["top5"].tolist()
    torch.testing.assert_close(fl[0].cpu(), g["final_logits"][0], atol=1e-3, rtol=0)
    eng.close()


@pytest.mark.parametrize("prec", [0, 2])
@pytest.mark.parametrize("arch,tag,nv", [("tiny-rn", "tinyrn", 3), ("RN50", "rn50", 2), ("RN50x64", "rn50x64", 1)])
def test_modified_resnet_matches_reference_fixture(L, dev, arch, tag, nv, prec):
    """ModifiedResNet image tower (model.py:94-154; BatchNorm folded, NHWC GEMM convolutions) vs the reference CLIP class."""
    from rlcf_amd.engine import Engine
    g, _ = load_golden("modules_rn")
    geo = synth.GEOMETRIES[arch]
    sd = synth.make_state_dict(geo, 11, device=dev)
    eng = Engine(geo, None, 4, 8, prec)
    eng.load_state_dict(L.STUDENT, sd)
    eng.finalize()
    views = synth.make_views(1000, nv, geo.image_resolution)
    f = eng.encode_image(L.STUDENT, views.to(dev)).cpu()
    ref = CR.l2_normalize(g[f"{tag}_image"])
    torch.testing.assert_close(f, ref, atol=2e-5, rtol=1e-4)
    if nv > 1:                   # chunked == unchunked, any order
        f2 = eng.encode_image(L.STUDENT, views.flip(0).to(dev)).cpu().flip(0)
        torch.testing.assert_close(f2, f, atol=1e-6, rtol=0)
    eng.close()


@pytest.mark.parametrize("ri,ro", [(32, 64), (224, 336), (64, 32), (17, 40)])
def test_bicubic_resample_vs_torch(L, dev, ri, ro):
    """The engine's reward-resolution change vs nn.functional.interpolate(mode='bicubic', align_corners=True)."""
    from rlcf_amd.engine import Engine
    geo = synth.ClipGeometry(64, ro, 1, 64, ro // 4 if ro % 4 == 0 else ro, 77, 1024, 64, 1, 1) if ro != 336 else synth.ClipGeometry(64, 336, 1, 64, 14, 77, 1024, 64, 1, 1)
    sd = synth.make_state_dict(geo, 3)
    eng = Engine(geo, None, 4, 8)
    eng.load_state_dict(L.STUDENT, sd)
    eng.finalize()
    x = synth.normal(12, "bic", (3, 3, ri, ri))
    f = eng.encode_image(L.STUDENT, x.to(dev)).cpu()
    ref = CR.l2_normalize(CR.encode_image(sd, torch.nn.functional.interpolate(x, size=ro, mode="bicubic", align_corners=True)))
    torch.testing.assert_close(f, ref, atol=1e-5, rtol=1e-4)
    eng.close()


def test_errors_are_loud(L, dev):
    from rlcf_amd.engine import TTAConfig
    eng, *_ = make_engine(("tiny", "tiny-r"), 8, 16, L.TEXT_SHARED)
    views = synth.make_views(1000, 8, 32).to(dev)
    with pytest.raises(L.RlcfError):     # int(8*0.1) == 0 selected views
        eng.tta_sample(views, TTAConfig(selection_p=0.1))
    with pytest.raises(L.RlcfError):
        eng.tta_sample(torch.cat([views, views]), TTAConfig(selection_p=0.5))   # more views than max_views
    with pytest.raises(L.RlcfError):     # the mix must name every reward slot
        eng.set_reward_mix([0.5, 0.5])
    with pytest.raises(L.RlcfError):     # the momentum EMA needs momentum in [0, 1]
        eng.momentum_update(eng.ln_params(), 1.5, 1.0, False)
    eng.close()
    from rlcf_amd.engine import Engine
    tiny, tr = synth.GEOMETRIES["tiny"], synth.GEOMETRIES["tiny-r"]
    with pytest.raises(L.RlcfError):     # at most RLCF_MAX_REWARDS reward models
        Engine(tiny, [tr] * 5, 8, 16)
    four = Engine(tiny, [tr] * 4, 8, 16)                     # ... and four are fine
    four.load_state_dict(L.STUDENT, synth.make_state_dict(tiny, 11))
    for m in range(4):
        four.load_state_dict(L.REWARD + m, synth.make_state_dict(tr, 23 + m))
    four.finalize()
    four.set_reward_mix([0.25] * 4)
    tokens = synth.make_token_bank(tiny, 16, seed=7, n_ctx=4)
    four.set_class_bank(tokens, 4, CR.ctx_from_tokens(synth.make_state_dict(tiny, 11), synth.ctx_token_ids_default(tiny, 4)), L.TEXT_SHARED)
    o = four.tta_sample(views, TTAConfig(selection_p=0.5))
    assert len(o["reward_image_features"]) == 4 and torch.isfinite(o["final_logits"]).all()
    four.close()
    # a ModifiedResNet student: the norm-layer path tunes its BatchNorms (row a-R); every-parameter tuning of it runs too (round 4)
    rn = synth.GEOMETRIES["tiny-rn32"]
    e2 = Engine(rn, tr, 8, 16)
    e2.load_state_dict(L.STUDENT, synth.make_state_dict(rn, 11)); e2.load_state_dict(L.REWARD, synth.make_state_dict(tr, 23))
    e2.finalize()
    tok2 = synth.make_token_bank(rn, 16, seed=7, n_ctx=4)
    e2.set_class_bank(tok2, 4, CR.ctx_from_tokens(synth.make_state_dict(rn, 11), synth.ctx_token_ids_default(rn, 4)), L.TEXT_SHARED)
    o = e2.tta_sample_ln(views, TTAConfig(selection_p=0.5))
    assert torch.isfinite(o["final_logits"]).all() and torch.isfinite(o["ln_grad"]).all() and o["ln_grad"].abs().max() > 0
    ov = e2.tta_sample_visual(views, TTAConfig(selection_p=0.5, lr=1e-4))
    assert torch.isfinite(ov["final_logits"]).all() and torch.isfinite(ov["vis_grad"]).all() and ov["vis_grad"].abs().max() > 0
    e2.close()


# ------------------------------------------------------------------------------ full geometry (BASELINE configs 0/1)
@pytest.mark.parametrize("name", ["tta_b16_n8", "tta_b16_n64", "tta_b16_rl14_s3"])
@pytest.mark.parametrize("mode,sparse,prec", [(2, True, 0), (1, True, 0), (0, False, 0), (2, True, 2), (0, False, 2)])
def test_vit_b16_tta_matches_reference_fixture(L, dev, name, mode, sparse, prec):
    """ViT-B/16 student + ViT-B/16 reward, C=1000, outputs of the REFERENCE itself
    (tests/golden/make_golden.py --only b16n8,b16n64): logits within 1e-3, identical top-1/top-5.  tta_b16_rl14_s3 is the
    setting of TPT/scripts/rlcf-prompt.sh: ViT-L/14 reward model, three tuning steps."""
    g, meta = load_golden(name)
    geo = synth.GEOMETRIES[meta["student"]]
    rgeo = synth.GEOMETRIES[meta["reward"]]
    from rlcf_amd.engine import Engine
    ssd = synth.make_state_dict(geo, meta["student_seed"], device=dev)
    rsd = synth.make_state_dict(rgeo, meta["reward_seed"], device=dev)
    eng = Engine(geo, rgeo, meta["n_views"], meta["n_cls"], prec)
    eng.load_state_dict(L.STUDENT, ssd)
    eng.load_state_dict(L.REWARD, rsd)
    eng.finalize()
    tokens = synth.make_token_bank(geo, meta["n_cls"], seed=meta["bank_seed"], n_ctx=meta["n_ctx"])
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(geo, meta["n_ctx"]), device=dev)].clone()
    eng.set_class_bank(tokens, meta["n_ctx"], ctx0, mode)
    rc = eng.reward_class_features().cpu()
    torch.testing.assert_close(rc[: g["reward_class_features"].shape[0]], g["reward_class_features"], atol=2e-5, rtol=1e-4)
    views = synth.make_views(meta["view_seed"], meta["n_views"], geo.image_resolution, device=dev)
    o = eng.tta_sample(views, _cfg_from_meta(meta, sparse))
    torch.cuda.synchronize()
    _check_against(o, g, meta, final_atol=1e-3)
    assert int(o["final_logits"].argmax()) == int(g["final_logits"].argmax())
    torch.testing.assert_close(o["reward_image_features"].cpu(), g["reward_image_features"], atol=2e-5, rtol=1e-4)
    if meta["tta_steps"] > 1 and sparse and mode == 2:        # the fused sample batch gives the same prediction
        big = Engine(geo, rgeo, meta["n_views"] * 2, meta["n_cls"], prec)
        big.load_state_dict(L.STUDENT, ssd); big.load_state_dict(L.REWARD, rsd); big.finalize()
        big.set_class_bank(tokens, meta["n_ctx"], ctx0, mode)
        top5 = big.tta_batch(torch.stack([views, synth.make_views(7, meta["n_views"], geo.image_resolution, device=dev)]), _cfg_from_meta(meta, True))
        assert int(top5[0, 0]) == int(g["top5"][0])
        big.close()
    eng.close()


@pytest.mark.parametrize("prec", [0, 2])
def test_modules_fixture(L, dev, prec):
    """encode_image of the reference CLIP class at ViT-B/16 and ViT-L/14 geometry (modules.npz)."""
    g, _ = load_golden("modules")
    from rlcf_amd.engine import Engine
    for arch, tag in (("ViT-B/16", "b16"), ("ViT-L/14", "l14")):
        geo = synth.GEOMETRIES[arch]
        sd = synth.make_state_dict(geo, 11, device=dev)
        eng = Engine(geo, None, 8, 8, prec)
        eng.load_state_dict(L.STUDENT, sd)
        eng.finalize()
        views = synth.make_views(1000, 2, geo.image_resolution, device=dev)
        views = torch.cat([views, views, views, views])[:8]      # 8 views: large enough for the split-f16 GEMM path
        f = eng.encode_image(L.STUDENT, views).cpu()
        ref = g[f"{tag}_image"]
        ref = ref / ref.norm(dim=-1, keepdim=True)
        torch.testing.assert_close(f, torch.cat([ref, ref, ref, ref]), atol=2e-5, rtol=1e-4)
        eng.close()


# ------------------------------------------------------------------------------ reference-shaped Python surface
def _harness_objects(dev, meta):
    import copy
    import types
    from rlcf_amd import clip_reward, clip_store, custom_clip, runtime
    runtime.reset_session()
    sg = synth.GEOMETRIES[meta["student"]]
    clip_store.register_checkpoint(meta["student"], sg, synth.make_state_dict(sg, meta["student_seed"]))
    ensemble = "+" in meta["reward"]
    if ensemble:        # get_reward_model(multiple_reward_models=1): CLIPRewardsMultiple over a fixed arch list (clip_reward.py:29-34)
        names = meta.get("reward_archs", "ViT-L/14@336px+ViT-L/14+ViT-B/16").split("+")    # the names the generator bound the members to
        for n, (geo, sd) in zip(names, synth.reward_members(meta["reward"], meta["reward_seeds"])):
            clip_store.register_checkpoint(n, geo, sd)
        clip_reward.ENSEMBLE_ARCHS[:] = names
    else:
        rg = synth.GEOMETRIES[meta["reward"]]
        clip_store.register_checkpoint(meta["reward"] + "#r", rg, synth.make_state_dict(rg, meta["reward_seed"]))
    bank = clip_store.SyntheticBank(sg, meta["n_cls"], meta["n_ctx"], meta["bank_seed"])
    clip_store.set_tokenizer(bank.tokenize)
    args = types.SimpleNamespace(tta_steps=meta["tta_steps"], selection_p=meta["selection_p"], gpu=0, tpt=True, print_freq=1000,
                                 min_entropy_reg=meta.get("min_entropy_reg", 0), min_entropy_w=meta.get("min_entropy_w", 0.2),
                                 reward_arch=meta["reward"] + "#r", multiple_reward_models=int(ensemble),
                                 weighted_scores=meta.get("weighted_scores", 1), sample_k=meta["sample_k"],
                                 reward_amplify=meta.get("reward_amplify", False), reward_process=True,
                                 process_batch=meta.get("process_batch", False))
    if str(meta.get("ctx_position", "end")) != "end":   # (get_coop has no ctx_position argument: the constructor it wraps does)
        model = custom_clip.ClipTestTimeTuning(dev, bank.classnames, None, arch=meta["student"], n_ctx=meta["n_ctx"], ctx_init="a_photo_of_a",
                                               ctx_position=str(meta["ctx_position"]))
    else:
        model = custom_clip.get_coop(meta["student"], "I", dev, meta["n_ctx"], str(meta.get("ctx_init", "a_photo_of_a")), classnames=bank.classnames)
    for name, p in model.named_parameters():            # tpt_cls_rl.py:103-105
        if "prompt_learner" not in name:
            p.requires_grad_(False)
    optimizer = torch.optim.AdamW(model.prompt_learner.parameters(), meta["lr"], weight_decay=meta["weight_decay"])
    optim_state = copy.deepcopy(optimizer.state_dict())
    reward_model = clip_reward.get_reward_model(dev, args)
    model.reset_classnames(bank.classnames, meta["student"])
    reward_model.set_class_features(tokenized_classes=model.prompt_learner.tokenized_prompts)
    return model, optimizer, optim_state, reward_model, args


@pytest.mark.parametrize("name", ["tta_tiny_front", "tta_tiny_middle", "tta_tiny_cls1", "tta_tiny_s1", "tta_tiny_s3", "tta_small_s1", "tta_tiny_rres", "tta_tiny_ens", "tta_tiny_ensmean",
                                  "tta_tiny_ensrn", "tta_tiny_rnreward", "tta_tiny_rnstudent"])
def test_reference_harness_runs_on_the_hip_path(L, dev, name):
    """The reference's main_worker/test_time_adapt_eval call sequence (tpt_cls_rl.py:94-190,219-279) with this
    package's classes in place of the reference's: same ctx update and final logits as the reference run."""
    from rlcf_amd import runtime, tpt_cls_rl
    g, meta = load_golden(name)
    model, optimizer, optim_state, reward_model, args = _harness_objects(dev, meta)
    views = synth.make_views(meta["view_seed"], meta["n_views"], synth.GEOMETRIES[meta["student"]].image_resolution)
    target = int(g["top5"][0])
    loader = [([v.unsqueeze(0) for v in views], torch.tensor([target]))]
    acc = tpt_cls_rl.test_time_adapt_eval(loader, model, optimizer, optim_state, None, args, reward_model=reward_model)
    assert acc == [100.0, 100.0]
    d = (model.prompt_learner.ctx.detach().cpu() - g["ctx_after"]).abs()
    assert (d > 1e-4).float().mean() < 0.01
    with torch.no_grad():
        out = model(views[:1].to(dev))
    torch.testing.assert_close(out.cpu(), g["final_logits"], atol=1e-3, rtol=0)
    runtime.reset_session()


def test_harness_images_per_pass(L, dev):
    """test_time_adapt_eval(images_per_pass=B): B test images per engine call give the accuracy of the one-by-one loop."""
    from rlcf_amd import runtime, tpt_cls_rl
    g, meta = load_golden("tta_tiny_s1")
    R = synth.GEOMETRIES[meta["student"]].image_resolution
    samples = [synth.make_views(1000 + i, meta["n_views"], R) for i in range(5)]
    res = {}
    for ipp in (1, 3):
        model, optimizer, optim_state, reward_model, args = _harness_objects(dev, meta)
        # targets: the reference's own prediction for sample 0, class 0 for the others (so that both top-1 and top-5 counts vary)
        loader = [([v.unsqueeze(0) for v in s], torch.tensor([int(g["top5"][0]) if i == 0 else 0])) for i, s in enumerate(samples)]
        res[ipp] = tpt_cls_rl.test_time_adapt_eval(loader, model, optimizer, optim_state, None, args, reward_model=reward_model, images_per_pass=ipp)
        runtime.reset_session()
    assert res[1] == res[3] and res[1][0] >= 20.0


def test_pretrained_prompt_load_flow(L, dev):
    """`--load` of the reference harness (tpt_cls_rl.py:95-101): ctx.copy_(pretrained); ctx_init_state = pretrained — WITHOUT a
    later reset_classnames.  The tuned result must be the oracle's from that starting prompt (not from the hand-written one)."""
    from rlcf_amd import runtime, tpt_cls_rl
    g, meta = load_golden("tta_tiny_s1")
    model, optimizer, optim_state, reward_model, args = _harness_objects(dev, meta)
    pre = synth.normal(31, "coop.ctx", tuple(model.prompt_learner.ctx.shape), 0.02).to(dev)
    with torch.no_grad():
        model.prompt_learner.ctx.copy_(pre)
        model.prompt_learner.ctx_init_state = pre
    views = synth.make_views(meta["view_seed"], meta["n_views"], 32).to(dev)
    model.reset()
    optimizer.load_state_dict(optim_state)
    tpt_cls_rl.test_time_tuning(model, views, optimizer, None, args, reward_model=reward_model)
    out = model(views[:1]).cpu()
    sg, rg = synth.GEOMETRIES[meta["student"]], synth.GEOMETRIES[meta["reward"]]
    ssd, rsd = synth.make_state_dict(sg, meta["student_seed"]), synth.make_state_dict(rg, meta["reward_seed"])
    tokens = synth.make_token_bank(sg, meta["n_cls"], seed=meta["bank_seed"], n_ctx=meta["n_ctx"])
    ref = RR.tta_sample(ssd, rsd, views.cpu(), tokens, pre.cpu(),
                        RR.TTAHyper(selection_p=meta["selection_p"], tta_steps=meta["tta_steps"], sample_k=meta["sample_k"], lr=meta["lr"],
                                    weight_decay=meta["weight_decay"]))
    torch.testing.assert_close(out, ref["final_logits"], atol=1e-3, rtol=0)
    assert (out - g["final_logits"]).abs().max() > 1e-2          # ... and it is not the hand-written prompt's result
    runtime.reset_session()


@pytest.mark.parametrize("name", ["tta_tiny_s1", "tta_tiny_ens"])
def test_autograd_route_matches_reference_gradient(L, dev, name):
    """model(images) is differentiable w.r.t. ctx exactly like the reference module: an unmodified copy of the
    reference loop (torch ops for the loss, loss.backward()) yields the reference's ctx.grad."""
    from rlcf_amd import runtime, tpt_cls_rl
    g, meta = load_golden(name)
    model, optimizer, optim_state, reward_model, args = _harness_objects(dev, meta)
    views = synth.make_views(meta["view_seed"], meta["n_views"], 32).to(dev)
    output = model(views)
    torch.testing.assert_close(output.detach().cpu(), g["logits"], atol=1e-3, rtol=0)
    output, idx = tpt_cls_rl.select_confident_samples(output, args.selection_p)
    assert idx.cpu().tolist() == g["selected_idx"].tolist()
    reward_model.set_image_features(views[idx])
    bs, K = output.shape[0], reward_model.sample_k
    value, index = torch.topk(output, K, dim=-1)
    flat = index.flatten()
    score = reward_model.CLIPScore(class_index=flat, pairwise=False)
    rewards = reward_model.rewards_post_process(score if reward_model.process_batch else score.reshape(bs, -1))
    rep = torch.repeat_interleave(output, K, dim=0)
    loss = torch.mean(rewards * torch.nn.functional.cross_entropy(rep, flat, reduction="none"))
    optimizer.zero_grad()
    loss.backward()
    gr, og = g["ctx_grad"], model.prompt_learner.ctx.grad.cpu()
    assert (og - gr).norm() / gr.norm() < 1e-3
    torch.testing.assert_close(tpt_cls_rl.avg_entropy(output.detach()).cpu(), RR.avg_entropy(output.detach().cpu()), atol=1e-5, rtol=1e-5)
    runtime.reset_session()


# ------------------------------------------------------------------------------ LayerNorm-tuning path (BASELINE configs[2])
@pytest.mark.parametrize("prec", [0, 2])
@pytest.mark.parametrize("name", ["ln_tiny_s1", "ln_tiny_s3", "ln_small_s1", "ln_b16_n8", "ln_l14_n8", "ln_l14_n64"])     # ln_l14_n64 = BASELINE configs[2] at full size
def test_ln_tuning_matches_reference_fixture(L, dev, name, prec):
    """rlcf_tta_sample_ln vs the reference's CLIPCLS_TTA(only_norm=True) + test_time_tuning run (TPT/tune_cls_rl.py)."""
    if not os.path.exists(os.path.join(GOLDEN, name + ".npz")):
        pytest.skip("fixture not generated")
    g, meta = load_golden(name)
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
    o = eng.tta_sample_ln(views, _cfg_from_meta(meta))
    torch.cuda.synchronize()
    c = lambda k: o[k].cpu()
    assert c("selected_idx").tolist() == g["selected_idx"].tolist()
    assert c("topk_idx").reshape(-1).tolist() == g["topk_idx"].reshape(-1).tolist()
    assert c("top5").tolist()[: g["top5"].numel()] == g["top5"].tolist()
    torch.testing.assert_close(c("logits"), g["logits"], atol=1e-3, rtol=0)
    torch.testing.assert_close(c("rewards"), g["rewards"].reshape(-1), atol=5e-5, rtol=1e-3)
    multi = meta["tta_steps"] > 1
    torch.testing.assert_close(c("final_logits"), g["final_logits"], atol=1e-3, rtol=0)
    if not multi:
        gr, og = g["ln_grad"], c("ln_grad")
        assert gr.norm() > 0 and (og - gr).norm() / gr.norm() < 2e-3
    d = (c("ln_after") - g["ln_after"]).abs()
    assert (d > 0.1 * meta["lr"]).float().mean() < 0.01
    # the prompt path still sees pristine LayerNorms afterwards
    o2 = eng.tta_sample_ln(views, _cfg_from_meta(meta))
    torch.testing.assert_close(o2["final_logits"], o["final_logits"], atol=2e-4, rtol=0)
    eng.close()


# ------------------------------------------------------------------------------ full image-encoder tuning (scripts/rlcf-tune.sh)
def _tensor_norms(sd, keys, vec, base=None):
    out, off = [], 0
    for k in keys:
        n = sd[k].numel()
        v = vec[off: off + n].double()
        if base is not None:
            v = v - base[k].reshape(-1).double().to(v.device)
        out.append(v.norm())
        off += n
    assert off == vec.numel()
    return torch.stack(out).float().cpu()


@pytest.mark.parametrize("prec", [0, 2])
@pytest.mark.parametrize("name", ["vis_tiny_s1", "vis_tiny_s3", "vis_tinyp6_s3", "vis_small_s1", "vis_b16_s3"])
def test_visual_tuning_matches_reference_fixture(L, dev, name, prec):
    """rlcf_tta_sample_visual vs the reference's CLIPCLS_TTA(only_norm=False) + test_time_tuning run (TPT/tune_cls_rl.py, the
    configuration of scripts/rlcf-tune.sh): every visual parameter gets a gradient and an AdamW step."""
    if not os.path.exists(os.path.join(GOLDEN, name + ".npz")):
        pytest.skip("fixture not generated")
    g, meta = load_golden(name)
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
    base = eng.tta_sample_ln(views, _cfg_from_meta(meta))["final_logits"].clone()        # LayerNorm path before: must be unaffected after
    base_p = eng.tta_sample(views, _cfg_from_meta(meta))["final_logits"].clone()          # ... and so must the prompt path
    o = eng.tta_sample_visual(views, _cfg_from_meta(meta))
    torch.cuda.synchronize()
    c = lambda k: o[k].cpu()
    assert c("selected_idx").tolist() == g["selected_idx"].tolist()
    assert c("topk_idx").reshape(-1).tolist() == g["topk_idx"].reshape(-1).tolist()
    assert c("top5").tolist()[: g["top5"].numel()] == g["top5"].tolist()
    torch.testing.assert_close(c("logits"), g["logits"], atol=1e-3, rtol=0)
    torch.testing.assert_close(c("rewards"), g["rewards"].reshape(-1), atol=5e-5, rtol=1e-3)
    multi = meta["tta_steps"] > 1
    torch.testing.assert_close(c("final_logits"), g["final_logits"], atol=1e-3, rtol=0)
    keys = RR.visual_param_keys(ssd)
    grad, after = eng.merge_visual(o["ln_grad"], o["vis_grad"]), eng.merge_visual(o["ln_after"], o["vis_after"])
    torch.testing.assert_close(_tensor_norms(ssd, keys, grad), g["vis_grad_l2"], rtol=3e-3, atol=1e-9)
    torch.testing.assert_close(_tensor_norms(ssd, keys, after, ssd), g["vis_delta_l2"], rtol=0.01, atol=1e-7)
    if "vis_grad_sample" in g:
        gr, og = g["vis_grad_sample"], grad[::7].cpu()
        assert (og - gr).norm() / gr.norm() < 2e-3
        d = (after[::7].cpu() - g["vis_after_sample"]).abs()
        assert (d > 0.1 * meta["lr"]).float().mean() < 0.01
    # against the oracle on the same inputs (small geometries): the whole gradient vector
    if meta["student"] in ("tiny", "small") and not multi:
        ref = RR.tta_sample_ln({k: v.cpu() for k, v in ssd.items()}, {k: v.cpu() for k, v in rsd.items()}, views.cpu(), tokens,
                               RR.TTAHyper(selection_p=meta["selection_p"], tta_steps=meta["tta_steps"], sample_k=meta["sample_k"],
                                           lr=meta["lr"], weight_decay=meta["weight_decay"]), only_norm=False)
        assert (grad.cpu() - ref["ln_grad"]).norm() / ref["ln_grad"].norm() < 2e-3
    # the engine is back in its pristine state: the same call repeats, and the LayerNorm path gives what it gave before
    o2 = eng.tta_sample_visual(views, _cfg_from_meta(meta))
    torch.testing.assert_close(o2["final_logits"], o["final_logits"], atol=2e-4, rtol=0)
    torch.testing.assert_close(eng.tta_sample_ln(views, _cfg_from_meta(meta))["final_logits"], base, atol=1e-4, rtol=0)      # (float atomics in the LayerNorm-gradient reductions: last-bit run-to-run differences)
    torch.testing.assert_close(eng.tta_sample(views, _cfg_from_meta(meta))["final_logits"], base_p, atol=2e-4, rtol=0)
    torch.testing.assert_close(eng.visual_params(0), eng.visual_params(1), atol=0, rtol=0)
    # loading the adapted parameters (rlcf_engine_set_visual_params / set_ln_params: refreshed transposes and split copies) and
    # running plain inference on the clean view reproduces the call's own final logits; loading the checkpoint undoes it
    eng.set_ln_params(o["ln_after"]); eng.set_visual_params(o["vis_after"])
    lg = eng.logits(eng.encode_image(L.STUDENT, views[:1]), eng.text_features(ctx0))
    torch.testing.assert_close(lg, o["final_logits"], atol=2e-4, rtol=0)
    torch.testing.assert_close(eng.visual_params(0), o["vis_after"], atol=0, rtol=0)
    eng.set_ln_params(eng.ln_params(pristine=True)); eng.set_visual_params(eng.visual_params(2))
    torch.testing.assert_close(eng.tta_sample_ln(views, _cfg_from_meta(meta))["final_logits"], base, atol=1e-4, rtol=0)      # (float atomics in the LayerNorm-gradient reductions: last-bit run-to-run differences)
    torch.testing.assert_close(eng.visual_params(3), eng.visual_params(2), atol=0, rtol=0)      # momentum state untouched = checkpoint
    eng.close()


BN_NAMES = ["bn_tiny_train", "bn_tiny_train_s3", "bn_tiny_prior0", "bn_tiny_prior16_s3", "bn_rn50_train", "bn_rn50_prior16"]


@pytest.mark.parametrize("prec", [0, 2])
@pytest.mark.parametrize("name", BN_NAMES)
def test_bn_tuning_matches_reference_fixture(L, dev, name, prec):
    """Row a-R: a ModifiedResNet student under CLIPCLS_TTA(only_norm=True) -- rlcf_tta_sample_ln tunes its BatchNorm weights / biases --
    vs the reference's own run (tune_cls_rl.py harness body, nn.BatchNorm2d train mode or `_modified_bn_forward` under --prior_strength;
    tests/golden/make_golden.py groups bn / bnrn50): selection, sampled classes, rewards, the BatchNorm gradient, the adapted parameters,
    the running statistics the final inference saw and its logits (taken, as the reference takes them, with the norm layers still in
    train mode)."""
    g, meta = load_golden(name)
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
    eng.set_bn_prior_strength(meta["prior_strength"])
    keys = RR.visual_bn_keys({k: v for k, v in ssd.items()})
    assert int(eng.lib.rlcf_engine_ln_param_count(eng.h)) == sum(ssd[k].numel() for k in keys) == g["ln_grad"].numel()
    torch.testing.assert_close(eng.ln_params(pristine=True).cpu(), torch.cat([ssd[k].reshape(-1) for k in keys]).cpu(), atol=0, rtol=0)
    views = synth.make_views(meta["view_seed"], meta["n_views"], sg.image_resolution, device=dev)
    o = eng.tta_sample_ln(views, _cfg_from_meta(meta))
    torch.cuda.synchronize()
    c = lambda k: o[k].cpu()
    assert c("selected_idx").tolist() == g["selected_idx"].tolist()
    assert c("topk_idx").reshape(-1).tolist() == g["topk_idx"].reshape(-1).tolist()
    assert c("top5").tolist()[: g["top5"].numel()] == g["top5"].tolist()
    # first-pass logits: forward only.  Round 4 tightened this from 2e-3 to 3e-4 against the fixture and pinned it to the float64 value:
    # measured |HIP - f64| 4.7e-5 (train mode) / 1.6e-5 (prior 16) at RN50 in split-f16 mode, the reference's own float32 run 2.7e-5 /
    # 1.3e-5 (tests/golden/make_bn_f64.py; the old bar was simply loose, no reduction of the train-form BatchNorm costs 1e-3)
    torch.testing.assert_close(c("logits"), g["logits"], atol=3e-4, rtol=0)
    torch.testing.assert_close(c("rewards"), g["rewards"].reshape(-1), atol=5e-5, rtol=1e-3)
    # The float32 gradient of the REFERENCE is itself only good to `ref_err` of its norm here (2e-6 on the tiny towers, 1e-4 .. 7e-3 at
    # RN50: tests/golden/make_bn_f64.py has the why), so the HIP gradient is judged against the float64 value of the (reference-pinned)
    # oracle -- no further from it than twice the reference is -- and against the float32 fixture at the width of that noise band.
    z64 = np.load(os.path.join(GOLDEN, name + "_f64.npz"))
    assert (c("logits").double() - torch.from_numpy(z64["logits"])).abs().max() < max(2e-4, 4 * float(z64["ref_logit_err"]))
    g64, ref_err = torch.from_numpy(z64["ln_grad"]), float(z64["ref_err"])
    gr, og = g["ln_grad"], c("ln_grad")
    assert gr.norm() > 0
    assert (og.double() - g64).norm() / g64.norm() < max(2e-3, 2 * ref_err)
    assert (og - gr).norm() / gr.norm() < max(3e-3, 3 * ref_err)
    if meta["tta_steps"] == 1:          # Adam's first step is -lr sign(g): all but the elements whose gradient is ~0 move alike
        d = (c("ln_after") - g["ln_after"]).abs()
        assert (d > 0.1 * meta["lr"]).float().mean() < (0.01 if ref_err < 1e-4 else 0.05)
    st, gs = eng.bn_stats().cpu(), g["bn_stats_after"]
    assert st.numel() == gs.numel()
    torch.testing.assert_close(st, gs, atol=2e-4, rtol=2e-3)
    if meta["prior_strength"] >= 0:                       # `_modified_bn_forward` never writes the running statistics
        torch.testing.assert_close(st, eng.bn_stats(pristine=True).cpu(), atol=0, rtol=0)
    # final logits depend on the gradient (one AdamW step of -lr sign(g) per element): against the float32 fixture at the width of its own
    # noise band, and against the float64 value no further than 1e-3 / three times the reference's distance (5.4e-4 at RN50 in train mode;
    # measured 8.6e-4 in split-f16 mode)
    torch.testing.assert_close(c("final_logits"), g["final_logits"], atol=5e-3 if ref_err > 1e-3 else 1e-3, rtol=0)
    f64 = torch.from_numpy(z64["final_logits"])
    assert (c("final_logits").double() - f64).abs().max() < max(1e-3, 3 * (g["final_logits"].double() - f64).abs().max().item())
    # every sample starts from the checkpoint's parameters AND running statistics; the frozen-student prompt path is untouched
    o2 = eng.tta_sample_ln(views, _cfg_from_meta(meta))
    torch.testing.assert_close(o2["final_logits"], o["final_logits"], atol=2e-4, rtol=0)
    torch.testing.assert_close(eng.ln_params().cpu(), eng.ln_params(pristine=True).cpu(), atol=0, rtol=0)
    eng.close()


@pytest.mark.parametrize("prior", [-1, 0, 16])
@pytest.mark.parametrize("prec", [0, 2])
def test_encode_image_bn_matches_oracle(L, dev, prior, prec):
    """rlcf_engine_encode_image_bn: the ResNet student's image features with its BatchNorms on batch statistics (train mode, or the prior
    blend of `_modified_bn_forward`) and the LIVE tunable parameters, against the oracle's train-form encode (pinned to the reference by the
    bn_* fixtures); the running statistics follow torch's momentum update in train mode and stay put under a prior."""
    eng, ssd, rsd, tokens, _ = make_engine(("tiny-rn", "tiny-r"), 8, 16, L.TEXT_SHARED, prec=prec)
    eng.set_bn_prior_strength(prior)
    keys = RR.visual_bn_keys(ssd)
    new = {k: ssd[k] * (1.0 + 0.05 * synth.normal(5, "bn." + k, tuple(ssd[k].shape))) + 0.02 for k in keys}      # perturbed gamma / beta
    eng.set_ln_params(torch.cat([new[k].reshape(-1) for k in keys]))
    views = synth.make_views(1001, 8, synth.GEOMETRIES["tiny-rn"].image_resolution)
    mode = CR.BNMode("prior", prior / (prior + 1.0)) if prior >= 0 else CR.BNMode("train")
    prev = CR.set_bn_mode(mode)
    try:
        sd = dict(ssd); sd.update(new)
        ref = CR.l2_normalize(CR.encode_image(sd, views))
    finally:
        CR.set_bn_mode(prev)
    got = eng.encode_image_bn(views.to(dev)).cpu()
    torch.testing.assert_close(got, ref, atol=2e-5, rtol=1e-4)
    st = [mode.stats.get(b, (ssd[b + ".running_mean"], ssd[b + ".running_var"])) for b in RR.visual_bn_stat_keys(ssd)]
    torch.testing.assert_close(eng.bn_stats().cpu(), torch.cat([torch.cat([a.reshape(-1), b.reshape(-1)]) for a, b in st]), atol=1e-5, rtol=1e-4)
    eng.close()


def test_bn_tuning_with_resnet_reward_matches_oracle(L, dev):
    """A ModifiedResNet student whose BatchNorms are tuned, scored by a ModifiedResNet REWARD model: the reward
    model's inference pass runs between the student's train-form forward and its backward and shares the tower scratch with it (the
    student's saved attention-pool tensors must survive it).  Against the oracle (pinned to the reference by the bn_* fixtures)."""
    from rlcf_amd.engine import TTAConfig
    N, n_cls = 8, 16
    cfg = TTAConfig(selection_p=0.5, lr=1e-3, tta_steps=2)
    eng, ssd, rsd, tokens, _ = make_engine(("tiny-rn", "tiny-rn"), N, n_cls, L.TEXT_SHARED, prec=2)
    views = synth.make_views(1003, N, synth.GEOMETRIES["tiny-rn"].image_resolution)
    ref = RR.tta_sample_ln(ssd, rsd, views, tokens, RR.TTAHyper(selection_p=0.5, tta_steps=2, sample_k=cfg.sample_k, lr=1e-3,
                                                                 weight_decay=cfg.weight_decay))
    o = eng.tta_sample_ln(views.to(dev), cfg)
    assert o["selected_idx"].cpu().tolist() == ref["selected_idx"].tolist()
    assert o["topk_idx"].cpu().reshape(-1).tolist() == ref["topk_idx"].reshape(-1).tolist()
    gr, og = ref["ln_grad"], o["ln_grad"].cpu()
    assert gr.norm() > 0 and (og - gr).norm() / gr.norm() < 2e-3
    torch.testing.assert_close(o["final_logits"].cpu(), ref["final_logits"], atol=5e-3, rtol=0)
    torch.testing.assert_close(eng.bn_stats().cpu(), ref["bn_stats_after"], atol=2e-4, rtol=2e-3)
    eng.close()


def test_bn_tuning_batch_and_refusals(L, dev):
    """rlcf_tta_batch_ln with a ResNet student runs the samples one by one (the batch statistics couple one sample's views);
    every-parameter tuning of a ResNet student is built since round 4 (tests/test_gpu_round4.py) — its layout call answers."""
    from rlcf_amd.engine import TTAConfig
    N, n_cls = 8, 16
    cfg = TTAConfig(selection_p=0.5, lr=1e-3, tta_steps=1)
    eng, *_ = make_engine(("tiny-rn", "tiny-r"), N, n_cls, L.TEXT_SHARED, prec=2)
    vs = torch.stack([synth.make_views(1000 + i, N, synth.GEOMETRIES["tiny-rn"].image_resolution) for i in range(3)]).to(dev)
    ref = [eng.tta_sample_ln(vs[i], cfg) for i in range(3)]
    top5, fl = eng.tta_batch_ln(vs, cfg, want_logits=True)
    for i in range(3):
        assert top5[i].tolist() == ref[i]["top5"].tolist()
        torch.testing.assert_close(fl[i], ref[i]["final_logits"][0], atol=1e-5, rtol=0)
    g, meta = load_golden("bn_tiny_train")
    torch.testing.assert_close(fl[0].cpu(), g["final_logits"][0], atol=5e-3, rtol=0)
    lay = eng.visual_layout()                      # (round 4: the flat vector of a ModifiedResNet student — convolutions, downsample.1, attention pool)
    assert lay[0][0] == "visual.conv1.weight" and lay[-1][0] == "visual.attnpool.c_proj.bias"
    eng.close()
    # the single-pass f16 performance mode has no tuning paths: refused, not run in reduced precision
    e16, *_ = make_engine(("tiny-rn", "tiny-r"), N, n_cls, L.TEXT_SHARED, prec=L.PREC_F16)
    with pytest.raises(L.RlcfError, match="RLCF_PREC_F16"):
        e16.tta_sample_ln(vs[0], cfg)
    e16.close()


@pytest.mark.parametrize("steps", [1, 3])
@pytest.mark.parametrize("prec", [0, 2])
@pytest.mark.parametrize("geo,reward,n_cls,p", [("tiny", "tiny-r", 16, 0.5), ("small", "small", 40, 0.25)])
def test_ln_tuning_sample_batch_equals_per_sample(L, dev, geo, reward, n_cls, p, prec, steps):
    """rlcf_tta_batch_ln: B samples per tower pass (grouped LayerNorm-gradient reductions, per-sample AdamW and clean-view
    inference) give every sample the result it gets alone; sample 0 of the tiny case is the reference's own ln_tiny_s1 run."""
    from rlcf_amd.engine import TTAConfig
    N, B = 8, 3
    R = synth.GEOMETRIES[geo].image_resolution
    cfg = TTAConfig(selection_p=p, lr=1e-3, tta_steps=steps)
    vs = torch.stack([synth.make_views(1000 + i, N, R) for i in range(B + 2)]).to(dev)     # 5 samples: 3 fused + 2 fused
    one, *_ = make_engine((geo, reward), N, n_cls, L.TEXT_SHARED, prec=prec)
    ref = [one.tta_sample_ln(vs[i], cfg) for i in range(B + 2)]
    one.close()
    big, *_ = make_engine((geo, reward), N * B, n_cls, L.TEXT_SHARED, prec=prec)
    top5, fl = big.tta_batch_ln(vs, cfg, want_logits=True)
    for i in range(B + 2):
        if steps == 1:
            assert top5[i].tolist() == ref[i]["top5"].tolist()
        else:
            assert top5[i][0].item() == ref[i]["top5"][0].item()
        torch.testing.assert_close(fl[i], ref[i]["final_logits"][0], atol=5e-4 if steps == 1 else 1e-3, rtol=0)
    if geo == "tiny":
        g, meta = load_golden("ln_tiny_s1" if steps == 1 else "ln_tiny_s3")
        assert (meta["n_views"], meta["selection_p"], meta["lr"], meta["tta_steps"]) == (N, p, 1e-3, steps)
        torch.testing.assert_close(fl[0].cpu(), g["final_logits"][0], atol=1e-3, rtol=0)
    # the engine is left in the reset state
    o = big.tta_sample_ln(vs[1], cfg)
    torch.testing.assert_close(o["final_logits"][0], ref[1]["final_logits"][0], atol=5e-4, rtol=0)
    big.close()


def test_cls_tta_harness_surface(L, dev):
    """TPT/tune_cls_rl.py call sequence (:67-87, :206-227) with rlcf_amd.custom_clip.CLIPCLS_TTA."""
    import copy
    import types
    from rlcf_amd import clip_reward, clip_store, custom_clip, runtime, tpt_cls_rl
    g, meta = load_golden("ln_tiny_s1")
    runtime.reset_session()
    sg, rg = synth.GEOMETRIES[meta["student"]], synth.GEOMETRIES[meta["reward"]]
    clip_store.register_checkpoint("tiny", sg, synth.make_state_dict(sg, meta["student_seed"]))
    clip_store.register_checkpoint("tiny-r", rg, synth.make_state_dict(rg, meta["reward_seed"]))
    bank = clip_store.SyntheticBank(sg, meta["n_cls"], meta["n_ctx"], meta["bank_seed"])
    clip_store.set_tokenizer(bank.tokenize)
    args = types.SimpleNamespace(tta_steps=1, selection_p=meta["selection_p"], gpu=0, tpt=True, print_freq=1000, min_entropy_reg=0,
                                 min_entropy_w=0.2, reward_arch="tiny-r", multiple_reward_models=0, sample_k=meta["sample_k"],
                                 reward_amplify=False, reward_process=True, process_batch=False)
    model = custom_clip.CLIPCLS_TTA(dev, bank.classnames, arch="tiny", prompt_prefix="a_photo_of_a", only_visual=True, only_norm=True)
    reward_model = clip_reward.get_reward_model(dev, args)
    reward_model.set_class_features(tokenized_classes=model.tokenized_prompts)
    optimizer = torch.optim.AdamW(model.parameters(), meta["lr"], weight_decay=meta["weight_decay"])
    optim_state = copy.deepcopy(optimizer.state_dict())
    views = synth.make_views(meta["view_seed"], meta["n_views"], 32).to(dev)
    model.reset()
    optimizer.load_state_dict(optim_state)
    model.train()
    tpt_cls_rl.test_time_tuning(model, views, optimizer, None, args, reward_model=reward_model)
    model.eval()
    out = model(views[:1])
    torch.testing.assert_close(out.cpu(), g["final_logits"], atol=1e-3, rtol=0)
    d = (model.ln.detach().cpu() - g["ln_after"]).abs()
    assert (d > 0.1 * meta["lr"]).float().mean() < 0.01
    # only_visual=False: in the reference the flag changes nothing that is computed (parameters() ignores it, the class features are
    # cached under no_grad: tests/golden/make_golden.py --only onlyvisual checks the reference's own runs bit for bit against the
    # only_visual=True fixtures) — the mirror serves it on the same path and reproduces the same fixture
    model2 = custom_clip.CLIPCLS_TTA(dev, bank.classnames, arch="tiny", prompt_prefix="a_photo_of_a", only_visual=False, only_norm=True)
    optimizer2 = torch.optim.AdamW(model2.parameters(), meta["lr"], weight_decay=meta["weight_decay"])
    model2.reset()
    model2.train()
    tpt_cls_rl.test_time_tuning(model2, views, optimizer2, None, args, reward_model=reward_model)
    model2.eval()
    torch.testing.assert_close(model2(views[:1]).cpu(), g["final_logits"], atol=1e-3, rtol=0)
    runtime.reset_session()


@pytest.mark.parametrize("name", ["bn_tiny_train", "bn_tiny_prior16_s3"])
def test_cls_tta_harness_resnet_student(L, dev, name):
    """TPT/tune_cls_rl.py call sequence with rlcf_amd.custom_clip.CLIPCLS_TTA(arch = a ModifiedResNet, only_norm=True) and
    --prior_strength: model.train() -> test_time_tuning -> model.eval() -> model(image), against the reference's own run of it."""
    import copy
    import types
    from rlcf_amd import clip_reward, clip_store, custom_clip, runtime, tpt_cls_rl
    g, meta = load_golden(name)
    runtime.reset_session()
    sg, rg = synth.GEOMETRIES[meta["student"]], synth.GEOMETRIES[meta["reward"]]
    clip_store.register_checkpoint("tiny-rn", sg, synth.make_state_dict(sg, meta["student_seed"]))
    clip_store.register_checkpoint("tiny-r", rg, synth.make_state_dict(rg, meta["reward_seed"]))
    bank = clip_store.SyntheticBank(sg, meta["n_cls"], meta["n_ctx"], meta["bank_seed"])
    clip_store.set_tokenizer(bank.tokenize)
    args = types.SimpleNamespace(tta_steps=meta["tta_steps"], selection_p=meta["selection_p"], gpu=0, tpt=True, print_freq=1000,
                                 min_entropy_reg=0, min_entropy_w=0.2, reward_arch="tiny-r", multiple_reward_models=0,
                                 sample_k=meta["sample_k"], reward_amplify=False, reward_process=True, process_batch=False,
                                 prior_strength=meta["prior_strength"])
    model = custom_clip.CLIPCLS_TTA(dev, bank.classnames, arch="tiny-rn", prompt_prefix="a_photo_of_a", only_visual=True, only_norm=True)
    assert model.resnet and model.ln.numel() == g["ln_after"].numel()
    model.set_prior_strength(args.prior_strength)                      # tune_cls_rl.py:73-76
    reward_model = clip_reward.get_reward_model(dev, args)
    reward_model.set_class_features(tokenized_classes=model.tokenized_prompts)
    optimizer = torch.optim.AdamW(model.parameters(), meta["lr"], weight_decay=meta["weight_decay"])
    optim_state = copy.deepcopy(optimizer.state_dict())
    views = synth.make_views(meta["view_seed"], meta["n_views"], sg.image_resolution).to(dev)
    for _ in range(2):                                                 # the second pass starts from the same reset state
        model.reset()
        optimizer.load_state_dict(optim_state)
        model.train()
        tpt_cls_rl.test_time_tuning(model, views, optimizer, None, args, reward_model=reward_model)
        model.eval()
        out = model(views[:1])
        torch.testing.assert_close(out.cpu(), g["final_logits"], atol=2e-3, rtol=0)
        assert out.topk(5).indices[0].tolist() == g["top5"].tolist()
    # (only_norm=False on a ModifiedResNet — the parser defaults — is built since round 4: tests/test_gpu_round4.py)
    runtime.reset_session()


def test_cls_tta_full_visual_harness_with_momentum(L, dev):
    """TPT/tune_cls_rl.py:206-240 over three consecutive samples with the reference's DEFAULT CLIPCLS_TTA(only_norm=False) (what
    scripts/rlcf-tune.sh builds) and momentum_update=True, update_freq=2: clean-view logits, the adapted weights and the moving
    reset state (per-tensor norms of their distance from the checkpoint) against the reference's run."""
    import copy
    import types
    from rlcf_amd import clip_reward, clip_store, custom_clip, runtime, tpt_cls_rl
    g, meta = load_golden("vis_tiny_momentum")
    runtime.reset_session()
    sg, rg = synth.GEOMETRIES[meta["student"]], synth.GEOMETRIES[meta["reward"]]
    ssd = synth.make_state_dict(sg, meta["student_seed"])
    clip_store.register_checkpoint("tiny", sg, ssd)
    clip_store.register_checkpoint("tiny-r", rg, synth.make_state_dict(rg, meta["reward_seed"]))
    bank = clip_store.SyntheticBank(sg, meta["n_cls"], meta["n_ctx"], meta["bank_seed"])
    clip_store.set_tokenizer(bank.tokenize)
    args = types.SimpleNamespace(tta_steps=meta["tta_steps"], selection_p=meta["selection_p"], gpu=0, tpt=True, print_freq=1000,
                                 min_entropy_reg=0, min_entropy_w=0.2, reward_arch="tiny-r", multiple_reward_models=0,
                                 sample_k=meta["sample_k"], reward_amplify=False, reward_process=True, process_batch=False)
    model = custom_clip.CLIPCLS_TTA(dev, bank.classnames, arch="tiny", prompt_prefix="a_photo_of_a", momentum_update=True,
                                    update_freq=meta["update_freq"], update_w=meta["update_w"], momentum=meta["momentum"])
    assert not model.only_norm and len(model.parameters()) == 2
    reward_model = clip_reward.get_reward_model(dev, args)
    reward_model.set_class_features(tokenized_classes=model.tokenized_prompts)
    optimizer = torch.optim.AdamW(model.parameters(), meta["lr"], weight_decay=meta["weight_decay"])
    optim_state = copy.deepcopy(optimizer.state_dict())
    keys = RR.visual_param_keys(ssd)
    for i in range(meta["n_samples"]):
        views = synth.make_views(1000 + i, meta["n_views"], 32).to(dev)
        model.reset()
        optimizer.load_state_dict(optim_state)
        model.train()
        tpt_cls_rl.test_time_tuning(model, views, optimizer, None, args, reward_model=reward_model)
        model.eval()
        torch.testing.assert_close(model(views[:1]).cpu(), g[f"final_logits_{i}"], atol=1e-3, rtol=0)
        eng = runtime.SESSION.engine()
        after = eng.merge_visual(model.ln.data, model.vis.data)
        torch.testing.assert_close(_tensor_norms(ssd, keys, after, ssd), g[f"vis_delta_l2_{i}"], rtol=0.01, atol=1e-7)
        model.momentum_update_model()
        reset_state = eng.merge_visual(eng.ln_params(pristine=True), eng.visual_params(1))
        torch.testing.assert_close(_tensor_norms(ssd, keys, reset_state, ssd), g[f"vis_reset_delta_l2_{i}"], rtol=0.01, atol=1e-7)
    torch.testing.assert_close(eng.visual_params(0), eng.visual_params(1), atol=0, rtol=0)        # live copy follows the reset state
    runtime.reset_session()


def test_cls_tta_momentum_update_matches_reference(L, dev):
    """TPT/tune_cls_rl.py:206-240 over three consecutive samples with CLIPCLS_TTA(momentum_update=True, update_freq=2,
    update_w=0.5, momentum=0.9): tuned LayerNorms, the moving reset state and the clean-view logits against the reference's run."""
    import copy
    import types
    from rlcf_amd import clip_reward, clip_store, custom_clip, runtime, tpt_cls_rl
    g, meta = load_golden("ln_tiny_momentum")
    runtime.reset_session()
    sg, rg = synth.GEOMETRIES[meta["student"]], synth.GEOMETRIES[meta["reward"]]
    clip_store.register_checkpoint("tiny", sg, synth.make_state_dict(sg, meta["student_seed"]))
    clip_store.register_checkpoint("tiny-r", rg, synth.make_state_dict(rg, meta["reward_seed"]))
    bank = clip_store.SyntheticBank(sg, meta["n_cls"], meta["n_ctx"], meta["bank_seed"])
    clip_store.set_tokenizer(bank.tokenize)
    args = types.SimpleNamespace(tta_steps=meta["tta_steps"], selection_p=meta["selection_p"], gpu=0, tpt=True, print_freq=1000,
                                 min_entropy_reg=0, min_entropy_w=0.2, reward_arch="tiny-r", multiple_reward_models=0,
                                 sample_k=meta["sample_k"], reward_amplify=False, reward_process=True, process_batch=False)
    model = custom_clip.CLIPCLS_TTA(dev, bank.classnames, arch="tiny", prompt_prefix="a_photo_of_a", only_visual=True, only_norm=True,
                                    momentum_update=True, update_freq=meta["update_freq"], update_w=meta["update_w"],
                                    momentum=meta["momentum"])
    reward_model = clip_reward.get_reward_model(dev, args)
    reward_model.set_class_features(tokenized_classes=model.tokenized_prompts)
    optimizer = torch.optim.AdamW(model.parameters(), meta["lr"], weight_decay=meta["weight_decay"])
    optim_state = copy.deepcopy(optimizer.state_dict())
    for i in range(meta["n_samples"]):
        views = synth.make_views(1000 + i, meta["n_views"], 32).to(dev)
        model.reset()
        optimizer.load_state_dict(optim_state)
        model.train()
        tpt_cls_rl.test_time_tuning(model, views, optimizer, None, args, reward_model=reward_model)
        model.eval()
        torch.testing.assert_close(model(views[:1]).cpu(), g[f"final_logits_{i}"], atol=1e-3, rtol=0)
        d = (model.ln.detach().cpu() - g[f"ln_after_{i}"]).abs()
        assert (d > 0.1 * meta["lr"]).float().mean() < 0.01
        model.momentum_update_model()
        reset_state = runtime.SESSION.engine().ln_params(pristine=True).cpu()
        # Adam sign flips on ~0 gradients can move single elements by 2*lr; the EMA damps them by (1-momentum)*update_w
        torch.testing.assert_close(reset_state, g[f"ln_reset_{i}"], atol=2.5 * meta["lr"] * (1 - meta["momentum"]) * meta["update_w"], rtol=1e-6)
    assert (reset_state - runtime.SESSION.engine().ln_params(pristine=False).cpu()).abs().max() == 0      # live copy follows the reset state
    runtime.reset_session()


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("prec", [0, 2])
def test_ragged_odd_sizes_vs_oracle(L, dev, mode, prec):
    """Sizes that are multiples of nothing (C=37 classes, N=10 views -> n_sel=3, K=2): HIP path vs the CPU oracle,
    prompt tuning (sparse and dense backward) and LayerNorm tuning."""
    from rlcf_amd.engine import TTAConfig
    n_cls, N = 37, 10
    eng, ssd, rsd, tokens, ctx0 = make_engine(("tiny", "tiny-r"), N, n_cls, mode, prec=prec)
    views = synth.make_views(4250, N, 32)          # a seed with non-zero CLIP rewards (non-trivial gradient)
    hp = RR.TTAHyper(selection_p=0.3, sample_k=2, tta_steps=2)
    ref = RR.tta_sample(ssd, rsd, views, tokens, ctx0, hp)
    for sparse in (True, False):
        o = eng.tta_sample(views.to(dev), TTAConfig(selection_p=0.3, sample_k=2, tta_steps=2, sparse_backward=sparse))
        assert o["selected_idx"].cpu().tolist() == ref["selected_idx"].tolist()
        assert o["topk_idx"].cpu().tolist() == ref["topk_idx"].tolist()
        assert o["top5"].cpu().tolist() == ref["top5"].tolist()
        torch.testing.assert_close(o["logits"].cpu(), ref["logits"], atol=1e-3, rtol=0)
        torch.testing.assert_close(o["final_logits"].cpu(), ref["final_logits"], atol=1e-3, rtol=0)
        gr, og = ref["ctx_grad"], o["ctx_grad"].cpu()
        assert gr.norm() > 0 and (og - gr).norm() / gr.norm() < 1e-3
    ln = RR.tta_sample_ln(ssd, rsd, views, tokens, RR.TTAHyper(selection_p=0.3, sample_k=2, lr=1e-3))
    o = eng.tta_sample_ln(views.to(dev), TTAConfig(selection_p=0.3, sample_k=2, lr=1e-3))
    assert o["selected_idx"].cpu().tolist() == ln["selected_idx"].tolist()
    assert o["top5"].cpu().tolist() == ln["top5"].tolist()
    torch.testing.assert_close(o["final_logits"].cpu(), ln["final_logits"], atol=1e-3, rtol=0)
    assert (o["ln_grad"].cpu() - ln["ln_grad"]).norm() / ln["ln_grad"].norm() < 2e-3
    eng.close()


def test_full_size_properties(L, dev):
    """BASELINE configs[1] sizes (ViT-B/16 + ViT-B/16, N=64, C=1000), size-independent properties: (i) idempotence — the
    per-sample reset makes a repeated sample bit-for-bit reproducible up to the float atomics of the backward; (ii) the fused
    8-images-per-pass path gives every image the result it gets alone; (iii) the three text layouts (reference graph,
    EOT-packed, shared prefix) agree."""
    from rlcf_amd.engine import Engine, TTAConfig
    geo = synth.GEOMETRIES["ViT-B/16"]
    ssd = synth.make_state_dict(geo, 11, device=dev)
    rsd = synth.make_state_dict(geo, 23, device=dev)
    tokens = synth.make_token_bank(geo, 1000, seed=7, n_ctx=4)
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(geo, 4), device=dev)].clone()
    cfg = TTAConfig(selection_p=0.1)
    vs = torch.stack([synth.make_views(s, 64, 224, device=dev) for s in (1113, 1101, 1000)])
    results = {}
    for mode in (L.TEXT_SHARED, L.TEXT_PACKED, L.TEXT_DENSE):
        eng = Engine(geo, geo, 64 * 3, 1000, L.PREC_F16X3)
        eng.load_state_dict(L.STUDENT, ssd); eng.load_state_dict(L.REWARD, rsd); eng.finalize()
        eng.set_class_bank(tokens, 4, ctx0, mode)
        a = eng.tta_sample(vs[0], cfg, want_intermediates=False)
        b = eng.tta_sample(vs[0], cfg, want_intermediates=False)
        torch.testing.assert_close(a["final_logits"], b["final_logits"], atol=2e-4, rtol=0)          # (i)
        assert a["top5"].tolist() == b["top5"].tolist()
        top5, fl = eng.tta_batch(vs, cfg, want_logits=True)                                           # (ii) fused pass of 3 images
        torch.testing.assert_close(fl[0], a["final_logits"][0], atol=2e-4, rtol=0)
        assert top5[0].tolist() == a["top5"].tolist()
        results[mode] = fl.cpu()
        eng.close()
    torch.testing.assert_close(results[L.TEXT_PACKED], results[L.TEXT_SHARED], atol=1e-3, rtol=0)    # (iii)
    torch.testing.assert_close(results[L.TEXT_DENSE], results[L.TEXT_SHARED], atol=1e-3, rtol=0)


def test_config5_rn50x64_student_vit_l14_reward(L, dev):
    """BASELINE configs[4]: RN50x64 student (448^2 views, frozen ModifiedResNet image encoder, prompt tuning) + ViT-L/14 reward (the
    selected views are resampled 448 -> 224 with the bicubic kernel), N=32.  Full geometry, 200 classes; size-independent properties:
    per-sample reset reproducibility, f32 and split-f16 precision agree, rewards of a view sum to zero (baseline subtraction), and
    the student features of view 0 equal the reference-generated RN50x64 fixture."""
    from rlcf_amd.engine import Engine, TTAConfig
    sg, rg = synth.GEOMETRIES["RN50x64"], synth.GEOMETRIES["ViT-L/14"]
    ssd = synth.make_state_dict(sg, 11, device=dev)
    rsd = synth.make_state_dict(rg, 23, device=dev)
    tokens = synth.make_token_bank(sg, 200, seed=7, n_ctx=4)
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(sg, 4), device=dev)].clone()
    views = synth.make_views(1000, 32, 448, device=dev)
    cfg = TTAConfig(selection_p=0.1)
    out = {}
    for prec in (L.PREC_F32, L.PREC_F16X3):
        eng = Engine(sg, rg, 32, 200, prec)
        eng.load_state_dict(L.STUDENT, ssd); eng.load_state_dict(L.REWARD, rsd); eng.finalize()
        eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
        if prec == L.PREC_F32:
            g, _ = load_golden("modules_rn")
            f = eng.encode_image(L.STUDENT, views[:1]).cpu()
            torch.testing.assert_close(f, CR.l2_normalize(g["rn50x64_image"]), atol=2e-5, rtol=1e-4)
        a = eng.tta_sample(views, cfg)
        b = eng.tta_sample(views, cfg, want_intermediates=False)
        torch.cuda.synchronize()
        assert a["top5"].tolist() == b["top5"].tolist()
        torch.testing.assert_close(a["final_logits"], b["final_logits"], atol=2e-4, rtol=0)
        assert a["selected_idx"].numel() == 3 and a["topk_idx"].shape == (3, 3)
        torch.testing.assert_close(a["rewards"].view(3, 3).sum(1).cpu(), torch.zeros(3), atol=1e-5, rtol=0)
        out[prec] = {k: v.cpu() for k, v in a.items() if torch.is_tensor(v)}
        eng.close()
    x, y = out[L.PREC_F32], out[L.PREC_F16X3]
    assert x["selected_idx"].tolist() == y["selected_idx"].tolist() and x["topk_idx"].tolist() == y["topk_idx"].tolist()
    assert x["top5"].tolist() == y["top5"].tolist()
    torch.testing.assert_close(x["logits"], y["logits"], atol=1e-3, rtol=0)
    torch.testing.assert_close(x["final_logits"], y["final_logits"], atol=1e-3, rtol=0)


# ------------------------------------------------------------------------------ size-independent properties at BASELINE configs[1] size
def test_full_size_properties_view_permutation_and_reward_baseline(L, dev):
    """ViT-B/16 + ViT-B/16, N = 64 views, 1000 classes (BASELINE configs[1]), where the oracle is too slow to run in a test:
    (1) views are independent units until the selection: permuting views 1..63 permutes the selected indices and leaves the adapted
        prompt, the final logits and the top-5 unchanged;
    (2) with reward_process and per-view baselines the K rewards of every selected view sum to zero (tpt_cls_rl.py:63-67,
        clip_reward.py:152-165) and the loss gradient of a view's logits sums to zero over the classes;
    (3) the one-image call and the sample-batched call give the same predictions."""
    from rlcf_amd.engine import TTAConfig
    from rlcf_amd.engine import Engine
    N, C = 64, 1000
    geo = synth.GEOMETRIES["ViT-B/16"]
    ssd, rsd = synth.make_state_dict(geo, 11, device=dev), synth.make_state_dict(geo, 23, device=dev)
    eng = Engine(geo, geo, 2 * N, C, 2)
    eng.load_state_dict(L.STUDENT, ssd)
    eng.load_state_dict(L.REWARD, rsd)
    eng.finalize()
    tokens = synth.make_token_bank(geo, C, seed=7, n_ctx=4)
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(geo, 4), device=dev)].clone()
    eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
    cfg = TTAConfig(selection_p=0.1, sample_k=3, lr=7e-3, weight_decay=5e-4)
    views = synth.make_views(1113, N, 224, device=dev)          # the seed of the tta_b16_n64 fixture: non-zero CLIP rewards
    o = eng.tta_sample(views, cfg)
    g = torch.Generator().manual_seed(5)
    perm = torch.cat([torch.zeros(1, dtype=torch.long), 1 + torch.randperm(N - 1, generator=g)]).to(dev)
    o2 = eng.tta_sample(views[perm], cfg)
    sel, sel2 = o["selected_idx"].long(), o2["selected_idx"].long()
    assert sorted(perm[sel2].tolist()) == sorted(sel.tolist())                     # same views selected ...
    assert perm[sel2].tolist() == sel.tolist()                                      # ... in the same (ascending-entropy) order
    torch.testing.assert_close(o2["logits"], o["logits"][perm], atol=2e-5, rtol=0)
    torch.testing.assert_close(o2["ctx_after"], o["ctx_after"], atol=1e-6, rtol=0)
    torch.testing.assert_close(o2["final_logits"], o["final_logits"], atol=2e-4, rtol=0)
    assert o2["top5"].tolist() == o["top5"].tolist()
    n_sel = sel.numel()
    r = o["rewards"].view(n_sel, 3)
    assert r.abs().max() > 0
    torch.testing.assert_close(r.sum(1), torch.zeros(n_sel, device=dev), atol=2e-5, rtol=0)
    torch.testing.assert_close(o["dlogits"].sum(1), torch.zeros(n_sel, device=dev), atol=1e-6, rtol=0)
    both = torch.stack([views, views[perm]])
    top5, fl = eng.tta_batch(both, cfg, want_logits=True)
    assert top5[0].tolist() == o["top5"].tolist() and top5[1].tolist() == o["top5"].tolist()
    torch.testing.assert_close(fl[0], o["final_logits"][0], atol=2e-4, rtol=0)
    eng.close()


def test_full_size_properties_layernorm_tuning_l14(L, dev):
    """BASELINE configs[2] size (ViT-L/14 + ViT-L/14, N = 64, 1000 classes, LayerNorm tuning): the sample-batched call equals the
    one-image call, a permutation of views 1..63 leaves the result unchanged, and lr = 0 reproduces plain inference."""
    from rlcf_amd.engine import TTAConfig
    from rlcf_amd.engine import Engine
    N, C = 64, 1000
    geo = synth.GEOMETRIES["ViT-L/14"]
    ssd, rsd = synth.make_state_dict(geo, 11, device=dev), synth.make_state_dict(geo, 23, device=dev)      # (generated on the device: 2 x 428 M parameters)
    eng = Engine(geo, geo, 2 * N, C, 2)
    eng.load_state_dict(L.STUDENT, ssd)
    eng.load_state_dict(L.REWARD, rsd)
    eng.finalize()
    tokens = synth.make_token_bank(geo, C, seed=7, n_ctx=4)
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(geo, 4), device=dev)].clone()
    eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
    cfg = TTAConfig(selection_p=0.1, sample_k=3, lr=1e-4, weight_decay=5e-4)
    views = synth.make_views(1113, N, 224, device=dev)
    o = eng.tta_sample_ln(views, cfg)
    g = torch.Generator().manual_seed(9)
    perm = torch.cat([torch.zeros(1, dtype=torch.long), 1 + torch.randperm(N - 1, generator=g)]).to(dev)
    o2 = eng.tta_sample_ln(views[perm], cfg)
    assert perm[o2["selected_idx"].long()].tolist() == o["selected_idx"].long().tolist()
    torch.testing.assert_close(o2["final_logits"], o["final_logits"], atol=1e-3, rtol=0)
    top5, fl = eng.tta_batch_ln(torch.stack([views, views[perm]]), cfg, want_logits=True)
    assert top5[0].tolist() == o["top5"].tolist() and top5[1].tolist() == o["top5"].tolist()
    torch.testing.assert_close(fl[0], o["final_logits"][0], atol=1e-3, rtol=0)
    z = eng.tta_sample_ln(views, TTAConfig(selection_p=0.1, sample_k=3, lr=0.0, weight_decay=0.0))
    plain = eng.logits(eng.encode_image(L.STUDENT, views[:1]), eng.text_features(ctx0.to(dev)))
    torch.testing.assert_close(z["final_logits"], plain, atol=2e-4, rtol=0)
    assert (o["final_logits"] - plain).abs().max() > 1e-4          # ... and the tuned run did move the logits
    eng.close()


def test_full_size_properties_full_encoder_tuning_b16(L, dev):
    """scripts/rlcf-tune.sh size (ViT-B/16 + ViT-B/16, N = 64, 1000 classes, every visual parameter tuned, 3 steps): permuting views
    1..63 leaves the result unchanged, lr = 0 reproduces plain inference, the first-step gradient does not depend on the number of
    steps, and the LayerNorm part of that gradient is the LayerNorm-only path's gradient."""
    from rlcf_amd.engine import Engine, TTAConfig
    N, C = 64, 1000
    geo = synth.GEOMETRIES["ViT-B/16"]
    ssd, rsd = synth.make_state_dict(geo, 11, device=dev), synth.make_state_dict(geo, 23, device=dev)
    eng = Engine(geo, geo, N, C, 2)
    eng.load_state_dict(L.STUDENT, ssd)
    eng.load_state_dict(L.REWARD, rsd)
    eng.finalize()
    tokens = synth.make_token_bank(geo, C, seed=7, n_ctx=4)
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(geo, 4), device=dev)].clone()
    eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
    views = synth.make_views(1113, N, 224, device=dev)
    cfg = lambda steps, lr=1e-5: TTAConfig(selection_p=0.1, sample_k=3, lr=lr, weight_decay=5e-4 if lr else 0.0, tta_steps=steps)
    o3, o1 = eng.tta_sample_visual(views, cfg(3)), eng.tta_sample_visual(views, cfg(1))
    assert o1["vis_grad"].norm() > 0                                   # (float atomics in the attention backward: last-bit run-to-run differences)
    assert (o3["vis_grad"] - o1["vis_grad"]).norm() / o1["vis_grad"].norm() < 1e-5
    assert (o3["ln_grad"] - o1["ln_grad"]).norm() / o1["ln_grad"].norm() < 1e-5
    ln = eng.tta_sample_ln(views, cfg(1))
    assert (ln["ln_grad"] - o1["ln_grad"]).norm() / o1["ln_grad"].norm() < 1e-4
    g = torch.Generator().manual_seed(3)
    perm = torch.cat([torch.zeros(1, dtype=torch.long), 1 + torch.randperm(N - 1, generator=g)]).to(dev)
    p3 = eng.tta_sample_visual(views[perm], cfg(3))
    assert perm[p3["selected_idx"].long()].tolist() == o3["selected_idx"].long().tolist()
    torch.testing.assert_close(p3["final_logits"], o3["final_logits"], atol=1e-3, rtol=0)
    assert p3["top5"].tolist() == o3["top5"].tolist()
    z = eng.tta_sample_visual(views, cfg(3, lr=0.0))
    plain = eng.logits(eng.encode_image(L.STUDENT, views[:1]), eng.text_features(ctx0))
    torch.testing.assert_close(z["final_logits"], plain, atol=2e-4, rtol=0)
    torch.testing.assert_close(z["vis_after"], eng.visual_params(1), atol=0, rtol=0)
    assert (o3["final_logits"] - plain).abs().max() > 1e-4
    eng.close()
