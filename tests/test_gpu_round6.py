"""GPU tests, round-6 additions: samples in flight from ONE host thread (rlcf_lanes_*, rlcf_tta_lanes), the loop's hit counters as a
kernel (rlcf_top5_hits), the deferred-store form of the persistent single-pass f16 GEMM (bit-equality with the epilogue-only form),
the opt-in LayerNorm fold of RLCF_PREC_F16, NaN propagation of the stand-alone avg_entropy, and an RCCL pre-flight with two ranks
that runs wherever two GPUs are visible."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest
import torch

from rlcf_amd import _lib as L
from rlcf_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _st():
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------ samples in flight: one host thread, K lanes
@pytest.mark.parametrize("lanes", [2, 3])
@pytest.mark.parametrize("path", ["prompt", "ln"])
def test_lanes_submit_is_the_one_image_call_bit_for_bit(lanes, path):
    """rlcf_lanes_submit / rlcf_tta_lanes: sample i runs on lane i mod K as exactly rlcf_tta_batch (rlcf_tta_batch_ln) with one image —
    seven samples (ragged last round) over 2 and 3 lanes against the same calls one at a time on ONE engine: top-5 and final logits
    bit-identical.  The lanes are fresh engines: their first-call workspace growth runs on their own non-blocking streams."""
    from test_gpu_parity import make_engine
    from rlcf_amd.engine import Lanes, TTAConfig
    n_views, n_cls, n = 16, 40, 7
    cfg = TTAConfig(selection_p=0.25, lr=5e-3 if path == "prompt" else 1e-4)
    R = synth.GEOMETRIES["small"].image_resolution
    views = torch.stack([synth.make_views(3000 + i, n_views, R, device=DEV) for i in range(n)])
    ref, *_ = make_engine(("small", "small"), n_views, n_cls, L.TEXT_SHARED, prec=L.PREC_F16X3)
    one = ref.tta_batch if path == "prompt" else ref.tta_batch_ln
    exp = [one(views[i: i + 1], cfg, want_logits=True) for i in range(n)]
    exp5, expl = torch.cat([e[0] for e in exp]), torch.cat([e[1] for e in exp])
    engs = [make_engine(("small", "small"), n_views, n_cls, L.TEXT_SHARED, prec=L.PREC_F16X3)[0] for _ in range(lanes)]
    # (a) the lanes object, as the harness mirror drives it: submit sample by sample, one join at the end
    ln = Lanes(engs)
    top5 = torch.full((n, 5), -1, dtype=torch.int32, device=DEV)
    logits = torch.full((n, n_cls), float("nan"), device=DEV)
    took = [ln.submit(views[i], cfg, top5[i], norm_layers=(path == "ln"), final_logits=logits[i]) for i in range(n)]
    assert took == [i % lanes for i in range(n)]
    ln.join()
    torch.cuda.current_stream().synchronize()
    assert torch.equal(top5, exp5) and torch.equal(logits, expl)
    ln.close()
    # (b) the one-shot convenience on a whole batch
    top5b = torch.full((n, 5), -1, dtype=torch.int32, device=DEV)
    logb = torch.full((n, n_cls), float("nan"), device=DEV)
    arr = (C.c_void_p * lanes)(*[e.h for e in engs])
    a = cfg.c_args(n_views)
    L.check(L.lib().rlcf_tta_lanes(arr, lanes, views.data_ptr(), n, n_views, C.byref(a), logb.data_ptr(), top5b.data_ptr(), 1 if path == "ln" else 0, _st()),
            "rlcf_tta_lanes")
    torch.cuda.synchronize()
    assert torch.equal(top5b, exp5) and torch.equal(logb, expl)
    for e in engs + [ref]:
        e.close()


def test_lanes_refuse_an_engine_twice():
    from test_gpu_parity import make_engine
    e, *_ = make_engine(("tiny", "tiny-r"), 8, 16, L.TEXT_SHARED)
    arr = (C.c_void_p * 2)(e.h, e.h)
    assert not L.lib().rlcf_lanes_create(arr, 2)
    assert b"distinct" in L.lib().rlcf_last_error()
    e.close()


def test_top5_hits_accumulates_counts():
    g = torch.Generator().manual_seed(5)
    top5 = torch.stack([torch.randperm(50, generator=g)[:5] for _ in range(300)]).to(torch.int32).to(DEV)
    tgt = torch.randint(0, 50, (300,), generator=g).to(DEV)
    tgt[::3] = top5[::3, 0].long()
    tgt[1::7] = top5[1::7, 3].long()
    out = torch.zeros(2, device=DEV)
    for lo, hi in ((0, 100), (100, 300)):
        L.check(L.lib().rlcf_top5_hits(top5[lo:hi].contiguous().data_ptr(), tgt[lo:hi].contiguous().data_ptr(), hi - lo, out.data_ptr(), _st()))
    h1 = int((top5[:, 0].long() == tgt).sum())
    h5 = int((top5.long() == tgt[:, None]).any(1).sum())
    assert out.cpu().tolist() == [float(h1), float(h5)] and h5 > h1 > 0


def test_in_flight_falls_back_to_the_serial_loop_after_an_applied_ema():
    """Lane engines are built from the checkpoint; once a cross-sample EMA has moved the session engine's reset state the lanes would
    tune from another state than lane 0 — the in_flight gate then takes the serial loop (runtime.Session.reset_state_moved)."""
    from test_gpu_parity import make_engine
    eng, *_ = make_engine(("tiny", "tiny-r"), 8, 16, L.TEXT_SHARED)
    assert eng.reset_moved is False
    p = eng.ln_params()
    eng.momentum_update(p * 1.01, 0.9, 0.5, apply=False)
    assert eng.reset_moved is False
    eng.momentum_update(p * 1.01, 0.9, 0.5, apply=True)
    assert eng.reset_moved is True
    eng.reset_visual_state()
    assert eng.reset_moved is False
    eng.close()


# ------------------------------------------------------------------ persistent f16 GEMM: deferred stores
@pytest.mark.parametrize("M,N,K,epi", [(70000 + 37, 768, 768, 0), (66000, 2304, 768, 0), (66000, 3072, 768, 1), (70000 + 37, 768, 3072, 0), (66816, 1024, 1024, 0)])
def test_f16_gemm_deferred_stores_equal_epilogue_stores(M, N, K, epi, monkeypatch):
    """gemm_nt_f16_pp_kernel<.., DEFER = 1> (half of a tile's output leaves under the next tile's K loop, one store per K tile behind a
    counted wait) and <.., TS = 1> (full-line stores: every 32-row block of a wave's tile through a private LDS slab, 8 rows x 128 B per
    store instruction) write the SAME bits as the epilogue-only form with 32 rows x 32 B per instruction, on matrices with more tiles than workgroups (every workgroup carries
    pending stores across tiles), a ragged last row tile, both epilogues and both K depths; every element is written (NaN pre-fill)."""
    a = torch.randn(M, K, device=DEV).half()
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    b = torch.randn(N, device=DEV) * 0.1
    outs = []
    for d, ts in (("0", "0"), ("1", "0"), ("1", "0"), ("0", "1"), ("0", "1")):          # epilogue-only 32 x 32-B stores; deferred (twice); full-line stores through LDS (twice)
        monkeypatch.setenv("RLCF_F16_PP_DEFER", d)
        monkeypatch.setenv("RLCF_F16_PP_TSTORE", ts)
        c = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
        L.check(L.lib().rlcf_gemm_f16(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), None, N, None, N, c.data_ptr(), N, M, N, K, 1.0, epi, _st()))
        torch.cuda.current_stream().synchronize()
        outs.append(c)
    assert not torch.isnan(outs[1]).any()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    assert torch.equal(outs[0], outs[3]) and torch.equal(outs[3], outs[4]) and not torch.isnan(outs[3]).any()
    rows = torch.randint(0, M, (128,), device=DEV)
    ref = a[rows].double() @ w.double().t() + b.double()
    if epi == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    assert ((outs[1][rows].double() - ref).abs() / (ref.abs() + 1.0)).max().item() < 1e-2


# ------------------------------------------------------------------ RLCF_PREC_F16: the LayerNorm fold is an opt-in
def test_f16_lnfold_is_off_by_default_and_switchable(monkeypatch):
    """Default: f32 residual stream + layernorm_add_fwd (keeps the reference's top-1 on 32 / 32 stream samples); rlcf_engine_set_f16_lnfold(1)
    / RLCF_F16_LNFOLD=1 at create: f16 residual stream with folded LayerNorms.  Both stay within the mode's band of the split-f16
    engine, they differ from each other (the switch does something), and the setter and the environment agree bit for bit."""
    from test_gpu_parity import make_engine
    monkeypatch.delenv("RLCF_F16_LNFOLD", raising=False)
    N, n_cls = 64, 40
    ex, *_ = make_engine(("small", "small"), N, n_cls, L.TEXT_SHARED, prec=L.PREC_F16X3)
    eh, *_ = make_engine(("small", "small"), N, n_cls, L.TEXT_SHARED, prec=L.PREC_F16)
    views = synth.make_views(2001, N, 64).to(DEV)
    fx = ex.encode_image(L.STUDENT, views).clone()
    f_off = eh.encode_image(L.STUDENT, views).clone()
    eh.set_f16_lnfold(True)
    f_on = eh.encode_image(L.STUDENT, views).clone()
    eh.set_f16_lnfold(False)
    assert torch.equal(eh.encode_image(L.STUDENT, views), f_off)
    assert (f_off - fx).abs().max().item() < 5e-3 and (f_on - fx).abs().max().item() < 5e-3
    assert not torch.equal(f_on, f_off)
    monkeypatch.setenv("RLCF_F16_LNFOLD", "1")
    ee, *_ = make_engine(("small", "small"), N, n_cls, L.TEXT_SHARED, prec=L.PREC_F16)
    assert torch.equal(ee.encode_image(L.STUDENT, views), f_on)
    for e in (ex, eh, ee):
        e.close()


# ------------------------------------------------------------------ mirror conveniences: reference semantics at the edges
def test_avg_entropy_propagates_nan_and_keeps_the_dtype():
    from rlcf_amd import tpt_cls_rl
    x = torch.randn(6, 30, device=DEV) * 3
    ref = tpt_cls_rl.avg_entropy(x.clone().requires_grad_(True)).detach()       # (autograd input: the reference's torch expression)
    got = tpt_cls_rl.avg_entropy(x)
    torch.testing.assert_close(got, ref, atol=1e-5, rtol=1e-5)
    xn = x.clone()
    xn[2, 7] = float("nan")
    assert torch.isnan(tpt_cls_rl.avg_entropy(xn)) and torch.isnan(tpt_cls_rl.avg_entropy(xn.clone().requires_grad_(True)))
    xi = x.clone()
    xi[4, 1] = float("inf")
    assert torch.isnan(tpt_cls_rl.avg_entropy(xi))
    assert tpt_cls_rl.avg_entropy(x.half()).dtype == torch.float16


def test_accuracy_with_fewer_classes_than_topk_raises_like_the_reference():
    from rlcf_amd import tpt_cls_rl
    out, tgt = torch.randn(4, 3, device=DEV), torch.tensor([0, 1, 2, 0], device=DEV)
    with pytest.raises(RuntimeError):
        tpt_cls_rl.accuracy(out, tgt, topk=(1, 5))
    assert len(tpt_cls_rl.accuracy(out, tgt, topk=(1,))) == 1


# ------------------------------------------------------------------ RCCL with more than one rank, wherever two GPUs are visible
@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() < 2, reason="needs two visible GPUs (the first multi-GPU box a lease gives runs it)")
def test_bench_two_ranks_rccl_preflight():
    """`python bench.py --gpus 2 --dist-backend nccl`: two ranks, one GPU each, barrier + all_reduce(MAX) + all_gather over RCCL — the
    first time RCCL sees more than one rank on this code is inside the GPU suite of the first multi-GPU box, not in the driver's
    scaling run."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "nccl", "--steps", "8", "--warmup", "4",
                        "--no-cpu-baseline", "--sustain-seconds", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["distributed"]["rccl_ranks"] == 2 and line["value"] > 0
    assert len(line["distributed"]["timed_region_seconds_per_rank"]) == 2


def test_bench_refuses_more_ranks_than_visible_gpus_under_nccl():
    """`--gpus N` with N > visible devices under the RCCL backend is an error with a message, not ranks stacked on one device that die
    inside RCCL (bench.py; the gloo backend of the CPU / one-GPU smoke tests may still share a device)."""
    n = torch.cuda.device_count() + 1
    env = dict(os.environ, WORLD_SIZE=str(n), RANK="0", LOCAL_RANK=str(n - 1), MASTER_ADDR="127.0.0.1", MASTER_PORT="29599")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dist-backend", "nccl", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "visible" in (r.stderr + r.stdout)


# ------------------------------------------------------------------ the parity kernel's line-complete pair stores move bytes, not bits
_LINEST_PROBE = r"""
import hashlib, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from rlcf_amd import _lib as L, synth
from rlcf_amd.engine import Engine
g = synth.GEOMETRIES["ViT-B/16"]; dev = torch.device("cuda:0")
eng = Engine(g, g, 64, 8, L.PREC_F16X3)
sd = synth.make_state_dict(g, 11, device=dev)
eng.load_state_dict(L.STUDENT, sd); eng.load_state_dict(L.REWARD, sd); eng.finalize()
f = eng.encode_image(L.STUDENT, synth.make_views(4242, 64, g.image_resolution, device=dev))
print("FEATURES", hashlib.sha256(f.cpu().numpy().tobytes()).hexdigest(), float(f.abs().sum()))
"""


def test_x3_line_complete_pair_stores_bit_identical():
    """RLCF_X3_LINEST (read once per process): the in_proj / c_fc epilogues of the 256x256 split-f16 kernel write the interleaved hi / lo
    pair rows either as half lines (0) or as whole 128-byte lines after a lane exchange through the park slab (1, default).  One ViT-B/16
    image tower pass over 64 views (12 608 token rows: both products run on that kernel) must give the same feature BITS either way."""
    outs = []
    for v in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", _LINEST_PROBE, ROOT], capture_output=True, text=True, timeout=600, cwd=ROOT,
                           env=dict(os.environ, RLCF_X3_LINEST=v))
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append([x for x in r.stdout.splitlines() if x.startswith("FEATURES")][-1])
    assert outs[0] == outs[1], outs
    assert float(outs[0].split()[2]) > 0.0


# ------------------------------------------------------------------ attention forward: the output block as whole lines through LDS
def _attention_line_store_cases():
    from test_gpu_round3 import PAIR_CASES
    return PAIR_CASES


@pytest.mark.parametrize("prec_name", ["f16x3", "f16"])
@pytest.mark.parametrize("case", range(9))
def test_attention_whole_line_stores_bit_identical(prec_name, case):
    """attention_fwd_pair_kernel with only the operand-layout output requested (what the engine's image towers ask for): the 32-query block of
    a wave leaves either as 16-byte pieces per lane (RLCF_ATTN_LINEST=0, the accumulator layout) or transposed through the wave's LDS slab as
    whole 128-byte lines (default).  Same BITS in the output rows and in the log-sum-exp, rows of no sequence untouched — on every
    sequence-set of the round-3 parity cases (197 / 257 / 577 tokens, ragged, prefixes, one-query blocks), both operand forms."""
    from test_gpu_round3 import _seq_buf
    name, seqs, T, W = _attention_line_store_cases()[case]
    prec = L.PREC_F16X3 if prec_name == "f16x3" else L.PREC_F16
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    qkv = synth.normal(3, "att." + name, (T, 3 * W), 1.5).to(DEV)
    pairs = torch.empty(T, 3 * W, device=DEV)
    L.check(lib.rlcf_split_pairs(qkv.data_ptr(), pairs.data_ptr(), T * 3 * W, prec, st))
    sbuf = _seq_buf(L, seqs, DEV)
    mq = max(s[1] for s in seqs)
    got = {}
    try:
        for v in ("0", "1"):
            os.environ["RLCF_ATTN_LINEST"] = v
            op = torch.full((T, W), 7.25, device=DEV)                        # (rows no sequence owns must keep this)
            lse = torch.full((T, W // 64), -3.0, device=DEV)
            L.check(lib.rlcf_attention_fwd_pairs(pairs.data_ptr(), sbuf.data_ptr(), len(seqs), mq, W, None, op.data_ptr(), lse.data_ptr(), prec, st))
            torch.cuda.synchronize()
            got[v] = (op.cpu(), lse.cpu())
    finally:
        os.environ.pop("RLCF_ATTN_LINEST", None)
    assert torch.equal(got["0"][0].view(torch.int32), got["1"][0].view(torch.int32))
    assert torch.equal(got["0"][1].view(torch.int32), got["1"][1].view(torch.int32))
    rows = sorted({r for (qs, ql, _, _) in seqs for r in range(qs, qs + ql)})
    assert not torch.equal(got["1"][0][rows], torch.full((len(rows), W), 7.25))
