"""GPU parity, round-3 additions: the attention forward on producer-emitted operand pairs (attention_pair.hip: LDS-DMA staging,
ds_read_b64_tr_b16) against the f64 reference and against the round-2 kernel it replaces; the in_proj pair epilogue; the engine's
image tower with the old and the new hand-over; one-rank RCCL runs of bench.py and the eval driver."""
import json
import os
import subprocess
import sys

import pytest
import torch

from rlcf_amd import synth
from test_gpu_parity import _attn_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def _seq_buf(L, seqs, dev):
    sq = (L.Seq * len(seqs))(*[L.Seq(*s) for s in seqs])
    return torch.frombuffer(bytearray(bytes(sq)), dtype=torch.int32).to(dev)


def _unpack_pairs(buf, T, W):
    """interleaved pair matrix [T, W] (per 32 columns: 32 hi halves, 32 lo halves) -> float64 hi + lo"""
    h = buf.view(torch.float16).reshape(T, W // 32, 2, 32).double()
    return (h[:, :, 0] + h[:, :, 1]).reshape(T, W)


# non-causal sequence sets: ViT-B/16 (197 = 6 x 32 + 5), ViT-L/14 (257 = 8 blocks + the odd query -> one-wave tail launch),
# ViT-L/14@336 (577 = three 8-wave blocks), 50 / 17 tokens (4-wave and one-wave workgroups), the class-token-only last block
# (one query, prefix = the other 196 rows), ragged lengths, and a prefix in front of a multi-block sequence
PAIR_CASES = [
    ("vit197", [(i * 197, 197, 0, 0) for i in range(3)], 591, 128),
    ("vit257", [(i * 257, 257, 0, 0) for i in range(2)], 514, 128),
    ("vit577", [(0, 577, 0, 0)], 577, 64),
    ("vit50", [(0, 50, 0, 0), (50, 50, 0, 0), (100, 50, 0, 0)], 150, 128),
    ("tiny17", [(i * 17, 17, 0, 0) for i in range(4)], 68, 64),
    ("cls_only", [(i * 197, 1, i * 197 + 1, 196) for i in range(3)], 591, 128),
    ("ragged", [(0, 33, 0, 0), (33, 1, 0, 0), (34, 64, 0, 0), (98, 65, 0, 0), (163, 129, 0, 0)], 292, 64),
    ("prefix_multi", [(40, 130, 0, 40), (170, 37, 0, 40), (0, 40, 0, 0)], 207, 64),
    ("keys64", [(0, 64, 0, 0), (64, 128, 0, 0), (192, 32, 0, 0)], 224, 64),
]


@pytest.mark.parametrize("name,seqs,T,W", PAIR_CASES)
def test_attention_pairs_split_f16(L, dev, name, seqs, T, W):
    """attention_fwd_pair_kernel (3 f16 MFMAs per product on operands split by the PRODUCER) against the f64 reference: f32-grade,
    the same bar as the round-2 kernel; f32 output, pair output and log-sum-exp."""
    qkv = synth.normal(3, "att." + name, (T, 3 * W), 1.5)
    qd = qkv.to(dev)
    pairs = torch.empty(T, 3 * W, device=dev)                     # a pair row is as long as an f32 row
    L.check(L.lib().rlcf_split_pairs(qd.data_ptr(), pairs.data_ptr(), T * 3 * W, L.PREC_F16X3, st()))
    sbuf = _seq_buf(L, seqs, dev)
    mq = max(s[1] for s in seqs)
    out = torch.zeros(T, W, device=dev)
    op = torch.zeros(T, W, device=dev)
    lse = torch.zeros(T, W // 64, device=dev)
    L.check(L.lib().rlcf_attention_fwd_pairs(pairs.data_ptr(), sbuf.data_ptr(), len(seqs), mq, W, out.data_ptr(), op.data_ptr(),
                                             lse.data_ptr(), L.PREC_F16X3, st()))
    torch.cuda.synchronize()
    # the operands ARE the rounded pairs: reference on hi + lo (22-bit operands) and on the f32 values
    q22 = _unpack_pairs(pairs.cpu(), T, 3 * W)
    ref = _attn_ref(qkv.double(), seqs, W, 0)
    torch.testing.assert_close(out.cpu().double(), ref, atol=5e-6, rtol=1e-5)
    torch.testing.assert_close(out.cpu().double(), _attn_ref(q22, seqs, W, 0), atol=5e-6, rtol=1e-5)
    rows = sorted({r for (qs, ql, _, _) in seqs for r in range(qs, qs + ql)})
    torch.testing.assert_close(_unpack_pairs(op.cpu(), T, W)[rows], out.cpu().double()[rows], atol=2e-6, rtol=1e-6)
    # log-sum-exp of the scaled scores
    H = W // 64
    for (qs, ql, ps, pl) in seqs[:2]:
        kr = list(range(ps, ps + pl)) + list(range(qs, qs + ql))
        q = qkv[qs:qs + ql, :W].double().reshape(ql, H, 64).transpose(0, 1)
        k = qkv[kr, W:2 * W].double().reshape(len(kr), H, 64).transpose(0, 1)
        want = torch.logsumexp((q * 0.125) @ k.transpose(-1, -2), -1).transpose(0, 1)
        torch.testing.assert_close(lse[qs:qs + ql].cpu().double(), want, atol=2e-5, rtol=1e-5)
    # ... and the kernel it replaces, on the same values
    old = torch.zeros(T, W, device=dev)
    L.check(L.lib().rlcf_attention_fwd(qd.data_ptr(), sbuf.data_ptr(), len(seqs), mq, W, 0, old.data_ptr(), None, L.PREC_F16X3, st()))
    torch.testing.assert_close(out.cpu()[rows], old.cpu()[rows], atol=3e-6, rtol=1e-5)


@pytest.mark.parametrize("name,seqs,T,W", PAIR_CASES[:6])
def test_attention_pairs_single_f16(L, dev, name, seqs, T, W):
    """RLCF_PREC_F16 form (plain f16 operands, one MFMA per product): exact against a reference on the SAME f16-rounded operands up to
    the f16 rounding of P (carried times 2^6) — the bar of the round-2 single-pass kernel."""
    qkv = synth.normal(3, "att." + name, (T, 3 * W), 1.5)
    qd = qkv.to(dev)
    q16 = torch.empty(T, 3 * W, dtype=torch.float16, device=dev)
    L.check(L.lib().rlcf_split_pairs(qd.data_ptr(), q16.data_ptr(), T * 3 * W, L.PREC_F16, st()))
    assert torch.equal(q16.cpu(), qkv.half())
    sbuf = _seq_buf(L, seqs, dev)
    mq = max(s[1] for s in seqs)
    out = torch.zeros(T, W, device=dev)
    o16 = torch.zeros(T, W, dtype=torch.float16, device=dev)
    L.check(L.lib().rlcf_attention_fwd_pairs(q16.data_ptr(), sbuf.data_ptr(), len(seqs), mq, W, out.data_ptr(), o16.data_ptr(), None,
                                             L.PREC_F16, st()))
    ref = _attn_ref(qkv.half().double(), seqs, W, 0)
    torch.testing.assert_close(out.cpu().double(), ref, atol=4e-3, rtol=2e-3)
    rows = sorted({r for (qs, ql, _, _) in seqs for r in range(qs, qs + ql)})
    torch.testing.assert_close(o16.cpu().float()[rows], out.cpu()[rows], atol=2e-3, rtol=2e-3)
    old = torch.zeros(T, W, device=dev)
    L.check(L.lib().rlcf_attention_fwd(qd.data_ptr(), sbuf.data_ptr(), len(seqs), mq, W, 0, old.data_ptr(), None, L.PREC_F16, st()))
    # (the two kernels round P to f16 against different exponent references — the round-3 kernel moves it lazily — so they agree to
    # the f16 rounding of P, not to the last bit)
    torch.testing.assert_close(out.cpu()[rows], old.cpu()[rows], atol=3e-3, rtol=2e-3)


def test_attention_pairs_spiked_scores(L, dev):
    """a key whose score towers over the rest in a LATE chunk (the online-softmax rescale path with a large jump) and a row of
    identical scores: against the f64 reference."""
    T, W = 197, 64
    qkv = synth.normal(5, "att.spike", (T, 3 * W), 1.0)
    qkv[150, W:2 * W] = qkv[7, :W] * 6.0          # key 150 aligned with query 7: score ~ 6 |q|^2 / 8
    qkv[100, :W] = 0.0                            # query 100: all scores equal
    seqs = [(0, 197, 0, 0)]
    qd = qkv.to(dev)
    pairs = torch.empty(T, 3 * W, device=dev)
    L.check(L.lib().rlcf_split_pairs(qd.data_ptr(), pairs.data_ptr(), T * 3 * W, L.PREC_F16X3, st()))
    out = torch.zeros(T, W, device=dev)
    L.check(L.lib().rlcf_attention_fwd_pairs(pairs.data_ptr(), _seq_buf(L, seqs, dev).data_ptr(), 1, 197, W, out.data_ptr(), None, None,
                                             L.PREC_F16X3, st()))
    torch.testing.assert_close(out.cpu().double(), _attn_ref(qkv.double(), seqs, W, 0), atol=5e-6, rtol=1e-5)


def test_gemm_pair_epilogue_matches_split_of_f32_output(L, dev):
    """in_proj shape through rlcf_gemm_f16x3 with the pair-only output (epilogue kind 4 of the DMA-ring kernels): the pairs are the
    split of exactly the f32 values the f32-output epilogue writes."""
    M, N, K = 1182, 384, 128
    a = synth.normal(7, "pe.a", (M, K), 1.0).to(dev)
    w = synth.normal(7, "pe.w", (N, K), 1.0).to(dev)
    bias = synth.normal(7, "pe.b", (N,), 0.5).to(dev)
    a2 = torch.empty(M, K, device=dev); w2 = torch.empty(N, K, device=dev)
    L.check(L.lib().rlcf_split_pairs(a.data_ptr(), a2.data_ptr(), M * K, L.PREC_F16X3, st()))
    L.check(L.lib().rlcf_split_pairs(w.data_ptr(), w2.data_ptr(), N * K, L.PREC_F16X3, st()))
    c = torch.zeros(M, N, device=dev)
    chi = torch.zeros(M, N, dtype=torch.float16, device=dev)
    clo = torch.zeros(M, N, dtype=torch.float16, device=dev)
    lo = lambda t: t.data_ptr() + 64
    L.check(L.lib().rlcf_gemm_f16x3(a2.data_ptr(), lo(a2), 2 * K, w2.data_ptr(), lo(w2), 2 * K, bias.data_ptr(), None, 0, None, 0, c.data_ptr(), N,
                                    None, None, 0, M, N, K, 1.0, 0, st()))
    # (the C ABI writes the pair as two arrays; the engine asks the same epilogue for the interleaved layout — covered by
    # test_image_tower_old_and_new_attention_agree)
    L.check(L.lib().rlcf_gemm_f16x3(a2.data_ptr(), lo(a2), 2 * K, w2.data_ptr(), lo(w2), 2 * K, bias.data_ptr(), None, 0, None, 0, None, 0,
                                    chi.data_ptr(), clo.data_ptr(), N, M, N, K, 1.0, 0, st()))
    whi = torch.empty(M, N, dtype=torch.float16, device=dev)
    wlo = torch.empty(M, N, dtype=torch.float16, device=dev)
    L.check(L.lib().rlcf_split_f16x2(c.data_ptr(), whi.data_ptr(), wlo.data_ptr(), M * N, st()))
    assert torch.equal(chi.cpu().view(torch.int16), whi.cpu().view(torch.int16))
    assert torch.equal(clo.cpu().view(torch.int16), wlo.cpu().view(torch.int16))
    ref = a.cpu().double() @ w.cpu().double().t() + bias.cpu().double()
    torch.testing.assert_close(c.cpu().double(), ref, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("geo", ["ViT-B/16", "ViT-L/14"])
def test_image_tower_old_and_new_attention_agree(L, dev, geo):
    """engine image tower (split-f16 pipeline) with the producer-emitted pairs against the same tower with the round-2 f32 hand-over
    (RLCF_ATTN_OLD=1, a fresh process: the switch is read once): features agree to f32 round-off, and both agree with the fixture
    tests' bar against the reference (checked elsewhere).  8 views so the 256x256 / 256x128 GEMM tiles with the pair epilogue run."""
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from rlcf_amd import synth, _lib\nfrom rlcf_amd.engine import Engine\n"
        "g = synth.GEOMETRIES[%r]\n"
        "e = Engine(g, None, 8, 16, _lib.PREC_F16X3)\n"
        "e.load_state_dict(_lib.STUDENT, synth.make_state_dict(g, 11, device='cuda'))\n"
        "e.finalize()\n"
        "v = synth.make_views(1003, 8, g.image_resolution, device='cuda')\n"
        "f = e.encode_image(_lib.STUDENT, v)\n"
        "torch.cuda.synchronize()\n"
        "torch.save(f.cpu(), sys.argv[1])\n" % (ROOT, geo))
    import tempfile
    outs = []
    for old in ("0", "1"):
        with tempfile.NamedTemporaryFile(suffix=".pt") as tf:
            env = dict(os.environ, RLCF_ATTN_OLD=old)
            r = subprocess.run([sys.executable, "-c", code, tf.name], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(torch.load(tf.name))
    torch.testing.assert_close(outs[0], outs[1], atol=2e-6, rtol=1e-5)


# ------------------------------------------------------------------------------ two-stream overlap of the one-image call
_OVERLAP_SNIPPET = r"""
import sys, torch
sys.path.insert(0, %(root)r)
from rlcf_amd import synth, _lib
from rlcf_amd.engine import Engine, TTAConfig
from oracle import clip_ref as CR
g = synth.GEOMETRIES["ViT-B/16"]
rewards = [g] * %(n_rewards)d
eng = Engine(g, rewards if len(rewards) > 1 else g, 16, 24, _lib.PREC_F16X3)
ssd_d = synth.make_state_dict(g, 11, device="cuda")
eng.load_state_dict(_lib.STUDENT, ssd_d)
for m in range(len(rewards)):
    eng.load_state_dict(_lib.REWARD + m, synth.make_state_dict(g, 23 + m, device="cuda"))
eng.finalize()
if len(rewards) > 1: eng.set_reward_mix([0.5, 0.2, 0.3][:len(rewards)])
tokens = synth.make_token_bank(g, 24, seed=7, n_ctx=4)
ctx0 = CR.ctx_from_tokens({"token_embedding.weight": ssd_d["token_embedding.weight"].cpu()}, synth.ctx_token_ids_default(g, 4))
eng.set_class_bank(tokens, 4, ctx0, %(text_mode)s)
cfg = TTAConfig(selection_p=0.4, sample_k=3)          # 6 views x 3 classes: 18 prompts x 77 rows = 1386 > 512 in the dense layout
outs = []
for rep in range(3):
    v = synth.make_views(1000 + rep, 16, g.image_resolution, device="cuda")
    o = eng.tta_sample(v, cfg)
    torch.cuda.synchronize()
    rif = o["reward_image_features"]
    rif = torch.cat([r.flatten() for r in rif]) if isinstance(rif, list) else rif.flatten()
    outs.append({k: o[k].cpu() for k in ("final_logits", "ctx_after", "rewards", "clip_score", "topk_idx", "ctx_grad")} | {"rif": rif.cpu()})
torch.save(outs, sys.argv[1])
"""


@pytest.mark.parametrize("text_mode,n_rewards", [("_lib.TEXT_DENSE", 1), ("_lib.TEXT_SHARED", 3)])
def test_one_image_call_overlap_is_bitwise_the_single_stream_result(L, dev, text_mode, n_rewards):
    """engine_tta_sample runs the reward towers of the selected views on a second stream beside the student's sparse text forward.  In
    the dense text layout the main stream's GEMMs re-split their A operand (1 386 rows > 512) while the side stream's reward tower
    reads ITS split operand: since round 3 the side stream owns a second buffer (round-2 advisory: both used e->a_hi).  Bitwise
    equality with RLCF_NO_OVERLAP=1 over three samples, also with a 3-member reward ensemble (the side stream rewrites its buffer
    once per member)."""
    import tempfile
    code = _OVERLAP_SNIPPET % dict(root=ROOT, n_rewards=n_rewards, text_mode=text_mode)
    res = []
    for no in ("0", "1"):
        with tempfile.NamedTemporaryFile(suffix=".pt") as tf:
            env = dict(os.environ, RLCF_NO_OVERLAP=no)
            r = subprocess.run([sys.executable, "-c", code, tf.name], env=env, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-3000:]
            res.append(torch.load(tf.name))
    for a, b in zip(*res):
        for k in a:
            assert torch.equal(a[k], b[k]), k


# ------------------------------------------------------------------------------ the two-stream sample-batched pass
_PIPE_SNIPPET = r"""
import sys, torch
sys.path.insert(0, %(root)r)
from rlcf_amd import synth, _lib
from rlcf_amd.engine import Engine, TTAConfig
from oracle import clip_ref as CR
g = synth.GEOMETRIES[%(geo)r]
B, N, C = %(B)d, %(N)d, %(C)d
eng = Engine(g, g, B * N, C, %(prec)s)
ssd_d = synth.make_state_dict(g, 11, device="cuda")
eng.load_state_dict(_lib.STUDENT, ssd_d)
eng.load_state_dict(_lib.REWARD, synth.make_state_dict(g, 23, device="cuda"))
eng.finalize()
tokens = synth.make_token_bank(g, C, seed=7, n_ctx=4)
ctx0 = CR.ctx_from_tokens({"token_embedding.weight": ssd_d["token_embedding.weight"].cpu()}, synth.ctx_token_ids_default(g, 4))
eng.set_class_bank(tokens, 4, ctx0, _lib.TEXT_SHARED)
cfg = TTAConfig(selection_p=%(p)s, sample_k=3)
outs = []
for rep in range(2):
    vs = torch.stack([synth.make_views(1000 + rep * B + b, N, g.image_resolution, device="cuda") for b in range(B)])
    top5, fl = eng.tta_batch(vs, cfg, want_logits=True)
    torch.cuda.synchronize()
    outs.append((top5.cpu(), fl.cpu()))
torch.save(outs, sys.argv[1])
"""


@pytest.mark.parametrize("geo,B,N,C,p,prec,parts", [("small", 8, 8, 16, "0.5", "_lib.PREC_F16X3", "2"), ("small", 9, 8, 16, "0.5", "_lib.PREC_F16X3", "4"),
                                                     ("ViT-B/16", 16, 16, 40, "0.25", "_lib.PREC_F16X3", "4"), ("ViT-B/16", 8, 16, 40, "0.25", "_lib.PREC_F16", "2")])
def test_batch_pipeline_equals_single_stream(L, dev, geo, B, N, C, p, prec, parts):
    """rlcf_tta_batch with the pass cut into parts on two streams (student tower of part k+1 beside everything behind the tower of
    part k, side-stream scratch of its own) against the same pass on one stream (RLCF_BATCH_PARTS=1): the samples are independent, so
    results agree to the round-off of the GEMM forms the part sizes select (split-K and tile choice follow M: logits within 5e-4 at
    logit scale 100, same top-1); odd part sizes and the f16 mode included.  (Measured on configs[1]: 115.7 images/s on one stream,
    114.1 / 109.3 in 2 / 4 parts — a workgroup of the small kernels behind the tower blocks a CU for the 139-KB GEMM workgroups just
    as it does alone, so nothing is hidden; the one-stream pass stays the default, RLCF_BATCH_PARTS=n selects the pipelined one.)"""
    import tempfile
    code = _PIPE_SNIPPET % dict(root=ROOT, geo=geo, B=B, N=N, C=C, p=p, prec=prec)
    res = []
    for pe in (parts, "1"):
        with tempfile.NamedTemporaryFile(suffix=".pt") as tf:
            env = dict(os.environ, RLCF_BATCH_PARTS=pe)
            r = subprocess.run([sys.executable, "-c", code, tf.name], env=env, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-3000:]
            res.append(torch.load(tf.name))
    for (t_a, f_a), (t_b, f_b) in zip(*res):
        assert torch.equal(t_a[:, 0], t_b[:, 0])
        # Adam's first step is -lr * sign(g): a gradient element at round-off level flips with the GEMM form and moves one prompt
        # element by 2 lr, i.e. a sample's logits by ~2e-3 — the bar the reference fixtures use as well
        tol = 5e-3 if "X3" in prec else 5e-2
        torch.testing.assert_close(f_a, f_b, atol=tol, rtol=1e-5)


# ------------------------------------------------------------------------------ the RCCL code path, executed with one rank
def _run_json(cmd, env=None, timeout=900):
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def _torchrun(port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)]


def test_bench_runs_the_rccl_path_with_one_rank(L, dev):
    """bench.py launched the way the driver launches N > 1 (torch.distributed.run, one rank per GPU) with the RCCL backend and ONE rank:
    init_process_group("nccl", device_id=...), both barriers and the CUDA-tensor all_reduce(MAX) of the elapsed time execute; the line
    reports the rank count RCCL saw and is otherwise the line of the plain single-process run (same first prediction, same executed
    FLOPs, same configuration)."""
    small = ["--steps", "4", "--warmup", "2", "--batch", "2", "--classes", "50", "--no-cpu-baseline", "--sustain-seconds", "0", "--no-f16-line"]
    plain = _run_json([sys.executable, "bench.py"] + small)
    rccl = _run_json(_torchrun(29541) + ["bench.py", "--gpus", "1", "--dist-backend", "nccl"] + small, env={"RLCF_FORCE_DIST": "1"})
    assert rccl["distributed"]["rccl_ranks"] == 1 and "RCCL" in rccl["distributed"]["backend"]
    assert "distributed" not in plain
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "top1_first", "config"):
        assert rccl[k] == plain[k], k
    assert abs(rccl["flops_exec_per_image"] - plain["flops_exec_per_image"]) <= 1e-6 * plain["flops_exec_per_image"]
    assert rccl["value"] > 0 and rccl["roofline"]["frac"] > 0


def test_eval_driver_runs_the_rccl_path_with_one_rank(L, dev):
    """python -m rlcf_amd.eval under torch.distributed.run with backend nccl and one rank: the end-of-dataset all_reduce of the hit
    counters, the all_gather of the int64 top-5 block and the all_reduce(MAX) of the time run on RCCL with CUDA tensors; predictions are
    those of the plain run."""
    args = ["--total-images", "6", "--images-per-pass", "3", "--views", "8", "--classes", "40", "--selection-p", "0.5"]
    plain = _run_json([sys.executable, "-m", "rlcf_amd.eval"] + args)
    rccl = _run_json(_torchrun(29542) + ["-m", "rlcf_amd.eval", "--gpus", "1", "--dist-backend", "nccl"] + args, env={"RLCF_FORCE_DIST": "1"})
    assert rccl["distributed"] == {"backend": "nccl", "ranks": 1, "collectives": rccl["distributed"]["collectives"]}
    assert plain["distributed"] is None
    assert rccl["predictions_sha256"] == plain["predictions_sha256"] and rccl["top5"] == plain["top5"]
    assert (rccl["images"], rccl["acc1"], rccl["acc5"]) == (plain["images"], plain["acc1"], plain["acc5"])


# ------------------------------------------------------------------------------ bit-reproducible image-tower backward
@pytest.mark.parametrize("geo,n_cls", [("tiny", 16), ("ViT-B/16", 1000)])
def test_backbone_tuning_bit_reproducible(L, dev, geo, n_cls):
    """LayerNorm tuning (one image and the sample-batched call) and every-parameter tuning of the image encoder give the same BITS on
    every run: dK / dV of the attention backward are parked per query block and added in block order, the LayerNorm / ln_pre parameter
    gradients and the bias column sums are per-wave partial sums added in a fixed order, d feat is a fixed-order reduction (no float
    atomics left on the path in RLCF_PREC_F16X3)."""
    from rlcf_amd.engine import TTAConfig
    from test_gpu_parity import make_engine
    N = 8
    eng, *_ = make_engine((geo, geo if geo != "tiny" else "tiny-r"), N * 2, n_cls, L.TEXT_SHARED, prec=L.PREC_F16X3)
    cfg = TTAConfig(selection_p=0.5, tta_steps=2, lr=1e-4)
    R = synth.GEOMETRIES[geo].image_resolution
    vs = torch.stack([synth.make_views(1000 + i, N, R, device=dev) for i in range(2)])      # (seed 1000: the ln_b16_n8 fixture's views, non-zero rewards)
    runs = [eng.tta_sample_ln(vs[0], cfg) for _ in range(3)]
    assert runs[0]["ln_grad"].abs().max() > 0
    for o in runs[1:]:
        for k in ("ln_grad", "ln_after", "final_logits"):
            assert torch.equal(o[k], runs[0][k]), k
    b0 = eng.tta_batch_ln(vs, cfg, want_logits=True)[1].clone()
    for _ in range(2):
        assert torch.equal(eng.tta_batch_ln(vs, cfg, want_logits=True)[1], b0)
    full = [eng.tta_sample_visual(vs[0], cfg) for _ in range(3)]
    assert full[0]["vis_grad"].abs().max() > 0
    for o in full[1:]:
        for k in ("ln_grad", "vis_grad", "vis_after", "final_logits"):
            assert torch.equal(o[k], full[0][k]), k
    eng.close()


# ------------------------------------------------------------------------------ bench.py --config: the other single-GPU BASELINE configs
@pytest.mark.parametrize("config,classes", [(0, 50), (2, 50), (4, 50)])
def test_bench_config_switch_emits_the_contract_line(L, dev, config, classes):
    """`bench.py --config N`: configs[0] (N = 8 views, selection_p 0.5), configs[2] (ViT-L/14 LayerNorm tuning through
    rlcf_tta_batch_ln) and configs[4] (RN50x64 @448 student: the implicit 3x3 convolutions and the pair-to-pair hand-over of the ResNet
    tower) emit the same JSON shape as the headline config: metric / value / roofline with a per-kernel table that names backward
    kernels where the configuration has them."""
    line = _run_json([sys.executable, "bench.py", "--config", str(config), "--steps", "2", "--warmup", "1", "--classes", str(classes),
                      "--no-cpu-baseline", "--sustain-seconds", "0"])
    assert line["metric"] and line["unit"] == "images/s" and line["value"] > 0 and line["n_gpus"] == 1 and line["steps"] == 2
    assert line["higher_is_better"] is True and line["config"]["classes"] == classes
    arch = {0: "ViT-B/16 student", 2: "ViT-L/14 student", 4: "RN50x64 student"}[config]
    assert arch in line["config"]["workload"] and line["config"]["views"] == {0: 8, 2: 64, 4: 32}[config]
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and roof["achieved"] > 0
    kinds = {r["kernel"] for r in roof["per_kernel_all_launches"]}
    assert any("gemm" in k for k in kinds)
    if config == 2:
        assert any("attention backward" in k for k in kinds) and any("LayerNorm backward" in k for k in kinds)


# ------------------------------------------------------------------------------ implicit 3x3 convolution (op level)
@pytest.mark.parametrize("n,H,W,Cin,Cout,epi,res", [(16, 56, 56, 64, 64, 3, False), (4, 28, 28, 32, 4096, 0, True), (40, 15, 15, 96, 1536, 3, True),
                                                    (52, 10, 24, 128, 1024, 0, False)])
def test_conv3x3_implicit_gemm_matches_conv2d(L, dev, n, H, W, Cin, Cout, epi, res):
    """rlcf_conv3x3_nhwc_f16x3 (the 256x256 split-f16 GEMM gathering every tap's K tile from the activation's operand pairs, zero page
    outside the image) against torch's conv2d in float64: square / odd / non-square maps, M tails, bias + identity + ReLU."""
    x = synth.normal(3, "cv.x", (n, H, W, Cin)).to(dev)
    w = (synth.normal(3, "cv.w", (Cout, 3, 3, Cin)) * (9 * Cin) ** -0.5).to(dev)
    b = synth.normal(3, "cv.b", (Cout,), 0.1).to(dev)
    r = synth.normal(3, "cv.r", (n, H, W, Cout)).to(dev) if res else None
    y = torch.empty(n, H, W, Cout, device=dev)
    L.check(L.lib().rlcf_conv3x3_nhwc_f16x3(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), n, H, W, Cin, Cout,
                                            epi, torch.cuda.current_stream().cuda_stream))
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(), padding=1).permute(0, 2, 3, 1)
    if res:
        ref = ref + r.double()
    if epi == 3:                              # RLCF_EPI_RELU
        ref = ref.clamp_min(0)
    torch.testing.assert_close(y.double(), ref, atol=2e-5, rtol=1e-5)
    with pytest.raises(L.RlcfError):          # too few tiles for the 256x256 kernel: refused, not silently something else
        L.check(L.lib().rlcf_conv3x3_nhwc_f16x3(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), 1, H, W, Cin, Cout, 0,
                                                torch.cuda.current_stream().cuda_stream))
