"""Host-side mirror of the RLCF prompt-tuning entry (TPT/tpt_cls_rl.py): `select_confident_samples`
(:32-35), `avg_entropy` (:38-44), `test_time_tuning` (:47-79), `test_time_adapt_eval` (:219-279) and
`accuracy` (TPT/utils/tools.py:84-98).  Same signatures; the arithmetic is one call into the HIP engine."""
from __future__ import annotations

import contextlib
import math
import os
import time

import torch

from . import _lib as L
from . import runtime
from .datautils import _upload
from .engine import TTAConfig


def select_confident_samples(logits, top):
    """TPT/tpt_cls_rl.py:32-35 -> (logits[idx], idx): the int(N*top) lowest-entropy rows, ascending."""
    logits = logits.contiguous().float()
    n, c = logits.shape
    n_sel = int(n * top)
    ent = torch.empty(n, device=logits.device)
    idx = torch.empty(max(n_sel, 1), dtype=torch.int32, device=logits.device)
    L.check(L.lib().rlcf_entropy_select(logits.data_ptr(), n, c, n_sel, ent.data_ptr(), idx.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "entropy_select")
    idx = idx[:n_sel].long()
    return logits[idx], idx


def avg_entropy(outputs):
    """TPT/tpt_cls_rl.py:38-44 (stand-alone convenience on device tensors; inside the tuning step the regulariser and its gradient
    come from the fused rlcf_reward_loss kernel).  One launch of avg_entropy_kernel (rlcf_avg_entropy); tensors with autograd history
    or off the GPU take the reference's torch expression."""
    if outputs.is_cuda and not outputs.requires_grad and outputs.dim() == 2 and outputs.shape[0] <= 4096:
        x = outputs.detach().float().contiguous()
        out = torch.empty((), device=x.device, dtype=torch.float32)
        L.check(L.lib().rlcf_avg_entropy(x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), torch.cuda.current_stream().cuda_stream), "avg_entropy")
        return out.to(outputs.dtype)               # (the reference's expression keeps the input dtype: fp16 logits under autocast)
    logits = outputs - outputs.logsumexp(dim=-1, keepdim=True)
    avg_logits = logits.logsumexp(dim=0) - math.log(logits.shape[0])
    avg_logits = torch.clamp(avg_logits, min=torch.finfo(avg_logits.dtype).min)
    return -(avg_logits * torch.exp(avg_logits)).sum(dim=-1)


def _config(args, optimizer, reward_model) -> TTAConfig:
    g = optimizer.param_groups[0]
    b1, b2 = g.get("betas", (0.9, 0.999))
    return TTAConfig(selection_p=args.selection_p, tta_steps=args.tta_steps, sample_k=reward_model.sample_k, lr=g["lr"],
                     weight_decay=g.get("weight_decay", 0.0), beta1=b1, beta2=b2, eps=g.get("eps", 1e-8),
                     reward_process=bool(reward_model.reward_process), process_batch=bool(reward_model.process_batch),
                     reward_amplify=bool(reward_model.amplify_rewards), clipscore_weight=reward_model.clipscore_weight,
                     min_entropy_reg=bool(getattr(args, "min_entropy_reg", 0)),
                     min_entropy_w=float(getattr(args, "min_entropy_w", 0.1)))


def test_time_tuning(model, inputs, optimizer, scaler, args, reward_model=None):
    """TPT/tpt_cls_rl.py:47-79.  `optimizer` supplies the AdamW hyper-parameters (its state is reset per
    sample by the harness, :255, so the fused step starts from step 1); `scaler` is accepted and unused:
    the HIP path computes in f32-grade precision, there is no loss scaling to apply."""
    # the tensor test_time_adapt_eval will pass to model(image) next (row 0 of `inputs`): consumed HERE, whatever happens below — an early
    # return or an exception must not leave it behind for an unrelated later call
    hint = getattr(model, "_clean_view_hint", None)
    model._clean_view_hint = None
    if reward_model is None:
        raise ValueError("RLCF needs a reward model (get_reward_model)")
    if args.tta_steps <= 0:
        return
    cfg = _config(args, optimizer, reward_model)
    eng = runtime.SESSION.engine(inputs.shape[0])
    if not hasattr(model, "prompt_learner"):               # CLIPCLS_TTA: image-encoder tuning (TPT/tune_cls_rl.py:31,217)
        full = not model.only_norm
        trk = model._track
        trk.check()
        # reset() just ran and nothing the mirror can see has touched the tensors since (no device round trip); an edit through `.data`
        # in between is invisible from the host: the device-side guard queued below reports it at the next entry point
        at_reset = trk.at_reset(model.ln, model.vis if full else None)
        if not at_reset and (not torch.equal(model.ln.data, model._ln_init) or (full and not torch.equal(model.vis.data, model._vis_init))):
            raise NotImplementedError("image-encoder tuning starts from the reset state (model.reset(), tune_cls_rl.py:210)")
        if at_reset:
            trk.guard(model.ln.data, model._ln_init, "CLIPCLS_TTA: the norm-layer parameters were not the reset state when test_time_tuning ran")
            # (every-parameter tuning: comparing the whole visual vector — 1e8 floats and a bool temporary of the same length — per test
            # image is a debugging aid, not a default: RLCF_GUARD_VISUAL=1.  The small norm-layer vector above is always guarded.  A guard
            # reports one sample late: the offending sample's prediction is already in the hit counts when it raises.)
            if full and os.environ.get("RLCF_GUARD_VISUAL", "0") == "1":
                trk.guard(model.vis.data, model._vis_init, "CLIPCLS_TTA: the visual parameters were not the reset state when test_time_tuning ran")
        out = (eng.tta_sample_visual if full else eng.tta_sample_ln)(inputs, cfg, skip_final=True)
        with torch.no_grad():
            model.ln.data.copy_(out["ln_after"])
            if full:
                model.vis.data.copy_(out["vis_after"])
                model.vis.grad = None
        trk.wrote()
        model.ln.grad = None
        return
    pl = model.prompt_learner
    trk = pl._track
    trk.check()
    at_reset = trk.at_reset(pl.ctx)                          # reset() just ran (no device round trip)
    ctx_in = None if at_reset or torch.equal(pl.ctx.data, pl.ctx_init_state) else pl.ctx.data
    if at_reset:
        trk.guard(pl.ctx.data, pl.ctx_init_state, "the prompt was not its reset state when test_time_tuning ran (a `.data` edit after reset())")
    # The harness asks for model(image) on the clean view right after this call (tpt_cls_rl.py:260-262).  The image tower is frozen, so
    # the engine already holds that view's features (row 0 of the N-view pass): the fused call finishes the sample — text features of
    # the adapted prompt, logits — and the result is kept for ClipTestTimeTuning.inference, which returns it when it is asked for
    # exactly that view with exactly this prompt (otherwise it computes as before).  Saves one 197-token tower pass per test image.
    out = eng.tta_sample(inputs, cfg, want_intermediates=False, skip_final=False, ctx_in=ctx_in)
    with torch.no_grad():
        pl.ctx.data.copy_(out["ctx_after"])
    pl.ctx.grad = None
    trk.wrote()
    # the hint is honoured only if it IS row 0 of this call's input (same storage start, one image of the same shape) — and the cache keeps
    # a COPY of that row (600 KB), not a view that would pin the whole N-view tensor until the next reset
    hint_key = None
    if hint is not None and hint.dim() == inputs.dim() and hint.shape[0] == 1 and hint.shape[1:] == inputs.shape[1:] and hint.device == inputs.device:
        hint_key = (hint.data_ptr(), tuple(hint.shape), hint._version)
    model._tuned_view_cache = (inputs[:1].clone(), trk.gen, pl.ctx._version, out["ctx_after"], out["final_logits"], hint_key)
    return


def accuracy(output, target, topk=(1,)):
    """TPT/utils/tools.py:84-98.  topk drawn from {1, 5} on device tensors: top5_kernel + accuracy_kernel (rlcf_accuracy; ties go to the
    lower class index); anything else takes the reference's torch expression."""
    if output.is_cuda and output.dim() == 2 and target.is_cuda and set(topk) <= {1, 5} and not output.requires_grad and output.shape[1] >= max(topk):
        # (fewer classes than max(topk): the reference's output.topk raises — the torch expression below does the same)
        x = output.detach().float().contiguous()
        t = target.detach().to(torch.int64).contiguous()
        B, C = x.shape
        scratch = torch.empty(B, 5, dtype=torch.int32, device=x.device)
        res = torch.empty(2, dtype=torch.float32, device=x.device)
        L.check(L.lib().rlcf_accuracy(x.data_ptr(), t.data_ptr(), B, C, scratch.data_ptr(), res.data_ptr(), torch.cuda.current_stream().cuda_stream),
                "accuracy")
        return [res[0:1] if k == 1 else res[1:2] for k in topk]
    with torch.no_grad():
        maxk = max(topk)
        batch_size = target.size(0)
        _, pred = output.topk(maxk, 1, True, True)
        pred = pred.t()
        correct = pred.eq(target.view(1, -1).expand_as(pred))
        return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / batch_size) for k in topk]


def _hits(top5: torch.Tensor, targets, hits: torch.Tensor) -> None:
    """hits[0] += top-1 hits, hits[1] += top-5 hits of the engine's top-5 rows against the labels (rlcf_top5_hits: one launch)."""
    dev = top5.device
    t = torch.stack([x.reshape(-1)[0] for x in targets]) if not isinstance(targets, torch.Tensor) else targets
    t = (_upload(t, dev) if not t.is_cuda else t.to(dev)).to(torch.int64).contiguous()
    L.check(L.lib().rlcf_top5_hits(top5.data_ptr(), t.data_ptr(), top5.shape[0], hits.data_ptr(), torch.cuda.current_stream().cuda_stream), "top5_hits")


def _eval_batched(val_loader, model, optimizer, args, reward_model, images_per_pass):
    """Throughput form of the harness: test images are independent units (per-sample reset, tpt_cls_rl.py:251-255), so
    `images_per_pass` of them share every tower pass inside the engine (rlcf_tta_batch / rlcf_tta_batch_ln); same predictions."""
    cfg = _config(args, optimizer, reward_model)
    prompt = hasattr(model, "prompt_learner")
    n, hits, buf, tgt = 0, None, [], []

    def flush():
        # (hit counts stay on the device until the end: reading them per pass would stop the host from preparing the next pass's views
        # while this one runs; labels go up on a side stream for the same reason)
        nonlocal n, hits, buf, tgt
        if not buf:
            return
        views = torch.stack(buf)
        eng = runtime.SESSION.engine(views.shape[0] * views.shape[1])
        top5 = (eng.tta_batch if prompt else eng.tta_batch_ln)(views, cfg)
        if hits is None:
            hits = torch.zeros(2, device=top5.device)
        _hits(top5, tgt, hits)
        n += len(buf)
        buf, tgt = [], []

    for images, target in val_loader:
        if isinstance(images, list):
            images = torch.cat([im.cuda(args.gpu, non_blocking=True) for im in images], dim=0)
        else:
            images = (images.squeeze(0) if images.dim() > 4 else images).cuda(args.gpu, non_blocking=True)
        buf.append(images)
        tgt.append(target.reshape(-1)[0])
        if len(buf) == images_per_pass:
            flush()
    flush()
    s1, s5 = (float(hits[0]) * 100.0, float(hits[1]) * 100.0) if n else (0.0, 0.0)
    return [round(x, 3) for x in [s1 / max(n, 1), s5 / max(n, 1)]]


def _eval_in_flight(val_loader, model, optimizer, args, reward_model, lanes):
    """One test image per engine call, as the reference feeds them, with `lanes` images IN FLIGHT: a sample's step is a 64-view tower pass
    that fills the chip followed by a long tail of few-row launches (the selected views through the reward model, the sampled classes'
    text passes, the final text pass) that does not — so sample i + 1 starts on another engine and another stream while sample i
    finishes.  Samples are independent units (per-sample reset, tpt_cls_rl.py:251-255): every sample's arithmetic is exactly the
    one-at-a-time call's (rlcf_tta_batch with one image), only the wall clock changes.  Lane k takes images k, k + lanes, ...
    ONE host thread (this one) enqueues every lane through rlcf_lanes_submit: the hand-offs between the producer's stream and the lanes
    are device-side events, there are no Python threads, queues or per-sample torch expressions (round 5 ran one Python thread per lane;
    its legs scattered by 30 % with the GIL's scheduling)."""
    from collections import deque
    from .engine import Lanes
    cfg = _config(args, optimizer, reward_model)
    prompt = hasattr(model, "prompt_learner")
    dev = torch.device("cuda", args.gpu)
    CHUNK = 256                                                  # samples per output block (top-5 rows live until the final count)
    ln, blocks, targets, n = None, [], [], 0
    in_queue = deque()                                           # one event per submitted sample: bounds how far the host runs ahead
    # Views that are ALREADY on the device (a device-side augmenter) are gathered on the LANE's stream, not on the loop's: the loop's stream then
    # carries no work at all, and no lane waits for it.  With the gather (`torch.cat`, the reference's own, tpt_cls_rl.py:246) on the loop's
    # stream every lane depended on a stream that shares a hardware queue with one of them in some stream assignments — its copy kernel then
    # sat behind that lane's whole step and the other lanes starved (round 6: one staged leg in three at 85 instead of 97 images/s).  A loader
    # that makes its views on the device inside next() (datautils.ViewPrefetcher: `on_device`) is advanced under the lane's stream as well.
    main = torch.cuda.current_stream(dev)
    on_device = bool(getattr(val_loader, "on_device", False))
    it = iter(val_loader)
    i = -1
    try:
        while True:
            lane_st = ln.streams[ln.next_lane] if ln is not None else None
            try:
                if lane_st is not None and on_device:
                    with torch.cuda.stream(lane_st):
                        images, target = next(it)
                else:
                    images, target = next(it)
            except StopIteration:
                break
            i += 1
            first = images[0] if isinstance(images, list) else images
            use_lane = lane_st is not None and first.is_cuda
            if use_lane and not on_device:
                lane_st.wait_stream(main)                        # (whatever produced the views on the loop's stream)
            with (torch.cuda.stream(lane_st) if use_lane else contextlib.nullcontext()):
                if isinstance(images, list):
                    if use_lane:
                        for im in images:
                            im.record_stream(lane_st)
                    images = torch.cat([im.cuda(args.gpu, non_blocking=True) for im in images], dim=0)
                else:
                    if use_lane:
                        images.record_stream(lane_st)
                    images = (images.squeeze(0) if images.dim() > 4 else images).cuda(args.gpu, non_blocking=True)
                if ln is None:                                   # engines are sized by the first image's view count
                    ln = Lanes(runtime.SESSION.lane_engines(lanes, images.shape[0]))
                if i % CHUNK == 0:
                    blocks.append(torch.empty(CHUNK, 5, dtype=torch.int32, device=dev))
                views = images.to(dev, torch.float32).contiguous()
                k = ln.submit(views, cfg, blocks[-1][i % CHUNK], norm_layers=not prompt)
            targets.append(target.reshape(-1)[0])
            n += 1
            ev = torch.cuda.Event()
            ev.record(ln.streams[k])
            in_queue.append(ev)
            if len(in_queue) > 4 * lanes:                        # (a sample's 64 views are 38 MB: at most 4 per lane wait in the queues)
                in_queue.popleft().synchronize()
        if ln is None:
            return [0.0, 0.0]
        ln.join()
        hits = torch.zeros(2, device=dev)
        for b, blk in enumerate(blocks):
            m = min(CHUNK, n - b * CHUNK)
            _hits(blk[:m], targets[b * CHUNK: b * CHUNK + m], hits)
        s1, s5 = float(hits[0]) * 100.0, float(hits[1]) * 100.0
    finally:
        if ln is not None:
            ln.close()                                           # (waits for the lanes; the engines get their side streams back)
    return [round(x, 3) for x in [s1 / max(n, 1), s5 / max(n, 1)]]


def _loop_option(value, args, name):
    """`images_per_pass` / `in_flight` of test_time_adapt_eval: the keyword if the caller gave one, else `args.<name>` (rlcf_amd.params adds
    --images_per_pass / --in_flight), else the environment (RLCF_IMAGES_PER_PASS / RLCF_IN_FLIGHT) — so that the reference's own main_worker,
    which calls test_time_adapt_eval(val_loader, model, optimizer, optim_state, scaler, args) and nothing more (TPT/tpt_cls_rl.py:187-188),
    reaches both forms after the import swap of INTEGRATION.md section A without an edit of the call — else 1 (the reference's loop)."""
    if value is None:
        value = getattr(args, name, None)
    if value is None:
        value = os.environ.get("RLCF_" + name.upper()) or 1
    value = int(value)
    if value < 1:
        raise ValueError(f"{name} must be >= 1, got {value}")
    return value


def test_time_adapt_eval(val_loader, model, optimizer, optim_state, scaler, args, device=None, reward_model=None, images_per_pass=None, in_flight=None):
    """TPT/tpt_cls_rl.py:219-279 and its twin TPT/tune_cls_rl.py:183-256 (CLIPCLS_TTA models: model.train() / model.eval() round the
    tuning step, :216-218, and model.momentum_update_model() after the clean-view inference, :240): per test image
    reset -> tune -> clean-view inference -> (EMA) -> top-1/top-5.
    `images_per_pass > 1` (not in the reference) hands that many test images to the engine at once; `in_flight > 1` (not in the reference
    either) keeps one image per engine call and runs that many samples side by side on their own engines and streams (_eval_in_flight).
    Both need independent samples: no cross-sample EMA, no per-sample encoder weights."""
    images_per_pass, in_flight = _loop_option(images_per_pass, args, "images_per_pass"), _loop_option(in_flight, args, "in_flight")
    backbone = not hasattr(model, "prompt_learner")                 # CLIPCLS_TTA: the tune_cls_rl.py form of the loop
    full_visual = not hasattr(model, "prompt_learner") and not model.only_norm      # per-sample weights: one sample per pass
    if images_per_pass > 1 and args.tta_steps > 0 and not getattr(model, "momentum_update", False) and not full_visual:
        model.eval()
        with torch.no_grad():
            model.reset()
        return _eval_batched(val_loader, model, optimizer, args, reward_model, images_per_pass)
    if in_flight > 1 and args.tta_steps > 0 and not getattr(model, "momentum_update", False) and not full_visual and not (
            backbone and runtime.SESSION.reset_state_moved()):      # (an earlier EMA run left lane 0 another reset state than the lanes built from the checkpoint: serial loop)
        model.eval()
        with torch.no_grad():
            model.reset()
        return _eval_in_flight(val_loader, model, optimizer, args, reward_model, in_flight)
    # The hit counts accumulate ON THE DEVICE (exact: sums of 0 / 100 in float32) and are read at print_freq and at the end: the
    # reference reads them after every image (float(acc1[0])), which makes the host wait for the GPU before it may draw the next
    # image's views — the one place where following the loop line by line would leave the GPU idle.  Same numbers.
    n, s1_t, s5_t = 0, None, None
    model.eval()
    with torch.no_grad():
        model.reset()
    end = time.time()
    for i, (images, target) in enumerate(val_loader):
        assert args.gpu is not None
        if isinstance(images, list):
            images = [im.cuda(args.gpu, non_blocking=True) for im in images]
            image = images[0]
        else:
            if images.dim() > 4:
                assert images.size(0) == 1
                images = images.squeeze(0)
            images = images.cuda(args.gpu, non_blocking=True)
            image = images
        # (the label goes up on a side stream: a host-memory copy issued on the compute stream makes the host wait for everything queued there)
        target = _upload(target, torch.device("cuda", args.gpu)) if not target.is_cuda and not target.is_pinned() else target.cuda(args.gpu, non_blocking=True)
        if args.tpt and isinstance(images, list):
            images = torch.cat(images, dim=0)
        if args.tta_steps > 0:
            with torch.no_grad():
                model.reset()
        optimizer.load_state_dict(optim_state)
        if backbone:
            model.train()                                           # tune_cls_rl.py:216
        elif args.tpt and image is not images and image.dim() == 4 and image.size(0) == 1:
            model._clean_view_hint = image                          # images = cat([image, views...]): row 0 of the step's input IS this tensor
        test_time_tuning(model, images, optimizer, scaler, args, reward_model=reward_model)
        if backbone:
            model.eval()                                            # tune_cls_rl.py:218
        with torch.no_grad():
            output = model(image)
        if backbone:
            model.momentum_update_model()                           # tune_cls_rl.py:240 (no-op unless momentum_update)
        acc1, acc5 = accuracy(output, target, topk=(1, 5))
        bs = image.size(0)
        n += bs
        s1_t = acc1[0] * bs if s1_t is None else s1_t + acc1[0] * bs
        s5_t = acc5[0] * bs if s5_t is None else s5_t + acc5[0] * bs
        if (i + 1) % getattr(args, "print_freq", 500) == 0:
            print(f"Test: [{i + 1}/{len(val_loader)}] Time {time.time() - end:6.3f} Acc@1 {float(s1_t) / n:6.2f} Acc@5 {float(s5_t) / n:6.2f}")
        elif (i + 1) % 32 == 0:
            torch.cuda.current_stream().synchronize()                # (bounds how far the host runs ahead of the device)
        end = time.time()
    s1, s5 = (float(s1_t), float(s5_t)) if n else (0.0, 0.0)
    for trk in (getattr(model, "_track", None), getattr(getattr(model, "prompt_learner", None), "_track", None)):
        if trk is not None:
            trk.check(wait=True)                                     # (the guards of the last samples: the loop is over, waiting costs nothing)
    return [round(x, 3) for x in [s1 / max(n, 1), s5 / max(n, 1)]]
