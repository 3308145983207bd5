"""Sharded evaluation driver: BASELINE configs[3] — a fixed stream of independent test images split over the GPUs of one node.

    python -m rlcf_amd.eval --gpus 1 --total-images 256
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        -m rlcf_amd.eval --gpus 8 --total-images 256

The reference evaluates one image at a time on one GPU (TPT/tpt_cls_rl.py:219-279; "only been tested under the single GPU
setting", TPT/params.py:92-93).  Test images are independent units — every sample starts from the reset prompt and the reset
optimizer state (:251-255) — so rank r takes the contiguous block shard_range(total, r, world) of the stream, runs the fused
per-sample step on it (rlcf_tta_batch), and the ONLY communication is at the end of the dataset: one all_reduce of
(top-1 hits, top-5 hits, n) and one all_gather of the top-5 predictions (RCCL on GPUs; gloo lets two ranks share one GPU in
tests).  Nothing on the data path is collective.  Inputs are the seeded synthetic stream of SURVEY section 8d (sample i ->
views seed 1000 + i, target = a seeded class id), so results do not depend on the placement.

Prints one JSON line on rank 0: accuracy, the predictions' digest, images/s (max over ranks of the shard time).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import torch

from . import _lib, shard, synth
from .engine import Engine, TTAConfig


def synthetic_targets(n: int, n_cls: int, seed: int = 5) -> torch.Tensor:
    """class id of stream sample i (seeded, placement-independent; the synthetic stream has no labels of its own)"""
    return synth.randint(seed, "eval.targets", n, 0, n_cls)


def run(args) -> dict:
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    local = shard.local_device(local, args.dist_backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or bool(os.environ.get("RLCF_FORCE_DIST"))     # RLCF_FORCE_DIST: the RCCL path with one rank (tests)
    if world > 1:                   # one process per GPU: keep each rank on its own slice of the host cores
        ncpu = os.cpu_count() or 1
        per = max(1, ncpu // world)
        torch.set_num_threads(min(per, 16))
        try:
            os.sched_setaffinity(0, set(range(local * per, (local + 1) * per)))
        except (AttributeError, OSError):
            pass
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group(args.dist_backend, **({"device_id": dev} if args.dist_backend == "nccl" else {}))
    geo, rgeo = synth.GEOMETRIES[args.arch], synth.GEOMETRIES[args.reward_arch]
    ssd, rsd = synth.make_state_dict(geo, 11, device=dev), synth.make_state_dict(rgeo, 23, device=dev)
    tokens = synth.make_token_bank(geo, args.classes, seed=7, n_ctx=args.n_ctx)
    ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(geo, args.n_ctx), device=dev)].clone()
    prec = {"f32": _lib.PREC_F32, "f16x3": _lib.PREC_F16X3}[args.precision]
    eng = Engine(geo, rgeo, args.views * args.images_per_pass, args.classes, prec)
    eng.load_state_dict(_lib.STUDENT, ssd)
    eng.load_state_dict(_lib.REWARD, rsd)
    eng.finalize()
    eng.set_class_bank(tokens, args.n_ctx, ctx0, _lib.TEXT_SHARED)
    cfg = TTAConfig(selection_p=args.selection_p, tta_steps=args.tta_steps, sample_k=args.sample_k, lr=args.lr, weight_decay=args.weight_decay)

    lo, hi = shard.shard_range(args.total_images, rank, world)
    targets = synthetic_targets(args.total_images, args.classes)
    top5 = torch.empty(hi - lo, 5, dtype=torch.int32, device=dev)
    keep = min(max(args.keep_logits, 0), args.total_images)             # final logits of stream samples [0, keep): parity checks at size
    if keep > shard.shard_range(args.total_images, 0, world)[1]:
        raise ValueError("--keep-logits must fit in rank 0's shard")
    kept = torch.empty(min(keep, hi - lo) if rank == 0 else 0, args.classes, device=dev)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for s in range(lo, hi, args.images_per_pass):                   # views are generated pass by pass: a shard need not fit in HBM
        e = min(s + args.images_per_pass, hi)
        views = torch.stack([synth.make_views(shard.sample_seed(args.first_seed, i), args.views, geo.image_resolution, device=dev)
                             for i in range(s, e)])
        if s < kept.shape[0]:
            t5, fl = eng.tta_batch(views, cfg, want_logits=True)
            top5[s - lo: e - lo] = t5
            n_k = min(e, kept.shape[0]) - s
            kept[s: s + n_k] = fl[:n_k]
        else:
            top5[s - lo: e - lo] = eng.tta_batch(views, cfg)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pred = top5.long().cpu()
    tgt = targets[lo:hi].view(-1, 1)
    h1, h5 = int((pred[:, :1] == tgt).any(1).sum()), int((pred == tgt).any(1).sum())
    cdev = dev if (use_dist and args.dist_backend == "nccl") else "cpu"
    acc1, acc5, n = shard.reduce_hits(h1, h5, hi - lo, device=cdev)          # the end-of-dataset reduction: 3 integers
    all_pred = pred
    if use_dist:
        # optional gather of the predictions (blocks differ by at most one row: pad to the largest block)
        q = -(-args.total_images // world)
        buf = torch.full((q, 5), -1, dtype=torch.int64, device=cdev)
        buf[: hi - lo] = pred.to(cdev)
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        all_pred = torch.cat([p[: shard.shard_range(args.total_images, r, world)[1] - shard.shard_range(args.total_images, r, world)[0]]
                              for r, p in enumerate(parts)]).cpu()
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        per_rank = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(per_rank, t)
        rank_seconds = [float(x.item()) for x in per_rank]
        dt = max(rank_seconds)
    else:
        rank_seconds = [dt]
    out = {"images": n, "acc1": round(acc1, 3), "acc5": round(acc5, 3), "n_gpus": world, "seconds": dt, "images_per_s": n / dt,
           "predictions_sha256": hashlib.sha256(all_pred.numpy().tobytes()).hexdigest(), "top5": all_pred.tolist(),
           "rank_seconds": rank_seconds, "first_seed": args.first_seed,
           "final_logits_first": kept.cpu().tolist() if kept.numel() else None,
           "distributed": ({"backend": args.dist_backend, "ranks": dist.get_world_size(),
                            "collectives": "all_reduce(SUM) of 3 hit counters, all_gather of the top-5 block, all_gather of the shard time"}
                           if use_dist else None),
           "config": {"arch": args.arch, "reward_arch": args.reward_arch, "views": args.views, "classes": args.classes,
                      "tta_steps": args.tta_steps, "images_per_pass": args.images_per_pass, "precision": args.precision,
                      "sharding": f"contiguous blocks over {world} rank(s), no data-path collective"}}
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return out if rank == 0 else {}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--total-images", type=int, default=256, help="length of the test stream (BASELINE configs[3]: 256)")
    ap.add_argument("--arch", default="ViT-B/16")
    ap.add_argument("--reward-arch", default="ViT-B/16")
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--classes", type=int, default=1000)
    ap.add_argument("--n-ctx", type=int, default=4)
    ap.add_argument("--selection-p", type=float, default=0.1)
    ap.add_argument("--tta-steps", type=int, default=1)
    ap.add_argument("--sample-k", type=int, default=3)
    ap.add_argument("--lr", type=float, default=7e-3)
    ap.add_argument("--weight-decay", type=float, default=5e-4)
    ap.add_argument("--images-per-pass", type=int, default=32)
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"])
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--first-seed", type=int, default=1000, help="view seed of stream sample 0 (sample i -> first_seed + i; SURVEY section 8d: 1000)")
    ap.add_argument("--keep-logits", type=int, default=0, help="also report the final logits of stream samples [0, K) (must lie in rank 0's shard)")
    ap.add_argument("--out", default="", help="also write the JSON record to this file")
    a = ap.parse_args(argv)
    rc = shard.self_launch(a.gpus, list(argv) if argv is not None else sys.argv[1:], module="rlcf_amd.eval")      # --gpus N without a launcher
    if rc is not None:
        sys.exit(rc)
    rec = run(a)
    if rec:
        line = json.dumps(rec)
        print(line)
        if a.out:
            with open(a.out, "w") as f:
                f.write(line + "\n")


if __name__ == "__main__":
    main()
