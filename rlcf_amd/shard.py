"""Sample sharding across the GPUs of one node.  Test images are independent units (per-sample reset,
reference TPT/tpt_cls_rl.py:251-255), so the data path needs no collective: rank r works on its own contiguous block.
The only communication is the end-of-dataset reduction of the hit counters (RCCL all_reduce on GPUs, gloo in tests)."""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(n_samples: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one sample and cover [0, n) exactly once."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(n_samples, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def sample_seed(base_seed: int, sample_index: int) -> int:
    """Synthetic inputs are seeded per SAMPLE, so results do not depend on the placement."""
    return base_seed + sample_index


def reduce_hits(top1_hits: int, top5_hits: int, n: int, device="cpu") -> Tuple[float, float, int]:
    """(acc@1 %, acc@5 %, n) over all ranks; a single all_reduce of 3 integers — latency-bound, off the data path."""
    import torch.distributed as dist
    t = torch.tensor([top1_hits, top5_hits, n], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    a, b, c = (int(x) for x in t.tolist())
    return 100.0 * a / max(c, 1), 100.0 * b / max(c, 1), c
