"""Sample sharding across the GPUs of one node.  Test images are independent units (per-sample reset,
reference TPT/tpt_cls_rl.py:251-255), so the data path needs no collective: rank r works on its own contiguous block.
The only communication is the end-of-dataset reduction of the hit counters (RCCL all_reduce on GPUs, gloo in tests)."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Tuple

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


def local_device(local_rank: int, backend: str) -> int:
    """The GPU index of local rank `local_rank`.  Under `nccl` (= RCCL) every rank needs a device of its own: more ranks than visible
    GPUs is an error HERE, with a message — stacked on one device they would die inside RCCL with a duplicate-device error.  `gloo`
    (CPU / one-GPU smoke tests) may share a device."""
    n = torch.cuda.device_count()
    if n <= 0:
        raise RuntimeError("no visible GPU (the RLCF HIP path has no CPU fallback)")
    if local_rank >= n and backend == "nccl":
        raise RuntimeError(f"local rank {local_rank} has no GPU of its own: {n} visible device(s) but the launch asks for more ranks "
                           f"(--gpus / --nproc-per-node > visible devices); RCCL needs one GPU per rank — lower --gpus, or use "
                           f"--dist-backend gloo to share a device in a smoke test")
    return local_rank % n


def self_launch(n_gpus: int, script_argv: List[str], module: Optional[str] = None) -> Optional[int]:
    """`python bench.py --gpus N` / `python -m rlcf_amd.eval --gpus N` typed WITHOUT a launcher: when N > 1 and no rank environment is
    present (WORLD_SIZE unset), start the same command line as N ranks — one process per GPU — under `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (the launch the driver itself uses) and return its exit code; return None
    when this process already IS a rank (or N == 1) and should simply run.  The port is a free one of the loopback interface, so two
    such commands can run side by side.  The ranks' stdout / stderr are inherited: rank 0's JSON line is this command's JSON line."""
    if n_gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port)]
    cmd += (["-m", module] if module else [os.path.abspath(sys.argv[0])]) + list(script_argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_gpus)))
    return subprocess.call(cmd, env=env)
