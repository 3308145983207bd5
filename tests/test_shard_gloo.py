"""CPU, world_size 2 over gloo: the N>1 path of bench.py / the eval harness (sharding + final reduction)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rlcf_amd import shard


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 256, 1000):
        for w in (1, 2, 3, 8):
            blocks = [shard.shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(11, rank, world)
    # stand-in for the per-sample step: sample i is "correct@1" iff i % 3 == 0, "correct@5" iff i % 2 == 0
    seeds = [shard.sample_seed(1000, i) for i in range(lo, hi)]
    h1 = sum(1 for i in range(lo, hi) if i % 3 == 0)
    h5 = sum(1 for i in range(lo, hi) if i % 2 == 0)
    a1, a5, n = shard.reduce_hits(h1, h5, hi - lo)
    dist.barrier()
    q.put((rank, seeds, a1, a5, n))
    dist.destroy_process_group()


def test_two_ranks_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    seeds = res[0][1] + res[1][1]
    assert seeds == [1000 + i for i in range(11)]                     # every sample exactly once, placement-independent seeds
    for _, _, a1, a5, n in res:                                       # both ranks hold the same reduced result
        assert n == 11 and abs(a1 - 100.0 * 4 / 11) < 1e-9 and abs(a5 - 100.0 * 6 / 11) < 1e-9


def test_self_launch_becomes_n_ranks(tmp_path):
    """`python <script> --gpus 2` typed without a launcher (no WORLD_SIZE): shard.self_launch starts the same command line as two ranks
    under torch.distributed.run on 127.0.0.1 (what bench.py / rlcf_amd.eval do first thing); inside a rank it is a no-op."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "probe.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "from rlcf_amd import shard\n"
        "rc = shard.self_launch(2, sys.argv[1:])\n"
        "if rc is not None:\n"
        "    print('parent rc', rc); sys.exit(rc)\n"
        "import torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        # one write() per rank: the ranks share the launcher's pipe and print()'s per-argument writes interleave
        "sys.stdout.write('rank %s of %s args %s master %s\\n' % (os.environ['RANK'], os.environ['WORLD_SIZE'], sys.argv[1:], os.environ['MASTER_ADDR'])); sys.stdout.flush()\n"
        "dist.barrier(); dist.destroy_process_group()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(script), "--gpus", "2", "--steps", "3"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    assert "rank 0 of 2 args ['--gpus', '2', '--steps', '3'] master 127.0.0.1" in out and "rank 1 of 2" in out and "parent rc 0" in out
    # a process that already is a rank (or N == 1) just runs
    assert shard.self_launch(1, []) is None
    os.environ["WORLD_SIZE"] = "2"
    try:
        assert shard.self_launch(2, []) is None
    finally:
        del os.environ["WORLD_SIZE"]
