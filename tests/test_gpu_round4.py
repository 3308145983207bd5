"""GPU parity, round-4 additions: BASELINE configs[3] run AS A WORKLOAD at full size (256 ViT-B/16 samples, N = 64, C = 1000) through
the sharded driver on one rank and on two ranks, its first eight samples pinned to the reference-generated stream; bench.py's
multi-rank timing record."""
import json
import os
import subprocess
import sys

import pytest
import torch

from test_gpu_parity import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
