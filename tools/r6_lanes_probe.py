"""Round 6: where does the in-flight loop's time go?  K engines (ViT-B/16 + ViT-B/16, N = 64, C = 1000), samples submitted round robin through
rlcf_lanes_submit from ONE thread: host seconds per submit (the call returns without waiting), images/s for K = 1 .. 5, and the same with
the host throttled to at most D samples queued per lane.  args: [weights fp32|fp16grid]"""
import os
import sys
import time
from collections import deque

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L, synth  # noqa: E402
from rlcf_amd.engine import Engine, Lanes, TTAConfig  # noqa: E402

dev = torch.device("cuda:0")
grid = len(sys.argv) > 1 and sys.argv[1] == "fp16grid"
geo = synth.GEOMETRIES["ViT-B/16"]
ssd, rsd = synth.make_state_dict(geo, 11, device=dev), synth.make_state_dict(geo, 23, device=dev)
if grid:
    ssd, rsd = synth.to_fp16_grid(ssd), synth.to_fp16_grid(rsd)
tokens = synth.make_token_bank(geo, 1000, seed=7, n_ctx=4)
ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(geo, 4), device=dev)].clone()
cfg = TTAConfig(selection_p=0.1, lr=7e-3, weight_decay=5e-4)
KMAX = 5
engs = []
for _ in range(KMAX):
    e = Engine(geo, geo, 64, 1000, L.PREC_F16X3)
    e.load_state_dict(L.STUDENT, ssd); e.load_state_dict(L.REWARD, rsd); e.finalize()
    e.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
    engs.append(e)
views = [synth.make_views(1000 + i, 64, 224, device=dev) for i in range(8)]
n = 96
top5 = torch.empty(n, 5, dtype=torch.int32, device=dev)
for K in (1, 2, 3, 4, 5):
    for depth in (4, 2):
        ln = Lanes(engs[:K])
        for i in range(2 * K):                                 # warm-up: workspaces of every lane
            ln.submit(views[i % 8], cfg, top5[i])
        ln.join(); torch.cuda.synchronize()
        q, sub = deque(), []
        t0 = time.perf_counter()
        for i in range(n):
            t1 = time.perf_counter()
            k = ln.submit(views[i % 8], cfg, top5[i])
            sub.append(time.perf_counter() - t1)
            ev = torch.cuda.Event(); ev.record(ln.streams[k]); q.append(ev)
            if len(q) > depth * K:
                q.popleft().synchronize()
        ln.join(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sub.sort()
        print(f"[lanes probe{' grid' if grid else ''}] K={K} queue depth {depth}/lane: {n / dt:6.1f} images/s; host per submit: median {sub[len(sub) // 2] * 1e3:.2f} ms  "
              f"p90 {sub[int(0.9 * len(sub))] * 1e3:.2f}  max {sub[-1] * 1e3:.2f}  sum {sum(sub) * 1e3:.0f} ms of {dt * 1e3:.0f}", flush=True)
        ln.close()
