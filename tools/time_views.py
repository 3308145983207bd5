"""Timing of device-side view generation: one 375x500 uint8 image -> 64 normalised 224x224 views (rlcf_make_views)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import make_views_golden as G
from rlcf_amd import datautils as D
img = torch.from_numpy(G.synth_image("views_imagenet", 375, 500)).cuda()
aug = D.AugMixAugmenter(None, None, n_views=63)
torch.manual_seed(0)
crops = [aug.preaugment(375, 500) for _ in range(63)]
for _ in range(3): D.make_views(img, crops)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): D.make_views(img, crops)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
t1 = time.perf_counter()
for _ in range(50): [aug.preaugment(375, 500) for _ in range(63)]
dp = (time.perf_counter() - t1) / 50
print(f"make_views 64 x 224^2 from 375x500: {dt*1e6:.0f} us per image (GPU, incl. launch + scratch alloc); crop sampling on host {dp*1e6:.0f} us; "
      f"output {64*3*224*224*4/1e6:.1f} MB -> {64*3*224*224*4/dt/1e9:.0f} GB/s written; upload {375*500*3/1e3:.0f} KB instead of 38.5 MB")
# the same with the AugMix op chains of the fine-grained sets (rlcf_make_views_augmix)
import numpy as np
np.random.seed(0)
plans = [D.draw_augmix_plan(1) for _ in crops]
for _ in range(3): D.make_views(img, crops, augmix_plans=plans)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): D.make_views(img, crops, augmix_plans=plans)
torch.cuda.synchronize(); da = (time.perf_counter() - t0) / 50
t1 = time.perf_counter()
for _ in range(50): [D.draw_augmix_plan(1) for _ in crops]
dq = (time.perf_counter() - t1) / 50
print(f"make_views + AugMix chains (63 views x 3 chains x <=3 ops): {da*1e6:.0f} us per image; plan sampling on host {dq*1e6:.0f} us")
# the hard_aug recipe (rlcf_make_views_hard): ColorJitter / RandomGrayscale / GaussianBlur between the resized crop and the flip
torch.manual_seed(0)
hp = D.HardAugParams()
drawn = [hp(375, 500) for _ in range(63)]
hcrops, hplans = [d[0] for d in drawn], [d[1] for d in drawn]
for _ in range(3): D.make_views(img, hcrops, hard_plans=hplans)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): D.make_views(img, hcrops, hard_plans=hplans)
torch.cuda.synchronize(); dh = (time.perf_counter() - t0) / 50
t1 = time.perf_counter()
for _ in range(50): [hp(375, 500) for _ in range(63)]
dhq = (time.perf_counter() - t1) / 50
nj, ng, nb = sum(p[0] is not None for p in hplans), sum(bool(p[5]) for p in hplans), sum(p[6] is not None for p in hplans)
print(f"make_views + hard_aug (63 views: {nj} colour-jittered, {ng} grayscale, {nb} blurred): {dh*1e6:.0f} us per image; parameter sampling on host {dhq*1e6:.0f} us")
for _ in range(3): D.make_views(img, hcrops, hard_plans=hplans, augmix_plans=plans)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): D.make_views(img, hcrops, hard_plans=hplans, augmix_plans=plans)
torch.cuda.synchronize(); dha = (time.perf_counter() - t0) / 50
print(f"make_views + hard_aug + AugMix chains: {dha*1e6:.0f} us per image")
