"""Times BASELINE configs[2]: ViT-L/14 student + ViT-L/14 reward, N=64 views, LayerNorm tuning (rlcf_tta_sample_ln).
args: ARCH CLASSES IMAGES_PER_PASS STEPS [full|fullonly]   (full: also rlcf_tta_sample_visual, every visual parameter tuned)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L, synth
from rlcf_amd.engine import Engine, TTAConfig
arch = sys.argv[1] if len(sys.argv) > 1 else "ViT-L/14"
n_cls = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda:0")
geo = synth.GEOMETRIES[arch]
rgeo = synth.GEOMETRIES[os.environ.get("REWARD_ARCH", arch)]             # (a ResNet ARCH: BatchNorm tuning, row a-R)
ssd, rsd = synth.make_state_dict(geo, 11, device=dev), synth.make_state_dict(rgeo, 23, device=dev)
if os.environ.get("GRID") == "1":                                          # GEMM weights on the fp16 grid, as a released checkpoint holds them
    ssd, rsd = synth.to_fp16_grid(ssd), synth.to_fp16_grid(rsd)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1            # test images per tower pass (rlcf_tta_batch_ln)
eng = Engine(geo, rgeo, 64 * B, n_cls, L.PREC_F16X3)
eng.load_state_dict(L.STUDENT, ssd); eng.load_state_dict(L.REWARD, rsd); eng.finalize()
tokens = synth.make_token_bank(geo, n_cls, seed=7, n_ctx=4)
ctx0 = ssd["token_embedding.weight"][torch.tensor(synth.ctx_token_ids_default(geo, 4), device=dev)].clone()
eng.set_class_bank(tokens, 4, ctx0, L.TEXT_SHARED)
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cfg = TTAConfig(selection_p=0.1, tta_steps=steps, sample_k=3, lr=1e-5, weight_decay=5e-4)
views = [synth.make_views(1000 + i, 64, geo.image_resolution, device=dev) for i in range(6)]
mode = sys.argv[5] if len(sys.argv) > 5 else ""
ln_runs = 1 if mode == "fullonly" else 4                  # (fullonly: profile runs, keep the LayerNorm path out of the kernel table)
for v in views[:2 if ln_runs > 1 else 1]: o = eng.tta_sample_ln(v, cfg)
torch.cuda.synchronize(); t0 = time.perf_counter()
for v in views[2:2 + ln_runs]: o = eng.tta_sample_ln(v, cfg)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / ln_runs
if B > 1:
    vs = torch.stack([synth.make_views(1000 + i, 64, geo.image_resolution, device=dev) for i in range(2 * B)])
    eng.tta_batch_ln(vs[:B], cfg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    top5 = eng.tta_batch_ln(vs, cfg)
    torch.cuda.synchronize(); dtb = (time.perf_counter() - t0) / (2 * B)
    print(f"{arch} LN-tuning, {steps} step(s), {B} images per pass: {dtb*1e3:.1f} ms/image ({1/dtb:.1f} images/s), flops_exec/image={eng.last_flops()/1e12:.2f} TF")
if mode in ("full", "fullonly"):          # every visual parameter tuned (CLIPCLS_TTA only_norm=False, scripts/rlcf-tune.sh)
    for v in views[:2]: of = eng.tta_sample_visual(v, cfg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for v in views[2:]: of = eng.tta_sample_visual(v, cfg)
    torch.cuda.synchronize(); dtf = (time.perf_counter() - t0) / 4
    print(f"{arch} full image-encoder tuning, {steps} step(s): {dtf*1e3:.1f} ms/image ({1/dtf:.1f} images/s), flops_exec/image={eng.last_flops()/1e12:.2f} TF, "
          f"top5={of['top5'].tolist()} |vis_grad|={of['vis_grad'].norm().item():.3e} nan={bool(torch.isnan(of['final_logits']).any())}")
print(f"{arch} LN-tuning: {dt*1e3:.1f} ms/image ({1/dt:.1f} images/s), flops_exec/image={eng.last_flops()/1e12:.2f} TF, "
      f"top5={o['top5'].tolist()} |ln_grad|={o['ln_grad'].norm().item():.3e} nan={bool(torch.isnan(o['final_logits']).any())}")
