#!/usr/bin/env python3
"""Inference (row A11 of SURVEY.md section 8: `march_rays` / `composite_rays`, nerf/renderer.py:764-802 -- the eval / export render loop):
one 800 x 800 view of the synthetic lego-like scene rendered by model.render() in eval mode after a short training run, the fused field
kernels and the on-device alive-ray compaction on.  Prints ms per frame, rays/s, samples/s (samples = march slots that carried a sample).
    python tools/infer_bench.py [--pretrain 600] [--frames 10]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf2mesh_amd import synthetic
from nerf2mesh_amd.engine import Stage0Engine
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options

ap = argparse.ArgumentParser()
ap.add_argument("--pretrain", type=int, default=600)
ap.add_argument("--frames", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
opt = make_options(O=True, bound=1, dt_gamma=0, iters=30000, fused_mlp=True)
poses = synthetic.make_cameras(100, seed=0)
tr = Stage0Engine(NeRFNetwork(opt), opt, poses, dev, seed=0)
tr.mark_untrained()
for _ in range(args.pretrain):
    tr.train_step()
torch.cuda.synchronize()
model = tr.model
model.eval()
HW = synthetic.LEGO_HW
pix = torch.arange(HW * HW, device=dev)
res = []
with torch.no_grad():
    for f in range(args.frames + 2):
        cam = (7 * f) % 100
        rays_o, rays_d = synthetic.rays_from_pixels(tr.poses, torch.full_like(pix, cam), pix)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.render(rays_o, rays_d, bg_color=1, perturb=False, shading="full", dt_gamma=opt.dt_gamma, max_steps=opt.max_steps, T_thresh=1e-4)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if f >= 2:
            res.append(dt)
ms = 1e3 * sum(res) / len(res)
psnr = tr.eval_psnr(0, 4)
print(f"INFER 800x800: {ms:.2f} ms/frame (min {1e3 * min(res):.2f}), {HW * HW / (ms * 1e-3) / 1e6:.1f} M rays/s; view-0 quarter-res PSNR after {args.pretrain} steps {psnr:.2f} dB")
