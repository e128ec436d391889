#!/usr/bin/env python3
"""Convergence check on the synthetic scene: PSNR of a held-out view while training stage 0, fused vs unfused MLPs.
    python tools/train_check.py [--steps 3000] [--unfused]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf2mesh_amd import synthetic
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options
from nerf2mesh_amd.trainer import Stage0Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--unfused", action="store_true")
ap.add_argument("--every", type=int, default=500)
ap.add_argument("--engine", action="store_true", help="the step executor (engine.Stage0Engine) instead of the autograd trainer")
args = ap.parse_args()
torch.manual_seed(0)
opt = make_options(O=True, bound=1, dt_gamma=0, iters=30000, fused_mlp=not args.unfused)
if args.engine:
    from nerf2mesh_amd.engine import Stage0Engine
    tr = Stage0Engine(NeRFNetwork(opt), opt, synthetic.make_cameras(100, seed=0), torch.device("cuda", 0), seed=0)
else:
    tr = Stage0Trainer(NeRFNetwork(opt), opt, synthetic.make_cameras(100, seed=0), torch.device("cuda"), seed=0)
tr.mark_untrained()
t0 = time.time()
for i in range(1, args.steps + 1):
    tr.train_step()
    if i % args.every == 0:
        torch.cuda.synchronize()
        occ = float((tr.model.density_grid > min(tr.model.mean_density, tr.model.density_thresh)).float().mean())
        print(f"step {i:5d}  {time.time()-t0:6.1f}s  loss(avg) {float(tr.loss_acc)/i:.5f}  rays/step {tr.num_rays:6d}  occupancy {occ:.4f}  "
              f"psnr(view0) {tr.eval_psnr(0):.2f}  psnr(view7) {tr.eval_psnr(7):.2f}", flush=True)
