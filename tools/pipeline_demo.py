#!/usr/bin/env python3
"""The whole nerf2mesh flow inside this package, on the synthetic lego-like scene:
   stage 0 (step executor) -> export_stage0 (density volume -> device marching cubes -> mesh_0.ply)
   -> stage 1 (rasterise / shade / antialias refinement of colours and vertex offsets on that mesh)
   -> export_stage1 (texture bake + mesh_0.obj/.mtl + feat{0,1}_0.jpg + mlp.json).
What the reference does with PyMCubes / pymeshlab / xatlas / cv2 in between and is NOT done here: mesh cleaning, decimation, UV
unwrapping (the bake uses the per-face grid atlas).  Prints one line per phase with its wall time."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf2mesh_amd import export, synthetic as S
from nerf2mesh_amd.engine import Stage0Engine
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options
from nerf2mesh_amd.trainer import Stage1Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/pipeline")
ap.add_argument("--iters0", type=int, default=3000)
ap.add_argument("--iters1", type=int, default=300)
ap.add_argument("--resolution", type=int, default=256)
ap.add_argument("--texture", type=int, default=1024)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)


def clock(label, t0, extra=""):
    torch.cuda.synchronize()
    print(f"[{label}] {time.perf_counter() - t0:7.3f} s  {extra}", flush=True)


opt = make_options(O=True, bound=1, dt_gamma=0, iters=args.iters0, fused_mlp=True)
poses = S.make_cameras(100, seed=0)
eng = Stage0Engine(NeRFNetwork(opt), opt, poses, dev, seed=0)
eng.mark_untrained()
eng.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.iters0 - 1):
    eng.train_step()
clock("stage 0", t0, f"{args.iters0} steps, PSNR(view 0, quarter res) {eng.eval_psnr():.2f} dB")

model = eng.model
t0 = time.perf_counter()
meshes = model.export_stage0(os.path.join(args.out, "mesh_stage0"), resolution=args.resolution)
v, t = meshes[0]
clock("export_stage0", t0, f"{args.resolution}^3 density volume -> {v.shape[0]} vertices, {t.shape[0]} triangles (raw iso-surface)")

rv, rt = export.read_ply(os.path.join(args.out, "mesh_stage0", "mesh_0.ply"))
opt1 = opt
opt1.stage, opt1.iters = 1, max(args.iters1, 501)
tr = Stage1Trainer(model, opt1, poses, torch.from_numpy(rv), torch.from_numpy(rt), dev)
tr.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
losses = [float(tr.train_step().detach()) for _ in range(args.iters1 - 1)]
clock("stage 1", t0, f"{args.iters1} steps (one 800x800 view each), loss {losses[0]:.5f} -> {sum(losses[-20:]) / 20:.5f}")

t0 = time.perf_counter()
out = model.export_stage1(os.path.join(args.out, "mesh_stage1"), h0=args.texture, w0=args.texture)
clock("export_stage1", t0, f"{args.texture}^2 atlas (ssaa {opt1.ssaa}), {int(out[0][2].sum())} covered texels; files: "
      + ", ".join(sorted(os.listdir(os.path.join(args.out, 'mesh_stage1')))))
