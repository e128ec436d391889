"""A/B timing of the binned backward on training-like inputs: python tools/bwd_ab.py (uses N2M_HIP_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf2mesh_amd import _lib as L, raymarching, synthetic as S
from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward
dev = torch.device("cuda")
poses = S.make_cameras(100, seed=0).to(dev)
bits = raymarching.packbits(S.scene_density_grid(H=128, device=dev), 10.0)
g = torch.Generator(device=dev).manual_seed(0)
xs = []; n = 0
while n < 2**18:
    o, d = S.random_rays(poses, 65536, g)
    nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.05)
    xyzs, _, _, _ = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, 0.0, 1024)
    xs.append(xyzs); n += xyzs.shape[0]
x = ((torch.cat(xs)[:2**18] + 1) / 2).contiguous()
for C, dt in ((1, torch.float32), (2, torch.float16)):
    enc = GridEncoder(level_dim=C, desired_resolution=2048).to(dev)
    grad = (torch.randn(16, 2**18, C, device=dev) * (torch.rand(1, 2**18, 1, device=dev) < 0.6)).to(dt)   # 40 % zero gradients like training
    gemb = torch.zeros(enc.host_offsets[-1], C, device=dev, dtype=dt)
    for _ in range(3): binned_backward(enc, grad, x, gemb, 16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): binned_backward(enc, grad, x, gemb, 16)
    torch.cuda.synchronize(); print(f"C={C} binned backward {1e6*(time.perf_counter()-t0)/20:8.1f} us/call  lib={os.environ.get('N2M_HIP_LIB','default')[-30:]}")
