#!/usr/bin/env python3
"""Micro-benchmarks of the hash-grid kernels: grid_encode forward / backward / TV on
 (a) incoherent inputs U[0,1)^3 and (b) coherent inputs = samples of march_rays_train on the synthetic scene,
with B in {2^18, 2^21}, lego tables (L=16, 2^19 rows/level), C=1 fp32 and C=2 fp16.  Timings come from the library's
hipEvent brackets (n2m_prof_*).  Usage: python tools/grid_bench.py [--levels]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nerf2mesh_amd import _lib as L, raymarching, synthetic as S
from nerf2mesh_amd.gridencoder import GridEncoder

ap = argparse.ArgumentParser()
ap.add_argument("--levels", action="store_true", help="time the backward per max_level (cumulative)")
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda")
p = L.ptr


def timed(name, fn, reps=args.reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    L.prof_reset(); L.prof_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    L.prof_enable(False)
    n, ms, by = L.prof_read(name)
    return 1e3 * ms / n, (by / n) / (ms / n * 1e-3) / 1e9


def coherent_samples(B):
    poses = S.make_cameras(100, seed=0).to(dev)
    bits = raymarching.packbits(S.scene_density_grid(H=128, device=dev), 10.0)
    g = torch.Generator(device=dev).manual_seed(0)
    xs = []
    n = 0
    while n < B:
        o, d = S.random_rays(poses, 65536, g)
        nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.05)
        xyzs, _, _, _ = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, 0.0, 1024)
        xs.append(xyzs); n += xyzs.shape[0]
    return ((torch.cat(xs)[:B] + 1) / 2).contiguous()


for C, dt in ((1, torch.float32), (2, torch.float16)):
    enc = GridEncoder(level_dim=C, desired_resolution=2048).to(dev)
    emb = enc.embeddings.detach().to(dt).contiguous()
    S_ = float(np.log2(enc.per_level_scale))
    dtid = L.F16 if dt == torch.float16 else L.F32
    for B in (2 ** 18, 2 ** 21):
        for kind in ("uniform", "coherent"):
            x = torch.rand(B, 3, device=dev) if kind == "uniform" else coherent_samples(B)
            out = torch.empty(B, 16 * C, device=dev, dtype=dt)
            us, gbs = timed("grid_encode_forward", lambda: L.call("n2m_grid_encode_forward_bm", p(x), p(emb), p(enc.offsets), p(out), B, 3, C, 16,
                                                                   16, S_, 16, 0, 0, 0, dtid, L.stream()))
            print(f"C={C} {str(dt)[6:]:8s} B=2^{int(np.log2(B))} {kind:9s} forward(bm)  {us:9.1f} us  {gbs:7.0f} GB/s algorithmic = {100*gbs/8000:5.1f}% of 8 TB/s")
            out_lm = torch.empty(16, B, C, device=dev, dtype=dt)
            us, gbs = timed("grid_encode_forward", lambda: L.call("n2m_grid_encode_forward", p(x), p(emb), p(enc.offsets), p(out_lm), B, 3, C, 16,
                                                                   16, S_, 16, None, 0, 0, 0, dtid, L.stream()))
            print(f"C={C} {str(dt)[6:]:8s} B=2^{int(np.log2(B))} {kind:9s} forward(lm)  {us:9.1f} us  {gbs:7.0f} GB/s algorithmic = {100*gbs/8000:5.1f}% of 8 TB/s")
            if B > 2 ** 18:
                continue
            grad = torch.randn(16, B, C, device=dev).to(dt)
            gemb = torch.zeros_like(emb)
            levels = (1, 3, 5, 8, 16) if args.levels else (16,)
            for ml in levels:
                us, gbs = timed("grid_encode_backward", lambda: L.call("n2m_grid_encode_backward", p(grad), p(x), p(emb), p(enc.offsets), p(gemb), B, 3,
                                                                        C, 16, ml, S_, 16, None, None, 0, 0, 0, dtid, L.stream()))
                print(f"C={C} {str(dt)[6:]:8s} B=2^{int(np.log2(B))} {kind:9s} backward max_level={ml:2d} {us:9.1f} us  {gbs:7.0f} GB/s algorithmic")
            from nerf2mesh_amd.gridencoder import binned_backward, binned_tv
            for ml in levels:
                us, gbs = timed("grid_encode_backward", lambda: binned_backward(enc, grad, x, gemb, ml))
                print(f"C={C} {str(dt)[6:]:8s} B=2^{int(np.log2(B))} {kind:9s} backward BINNED max_level={ml:2d} {us:9.1f} us  {gbs:7.0f} GB/s algorithmic")
            if C == 1:
                us, gbs = timed("grad_total_variation", lambda: binned_tv(enc, x, emb, gemb, 1e-8))
                print(f"C={C} {str(dt)[6:]:8s} B=2^{int(np.log2(B))} {kind:9s} TV BINNED    {us:9.1f} us  {gbs:7.0f} GB/s algorithmic")
                us, gbs = timed("grad_total_variation", lambda: L.call("n2m_grad_total_variation", p(x), p(emb), p(gemb), p(enc.offsets), 1e-8, B, 3, C,
                                                                        16, S_, 16, 0, 0, dtid, L.stream()))
                print(f"C={C} {str(dt)[6:]:8s} B=2^{int(np.log2(B))} {kind:9s} TV           {us:9.1f} us  {gbs:7.0f} GB/s algorithmic")
