#!/usr/bin/env python3
"""What would the SAMPLE ORDER buy?  The step's lookup and table backward walk the marched samples ray by ray (raymarching.cu:417-470 writes
them that way and composite_rays_train needs them that way); both are bound by scattered L2 requests on the fine hashed levels, where
consecutive samples of a ray share no cell.  This times the packed lookup and the pair backward on the same 2^18 marched samples in three
orders: ray order (what the step does), Morton order of the position (10 bits per axis), and Morton order inside blocks of 16 384 samples (what
a sort kept local to a chunk of rays would give).  Measurement only -- nothing in the library sorts samples.
Usage: python tools/order_bench.py [--reps 20]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nerf2mesh_amd import _lib as L, raymarching, synthetic as S
from nerf2mesh_amd.gridencoder import GridEncoder, _host_offsets, binned_backward_pair

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--B", type=int, default=2 ** 18)
args = ap.parse_args()
dev = torch.device("cuda")
p = L.ptr


def marched(B):
    poses = S.make_cameras(100, seed=0).to(dev)
    bits = raymarching.packbits(S.scene_density_grid(H=128, device=dev), 10.0)
    g = torch.Generator(device=dev).manual_seed(0)
    xs, n = [], 0
    while n < B:
        o, d = S.random_rays(poses, 65536, g)
        nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.05)
        xyzs, _, _, _ = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, 0.0, 1024)
        xs.append(xyzs); n += xyzs.shape[0]
    return torch.cat(xs)[:B].contiguous()          # in [-1, 1]


def morton_key(x01):
    q = (x01.clamp(0, 1) * 1023).to(torch.int64)

    def spread(v):
        v = (v | (v << 16)) & 0xFF0000FF
        v = (v | (v << 8)) & 0x0F00F00F
        v = (v | (v << 4)) & 0xC30C30C3
        v = (v | (v << 2)) & 0x49249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


B = args.B
torch.manual_seed(0)
e1 = GridEncoder(level_dim=1, desired_resolution=2048).to(dev)
e2 = GridEncoder(level_dim=2, desired_resolution=2048).to(dev)
with torch.no_grad():
    e1.embeddings.normal_(0, 0.1); e2.embeddings.normal_(0, 0.1)
rows = e1.embeddings.shape[0]
pk = torch.empty(rows, 2, dtype=torch.float32, device=dev)
pk[:, 0] = e1.embeddings.detach()[:, 0]
pk.view(torch.float16)[:, 2:] = e2.embeddings.detach().half()
ho = _host_offsets(e1)
S_, H0 = float(np.log2(e1.per_level_scale)), int(e1.base_resolution)
x_ray = marched(B)
key = morton_key(x_ray * 0.5 + 0.5)
x_mort = x_ray[torch.argsort(key)].contiguous()
blk = 16384
idx = torch.cat([i + torch.argsort(key[i:i + blk]) for i in range(0, B, blk)])
x_blk = x_ray[idx].contiguous()
g1 = torch.randn(16, B, 1, device=dev) * 1e-3
g2 = (torch.randn(16, B, 2, device=dev) * 1e-3).half()
# like a trained batch: the second half of every run of 11 samples carries no gradient (48 % dead in the step, profiles/r05_fill_stats.txt) -- in RAY order;
# the same per-sample gradients follow their samples into the other orders
dead = (torch.arange(B, device=dev) % 11) >= 6
g1[:, dead] = 0; g2[:, dead] = 0
perm_m, perm_b = torch.argsort(key), idx
t1 = torch.zeros_like(e1.embeddings)
t2 = torch.zeros(e2.embeddings.shape, device=dev, dtype=torch.float16)
emb = e1.embeddings.detach()
h1, h2 = torch.empty(16, B, device=dev), torch.empty(16, B, 2, device=dev, dtype=torch.float16)


def timed(fn):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / args.reps


def lookup(x):
    L.call("n2m_grid_encode_forward_packed", p(x), p(pk), p(e1.offsets), p(h1), p(h2), B, 16, 16, S_, H0, e1.gridtype_id, int(bool(e1.align_corners)),
           e1.interp_id, 0.5, 0.5, L.stream())


def distinct_cells(x, level):
    scale = float(np.float32(np.exp2(np.float32(level) * np.float32(S_)) * np.float32(H0)) - np.float32(1.0))
    c = torch.floor((x * 0.5 + 0.5) * scale + 0.5).to(torch.int64)
    k = (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]
    heads = torch.ones(B, dtype=torch.bool, device=dev)
    heads[1:] = k[1:] != k[:-1]
    return float(heads.float().mean()), int(torch.unique(k).numel())


print(f"B = {B} marched samples of the synthetic scene; times in us per call (torch events over {args.reps} calls)")
for name, x, pm in (("ray order (the step)", x_ray, None), ("Morton order", x_mort, perm_m), (f"Morton order inside blocks of {blk}", x_blk, perm_b)):
    a1 = g1 if pm is None else g1[:, pm].contiguous()
    a2 = g2 if pm is None else g2[:, pm].contiguous()
    x01 = (x * 0.5 + 0.5).contiguous()
    tl = timed(lambda: lookup(x))
    tb = timed(lambda: binned_backward_pair(e1, e2, a1, a2, x01, t1, t2, 16, tv=(emb, 1e-8, 1e-8, 1.0, None), overwrite=True))
    tn = timed(lambda: binned_backward_pair(e1, e2, a1, a2, x01, t1, t2, 16, tv=None, overwrite=True))
    L.call("n2m_grid_backward_merge_levels", 16)       # same-cell runs merged on every level (the step merges levels 0..8: finer ones do not repeat along a ray)
    try:
        tm = timed(lambda: binned_backward_pair(e1, e2, a1, a2, x01, t1, t2, 16, tv=None, overwrite=True))
    finally:
        L.call("n2m_grid_backward_merge_levels", 0)
    heads = "  ".join(f"L{l}: {distinct_cells(x, l)[0]:.2f}" for l in (4, 8, 10, 12, 15))
    print(f"{name:38s} lookup {tl:7.1f}   backward + TV {tb:7.1f}   backward {tn:7.1f}   backward, runs merged on all levels {tm:7.1f}   | run heads / samples  {heads}")
print("distinct cells / samples: " + "  ".join(f"L{l}: {distinct_cells(x_ray, l)[1] / B:.2f}" for l in (4, 8, 10, 12, 15)))
