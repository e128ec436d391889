#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE'S OWN kernels (oracle/_ref: the .cu sources under /root/reference
compiled for the host by oracle/build_ref.py).  Runs only in the build container (needs /root/reference); the GPU box
and CI read the committed fixtures.  Inputs are seeded; every array the kernels read or write is stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref, oracle as orc   # noqa: E402  (oracle is used only for host-side level offsets / packbits inputs)
from nerf2mesh_amd import synthetic as S      # noqa: E402

assert build_ref.build(verbose=False), "needs /root/reference"
rm, ge, sh = build_ref.load()


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print(name, {k: v.shape for k, v in arrays.items()})


# ------------------------------------------------------------------------------------------------ freqencoder
fq = build_ref.load_freq()
rng_f = np.random.default_rng(2025)          # own stream: this section can be regenerated alone (--only-freq)
xf = (rng_f.normal(size=(64, 3)) * 2).astype(np.float32)
arr = {"inputs": xf}
for deg in (1, 4, 6):
    C = 3 + 2 * deg * 3
    out = np.zeros((64, C), np.float32)
    fq.freq_encode_forward(t(xf), 64, 3, deg, C, t(out))
    g = rng_f.normal(size=(64, C)).astype(np.float32)
    gi = np.zeros((64, 3), np.float32)
    fq.freq_encode_backward(t(g), t(out), 64, 3, deg, C, t(gi))
    arr[f"out{deg}"], arr[f"grad{deg}"], arr[f"grad_inputs{deg}"] = out, g, gi
save("freq", **arr)
if "--only-freq" in sys.argv:
    sys.exit(0)

rng = np.random.default_rng(2024)

# ---------------------------------------------------------------------------------------------- raymarching
poses = S.make_cameras(8, seed=7)
H = 64                                                      # small grid keeps the fixture tiny; same code path as 128
grid = S.scene_density_grid(H=H, cascade=1, bound=1.0).numpy()
bits = np.zeros(H ** 3 // 8, np.uint8)
rm.packbits(t(grid), H ** 3 // 8, 10.0, t(bits))
N = 600
o, d = S.random_rays(poses, N, torch.Generator().manual_seed(5))
o, d = o.numpy(), d.numpy()
d[:4, 0] = 0.0                                              # axis-parallel rays
aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
nears, fars = np.zeros(N, np.float32), np.zeros(N, np.float32)
rm.near_far_from_aabb(t(o), t(d), t(aabb), N, 0.05, t(nears), t(fars))
coords = rng.integers(0, 1024, (64, 3)).astype(np.int32)
mort = np.zeros(64, np.int32)
rm.morton3D(t(coords), 64, t(mort))
save("utils", grid=grid, thresh=np.float32(10.0), bits=bits, rays_o=o, rays_d=d, aabb=aabb, min_near=np.float32(0.05), nears=nears,
     fars=fars, coords=coords, morton=mort)

for name, cfg in {"march_lego": dict(bound=1.0, contract=False, dt_gamma=0.0, C=1),
                  "march_gamma": dict(bound=1.0, contract=False, dt_gamma=1 / 256, C=1)}.items():
    noises = rng.random(N).astype(np.float32)
    rays = np.zeros((N, 2), np.int32)
    counter = np.zeros(1, np.int32)
    args = (t(o), t(d), t(bits), cfg["bound"], cfg["contract"], cfg["dt_gamma"], 1024, N, cfg["C"], H, t(nears), t(fars))
    rm.march_rays_train(*args, None, None, None, t(rays), t(counter), t(noises))
    M = int(counter[0])
    xyzs, dirs, ts = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    rm.march_rays_train(*args, t(xyzs), t(dirs), t(ts), t(rays), t(counter), t(noises))
    save(name, noises=noises, rays=rays, xyzs=xyzs, ts=ts, dt_gamma=np.float32(cfg["dt_gamma"]), H=np.int32(H))
    if name == "march_lego":
        sig = (rng.random(M) * 60).astype(np.float32)
        rgb = rng.random((M, 3)).astype(np.float32)
        w, ws, dp, im = np.zeros(M, np.float32), np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
        rm.composite_rays_train_forward(t(sig), t(rgb), t(ts), t(rays), M, N, 1e-4, False, t(w), t(ws), t(dp), t(im))
        gw, gws, gd, gi = (rng.normal(size=M).astype(np.float32), rng.normal(size=N).astype(np.float32),
                           rng.normal(size=N).astype(np.float32), rng.normal(size=(N, 3)).astype(np.float32))
        gs, gr = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
        rm.composite_rays_train_backward(t(gw), t(gws), t(gd), t(gi), t(sig), t(rgb), t(ts), t(rays), t(ws), t(dp), t(im), M, N, 1e-4,
                                         False, t(gs), t(gr))
        save("composite", sigmas=sig, rgbs=rgb, weights=w, weights_sum=ws, depth=dp, image=im, grad_weights=gw, grad_weights_sum=gws,
             grad_depth=gd, grad_image=gi, grad_sigmas=gs, grad_rgbs=gr)

# ---------------------------------------------------------------------------------------------- gridencoder
# small tables (log2_hashmap_size 12) exercise dense levels, hashed levels and the dense->hash switch
for name, (C, half) in {"grid_c1_f32": (1, False), "grid_c2_f16": (2, True)}.items():
    L, Hb, pls = 8, 4, 1.7
    offs = orc.level_offsets(3, L, pls, Hb, 12)
    S_ = float(np.log2(pls))
    emb = ((rng.random((int(offs[-1]), C), dtype=np.float32) * 2 - 1) * 0.5)
    emb = emb.astype(np.float16) if half else emb
    B = 257
    x = rng.random((B, 3), dtype=np.float32)
    x[0] = 0; x[1] = 1; x[2, 0] = -0.01
    out = np.zeros((L, B, C), emb.dtype)
    dy = np.zeros((B, L * 3 * C), emb.dtype)
    ge.grid_encode_forward(t(x), t(emb), t(offs), t(out), B, 3, C, L, L, S_, Hb, t(dy), 0, False, 0)
    g = (rng.normal(size=(L, B, C)) * (8 if half else 1)).astype(emb.dtype)
    gemb = np.zeros_like(emb)
    gin = np.zeros((B, 3), emb.dtype)
    ge.grid_encode_backward(t(g), t(x), t(emb), t(offs), t(gemb), B, 3, C, L, L, S_, Hb, t(dy), t(gin), 0, False, 0)
    extra = {}
    if not half:
        gtv = (rng.normal(size=emb.shape) * 1e-3).astype(np.float32)
        extra["tv_grad_in"] = gtv.copy()
        ge.grad_total_variation(t(x), t(emb), t(gtv), t(offs), 1e-2, B, 3, C, L, S_, Hb, 0, False)
        extra["tv_grad_out"] = gtv
    save(name, inputs=x, embeddings=emb, offsets=offs, S=np.float32(S_), H=np.int32(Hb), outputs=out, dy_dx=dy, grad=g,
         grad_embeddings=gemb, grad_inputs=gin, **extra)

# ------------------------------------------------------------------------------------------------ shencoder
v = rng.normal(size=(64, 3)).astype(np.float32)
v /= np.linalg.norm(v, axis=1, keepdims=True)
arr = {"inputs": v}
for deg in (1, 4, 8):
    out = np.zeros((64, deg * deg), np.float32)
    dy = np.zeros((64, 3 * deg * deg), np.float32)
    sh.sh_encode_forward(t(v), t(out), 64, 3, deg, t(dy))
    arr[f"out{deg}"] = out
    arr[f"dy{deg}"] = dy
save("sh", **arr)
