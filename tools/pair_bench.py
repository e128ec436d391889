#!/usr/bin/env python3
"""Micro-benchmark of the shared binned backward of both field tables (n2m_grid_encode_backward_binned_pair) on coherent inputs:
samples of march_rays_train on the synthetic scene, B = 2^18, lego tables.  Prints the torch-event time of the whole call with and
without the folded TV term; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
Usage: python tools/pair_bench.py [--reps 20] [--levels]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf2mesh_amd import raymarching, synthetic as S
from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward_pair

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--levels", action="store_true")
ap.add_argument("--B", type=int, default=2 ** 18)
ap.add_argument("--uniform", action="store_true", help="incoherent inputs U[0,1)^3 instead of ray samples")
ap.add_argument("--add", action="store_true", help="add onto the tables (overwrite = 0) instead of the training step's overwrite mode")
args = ap.parse_args()
dev = torch.device("cuda")


def coherent_samples(B):
    poses = S.make_cameras(100, seed=0).to(dev)
    bits = raymarching.packbits(S.scene_density_grid(H=128, device=dev), 10.0)
    g = torch.Generator(device=dev).manual_seed(0)
    xs, n = [], 0
    while n < B:
        o, d = S.random_rays(poses, 65536, g)
        nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.05)
        xyzs, _, _, _ = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, 0.0, 1024)
        xs.append(xyzs); n += xyzs.shape[0]
    return ((torch.cat(xs)[:B] + 1) / 2).contiguous()


B = args.B
torch.manual_seed(0)
e1 = GridEncoder(level_dim=1, desired_resolution=2048).to(dev)
e2 = GridEncoder(level_dim=2, desired_resolution=2048).to(dev)
x = torch.rand(B, 3, device=dev) if args.uniform else coherent_samples(B)
g1 = torch.randn(16, B, 1, device=dev) * 1e-3
g2 = (torch.randn(16, B, 2, device=dev) * 1e-3).half()
t1 = torch.zeros_like(e1.embeddings)
t2 = torch.zeros(e2.embeddings.shape, device=dev, dtype=torch.float16)
emb = e1.embeddings.detach()


def run(tv, ml=16):
    tvp = (emb, 1e-8, 1e-8, 1.0, None) if tv else None
    for _ in range(3):
        assert binned_backward_pair(e1, e2, g1, g2, x, t1, t2, ml, tv=tvp, overwrite=not args.add)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.reps):
        binned_backward_pair(e1, e2, g1, g2, x, t1, t2, ml, tv=tvp, overwrite=not args.add)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / args.reps


if os.environ.get("FILL_MODE"):
    import ctypes
    from nerf2mesh_amd import _lib as L0
    L0.call("n2m_debug_fill_times", int(os.environ["FILL_MODE"]), None)
print(f"pair backward B=2^18 coherent, TV folded : {run(True):8.1f} us")
print(f"pair backward B=2^18 coherent, no TV     : {run(False):8.1f} us")
if args.levels:
    for ml in (1, 2, 4, 6, 8, 12, 16):
        print(f"   max_level={ml:2d} (no TV): {run(False, ml):8.1f} us")


# phase stamps of one fill workgroup (shader clock): top -> entries -> barrier 1 -> barrier 3 -> barrier 4 -> stores issued
import ctypes
from nerf2mesh_amd import _lib as L
L.call("n2m_debug_fill_times", 1 | int(os.environ.get("FILL_MODE", "0")), None)
run(True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 116)()
L.call("n2m_debug_fill_times", 0, ctypes.addressof(buf))
names = ["entries(+TV)", "slot atomics+bar1", "scan+bar2+bar3", "stage+bar4", "log stores"]
for w, what in enumerate(("workgroup 3 (fine hashed level)", "workgroup grid/2+3 (coarse dense level)")):
    t = [[buf[w * 48 + i * 6 + j] for j in range(6)] for i in range(8)]
    print(what)
    for i in range(8):
        if t[i][0] == 0: continue
        d = [t[i][j + 1] - t[i][j] for j in range(5)]
        nxt = (t[i + 1][0] - t[i][5]) if i < 7 and t[i + 1][0] else 0
        print(f"  iter {i}: " + "  ".join(f"{n} {v}" for n, v in zip(names, d)) + f"  | total {t[i][5]-t[i][0]} (+{nxt} to next top) cycles")
for k, what in enumerate(("accumulate fp32 C=1", "accumulate fp16 C=2")):
    for w in range(2):
        t = [buf[96 + k * 10 + w * 5 + j] for j in range(5)]
        if t[0]:
            print(f"{what} item {w}: clear+directory {t[1]-t[0]}  walk {t[2]-t[1]}  barrier {t[3]-t[2]}  flush {t[4]-t[3]}  | total {t[4]-t[0]} cycles")
