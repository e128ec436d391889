#!/usr/bin/env python3
"""Stand-alone timing of the training marcher's two passes on a lego-like batch (synthetic scene occupancy, ~13k rays -> ~2^18 samples).
A/B: N2M_MARCH_RESOLVE=serial python tools/march_bench.py     (scalar-loop resolution instead of the prefix-maximum form)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf2mesh_amd import _lib as L, raymarching, synthetic as S

dev = torch.device("cuda")
poses = S.make_cameras(100, seed=0).to(dev)
bits = raymarching.packbits(S.scene_density_grid(H=128, device=dev), 10.0)
g = torch.Generator(device=dev).manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 13400
o, d = S.random_rays(poses, N, g)
aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev)
nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.05)
for dt_gamma in (0.0, 1 / 256):
    out = None
    for rep in range(3):
        torch.manual_seed(1)
        out = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, dt_gamma, 1024)
    torch.cuda.synchronize()
    L.prof_reset(); L.prof_enable(True)
    for rep in range(20):
        torch.manual_seed(1)
        out = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, dt_gamma, 1024)
    torch.cuda.synchronize()
    L.prof_enable(False)
    res = {}
    for k in ("march_rays_train_count", "march_rays_train_write"):
        n, ms, by = L.prof_read(k)
        res[k] = 1e3 * ms / max(n, 1)
    M = out[0].shape[0]
    chk = int(out[3].long().sum()) ^ int((out[0].double().sum() * 1e6).long())
    print(f"dt_gamma={dt_gamma:.5f} N={N} M={M} count {res['march_rays_train_count']:.1f} us  write {res['march_rays_train_write']:.1f} us  "
          f"resolve={os.environ.get('N2M_MARCH_RESOLVE', 'parallel')} checksum {chk}")
    # march-once form (count + recorded chunks, replay): whole call incl. its workspace clear, by stream events
    noises = torch.rand(N, device=dev)
    cap = int(M * 1.25)
    ws = torch.empty(int(L.lib().n2m_march_fused_workspace_bytes(N)), dtype=torch.uint8, device=dev)
    fo = raymarching.march_rays_train_fused(o, d, 1.0, False, bits, 1, 128, nears, fars, noises, dt_gamma, 1024, max_points=cap, workspace=ws)
    two = None
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for rep in range(3):
        two = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, dt_gamma, 1024, noises)
    e0.record()
    for rep in range(20):
        raymarching.march_rays_train_fused(o, d, 1.0, False, bits, 1, 128, nears, fars, noises, dt_gamma, 1024, max_points=cap, out=fo[:3],
                                           rays=fo[3], counter=fo[4], workspace=ws)
    e1.record()
    for rep in range(20):
        two = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, dt_gamma, 1024, noises)
    e2.record()
    torch.cuda.synchronize()
    Mf = int(fo[4].item())
    same = Mf == two[0].shape[0] and torch.equal(fo[0][:Mf], two[0]) and torch.equal(fo[2][:Mf], two[2]) and torch.equal(fo[3], two[3])
    print(f"   single-pass {1e3 * e0.elapsed_time(e1) / 20:.1f} us/call   two-pass wrapper (allocations + host read included) "
          f"{1e3 * e1.elapsed_time(e2) / 20:.1f} us/call   identical={same}")
