#!/usr/bin/env python3
"""Bit comparison of the partition-major and the tile-major table backward on marched samples: run once per N2M_BIN_PM setting with
`dump <file>`, then `cmp <a> <b>` prints the differing entries per level."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if sys.argv[1] == "dump":
    from nerf2mesh_amd import raymarching, synthetic as S
    from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward_pair
    dev = torch.device("cuda")
    poses = S.make_cameras(100, seed=0).to(dev)
    bits = raymarching.packbits(S.scene_density_grid(H=128, device=dev), 10.0)
    g = torch.Generator(device=dev).manual_seed(0)
    xs, n, B = [], 0, 2 ** 18 - (333 if '--ragged' in sys.argv else 0)
    while n < B:
        o, d = S.random_rays(poses, 65536, g)
        nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.05)
        xyzs, _, _, _ = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, 0.0, 1024)
        xs.append(xyzs); n += xyzs.shape[0]
    x = ((torch.cat(xs)[:B] + 1) / 2).contiguous()
    torch.manual_seed(0)
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).to(dev); e2 = GridEncoder(level_dim=2, desired_resolution=2048).to(dev)
    g1 = torch.randn(16, B, 1, device=dev) * 1e-3 * (torch.rand(1, B, 1, device=dev) < 0.7); g2 = (torch.randn(16, B, 2, device=dev) * 1e-3).half()
    if '--zeros' in sys.argv:
        dead = torch.rand(1, B, 1, device=dev) < 0.4
        g1 = g1 * ~dead; g2 = (g2.float() * ~dead).half()
    inp = f"/tmp/pm_vs_tile_inputs_{B}_{'--zeros' in sys.argv}.pt"                  # the first run fixes the inputs for the others
    if os.path.exists(inp):
        d = torch.load(inp); x, g1, g2 = d["x"].to(dev), d["g1"].to(dev), d["g2"].to(dev); e1.embeddings.data.copy_(d["emb"].to(dev))
    else:
        torch.save({"x": x.cpu(), "g1": g1.cpu(), "g2": g2.cpu(), "emb": e1.embeddings.detach().cpu()}, inp)
    t1 = torch.zeros_like(e1.embeddings); t2 = torch.zeros(e2.embeddings.shape, device=dev, dtype=torch.float16)
    tv = (e1.embeddings.detach(), 1e-8, 1e-8, 1.0, torch.tensor(1024.0, device=dev)) if "--tv" in sys.argv else None
    if '--affine' in sys.argv: x = x * 2 - 1
    assert binned_backward_pair(e1, e2, g1, g2, x, t1, t2, 16, tv=tv, overwrite=True, in_affine=(0.5, 0.5) if '--affine' in sys.argv else (1.0, 0.0))
    torch.save({"t1": t1.cpu(), "t2": t2.cpu(), "offs": list(e1.host_offsets)}, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    offs = a["offs"]
    for l in range(16):
        sl = slice(offs[l], offs[l + 1])
        d1 = (a["t1"][sl].view(torch.int32) != b["t1"][sl].view(torch.int32)).sum().item()
        d2 = (a["t2"][sl].view(torch.int16) != b["t2"][sl].view(torch.int16)).sum().item()
        m1 = (a["t1"][sl] - b["t1"][sl]).abs().max().item() / max(a["t1"][sl].abs().max().item(), 1e-30)
        print(f"level {l:2d} rows {offs[l+1]-offs[l]:7d}: fp32 differing {d1:7d} (max rel to level max {m1:.2e})   fp16 differing {d2:7d}")
