#!/usr/bin/env python3
"""Marching-cubes kernel timing: count + emit passes on an R^3 volume (sphere + tori), torch.cuda events around each C-ABI call.
Algorithmic bytes: count 4 B read + 4 B written per node; emit 8 B read per node (only workgroups with a crossing load the volume) + the
mesh written (12 B per vertex, 12 B per triangle)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf2mesh_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--R", type=int, nargs="+", default=[128, 256, 512])
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
for R in args.R:
    x = torch.linspace(-1, 1, R, device="cuda")
    X, Y, Z = torch.meshgrid(x, x, x, indexing="ij")
    d = torch.minimum(torch.sqrt(X * X + Y * Y + Z * Z) - 0.45, torch.sqrt((torch.sqrt(X * X + Y * Y) - 0.75) ** 2 + Z * Z) - 0.12)
    vol = (-d).contiguous()
    del X, Y, Z, d
    ws = torch.empty(int(L.lib().n2m_marching_cubes_workspace_bytes(R, R, R)), dtype=torch.uint8, device="cuda")
    totals = torch.zeros(2, dtype=torch.int64, device="cuda")
    s = L.stream()
    L.call("n2m_marching_cubes_count", vol.data_ptr(), R, R, R, 0.0, ws.data_ptr(), ws.numel(), totals.data_ptr(), s)
    nv, nt = totals.tolist()
    v = torch.empty(nv, 3, device="cuda")
    t = torch.empty(nt, 3, dtype=torch.int32, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tc = te = 0.0
    for r in range(args.reps + 2):
        ev[0].record()
        L.call("n2m_marching_cubes_count", vol.data_ptr(), R, R, R, 0.0, ws.data_ptr(), ws.numel(), totals.data_ptr(), s)
        ev[1].record()
        L.call("n2m_marching_cubes_emit", vol.data_ptr(), R, R, R, 0.0, ws.data_ptr(), ws.numel(), R - 1.0, 2.0, -1.0, v.data_ptr(), 0, nv, t.data_ptr(), nt, s)
        ev[2].record()
        torch.cuda.synchronize()
        if r >= 2:
            tc += ev[0].elapsed_time(ev[1])
            te += ev[1].elapsed_time(ev[2])
    tc, te = tc / args.reps * 1e-3, te / args.reps * 1e-3
    N = R ** 3
    bc, be = 8.0 * N, 8.0 * N + 12.0 * nv + 12.0 * nt
    print(f"R={R}: {nv} vertices, {nt} triangles | count {tc * 1e6:8.1f} us = {bc / tc / 1e9:7.1f} GB/s ({bc / tc / 8e12:.2f} of 8 TB/s) | "
          f"emit {te * 1e6:8.1f} us = {be / te / 1e9:7.1f} GB/s | total {1e3 * (tc + te):.3f} ms")
