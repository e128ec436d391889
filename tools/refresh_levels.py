"""Where the occupancy refresh's density query spends its time, by level: n2m_grid_encode_forward (fp32 density table) on the refresh's own
points -- the 128^3 cell centres in Morton order + jitter -- for growing max_level, and on the same points shuffled (no coherence at all).
    python tools/refresh_levels.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf2mesh_amd import _lib as L
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options

dev = torch.device("cuda")
torch.manual_seed(0)
model = NeRFNetwork(make_options(O=True, bound=1, dt_gamma=0, fused_mlp=True)).to(dev)
e = model.encoder
cells = model._cells()                                   # [128^3, 3] in [-1, 1], Morton order
hgs = 1.0 / 128
pts = cells * (1 - hgs) + (torch.rand_like(cells) * 2 - 1) * hgs
x01 = ((pts + 1) / 2).contiguous()
B = x01.shape[0]
out = torch.empty(16, B, dtype=torch.float32, device=dev)
import numpy as np
S = float(np.log2(e.per_level_scale))


def run(x, ml, reps=10):
    args = (L.ptr(x), L.ptr(e.embeddings), L.ptr(e.offsets), L.ptr(out), B, 3, 1, 16, ml, S, int(e.base_resolution), None, e.gridtype_id,
            int(bool(e.align_corners)), e.interp_id, L.F32, L.stream())
    for _ in range(2):
        L.call("n2m_grid_encode_forward", *args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        L.call("n2m_grid_encode_forward", *args)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


shuf = x01[torch.randperm(B, device=dev)].contiguous()
prev = prev_s = 0.0
print(f"{B} points; level resolution = 16 * 1.3819^l; the occupancy grid is 128^3")
for ml in range(1, 17):
    t, ts = run(x01, ml), run(shuf, ml)
    res = int(np.ceil(16 * e.per_level_scale ** (ml - 1)))
    print(f"levels 0..{ml - 1:2d}: Morton order {t:7.1f} us (+{t - prev:6.1f} for level {ml - 1:2d}, res {res:4d})   shuffled {ts:7.1f} us (+{ts - prev_s:6.1f})")
    prev, prev_s = t, ts
