"""n2m_rasterize_forward on the stage-1 bench frame (305 k faces, 1600 x 1600), per-kernel time by torch events around the call.
   python tools/raster_lab.py   """
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf2mesh_amd import _lib as L, synthetic as S
dev = torch.device("cuda")
v, f = S.scene_mesh(300000)
v, f = v.to(dev), f.to(dev)
poses = S.make_cameras(100, seed=0)
H = W = 1600
V, F = v.shape[0], f.shape[0]
clip, zbuf, rast = torch.empty(V, 4, device=dev), torch.empty(H * W, dtype=torch.int64, device=dev), torch.empty(H, W, 4, device=dev)
s = L.stream()
tot = 0.0
for vi in range(4):
    mvp = S.mvp_matrix(poses[vi], 800, 800).to(dev).contiguous()
    L.call("n2m_to_clip", L.ptr(v), L.ptr(mvp), V, L.ptr(clip), s)
    for _ in range(3):
        L.call("n2m_rasterize_forward", L.ptr(clip), L.ptr(f), V, F, H, W, L.ptr(zbuf), L.ptr(rast), s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        L.call("n2m_rasterize_forward", L.ptr(clip), L.ptr(f), V, F, H, W, L.ptr(zbuf), L.ptr(rast), s)
    b.record(); torch.cuda.synchronize()
    tot += a.elapsed_time(b) / 20
    cs = int((rast[..., 3] > 0).sum()), float(rast.double().sum())
print(f"{tot / 4 * 1e3:.1f} us per rasterize call (clear + small + big + resolve); last view covered {cs[0]}, checksum {cs[1]:.6f}")
