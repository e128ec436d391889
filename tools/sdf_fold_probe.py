"""Probe for folding the SDF recipe's six finite-difference copies into the batch's own table backward: how many (copy, level) pairs share
the centre sample's cell at the end of the schedule, and what the stacked-copies backward costs when only the others carry a gradient.
python tools/sdf_fold_probe.py [steps]"""
import sys
import numpy as np
import torch
from nerf2mesh_amd import _lib as L, synthetic
from nerf2mesh_amd.engine import Stage0Engine
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options

p = L.ptr
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1040
torch.manual_seed(0)
opt = make_options(O=True, iters=2000, fused_mlp=True, bound=1, dt_gamma=0, sdf=True)
dev = torch.device("cuda", 0)
model = NeRFNetwork(opt)
tr = Stage0Engine(model, opt, synthetic.make_cameras(100, seed=0), dev, seed=0)
tr.mark_untrained()
for _ in range(steps):
    tr.train_step()
torch.cuda.synchronize()
M = tr.last_num_points
sb = tr._sdf
e1 = model.encoder
print(f"M = {M}, eps = {opt.normal_anneal_epsilon:.3g}, max_level = {model.max_level}")
pts01 = sb["pts01"][:18 * M].view(M, 6, 3)
ctr = 0.5 * (pts01[:, 0] + pts01[:, 1])
ctr[:, 1] = 0.5 * (pts01[:, 2, 1] + pts01[:, 3, 1]); ctr[:, 2] = 0.5 * (pts01[:, 4, 2] + pts01[:, 5, 2])
S, H0 = float(np.log2(e1.per_level_scale)), int(e1.base_resolution)
same_all = torch.ones(M, 6, dtype=torch.bool, device=dev)
same_l = []
for l in range(16):
    scale = 2.0 ** (l * S) * H0 - 1.0
    cell = lambda x: torch.floor(x * scale + 0.5).long()
    same = (cell(pts01) == cell(ctr).unsqueeze(1)).all(-1)            # [M,6]
    same_l.append(same)
    same_all &= same
    print(f"level {l:2d}: copies in the centre's cell {same.float().mean().item():.4f}")
print(f"copies in the centre's cell on ALL levels: {same_all.float().mean().item():.4f}")
geo = (tr.Lv, tr.Lv, tr.S, tr.H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id)
M6 = 6 * M
need = L.lib().n2m_grid_binned_pair_workspace_bytes(M6, tr.Lv, tr.ho.ctypes.data)
ws = L.workspace(dev, need)
g = torch.randn(16, M6, device=dev) * 1e-3


def run(grad, B, pts, label):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(6):
        if it == 1:
            ev[0].record()
        L.call("n2m_grid_encode_backward_binned_pair", p(grad), None, p(pts), tr.ho.ctypes.data, p(tr.g1), None, B, *geo,
               None, 0.0, 0.0, 1.0, None, p(tr.optimizer.found_inf), 1.0, 0.0, 0, p(ws), ws.numel(), L.stream())
    ev[1].record()
    torch.cuda.synchronize()
    print(f"{label:60s} {ev[0].elapsed_time(ev[1]) / 5 * 1e3:8.1f} us")


run(g, M6, sb["pts01"], "stacked copies, every (copy, level) carries a gradient")
gz = g.clone()
for l in range(16):
    gz[l][same_l[l].reshape(-1)] = 0
run(gz, M6, sb["pts01"], "same, gradient zeroed where the copy shares the centre's cell")
run(torch.zeros_like(g), M6, sb["pts01"], "same, all gradients zero")
keep = (~same_all).reshape(-1)
ptsk = sb["pts01"][:18 * M].view(M6, 3)[keep].contiguous()
gk = gz[:, keep].contiguous()
print(f"copies that leave the centre's cell on some level: {int(keep.sum())} of {M6}")
run(gk, ptsk.shape[0], ptsk, "compacted: only those copies")
