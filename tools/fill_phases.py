"""Phase durations inside the partition-major fill, in the REAL step (bench state after --pretrain steps): shader-clock stamps of two
workgroups (a fine hashed level, a coarse dense one), first 4 tiles each, wave 0 and the last wave.
    python tools/fill_phases.py [--pretrain 1000]
"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--pretrain", type=int, default=1000)
args = ap.parse_args()
from nerf2mesh_amd import _lib as L, synthetic
from nerf2mesh_amd.engine import Stage0Engine
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options
torch.manual_seed(0)
opt = make_options(O=True, iters=30000, fused_mlp=True, bound=1, dt_gamma=0)
dev = torch.device("cuda", 0)
eng = Stage0Engine(NeRFNetwork(opt), opt, synthetic.make_cameras(100, seed=0), dev, seed=0)
eng.mark_untrained()
for _ in range(args.pretrain + 2):
    eng.train_step()
torch.cuda.synchronize()
names = ["entries (+TV, merge)", "slot atomics -> barrier 1", "scan + staging", "-> barrier 2", "copy-out issued"]
acc = {}
for rep in range(6):
    L.call("n2m_debug_fill_times", 1, None)
    eng.train_step()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 116)()
    L.call("n2m_debug_fill_times", 0, ctypes.addressof(buf))
    for w in range(2):
        for wave in range(2):
            for i in range(4):
                t = [buf[w * 48 + (i + 4 * wave) * 6 + j] for j in range(6)]
                if t[0] == 0:
                    continue
                d = [t[j + 1] - t[j] for j in range(5)]
                nxt = buf[w * 48 + (i + 1 + 4 * wave) * 6] - t[5] if i < 3 and buf[w * 48 + (i + 1 + 4 * wave) * 6] else None
                acc.setdefault((w, wave), []).append((d, nxt, t[5] - t[0]))
# stamps are shader-clock cycles (a tile of the 151 us fill lasts ~8.7 us = ~19.5 k cycles: ~2.2 GHz)
for (w, wave), rows in sorted(acc.items()):
    what = ("fine hashed level (workgroup 3)", "coarse dense level (workgroup grid/2+3)")[w]
    n = len(rows)
    mean = [sum(r[0][j] for r in rows) / n for j in range(5)]
    tile = sum(r[2] for r in rows) / n
    gaps = [r[1] for r in rows if r[1] is not None]
    print(f"{what}, {'wave 0' if wave == 0 else 'last wave'}: tile {tile / 1e3:.1f} k cycles (mean of {n} tiles); "
          + ", ".join(f"{names[j]} {mean[j] / 1e3:.1f} k = {100 * mean[j] / tile:.0f} %" for j in range(5))
          + (f"; to the next tile's top {sum(gaps) / len(gaps) / 1e3:.1f} k" if gaps else ""))
