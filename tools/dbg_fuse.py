import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
from test_engine import _run, _table_state
from nerf2mesh_amd.engine import Stage0Engine
os.environ["N2M_FUSE_ADAM"] = "0"
a, la = _run(Stage0Engine, 1, diffuse_step=4)
b, lb = _run(Stage0Engine, 1, diffuse_step=4)
print("losses", la, lb, "found_inf", float(a.optimizer.found_inf), float(b.optimizer.found_inf), "scale", float(a.optimizer.scale), float(b.optimizer.scale))
sa, sb = _table_state(a), _table_state(b)
offs = list(a.model.encoder.host_offsets)
for name in ("density", "colour"):
    for which, x, y in zip(("param", "exp_avg", "exp_avg_sq"), sa[name], sb[name]):
        out = []
        for l in range(16):
            sl = slice(offs[l], offs[l + 1])
            nd = (x[sl] != y[sl]).sum().item()
            rel = ((x[sl] - y[sl]).abs().max() / x[sl].abs().max().clamp_min(1e-30)).item()
            out.append(f"{l}:{nd}({rel:.1e})")
        print(name, which, " ".join(out))
