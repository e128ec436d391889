import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
from test_engine import _run
from nerf2mesh_amd.engine import Stage0Engine
from nerf2mesh_amd.trainer import Stage0Trainer
cfg = dict(sdf=True, iters=40, diffuse_step=10)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 22
a, la = _run(Stage0Trainer, steps, **cfg)
b, lb = _run(Stage0Engine, steps, **cfg)
b2, lb2 = _run(Stage0Engine, steps, **cfg)
a2, la2 = _run(Stage0Trainer, steps, **cfg)
print("steps", a.optimizer.steps.tolist(), b.optimizer.steps.tolist(), "scale", float(a.optimizer.scale), float(b.optimizer.scale))
print("loss diff per step (trainer-engine):", " ".join(f"{abs(x-y):.1e}" for x, y in zip(la, lb)))
print("loss diff per step (engine-engine):", " ".join(f"{abs(x-y):.1e}" for x, y in zip(lb, lb2)))
print("loss diff per step (trainer-trainer):", " ".join(f"{abs(x-y):.1e}" for x, y in zip(la, la2)))
rel = lambda p, q: ((p - q).norm() / p.norm().clamp_min(1e-30)).item()
for (n, p), (_, q), (_, r), (_, t) in zip(a.model.named_parameters(), b.model.named_parameters(), b2.model.named_parameters(), a2.model.named_parameters()):
    print(f"{n:30s} t-e {rel(p,q):.2e}  e-e {rel(q,r):.2e}  t-t {rel(p,t):.2e}")
