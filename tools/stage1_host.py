"""Is the stage-1 step GPU-bound or host-bound?  Host enqueue time per step against wall time per step (tools/cpu_bound.py for stage 1)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf2mesh_amd import synthetic
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options
from nerf2mesh_amd.trainer import Stage1Trainer

torch.manual_seed(0)
opt = make_options(O=True, bound=1, dt_gamma=0, stage=1, fused_mlp=True)
v, f = synthetic.scene_mesh(300000)
tr = Stage1Trainer(NeRFNetwork(opt), opt, synthetic.make_cameras(100, seed=0), v, f, torch.device("cuda", 0))
tr.preload()
for _ in range(30): tr.train_step()
torch.cuda.synchronize()
n = 60
t0 = time.perf_counter(); c0 = time.thread_time()
for _ in range(n): tr.train_step()
c1 = time.thread_time(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"stage-1: wall/step {1e3*(t2-t0)/n:.3f} ms  host loop/step {1e3*(t1-t0)/n:.3f} ms  host CPU/step {1e3*(c1-c0)/n:.3f} ms  drain {1e3*(t2-t1):.2f} ms")
if "--cprofile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n): tr.train_step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
