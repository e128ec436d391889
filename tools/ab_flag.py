"""A/B of a module-level switch inside one process: python tools/ab_flag.py nerf2mesh_amd.fused CONCURRENT_BACKWARD
(or of a Stage0Trainer attribute: python tools/ab_flag.py trainer overlap_march)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf2mesh_amd import synthetic
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options
from nerf2mesh_amd.trainer import Stage0Trainer
mod = None if sys.argv[1] == "trainer" else importlib.import_module(sys.argv[1]); name = sys.argv[2]
torch.manual_seed(0)
opt = make_options(O=True, bound=1, dt_gamma=0, iters=30000, fused_mlp=True)
tr = Stage0Trainer(NeRFNetwork(opt), opt, synthetic.make_cameras(100, seed=0), torch.device("cuda", 0), seed=0)
tr.mark_untrained()
for i in range(400): tr.train_step()
tot = {True: 0.0, False: 0.0}
n = {True: 0, False: 0}
order = [True, False, False, True] * 6          # ABBA blocks cancel the slow drift of the workload (rays/step grows as the grid prunes)
for val in order:
    setattr(tr if mod is None else mod, name, val)
    for i in range(5): tr.train_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40): tr.train_step()
    torch.cuda.synchronize()
    tot[val] += time.perf_counter() - t0; n[val] += 40
for val in (True, False):
    print(f"{name}={val}: {1e3*tot[val]/n[val]:.3f} ms/step over {n[val]} steps")
