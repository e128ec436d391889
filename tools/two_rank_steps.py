"""Per-step wall time of the step executor with two ranks on one GPU (debug aid: where do slow stretches come from?).
   N2M_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29581 tools/two_rank_steps.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nerf2mesh_amd import synthetic
from nerf2mesh_amd.engine import Stage0Engine
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options
from nerf2mesh_amd.parallel import init_from_env
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rank, world, local = init_from_env()
dev = torch.device("cuda", local % torch.cuda.device_count())
torch.cuda.set_device(dev)
torch.manual_seed(0)
opt = make_options(O=True, bound=1, dt_gamma=0, iters=30000, fused_mlp=True)
tr = Stage0Engine(NeRFNetwork(opt), opt, synthetic.make_cameras(100, seed=0), dev, rank=rank, world_size=world, seed=0)
tr.mark_untrained()
ts = []
for i in range(steps):
    t0 = time.perf_counter()
    tr.train_step()
    ts.append(1e3 * (time.perf_counter() - t0))
torch.cuda.synchronize()
dist.barrier()
if rank == 0:
    print("peer" if tr.peer is not None else "collective", "host ms per train_step call:")
    for i in range(0, steps, 16):
        print(f"  steps {i + 1:3d}..{i + 16:3d}: " + " ".join(f"{t:6.1f}" for t in ts[i:i + 16]))
dist.destroy_process_group()
