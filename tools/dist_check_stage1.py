#!/usr/bin/env python3
"""Multi-rank STAGE-1 check (launch with torch.distributed.run; N2M_DIST_BACKEND=gloo lets the ranks share one GPU): views shard over the
ranks, the gradients of the colour table (fp16), the colour networks and the vertex offsets are SUMMED with 1 / world folded into
the loss scale, FusedAdamAMP steps in lock-step.  After K steps every rank must hold bit-identical parameters / loss scale / step counts;
sync_refine_state() must leave every rank with the SUM of the per-rank face-error accumulators (nerf/renderer.py:924-943 feeds them,
nerf/utils.py:1204-1207 consumes them); broadcast_mesh() must hand rank 0's (here: thinned) mesh to everybody, after which training goes
on in lock-step.  Prints 'DIST_CHECK_S1 OK ...' on rank 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nerf2mesh_amd import synthetic
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options
from nerf2mesh_amd.parallel import init_from_env
from nerf2mesh_amd.trainer import Stage1Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
driver = sys.argv[2] if len(sys.argv) > 2 else "trainer"           # "engine": engine_stage1.Stage1Engine drives the trainer's state
rank, world, local = init_from_env()
device = torch.device("cuda", local % torch.cuda.device_count())
torch.cuda.set_device(device)
torch.manual_seed(0)
opt = make_options(O=True, bound=1, dt_gamma=0, stage=1, fused_mlp=True)
v, f = synthetic.scene_mesh(20000)
tr = Stage1Trainer(NeRFNetwork(opt), opt, synthetic.make_cameras(8, seed=0), v, f, device, H=200, W=200, rank=rank, world_size=world)
assert tr.amp_adam, "the fused AMP optimizer must stay on with more than one rank"


def stepper():
    if driver != "engine":
        return tr.train_step
    from nerf2mesh_amd.engine_stage1 import Stage1Engine
    assert Stage1Engine.supported(tr)
    return Stage1Engine(tr).train_step


step = stepper()


def digest():
    flat = torch.cat([p.detach().float().reshape(-1) for p in tr.model.parameters()])
    o = tr.optimizer
    return torch.stack([flat.double().sum(), flat.double().abs().sum(), o.scale.double().reshape(()), o.steps.double().sum()]).cpu(), flat


def same_everywhere(t):
    got = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(got, t)
    return all(torch.equal(g, got[0]) for g in got)


losses = [float(step()) for _ in range(steps)]
torch.cuda.synchronize()
d, flat = digest()
ok = bool(torch.isfinite(flat).all()) and all(l == l for l in losses) and same_everywhere(d.to(device))
moved = float(tr.model.vertices_offsets.detach().abs().sum()) > 0
# per-face errors: every rank has seen its own views only
own_err, own_cnt = tr.model.triangles_errors.clone(), tr.model.triangles_errors_cnt.clone()
tot_e, tot_c = own_err.clone(), own_cnt.clone()
dist.all_reduce(tot_e); dist.all_reduce(tot_c)
tr.sync_refine_state()
ok = ok and torch.equal(tr.model.triangles_errors, tot_e) and torch.equal(tr.model.triangles_errors_cnt, tot_c) and float(tot_c.sum()) > float(own_cnt.sum()) > 0
# rank 0 "refines" the mesh (stand-in for refine_and_decimate: every third face dropped); everybody takes it over and trains on
if rank == 0:
    keep = torch.arange(tr.model.triangles.shape[0], device=device) % 3 != 0
    tr.model.init_stage1(tr.model.vertices, tr.model.triangles[keep])
n_faces_before = f.shape[0]
tr.broadcast_mesh(src=0)
nf = torch.tensor([tr.model.triangles.shape[0]], device=device)
ok = ok and same_everywhere(nf) and int(nf) < n_faces_before and same_everywhere(tr.model.triangles.double().sum().reshape(1))
step = stepper()               # the mesh changed: the trainer rebuilt its state, the executor its buffers
more = [float(step()) for _ in range(2)]
torch.cuda.synchronize()
d2, flat2 = digest()
ok = ok and bool(torch.isfinite(flat2).all()) and same_everywhere(d2.to(device)) and all(l == l for l in more)
dist.barrier()
if rank == 0:
    print(f"DIST_CHECK_S1 {'OK' if ok else 'FAILED'} driver={driver} world={world} backend={dist.get_backend()} steps={steps} loss {losses[0]:.5f} -> {losses[-1]:.5f} "
          f"offsets_moved={moved} faces {n_faces_before} -> {int(nf)} digest={[float(x) for x in d2]}")
dist.destroy_process_group()
sys.exit(0 if ok else 1)
