#!/usr/bin/env python3
"""Multi-rank stage-0 trainer check (launch with torch.distributed.run; N2M_DIST_BACKEND=gloo lets N ranks share one GPU):
every rank trains the same model on its own rays with the RCCL/gloo gradient sum; after K steps all ranks must hold bit-identical
parameters, optimizer scale and step count, the loss must have dropped, and the result must track a single-rank run of the same
seed within training noise.  Prints 'DIST_CHECK OK ...' on rank 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nerf2mesh_amd import synthetic
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options
from nerf2mesh_amd.parallel import init_from_env
from nerf2mesh_amd.trainer import Stage0Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ENGINE = len(sys.argv) > 2 and sys.argv[2] == "engine"       # the step executor instead of the autograd trainer
rank, world, local = init_from_env()
device = torch.device("cuda", local % torch.cuda.device_count())
torch.cuda.set_device(device)
torch.manual_seed(0)
GARDEN = os.environ.get("N2M_DIST_RECIPE") == "garden"       # BASELINE config 4's recipe: 5 cascades, per-view near / far, entropy term
opt = (make_options(O=True, bound=16, dt_gamma=1 / 256, lambda_entropy=1e-3, enable_cam_near_far=True, scene="garden", iters=30000, fused_mlp=True)
       if GARDEN else make_options(O=True, bound=1, dt_gamma=0, iters=30000, fused_mlp=True))
poses = synthetic.make_cameras(100, seed=0)
if os.environ.get("N2M_DIST_BLIND_RANK") == str(rank):
    # this rank's cameras are moved 50 units back and turned around: every ray misses the scene, every batch of the rank has zero samples.
    # It must still issue the same collectives in the same order as the ranks that have samples (and apply the summed gradients).
    poses = poses.clone()
    poses[:, :3, 3] = poses[:, :3, 3] + poses[:, :3, 2] * 50.0
    poses[:, :3, :3] = poses[:, :3, :3] @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))
model = NeRFNetwork(opt)
if GARDEN:
    model.update_aabb(synthetic.pts_aabb("garden"))
if ENGINE:
    from nerf2mesh_amd.engine import Stage0Engine
    tr = Stage0Engine(model, opt, poses, device, rank=rank, world_size=world, seed=0)
else:
    tr = Stage0Trainer(model, opt, poses, device, rank=rank, world_size=world, seed=0)
tr.mark_untrained()
ASYM = os.environ.get("N2M_DIST_ASYM")                       # a torch-side write to density_grid on RANK 1 ONLY after step 10 (tests/test_parallel_gpu.py):
losses = []                                                  # "touch": same values, new version counter; "mark": one empty cell set to -1 (the lists differ)
for it in range(steps):
    losses.append(float(tr.train_step()))
    if ASYM and it == 9 and rank == 1:
        if ASYM == "touch":
            tr.model.density_grid.mul_(1.0)
        else:
            tr.model.density_grid[0, 0] = -1.0                # Morton cell 0 = the (-1,-1,-1) corner: empty, its bit is 0 either way
if hasattr(tr, "sync_parameters"):
    tr.sync_parameters()              # sharded optimizer: every rank owns 1/W of the table rows until the fp32 tensors are gathered
torch.cuda.synchronize()
if getattr(tr, "peer", None) is not None:
    tr.peer.check()                   # a wait that ran into its timeout would have left its mark
flat = torch.cat([p.detach().float().reshape(-1) for p in tr.model.parameters()])
digest = torch.stack([flat.double().sum(), flat.double().abs().sum(), tr.optimizer.scale.double() if hasattr(tr.optimizer, "scale") else torch.zeros((), device=device).double(),
                      tr.optimizer.step_count.double() if hasattr(tr.optimizer, "step_count") else torch.zeros((), device=device).double()]).cpu()
if rank == 0 and os.environ.get("N2M_DIST_DUMP_GRID"):     # occupancy state: bit field + density grid (sharded vs replicated refresh)
    torch.save({"bits": tr.model.density_bitfield.cpu(), "grid": tr.model.density_grid.cpu()}, os.environ["N2M_DIST_DUMP_GRID"])
if rank == 0 and os.environ.get("N2M_DIST_DUMP"):          # every 97th parameter, for run-to-run comparisons (tests/test_parallel_gpu.py)
    torch.save(flat[::97].cpu(), os.environ["N2M_DIST_DUMP"])
ok = torch.isfinite(flat).all().item() and all(l == l for l in losses)
if world > 1:
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    ok = ok and all(torch.equal(g, gathered[0]) for g in gathered)
    dist.barrier()
first, last = sum(losses[:5]) / 5, sum(losses[-5:]) / 5
if os.environ.get("N2M_DIST_BLIND_RANK") is None:
    ok = ok and last < first
else:                                  # the blind rank's own loss is the constant background error
    ok = ok and (last < first or os.environ.get("N2M_DIST_BLIND_RANK") == str(rank))
    blind = torch.tensor([float(tr.samples_seen)], device=device)
    allb = [torch.zeros_like(blind) for _ in range(world)]
    dist.all_gather(allb, blind)
    ok = ok and float(allb[int(os.environ["N2M_DIST_BLIND_RANK"])]) == 0 and sum(float(b) for b in allb) > 0
if os.environ.get("N2M_DIST_CKPT"):
    # every rank writes what a checkpoint would hold (optimizer.state_dict() is a collective for a sharded run: it gathers the moments)
    sd = tr.optimizer.state_dict()
    keep = {}
    for i in (0, 1):                                   # the two tables are the first two parameters (get_params order)
        st = sd["state"][i]
        keep[f"exp_avg.{i}"], keep[f"exp_avg_sq.{i}"] = st["exp_avg"].detach().cpu(), st["exp_avg_sq"].detach().cpu()
    keep["steps"] = sd["n2m_amp"]["steps"]
    torch.save(keep, f"{os.environ['N2M_DIST_CKPT']}.rank{rank}.pt")
    if world > 1:
        dist.barrier()
if rank == 0:
    print(f"DIST_CHECK {'OK' if ok else 'FAILED'} driver={type(tr).__name__} shard={getattr(tr, 'shard', False)} peer_store={getattr(tr, 'peer', None) is not None} refresh_sharded={bool(getattr(tr.model, 'refresh_shard', None)) and bool((getattr(tr.model, '_refresh_bufs', None) or {}).get('shard_ok'))} backend={dist.get_backend() if world > 1 else 'none'} world={world} steps={steps} loss {first:.5f} -> {last:.5f} digest={[float(x) for x in digest]}")
if world > 1:
    dist.destroy_process_group()
sys.exit(0 if ok else 1)
