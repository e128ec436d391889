"""world_size-2 tests of the data-parallel path on CPU (gloo): gradient averaging, bucket packing, replicated
occupancy jitter, view sharding.  The same GradSync code runs over RCCL on the GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf2mesh_amd.parallel import GradSync, shard_views
    torch.manual_seed(0)
    model = torch.nn.Module()
    model.table = torch.nn.Parameter(torch.zeros(1 << 20, 1))       # "big": own collective, reduced in place
    model.w1 = torch.nn.Parameter(torch.zeros(32, 19))
    model.w2 = torch.nn.Parameter(torch.zeros(1, 32))
    model.unused = torch.nn.Parameter(torch.zeros(5))               # grad stays None on every rank
    sync = GradSync(model, world)
    g = torch.Generator().manual_seed(100 + rank)
    grads = {n: torch.randn(p.shape, generator=g) for n, p in model.named_parameters() if n != "unused"}
    for n, p in model.named_parameters():
        p.grad = grads[n].clone() if n in grads else None
    ptr_before = model.table.grad.data_ptr()
    sync.all_reduce()
    assert model.table.grad.data_ptr() == ptr_before                 # no staging copy for the big tensor
    # expected mean of the per-rank gradients
    ok = True
    per_rank = []
    for r in range(world):   # replay every rank's draw sequence (one generator per rank, parameters in order)
        gr = torch.Generator().manual_seed(100 + r)
        per_rank.append({n: torch.randn(p.shape, generator=gr) for n, p in model.named_parameters() if n != "unused"})
    for n, p in model.named_parameters():
        if n == "unused":
            ok &= bool((p.grad == 0).all())
            continue
        exp = sum(d[n] for d in per_rank) / world
        ok &= torch.allclose(p.grad, exp, atol=1e-6)
    # inf on one rank reaches every rank (GradScaler then skips the step everywhere)
    for n, p in model.named_parameters():
        p.grad = torch.zeros_like(p)
    if rank == 1:
        model.w1.grad[0, 0] = float("inf")
    sync.all_reduce()
    ok &= bool(torch.isinf(model.w1.grad[0, 0]))
    # explicit-tensor SUM (the FusedAdamAMP path): own collective per big tensor in its own dtype, flat bucket for the rest
    big32 = torch.full((1 << 16,), float(rank + 1))
    big16 = torch.full((1 << 16, 2), float(rank + 1), dtype=torch.float16)
    small = [torch.full((7, 3), float(rank)), torch.zeros(()) + (1.0 if rank == 1 else 0.0), None]
    p16 = big16.data_ptr()
    sync.all_reduce_sum([big32, big16, None], small)
    tot = sum(range(1, world + 1))
    ok &= bool((big32 == tot).all()) and bool((big16 == tot).all()) and big16.dtype == torch.float16 and big16.data_ptr() == p16
    ok &= bool((small[0] == sum(range(world))).all()) and float(small[1]) == 1.0
    # identical jitter for the replicated occupancy refresh
    GradSync.sync_rng_for_grid_update(48)
    r = torch.rand(4)
    gathered = [torch.zeros(4) for _ in range(world)]
    dist.all_gather(gathered, r)
    ok &= all(torch.equal(gathered[0], t) for t in gathered)
    ok &= shard_views(10, rank, world) == list(range(rank, 10, world))
    # the sharded refresh's symmetric list check: every rank gets the SAME verdict, whichever rank differs
    from nerf2mesh_amd.parallel import all_ranks_hold
    same = torch.tensor([-1, 0, 12345, 678901234567], dtype=torch.int64)
    ok &= all_ranks_hold(same, world) is True
    mine = same.clone()
    mine[2] += rank                                                  # rank 1 holds a different list
    ok &= all_ranks_hold(mine, world) is False
    # broadcast_parameters makes replicas identical
    with torch.no_grad():
        model.w2.fill_(float(rank + 1))
    sync.broadcast_parameters(model, src=0)
    ok &= bool((model.w2 == 1).all())
    q.put((rank, bool(ok), sync.grad_bytes()))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_grad_sync_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
    assert [r[1] for r in res] == [True, True]
    assert res[0][2] == ((1 << 20) + 32 * 19 + 32 + 5) * 4


def test_lego_gradient_volume():
    """73.5 MB of fp32 gradients per iteration for the lego network."""
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    from nerf2mesh_amd.parallel import GradSync
    m = NeRFNetwork(make_options(O=True, bound=1, dt_gamma=0))
    s = GradSync(m, 1)
    assert s.grad_bytes() == 18367240 * 4
    assert len(s.big) == 2 and sum(p.numel() for p in s.small) == 7648
