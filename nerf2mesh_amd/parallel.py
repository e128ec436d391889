"""Data-parallel stage-0 training across the GPUs of one node: one process per GPU, rays sharded by rank,
one gradient exchange per iteration over RCCL/xGMI.

The reference is single-GPU (its DDP scaffolding is never initialised, nerf/utils.py:517-519), so this is new
functionality.  Every sample is independent given the shared parameters and occupancy grid; rank r draws its
own rays and the only data-path collective is the sum-all-reduce (/ world) of the gradients:

  * the two hash tables' gradients (24.5 MB + 49.0 MB fp32 for lego) are reduced IN PLACE, one collective
    each -- already far above the xGMI latency regime, no staging copy;
  * the MLP weights' gradients (7 648 floats) are packed into one flat bucket -> one small collective.

`torch.distributed` backend "nccl" is RCCL on ROCm; the CPU tests drive the same code with "gloo".
The occupancy grid must stay identical on all ranks: its refresh is made deterministic by seeding the jitter
identically everywhere (`sync_rng_for_grid_update`), which costs no communication because the parameters are
already identical.
"""
import torch
import torch.distributed as dist

BIG = 1 << 20   # tensors with at least this many elements get their own collective


class GradSync:
    def __init__(self, module, world_size, process_group=None):
        self.world = world_size
        self.group = process_group
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.big = [p for p in self.params if p.numel() >= BIG]
        self.small = [p for p in self.params if p.numel() < BIG]
        n_small = sum(p.numel() for p in self.small)
        self.bucket = None
        if n_small:
            p0 = self.small[0]
            self.bucket = torch.zeros(n_small, dtype=torch.float32, device=p0.device)

    @torch.no_grad()
    def all_reduce(self):
        """grad <- mean over ranks of grad, for every parameter. Collectives are issued asynchronously and waited
        together so the small bucket rides along with the two large ones."""
        if self.world <= 1:
            return
        works = []
        for p in self.big:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            works.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if self.bucket is not None:
            off = 0
            for p in self.small:
                n = p.numel()
                if p.grad is None:
                    self.bucket[off:off + n].zero_()
                else:
                    self.bucket[off:off + n].copy_(p.grad.reshape(-1))
                off += n
            works.append(dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        inv = 1.0 / self.world
        for p in self.big:
            p.grad.mul_(inv)
        if self.bucket is not None:
            off = 0
            for p in self.small:
                n = p.numel()
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                p.grad.copy_(self.bucket[off:off + n].view_as(p.grad)).mul_(inv)
                off += n

    @torch.no_grad()
    def all_reduce_sum_begin(self, big, small):
        """Issue the in-place SUM over ranks of explicit tensors and return a token for all_reduce_sum_end: every tensor of `big`
        gets its own collective in its own dtype (an fp16 gradient travels as fp16: half the bytes over xGMI), the tensors of
        `small` ride together in one flat fp32 bucket.  No averaging here -- the caller folds 1/world into the loss scale, which
        saves a pass over the tables.  Work enqueued on the current stream between begin and end overlaps the collectives."""
        if self.world <= 1:
            return None
        works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for t in big if t is not None]
        small = [t for t in small if t is not None]
        n = sum(t.numel() for t in small)
        flat = None
        if n:
            if self.bucket is None or self.bucket.numel() < n or self.bucket.device != small[0].device:
                self.bucket = torch.zeros(n, dtype=torch.float32, device=small[0].device)
            flat = self.bucket[:n]
            torch.cat([t.reshape(-1).float() for t in small], out=flat)
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return works, small, flat

    @torch.no_grad()
    def all_reduce_sum_end(self, token):
        if token is None:
            return
        works, small, flat = token
        for w in works:
            w.wait()
        if flat is not None:
            off = 0
            for t in small:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()

    def all_reduce_sum(self, big, small):
        self.all_reduce_sum_end(self.all_reduce_sum_begin(big, small))

    def grad_bytes(self):
        return sum(p.numel() for p in self.params) * 4

    @staticmethod
    def sync_rng_for_grid_update(step, base=0x5EED):
        """Same jitter on every rank for the occupancy refresh at `step` (density_grid/bitfield stay replicated)."""
        torch.manual_seed(base + step)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(base + step)

    @torch.no_grad()
    def broadcast_parameters(self, module, src=0):
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=self.group)


def shard_views(n_views, rank, world):
    """Stage-1 sharding: views rank, rank+world, ... (one full image per rank per step)."""
    return list(range(rank, n_views, world))


def init_from_env(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world, local_rank). world == 1 needs no process group."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # N2M_DIST_BACKEND=gloo lets the whole multi-rank trainer run on a box with fewer GPUs than ranks (test use)
            backend = os.environ.get("N2M_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
