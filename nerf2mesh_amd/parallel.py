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


def shard_refresh_default():
    """Whether the occupancy refresh's density query is dealt to the ranks (renderer.update_extra_state).  N2M_SHARD_REFRESH=1 / 0 decides; unset:
    on whenever there is more than one rank.  (Rounds 4-5 kept it off over gloo -- the test mode in which two ranks time-share ONE GPU -- because
    "a refresh step now and then stalls for 10-40 s" there.  Round 6 measured where those stalls sit (tools/two_rank_steps.py, 320 steps per mode,
    profiles/r06_two_ranks_gloo_stalls.txt): in ORDINARY steps, at the same rate with the replicated refresh (40.4 s at step 36, 0.9 s at step 307)
    as with the sharded one (3.2 s at step 54, 9.4 s at step 228); every refresh step takes its 60-70 ms in both.  They belong to gloo's
    host-blocking collectives between two processes that share a device, not to the refresh.)"""
    import os
    flag = os.environ.get("N2M_SHARD_REFRESH")
    if flag is not None:
        return flag != "0"
    return dist.is_initialized() and dist.get_world_size() > 1


def all_ranks_hold(summary, world):
    """True on EVERY rank iff every rank passed the same `summary` (a small 1-D integer tensor); a collective all ranks must enter together.
    The sharded occupancy refresh decides with it, at every refresh, whether the ranks hold the same lists of cells (renderer.update_extra_state):
    the decision is made from the gathered summaries, i.e. identically everywhere -- no rank ever falls back (or stays) on its own.
    (gloo wants a flat output tensor; RCCL takes either.)"""
    flat = summary.reshape(-1)
    out = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat)
    every = out.view(world, -1).cpu()                      # one host read per refresh (every 16 steps)
    return bool((every == every[:1]).all())


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


# ------------------------------------------------------------------------------------------------ peer-store exchange (include/n2m_peer.h)
class _DevArray:
    """A raw device range as something torch.as_tensor() takes without a copy (__cuda_array_interface__, version 2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2, "strides": None}


class PeerMemory:
    """One device allocation per rank that every other rank of the node maps (hipIpcGetMemHandle / hipIpcOpenMemHandle through
    n2m_peer_export / n2m_peer_import).  COLLECTIVE constructor: the handles travel through all_gather_object.  ptrs[r] is rank r's
    allocation as THIS process addresses it (its own: the local pointer)."""

    def __init__(self, nbytes, fine_grained, rank, world, group=None):
        import ctypes
        from . import _lib as L
        self.rank, self.world, self.nbytes = rank, world, int(nbytes)
        p = ctypes.c_void_p()
        L.call("n2m_peer_alloc", self.nbytes, 1 if fine_grained else 0, ctypes.byref(p))
        self.local = int(p.value)
        h = ctypes.create_string_buffer(64)
        L.call("n2m_peer_export", self.local, h)
        handles = [None] * world
        dist.all_gather_object(handles, bytes(h.raw), group=group)
        self.ptrs, self._mapped = [], []
        for r in range(world):
            if r == rank:
                self.ptrs.append(self.local)
                continue
            q = ctypes.c_void_p()
            L.call("n2m_peer_import", ctypes.create_string_buffer(handles[r], 64), ctypes.byref(q))
            self.ptrs.append(int(q.value))
            self._mapped.append(int(q.value))

    def tensor(self, dtype, numel, offset_bytes=0, device=None):
        """The rank's OWN allocation (a range of it) as a torch tensor -- no copy, no ownership: keep this object alive."""
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        assert offset_bytes + nbytes <= self.nbytes
        t = torch.as_tensor(_DevArray(self.local + offset_bytes, nbytes), device=device or torch.device("cuda", torch.cuda.current_device()))
        assert t.data_ptr() == self.local + offset_bytes, "torch copied the range instead of aliasing it"
        return t.view(dtype)

    def close(self):
        from . import _lib as L
        for q in self._mapped:
            L.call("n2m_peer_unmap", q)
        self._mapped = []
        if self.local:
            L.call("n2m_peer_free", self.local)
            self.local = 0


class PeerExchange:
    """The sharded optimizer's exchange without a collective in the data path (include/n2m_peer.h):

      gradients   the table backward's flush stores every row into the slot its OWNER keeps for this rank (`route()` names the slots to
                  n2m_grid_backward_peer_route); signal_grad(half) behind each level half; the owner waits for all W signals of a half
                  and sums its W slots in rank order (reduce(half, g1, g2)) into the slices its Adam pass reads.
      parameters  after Adam, push_rows(half, packed) stores the rank's refreshed packed rows into every rank's packed table and signals;
                  wait_rows(half) in front of whatever reads the table next (the lookup of that level half).

    Row layout = engine.Stage0Engine's shards: coarse half [0, split) in W chunks of rows_c, fine half in chunks of rows_f.  Flags are
    epoch counters (one step = one epoch, begin_step()): a wait is "flag >= epoch", nothing is ever reset.  Why a slot is free when it is
    written again: a rank writes step e+1's gradients only after its step-e+1 lookup, which waited for every owner's step-e rows, which an
    owner sends after the Adam pass that follows its step-e reduction.  Why the packed table is quiet while a rank reads it: an owner sends
    step-e rows after its reduction, which waited for every rank's step-e gradient signals, which follow that rank's last read of the table
    (its table backward's TV stencil).  A rank without samples stores zeros into its slots (zero_slots) after waiting for the rows.
    Built and tested between two processes on ONE GPU; not run over xGMI -- opt-in (N2M_PEER_STORE=1)."""
    GF, GC, RC, RF, SM = 0, 1, 2, 3, 4     # flag kinds: gradients fine / coarse half, rows coarse / fine half, the small bucket
    KINDS = 5

    def __init__(self, rank, world, rows, split, rows_c, rows_f, device, group=None, timeout_ms=None, small_n=0):
        import os
        # rows_c = rows of a coarse CHUNK (slot size); the last rank's chunk may be shorter (chunks padded to a multiple of four rows)
        assert (world - 1) * rows_c < split <= world * rows_c and rows - split == world * rows_f and world <= 8
        self.rank, self.world, self.rows, self.split, self.rows_c, self.rows_f, self.device = rank, world, rows, split, rows_c, rows_f, device
        al = lambda n: (n + 255) & ~255
        n = {"c": rows_c, "f": rows_f}
        # staging: per half one fp32 region [W slots][rows of the half] and one fp16-pair region; then the packed table (8 bytes per row)
        self.off, o = {}, 0
        for h in ("c", "f"):
            self.off["s1" + h] = o; o += al(world * n[h] * 4)
            self.off["s2" + h] = o; o += al(world * n[h] * 4)
        self.off["pk"] = o; o += al(rows * 8)
        self.small_n = int(small_n)            # fp32 words every rank hands to every rank per step (MLP weight gradients + the non-finite flag)
        self.off["sm"] = o; o += al(world * max(self.small_n, 1) * 4)
        self.data = PeerMemory(o, False, rank, world, group)
        self.flags = PeerMemory(4096, True, rank, world, group)          # [KINDS][W] uint32 epoch counters, then the error word
        self.epoch = 0
        self.timeout_ms = int(timeout_ms if timeout_ms is not None else os.environ.get("N2M_PEER_TIMEOUT_MS", "10000"))
        self._err_off = self.KINDS * world * 4
        self._n = n
        self.packed = self.data.tensor(torch.float32, rows * 2, self.off["pk"], device).view(rows, 2)
        self._flag_t = self.flags.tensor(torch.int32, self.KINDS * world + 1, 0, device)
        dist.barrier(group=group)

    # ---- gradients
    def route(self):
        from . import _lib as L
        r = L.PeerRoute()
        r.world, r.split_row, r.rows_c, r.rows_f = self.world, self.split, self.rows_c, self.rows_f
        for hi, h in enumerate(("c", "f")):
            for owner in range(self.world):
                base = self.data.ptrs[owner]
                r.g1[hi][owner] = base + self.off["s1" + h] + self.rank * self._n[h] * 4
                r.g2[hi][owner] = base + self.off["s2" + h] + self.rank * self._n[h] * 4
        return r

    def begin_step(self):
        self.epoch += 1
        return self.epoch

    def _signal(self, kind):
        import ctypes
        from . import _lib as L
        p = L.PeerPtrs()
        p.count = self.world
        for dst in range(self.world):
            p.ptr[dst] = self.flags.ptrs[dst] + (kind * self.world + self.rank) * 4
        L.call("n2m_peer_signal", ctypes.byref(p), self.epoch, L.stream())

    def _wait(self, kind):
        from . import _lib as L
        L.call("n2m_peer_wait", self.flags.local + kind * self.world * 4, self.world, 1, self.epoch, self.timeout_ms, self.flags.local + self._err_off,
               L.stream())

    def signal_grad(self, half):
        self._signal(self.GF if half == "f" else self.GC)

    def zero_slots(self):
        """A rank whose batch is empty: zeros into its slot on every owner, both halves, both tables."""
        import ctypes
        from . import _lib as L
        for h in ("f", "c"):
            for key in ("s1", "s2"):
                p = L.PeerPtrs()
                p.count = self.world
                for owner in range(self.world):
                    p.ptr[owner] = self.data.ptrs[owner] + self.off[key + h] + self.rank * self._n[h] * 4
                L.call("n2m_peer_copy", None, ctypes.byref(p), self._n[h] * 4, L.stream())
            self.signal_grad(h)

    def reduce(self, half, g1, g2):
        """Owner side: wait for every rank's signal of this half, then g1 / g2 = the W slots summed in rank order."""
        from . import _lib as L
        self._wait(self.GF if half == "f" else self.GC)
        n = self._n[half]
        L.call("n2m_peer_reduce_slices", self.data.local + self.off["s1" + half], self.data.local + self.off["s2" + half], self.world, n, L.ptr(g1), L.ptr(g2),
               None, L.stream())

    # ---- the fused form: n2m_adam_step_peer sums the slots in its gradient load and stores the packed rows to every rank itself
    def wait_grad(self, half):
        """Owner side of the fused form: the W ranks' rows of this half have arrived (the wait reduce() starts with)."""
        self._wait(self.GF if half == "f" else self.GC)

    def signal_rows(self, half):
        """Behind n2m_adam_step_peer: this rank's refreshed packed rows of the half are in every rank's table (push_rows() without its copy)."""
        self._signal(self.RC if half == "c" else self.RF)

    def adam_peer(self, entries):
        """N2mAdamPeer for a descriptor whose entry k takes its gradient from the slots of (table, half) = entries[k] ("s1" | "s2", "c" | "f"),
        or from its own grad pointer where entries[k] is None.  Pointers are fixed for the life of the exchange."""
        from . import _lib as L
        a = L.AdamPeer()
        a.world = self.world
        for k, ent in enumerate(entries):
            if ent is None:
                continue
            key, h = ent
            for sl in range(self.world):
                a.slots[k][sl] = self.data.local + self.off[key + h] + sl * self._n[h] * 4
        a.packed_local = self.data.local + self.off["pk"]
        r = 0
        for dst in range(self.world):
            if dst != self.rank:
                a.packed_remote[r] = self.data.ptrs[dst] + self.off["pk"]
                r += 1
        a.n_remote = r
        return a

    # ---- the small bucket: every rank's copy to every rank, summed in rank order everywhere (bit-identical results without a collective)
    def all_sum_small(self, t):
        """t (contiguous fp32, small_n values) <- sum over ranks of t, in rank order."""
        import ctypes
        from . import _lib as L
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == self.small_n
        p = L.PeerPtrs()
        p.count = self.world
        for dst in range(self.world):
            p.ptr[dst] = self.data.ptrs[dst] + self.off["sm"] + self.rank * self.small_n * 4
        L.call("n2m_peer_copy", t.data_ptr(), ctypes.byref(p), self.small_n * 4, L.stream())
        self._signal(self.SM)
        self._wait(self.SM)
        L.call("n2m_peer_reduce_slices", self.data.local + self.off["sm"], None, self.world, self.small_n, t.data_ptr(), None, None, L.stream())

    # ---- parameters
    def push_rows(self, half, row0, n):
        """This rank's refreshed packed rows [row0, row0 + n) into every other rank's packed table, then the signal (own flag included)."""
        import ctypes
        from . import _lib as L
        p = L.PeerPtrs()
        k = 0
        for dst in range(self.world):
            if dst != self.rank:
                p.ptr[k] = self.data.ptrs[dst] + self.off["pk"] + row0 * 8
                k += 1
        p.count = k
        if k:
            L.call("n2m_peer_copy", self.data.local + self.off["pk"] + row0 * 8, ctypes.byref(p), n * 8, L.stream())
        self._signal(self.RC if half == "c" else self.RF)

    def wait_rows(self, half):
        self._wait(self.RC if half == "c" else self.RF)

    class _Token:
        def __init__(self, ex, half):
            self.ex, self.half = ex, half

        def wait(self):
            self.ex.wait_rows(self.half)

    def rows_tokens(self):
        return {"c": self._Token(self, "c"), "f": self._Token(self, "f")}

    def fold_error_into(self, found_inf):
        """Device side: found_inf += (error word != 0), so that the optimizer skips a step whose exchange timed out (its slots are stale);
        and the word starts its way to pinned host memory for raise_if_failed()."""
        word = self._flag_t[self.KINDS * self.world:self.KINDS * self.world + 1]
        found_inf.add_((word != 0).to(found_inf.dtype).view_as(found_inf))
        if getattr(self, "_err_host", None) is None:
            self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._err_host.copy_(word, non_blocking=True)

    def raise_if_failed(self):
        """Host side, non-blocking: raises when an EARLIER fold_error_into() has delivered a non-zero error word (normally the last step's)."""
        h = getattr(self, "_err_host", None)
        if h is not None and int(h[0]):
            raise RuntimeError(f"peer-store exchange: rank {self.rank} timed out waiting for rank {int(h[0]) - 1} (epoch <= {self.epoch}); the step "
                               "was skipped on the device (found_inf)")

    def check(self):
        """Host read of the error word: raises when a wait ran into its timeout (a peer never signalled)."""
        torch.cuda.synchronize()
        e = int(self._flag_t[self.KINDS * self.world])
        if e:
            raise RuntimeError(f"peer-store exchange: rank {self.rank} timed out waiting for rank {(e - 1)} (epoch {self.epoch})")

    def close(self):
        torch.cuda.synchronize()
        self.packed = self._flag_t = None
        self.flags.close()
        self.data.close()
