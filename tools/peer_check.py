#!/usr/bin/env python3
"""Two processes, one GPU: the n2m_peer_* primitives of include/n2m_peer.h (launch with torch.distributed.run, N2M_DIST_BACKEND=gloo).
  1. mapping: rank r fills its own buffer with r + 1, every rank then stores into the OTHER rank's buffer through the mapped pointer
     (n2m_peer_copy), signals, waits, and reads what the peer stored into its own;
  2. the slot sum (n2m_peer_reduce_slices) == the rank-order sum of the W slots, fp32 and fp16 pairs, an odd number of rows per slot;
  3. a wait for an epoch nobody signals returns after its timeout with the error word set (and PeerExchange.check() raises).
Prints 'PEER_CHECK OK' on rank 0."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nerf2mesh_amd import _lib as L
from nerf2mesh_amd.parallel import PeerExchange, PeerMemory, init_from_env

rank, world, local = init_from_env()
dev = torch.device("cuda", local % torch.cuda.device_count())
torch.cuda.set_device(dev)
ok = True

# ---- 1. mapping + signal / wait
n = 1 << 16
mem = PeerMemory(n * 4 * world, False, rank, world)
flags = PeerMemory(4096, True, rank, world)
mine = mem.tensor(torch.float32, n * world, 0, dev)
mine.fill_(-1.0)
src = torch.full((n,), float(rank + 1), device=dev)
torch.cuda.synchronize()
dist.barrier()
p = L.PeerPtrs()
p.count = world
for dst in range(world):
    p.ptr[dst] = mem.ptrs[dst] + rank * n * 4            # slot `rank` of every rank's buffer
L.call("n2m_peer_copy", src.data_ptr(), ctypes.byref(p), n * 4, L.stream())
f = L.PeerPtrs()
f.count = world
for dst in range(world):
    f.ptr[dst] = flags.ptrs[dst] + rank * 4
L.call("n2m_peer_signal", ctypes.byref(f), 7, L.stream())
L.call("n2m_peer_wait", flags.local, world, 1, 7, 5000, flags.local + 1024, L.stream())
torch.cuda.synchronize()
want = torch.cat([torch.full((n,), float(r + 1)) for r in range(world)])
ok = ok and torch.equal(mine.cpu(), want)
ft = flags.tensor(torch.int32, 512, 0, dev)
ok = ok and int(ft[256]) == 0 and all(int(ft[r]) == 7 for r in range(world))

# ---- 2. slot sum, odd rows
rows = 1001
g = torch.Generator(device=dev).manual_seed(3)
s1 = torch.randn(world, rows, device=dev, generator=g)
s2 = torch.randn(world, rows, 2, device=dev, generator=g).half()
o1, o2 = torch.empty(rows, device=dev), torch.empty(rows, 2, device=dev, dtype=torch.float16)
L.call("n2m_peer_reduce_slices", s1.data_ptr(), s2.data_ptr(), world, rows, o1.data_ptr(), o2.data_ptr(), None, L.stream())
w1, w2 = torch.zeros(rows, device=dev), torch.zeros(rows, 2, device=dev)
for r in range(world):
    w1 = w1 + s1[r]
    w2 = w2 + s2[r].float()
ok = ok and torch.equal(o1, w1) and torch.equal(o2, w2.half())

# ---- 3. timeout
dist.barrier()
t0 = time.time()
L.call("n2m_peer_wait", flags.local, world, 1, 99, 300, flags.local + 1024, L.stream())      # epoch 99 is never signalled
torch.cuda.synchronize()
dt = time.time() - t0
ok = ok and int(ft[256]) == 1 and 0.25 <= dt < 5.0

# ---- the exchange object end to end on a toy layout: every rank contributes its rank + 1 everywhere
rows_c, rows_f = 512, 1024
ex = PeerExchange(rank, world, world * (rows_c + rows_f), world * rows_c, rows_c, rows_f, dev, timeout_ms=3000, small_n=777)
ex.begin_step()
ex.zero_slots()
for h, nrow in (("f", rows_f), ("c", rows_c)):
    g1, g2 = torch.full((nrow, 1), 5.0, device=dev), torch.full((nrow, 2), 5.0, device=dev, dtype=torch.float16)
    ex.reduce(h, g1, g2)
    torch.cuda.synchronize()
    ok = ok and float(g1.abs().sum()) == 0.0 and float(g2.float().abs().sum()) == 0.0
small = torch.arange(777, device=dev, dtype=torch.float32) * (rank + 1)          # the small bucket: same sum, same bits, on every rank
ex.all_sum_small(small)
torch.cuda.synchronize()
ok = ok and torch.equal(small.cpu(), torch.arange(777, dtype=torch.float32) * sum(r + 1 for r in range(world)))
ex.packed.fill_(float(rank + 1))
torch.cuda.synchronize()
dist.barrier()
ranges = {"c": (rank * rows_c, rows_c), "f": (world * rows_c + rank * rows_f, rows_f)}
for h in ("c", "f"):
    ex.push_rows(h, *ranges[h])
for h in ("c", "f"):
    ex.wait_rows(h)
ex.check()
pk = ex.packed.cpu()
for r in range(world):
    ok = ok and bool((pk[r * rows_c:(r + 1) * rows_c] == r + 1).all()) and bool((pk[world * rows_c + r * rows_f:world * rows_c + (r + 1) * rows_f] == r + 1).all())
ex.begin_step()
ex._wait(ex.GF)                      # nobody signalled epoch 2
try:
    ex.check()
    ok = False
except RuntimeError:
    pass

res = torch.tensor([1.0 if ok else 0.0], device=dev)
dist.all_reduce(res, op=dist.ReduceOp.MIN)
dist.barrier()
del mine, ft
ex.close(); mem.close(); flags.close()
if rank == 0:
    print(f"PEER_CHECK {'OK' if float(res) == 1.0 else 'FAILED'} world={world} timeout wait took {dt:.2f} s")
dist.destroy_process_group()
sys.exit(0 if float(res) == 1.0 else 1)
