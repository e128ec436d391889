"""Which collectives does the gloo backend run on CUDA tensors here?  (two ranks on one GPU: the multi-rank GPU tests use this set-up)"""
import os, torch, torch.distributed as dist, torch.multiprocessing as mp
def run(rank, world):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29618"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    x = (torch.arange(8, dtype=torch.float32, device=dev) + rank)
    for name, fn in (("all_reduce", lambda: dist.all_reduce(x.clone())),
                     ("reduce_scatter_tensor f32", lambda: dist.reduce_scatter_tensor(torch.zeros(4, device=dev), x)),
                     ("reduce_scatter_tensor f16", lambda: dist.reduce_scatter_tensor(torch.zeros(4, device=dev, dtype=torch.float16), x.half())),
                     ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(torch.zeros(16, device=dev), x)),
                     ("async reduce_scatter", lambda: dist.reduce_scatter_tensor(torch.zeros(4, device=dev), x, async_op=True).wait())):
        try:
            fn(); torch.cuda.synchronize()
            if rank == 0: print(name, "ok")
        except Exception as e:
            if rank == 0: print(name, "FAIL", type(e).__name__, str(e)[:120])
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(run, args=(2,), nprocs=2)
