"""Is the stage-0 step GPU-bound or host-bound?  Times (a) the host's enqueue time per step, (b) the time the host spends
blocked on the sample-count read-back (pipelined march), (c) the drain time after the last step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf2mesh_amd import _lib, synthetic, raymarching as R
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options
from nerf2mesh_amd.trainer import Stage0Trainer

torch.manual_seed(0)
opt = make_options(O=True, bound=1, dt_gamma=0, iters=30000, fused_mlp=True)
ENGINE = "--autograd" not in sys.argv
if ENGINE:
    from nerf2mesh_amd.engine import Stage0Engine
    tr = Stage0Engine(NeRFNetwork(opt), opt, synthetic.make_cameras(100, seed=0), torch.device("cuda", 0), seed=0)
else:
    tr = Stage0Trainer(NeRFNetwork(opt), opt, synthetic.make_cameras(100, seed=0), torch.device("cuda", 0), seed=0)
tr.mark_untrained()
tr.pipeline = "--no-pipeline" not in sys.argv
wait = [0.0]
if ENGINE:
    orig = tr._finish
    def timed(b):
        t0 = time.perf_counter(); b.count_ready.synchronize(); wait[0] += time.perf_counter() - t0
        return orig(b)
    tr._finish = timed
else:
    orig = R.march_rays_train_finish
    def timed(t):
        t0 = time.perf_counter(); t.event.synchronize(); wait[0] += time.perf_counter() - t0
        return orig(t)
    R.march_rays_train_finish = timed
for i in range(400): tr.train_step()
torch.cuda.synchronize()
wait[0] = 0; t0 = time.perf_counter(); c0 = time.thread_time()
for i in range(200): tr.train_step()
c1 = time.thread_time(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host CPU time of the stepping thread (excludes blocking waits): {1e3*(c1-c0)/200:.3f} ms/step")
print(f"driver={type(tr).__name__} pipeline={tr.pipeline} wall/step {1e3*(t2-t0)/200:.3f} ms  host loop/step {1e3*(t1-t0)/200:.3f}  blocked-on-count/step {1e3*wait[0]/200:.3f}  drain {1e3*(t2-t1):.2f} ms")
if "--cprofile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(200): tr.train_step()
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(170)
