#!/usr/bin/env python3
"""Stage-0 training throughput of the nerf2mesh hot path on MI355X (BASELINE.json: "train rays/sec & samples/sec,
Lego 800x800 stage-0").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one full stage-0 iteration of the reference's loop (nerf/utils.py:1152-1180) on the lego recipe
(`-O --bound 1 --dt_gamma 0`, scripts/runall_syn.sh:1): [every 16th step: 128^3 occupancy refresh] -> 4096..N adaptive
rays over 100 synthetic 800x800 views -> near/far -> occupancy march -> 2 hash-grid encodes + 3 tiny MLPs (fp16 autocast)
-> compositing -> MSE/mask/specular loss -> backward -> TV gradient -> Adam over 18.4 M parameters.  Nothing is skipped
inside the timed region.  Data: synthetic lego-like scene (nerf2mesh_amd/synthetic.py), random-init network that is
first trained for `--pretrain` untimed iterations so that the occupancy grid is in its pruned, steady state.

WHICH step: the reference trains its first `diffuse_step` = 1000 iterations with diffuse-only shading (main.py:59,
nerf/utils.py:669-672: no specular branch, no specular loss) and the other 29 000 of 30 000 with shading = 'full' (specular MLP +
lambda_specular * mean(sum(specular^2)), nerf/utils.py:733-737).  The default `--pretrain 1000` puts warm-up and timed region BEHIND that
switch: the line reports the steady-state step (`config.shading` = "full").  `--diffuse` times the warm-up phase instead (pretrain 300,
steps 306.. as rounds 1-2 reported) and says so in `config.shading`.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant HIP kernel of the step: algorithmic bytes per launch / mean launch duration, measured with
                  hipEvents attached to the kernel dispatches on the launch stream inside the timed region (n2m_prof_*), against the
                  8 TB/s HBM peak and against a stream copy timed in the same process (peak_measured)
  cpu_baseline -- the CPU oracle (oracle/n2m_oracle.c, OpenMP, 32 threads) + PyTorch-CPU MLPs timed on a bounded sample of the same
                  workload (rank 0, N=1 only); cpu_baseline_reference -- the same iteration on the reference's own kernels compiled for the
                  host (oracle/_ref, one core)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); measured float4-copy ceiling is 6290 GB/s


# device kernels behind each library entry point, for the PMC traffic lookup: (substrings of the kernel names, how their
# per-launch byte counts combine into one call of the entry point)
_DEVICE_KERNELS = {"grid_encode_backward": (("pm_fill_pair_kernel", "pm_accumulate_both_kernel"), "sum"),     # one call = fill + the accumulate of both tables
                   "grid_encode_forward_packed": (("grid_forward3_packed_kernel",), "sum"),                  # one call = both tables (packed copy)
                   "adam_step": (("adam_kernel",), "sum"),
                   "mlp_backward": (("field_backward_pc_kernel<true, true>", "field_backward_pc_kernelILb1ELb1", "dw_finalize_kernel"), "sum"),
                   "mlp_forward": (("field_forward_kernel<true, true>", "field_forward_kernelILb1ELb1"), "mean"),
                   "march_rays_train_count": (("march_train_record_kernel",), "mean"),                      # one march per ray: count + recorded chunks,
                   "march_rays_train_write": (("march_train_replay_kernel",), "mean")}                      # then the replay that writes the samples


def pmc_traffic(entry_point):
    """(bytes, source) -- memory-side bytes per call of `entry_point` from the newest committed rocprofv3 --pmc passes
    (profiles/r*_pmc_traffic.json, made by tools/pmc_traffic.py in separate FETCH_SIZE / WRITE_SIZE passes over this same bench
    command) summed over the device kernels behind the entry point.  Counters cannot be read from inside the timed process, hence
    the file; (None, reason) when no file names the kernels this build launches.  The figure is an UPPER BOUND on HBM traffic:
    FETCH_SIZE is doubled per the gfx950 correction of MI355X_MICROARCH.md (calibrated there on wide coalesced reads, not on 8/16-byte
    gathers) and both counters include requests the Infinity Cache serves."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")))
    if not files or entry_point not in _DEVICE_KERNELS:
        return None, "no profiles/r*_pmc_traffic.json"
    path = files[-1]
    keys, how = _DEVICE_KERNELS[entry_point]
    vals = [v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"] for name, v in json.load(open(path)).items()
            if any(k in name for k in keys) and v.get("fetch_bytes_per_launch") is not None and v.get("write_bytes_per_launch") is not None]
    if not vals:
        return None, f"{os.path.basename(path)} has no kernel of {entry_point}"
    return float(sum(vals) if how == "sum" else sum(vals) / len(vals)), f"profiles/{os.path.basename(path)} (upper bound: FETCH_SIZE x2 + WRITE_SIZE, Infinity-Cache hits included)"


def cpu_baseline(n_rays=32768, reps=4, threads=None):
    """One stage-0 iteration's kernels on the host: oracle (C, OpenMP) for march/encode/composite/TV, torch-CPU for the
    MLPs.  Returns dict for the JSON line.  Bounded: ~n_rays rays of the same synthetic workload.
    threads: OpenMP/torch threads (shared runtime).  Default min(cores, 32): with all 256 hardware threads of the GPU box the
    fork/join cost of the many small parallel regions made the same code 40x slower (measured 3.3 k vs 610 k samples/s)."""
    from oracle import oracle as orc
    from nerf2mesh_amd import synthetic as S
    cores = min(os.cpu_count() or 1, 32) if threads is None else int(threads)
    torch.set_num_threads(cores)
    poses = S.make_cameras(100, seed=0)
    grid = S.scene_density_grid(H=128)
    bits = orc.packbits(grid.numpy(), 10.0)
    o, d = S.random_rays(poses, n_rays, torch.Generator().manual_seed(1))
    # rays that hit the object dominate the work; keep the natural mix
    o, d = o.numpy(), d.numpy()
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    pls = float(np.exp2(np.log2(2048 / 16) / 15))
    offs = orc.level_offsets(3, 16, pls, 16, 19)
    S_ = float(np.log2(pls))
    rng = np.random.default_rng(0)
    emb1 = ((rng.random((int(offs[-1]), 1), dtype=np.float32) * 2 - 1) * 1e-4)
    emb2 = ((rng.random((int(offs[-1]), 2), dtype=np.float32) * 2 - 1) * 1e-4)
    sigma_net = torch.nn.Sequential(torch.nn.Linear(19, 32, bias=False), torch.nn.ReLU(), torch.nn.Linear(32, 1, bias=False))
    color_net = torch.nn.Sequential(torch.nn.Linear(35, 64, bias=False), torch.nn.ReLU(), torch.nn.Linear(64, 64, bias=False),
                                    torch.nn.ReLU(), torch.nn.Linear(64, 6, bias=False))
    spec_net = torch.nn.Sequential(torch.nn.Linear(6, 32, bias=False), torch.nn.ReLU(), torch.nn.Linear(32, 3, bias=False))
    best, M = None, 0
    for _ in range(reps):
        t0 = time.perf_counter()
        nears, fars = orc.near_far_from_aabb(o, d, aabb, 0.05)
        noises = rng.random(n_rays).astype(np.float32)
        xyzs, dirs, ts, rays = orc.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, noises, 0.0, 1024)
        M = xyzs.shape[0]
        x01 = (xyzs + 1) / 2
        f1 = orc.grid_encode_forward(x01, emb1, offs, S_, 16, sample_major=True)
        f2 = orc.grid_encode_forward(x01, emb2, offs, S_, 16, sample_major=True)
        xt = torch.from_numpy(xyzs)
        h1 = torch.from_numpy(f1).requires_grad_(True)
        h2 = torch.from_numpy(f2).requires_grad_(True)
        sig = torch.exp(sigma_net(torch.cat([xt, h1], -1))[:, 0])
        geo = torch.sigmoid(color_net(torch.cat([xt, h2], -1)))
        dn = torch.from_numpy(dirs)
        dn = dn / dn.norm(dim=-1, keepdim=True)
        spec = torch.sigmoid(spec_net(torch.cat([dn, geo[:, 3:]], -1)))
        rgb = (spec + geo[:, :3]).clamp(0, 1)
        w, ws, dp, im = orc.composite_rays_train_forward(sig.detach().numpy(), rgb.detach().numpy(), ts, rays)
        gi = (2 * (im - 0.5) / im.size).astype(np.float32)
        gs, gr = orc.composite_rays_train_backward(np.zeros(M, np.float32), np.zeros(n_rays, np.float32), np.zeros(n_rays, np.float32),
                                                   gi, sig.detach().numpy(), rgb.detach().numpy(), ts, rays, ws, dp, im)
        torch.autograd.backward([sig, rgb], [torch.from_numpy(gs), torch.from_numpy(gr)])
        g1 = orc.grid_encode_backward(h1.grad.numpy(), x01, emb1, offs, S_, 16, sample_major=True)
        orc.grid_encode_backward(h2.grad.numpy(), x01, emb2, offs, S_, 16, sample_major=True)
        orc.grad_total_variation(x01, emb1, g1, offs, 1e-8, S_, 16)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": M / best, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"1 stage-0 iteration (march+2 encodes+MLPs+composite fwd/bwd+TV, no Adam/occupancy refresh) on {n_rays} rays = "
                      f"{M} samples, best of {reps}; oracle C/OpenMP + torch-CPU MLPs"}


def cpu_baseline_reference(n_rays=2048, reps=2):
    """The same iteration on the REFERENCE's own kernels: raymarching.cu / gridencoder.cu compiled for the host by oracle/build_ref.py
    (oracle/_ref: every arithmetic statement is the reference's; a launch is a serial sweep over blockIdx / threadIdx, so this is ONE core)
    + torch-CPU MLPs.  kind = "reference".  Bounded: ~n_rays rays of the same synthetic workload."""
    from oracle import build_ref
    from oracle import oracle as orc
    from nerf2mesh_amd import synthetic as S
    if not build_ref.available():
        return {"value": None, "unit": "samples/s", "cores": 1, "kind": "reference", "sample": "oracle/_ref is not built on this box"}
    rm, ge, _ = build_ref.load()
    threads_before = torch.get_num_threads()
    torch.set_num_threads(1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    poses = S.make_cameras(100, seed=0)
    bits = orc.packbits(S.scene_density_grid(H=128).numpy(), 10.0)
    o, d = S.random_rays(poses, n_rays, torch.Generator().manual_seed(1))
    N = n_rays
    aabb = t(np.array([-1, -1, -1, 1, 1, 1], np.float32))
    pls = float(np.exp2(np.log2(2048 / 16) / 15))
    offs = t(orc.level_offsets(3, 16, pls, 16, 19))
    S_ = float(np.log2(pls))
    rng = np.random.default_rng(0)
    rows = int(offs[-1])
    emb1 = t((rng.random((rows, 1), dtype=np.float32) * 2 - 1) * 1e-4)
    emb2 = t(((rng.random((rows, 2), dtype=np.float32) * 2 - 1) * 1e-4).astype(np.float16))      # the colour table is cast to half (grid.py:45)
    sigma_net = torch.nn.Sequential(torch.nn.Linear(19, 32, bias=False), torch.nn.ReLU(), torch.nn.Linear(32, 1, bias=False))
    color_net = torch.nn.Sequential(torch.nn.Linear(35, 64, bias=False), torch.nn.ReLU(), torch.nn.Linear(64, 64, bias=False),
                                    torch.nn.ReLU(), torch.nn.Linear(64, 6, bias=False))
    spec_net = torch.nn.Sequential(torch.nn.Linear(6, 32, bias=False), torch.nn.ReLU(), torch.nn.Linear(32, 3, bias=False))
    best, M = None, 0
    for _ in range(reps):
        t0 = time.perf_counter()
        nears, fars = torch.zeros(N), torch.zeros(N)
        rm.near_far_from_aabb(o, d, aabb, N, 0.05, nears, fars)
        noises = torch.rand(N)
        rays, counter = torch.zeros(N, 2, dtype=torch.int32), torch.zeros(1, dtype=torch.int32)
        margs = (o, d, t(bits), 1.0, False, 0.0, 1024, N, 1, 128, nears, fars)
        rm.march_rays_train(*margs, None, None, None, rays, counter, noises)
        M = int(counter[0])
        xyzs, dirs, ts = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
        rm.march_rays_train(*margs, xyzs, dirs, ts, rays, counter, noises)
        x01 = (xyzs + 1) / 2
        f1 = torch.zeros(16, M, 1); f2 = torch.zeros(16, M, 2, dtype=torch.float16)
        ge.grid_encode_forward(x01, emb1, offs, f1, M, 3, 1, 16, 16, S_, 16, None, 0, False, 0)
        ge.grid_encode_forward(x01, emb2, offs, f2, M, 3, 2, 16, 16, S_, 16, None, 0, False, 0)
        h1 = f1.permute(1, 0, 2).reshape(M, 16).requires_grad_(True)
        h2 = f2.permute(1, 0, 2).reshape(M, 32).float().requires_grad_(True)
        sig = torch.exp(sigma_net(torch.cat([xyzs, h1], -1))[:, 0])
        geo = torch.sigmoid(color_net(torch.cat([xyzs, h2], -1)))
        dn = dirs / dirs.norm(dim=-1, keepdim=True)
        spec = torch.sigmoid(spec_net(torch.cat([dn, geo[:, 3:]], -1)))
        rgb = (spec + geo[:, :3]).clamp(0, 1)
        w, ws, dp, im = torch.zeros(M), torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
        sd, rd = sig.detach().contiguous(), rgb.detach().contiguous()
        rm.composite_rays_train_forward(sd, rd, ts, rays, M, N, 1e-4, False, w, ws, dp, im)
        gi = (2 * (im - 0.5) / im.numel()).contiguous()
        gs, gr = torch.zeros(M), torch.zeros(M, 3)
        rm.composite_rays_train_backward(torch.zeros(M), torch.zeros(N), torch.zeros(N), gi, sd, rd, ts, rays, ws, dp, im, M, N, 1e-4, False, gs, gr)
        torch.autograd.backward([sig, rgb], [gs, gr])
        g1, g2 = torch.zeros_like(emb1), torch.zeros_like(emb2)
        d1 = h1.grad.reshape(M, 16, 1).permute(1, 0, 2).contiguous()
        d2 = h2.grad.reshape(M, 16, 2).permute(1, 0, 2).contiguous().half()
        ge.grid_encode_backward(d1, x01, emb1, offs, g1, M, 3, 1, 16, 16, S_, 16, None, None, 0, False, 0)
        ge.grid_encode_backward(d2, x01, emb2, offs, g2, M, 3, 2, 16, 16, S_, 16, None, None, 0, False, 0)
        ge.grad_total_variation(x01, emb1, g1, offs, 1e-8, M, 3, 1, 16, S_, 16, 0, False)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    torch.set_num_threads(threads_before)
    return {"value": M / best, "unit": "samples/s", "cores": 1, "kind": "reference",
            "sample": f"1 stage-0 iteration (march+2 encodes+MLPs+composite fwd/bwd+TV, no Adam/occupancy refresh) on {n_rays} rays = {M} samples, "
                      f"best of {reps}: {best:.2f} s; the reference's raymarching.cu / gridencoder.cu compiled for the host (oracle/_ref, serial sweep) "
                      "+ torch-CPU MLPs on one thread"}


def measured_stream_peak(device, mbytes=1024, reps=20):
    """GB/s of a device-to-device copy of `mbytes` MB (read + write counted) by the library's own 16-bytes-per-lane grid-stride kernel
    (n2m_stream_copy, csrc/runtime.hip -- the form MI355X_MICROARCH.md measures 6.29 TB/s with), in THIS process on THIS device, best of a
    few grid sizes: the practical streaming ceiling the nominal 8 TB/s is never reached at (SURVEY 8d asks for it as a second denominator).
    Until round 4 this timed torch.Tensor.copy_ (5.2 TB/s) -- below what the step's own Adam kernel streams at, so fractions of it flattered."""
    from nerf2mesh_amd import _lib
    nbytes = mbytes * (1 << 20)
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    b = torch.empty_like(a)
    best = 0.0
    for wg in (0, 1024, 4096, 8192):
        for _ in range(3):
            _lib.call("n2m_stream_copy", a.data_ptr(), b.data_ptr(), nbytes, wg, _lib.stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _lib.call("n2m_stream_copy", a.data_ptr(), b.data_ptr(), nbytes, wg, _lib.stream())
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    assert torch.equal(a[-4096:], b[-4096:])
    return best


def dropin_reference_loop(device, steps=50, warmup=20, pretrain=300, fused_field=False, return_losses=False):
    """What a reference user gets by putting nerf2mesh_amd/backends/_*.py on sys.path and changing nothing else: the reference's OWN
    `Trainer.train_step` / `post_train_step` (nerf/utils.py:628-823) inside the loop of `train_one_epoch` (:1152-1180: occupancy refresh every
    16 steps, zero_grad, scaler.scale(loss).backward(), TV, scaler.step, scaler.update, LambdaLR step, loss.item()) with main.py:221's
    torch.optim.Adam(eps=1e-15) and GradScaler, over the unchanged nerf/renderer.py + nerf/network.py + autograd wrappers -- all from the
    byte-compiled copy of the reference's Python (oracle/_ref/pyref; the checkout itself where present) -- on libn2m_hip.so.  Lego recipe
    (-O --bound 1 --dt_gamma 0), the synthetic views, adaptive num_rays.  A reported comparison like cpu_baseline: nothing of it is shipped."""
    import types
    from oracle import ref_python as RP
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.options import make_options
    if not RP.available():
        return {"value": None, "note": "reference Python not available on this box (oracle/_ref/pyref not built)"}
    ns = RP.load("hip")
    RP.use_backend("hip")
    if fused_field:      # opt-in: the fused MFMA field behind the unchanged class (`install(fused_mlp=True)`)
        from nerf2mesh_amd import backends
        backends.fuse_field(ns.network.NeRFNetwork)
    torch.manual_seed(0)
    d = dict(vars(RP.reference_opt()))
    d.update(vars(make_options(O=True, bound=1, dt_gamma=0, iters=30000)))
    for k in ("scene", "fused_mlp", "enable_cam_near_far"):
        d.pop(k, None)
    d.update(bound=1.0, data_format="nerf", lambda_depth=0.0)
    opt = types.SimpleNamespace(**d)
    model = ns.network.NeRFNetwork(opt).cuda()
    optimizer = torch.optim.Adam(model.get_params(opt.lr), eps=1e-15)                                     # main.py:221
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda it: 0.01 + 0.99 * (it / 500) if it <= 500 else 0.1 ** ((it - 500) / (opt.iters - 500)))
    T = ns.utils.Trainer
    me = types.SimpleNamespace(opt=opt, model=model, global_step=0, device=device, criterion=torch.nn.MSELoss(reduction="none"), optimizer=optimizer,
                               scaler=torch.cuda.amp.GradScaler(enabled=True), tmp_xyzs=None)
    poses = synthetic.make_cameras(100, seed=0).to(device)
    images = synthetic.preload_images(poses, synthetic.boxes(device, "lego"))
    gen = torch.Generator(device=device).manual_seed(0)
    model.train()
    samples = rays = 0

    losses = []

    def step():
        nonlocal samples, rays
        if me.global_step % opt.update_extra_interval == 0:                                               # nerf/utils.py:1155-1156
            model.update_extra_state()
        me.global_step += 1
        optimizer.zero_grad()
        o, dd, rgba = synthetic.random_batch(poses, images, int(opt.num_rays), gen)
        n = o.shape[0]
        _, _, loss = T.train_step(me, {"rays_o": o, "rays_d": dd, "index": [0], "images": rgba})
        me.scaler.scale(loss).backward()                                                                  # :1172
        T.post_train_step(me)                                                                             # :1174 (unscale + in-place TV)
        me.scaler.step(optimizer)
        me.scaler.update()
        scheduler.step()
        losses.append(loss.item())                                                                        # :1182 (the loop's host read-back)
        samples += int(me.tmp_xyzs.shape[0]) if me.tmp_xyzs is not None else 0
        rays += n
    for _ in range(pretrain + warmup):
        step()
    torch.cuda.synchronize()
    samples = rays = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"ms_per_step": 1e3 * dt / steps, "value": samples / dt, "unit": "samples/s", "rays_per_sec": rays / dt, "steps": steps,
            "samples_per_step": samples / steps, "first_timed_step": pretrain + warmup + 1,
            **({"losses": losses} if return_losses else {}),
            "field": ("fused MFMA field behind the unchanged NeRFNetwork (backends.fuse_field, opt-in)" if fused_field else
                      "the reference's own nn.Linear graph"),
            "what": "the reference's unchanged Python (Trainer.train_step / post_train_step, render, NeRFNetwork, autograd wrappers; byte-compiled copy "
                    "oracle/_ref/pyref) + torch.optim.Adam + GradScaler over nerf2mesh_amd/backends/_*.py (libn2m_hip.so): the drop-in path of "
                    "INTEGRATION.md section A, nothing fused, nothing restated; shading "
                    + ("full" if pretrain + warmup + 1 >= int(opt.diffuse_step) else "diffuse (step < diffuse_step)")}


def cpu_baseline_stage1(v, f, reps=2):
    """The scalar C rasteriser / interpolator / antialiaser of oracle/n2m_raster_oracle.c (SURVEY 8d's stage-1 CPU baseline) on ONE view
    of the same mesh at the same resolution: rasterize (bbox form, bit-identical to the brute-force oracle) + interpolate(xyz) +
    interpolate(ones) + antialias(alpha) + antialias(rgb), forward only, one core."""
    from oracle import oracle as orc
    from nerf2mesh_amd import synthetic as S
    poses = S.make_cameras(100, seed=0)
    mvp = S.mvp_matrix(poses[0], 800, 800)
    vc = (torch.cat([v.cpu(), torch.ones(v.shape[0], 1)], 1) @ mvp.T).numpy()
    vn, fn = v.cpu().numpy(), f.cpu().numpy()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        rast = orc.rasterize(vc, fn, 1600, 1600, bbox=True)
        xyz = orc.interpolate(vn, rast, fn)
        mask = orc.interpolate(np.ones((vn.shape[0], 1), np.float32), rast, fn)
        orc.antialias(mask, rast, vc, fn)
        orc.antialias(xyz, rast, vc, fn)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": 800 * 800 / best, "unit": "output pixels/s", "cores": 1, "kind": "port",
            "sample": f"raster operators of ONE 800x800 view at 1600x1600 (rasterize + 2 interpolate + 2 antialias, forward only, no shading / "
                      f"backward / optimizer), {fn.shape[0]} faces, best of {reps}: {best:.2f} s; scalar C (oracle/n2m_raster_oracle.c)"}


def bench_stage1(args, rank, world, device):
    """One step = render_stage1 of one 800x800 view at 1600x1600 (ssaa 2) on a ~300k-face mesh + loss + backward + Adam
    (nerf/utils.py:708-721, nerf/renderer.py:816-921); views shard across ranks."""
    from nerf2mesh_amd import _lib, synthetic
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    from nerf2mesh_amd.trainer import Stage1Trainer
    torch.manual_seed(0)
    opt = make_options(O=True, bound=1, dt_gamma=0, stage=1, fused_mlp=not args.unfused)
    v, f = synthetic.scene_mesh(300000)
    tr = Stage1Trainer(NeRFNetwork(opt), opt, synthetic.make_cameras(100, seed=0), v, f, device, rank=rank, world_size=world)
    tr.preload()                     # rays + ground truth of all views on the device, as the reference's --preload
    from nerf2mesh_amd.engine_stage1 import Stage1Engine
    use_engine = not args.autograd and Stage1Engine.supported(tr)
    stepper = Stage1Engine(tr) if use_engine else tr      # the fixed launch sequence (engine_stage1.py) or the autograd trainer (--autograd)
    for _ in range(args.warmup):
        stepper.train_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if not args.no_prof:
        _lib.prof_reset(); _lib.prof_enable(args.prof_every)
    c0 = tr.covered_seen
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stepper.train_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    stats = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dt = float(stats[0])
    if rank == 0:
        px = 800 * 800 * args.steps * world
        HW = 1600.0 * 1600.0
        cov = (tr.covered_seen - c0) / args.steps            # covered (shaded) full-resolution pixels per view, measured
        V, F = float(v.shape[0]), float(f.shape[0])
        # ALGORITHMIC bytes per launch with the MEASURED coverage (SURVEY 8d's formulas; rounds 1-2 counted every pixel as covered, which
        # put interpolate above the HBM peak): per pixel what every pixel costs (the rast read, the output write), per COVERED pixel the
        # triangle's indices and vertex attributes.  interpolate runs twice per step (A = 3 positions, A = 1 mask): mean of the two.
        interp_f = lambda A: HW * (16 + 4 * A) + cov * (12 + 12 * A)
        # backward: the one launch per step is the mask's (A = 1; the positions are detached, nerf/renderer.py:878): rast read + grad_rast
        # written for every pixel, d_out / indices / attributes for covered pixels only
        interp_b = lambda A, rast_grad: HW * 16 + (16 * HW if rast_grad else 0) + cov * (4 * A + 12 + 12 * A)
        model = {"rasterize": 16 * V + 12 * F + 16 * HW, "rasterize_backward": 16 * HW + cov * 16 + 16 * V,
                 "interpolate_forward": 0.5 * (interp_f(3) + interp_f(1)), "interpolate_backward": interp_b(1, True),
                 "antialias_forward": 0.5 * ((8 * 1 + 16) + (8 * 3 + 16)) * HW, "antialias_backward": 0.5 * ((12 * 1 + 16) + (12 * 3 + 16)) * HW}
        kernels = {}
        for name in ("rasterize", "rasterize_backward", "interpolate_forward", "interpolate_backward", "antialias_forward", "antialias_backward",
                     "mlp_forward", "mlp_backward", "grid_encode_forward", "grid_encode_backward"):
            n, ms, by = _lib.prof_read(name)
            if n:
                seen = _lib.prof_seen(name)
                by_launch = model.get(name, by / n)
                gbps = by_launch / (ms / n * 1e-3) / 1e9 if ms > 0 else None
                kernels[name] = {"launches": n, "launches_seen": seen, "avg_us": 1e3 * ms / n, "ms_per_step": (ms / n) * seen / args.steps,
                                 "algo_bytes_per_launch": by_launch, "GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS if gbps else None}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"]) if kernels else None
        roof = None
        if dom:
            k = kernels[dom]
            roof = {"kernel": dom, "bound": "hbm", "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["frac_of_hbm_peak"],
                    "traffic": None, "avg_us": k["avg_us"], "algo_bytes_per_launch": k["algo_bytes_per_launch"],
                    "note": "algorithmic bytes per SURVEY.md 8d; raster operators with the measured covered-pixel count of the timed views"}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline_stage1(v, f)
            except Exception as e:
                cpu = {"value": None, "unit": "output pixels/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps({"metric": "stage1_train_pixels_per_sec", "value": px / dt, "unit": "output pixels/s (800x800 views, rendered at 1600x1600)",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 raster, f16 autocast shading",
                          "data": "synthetic", "config": {"workload": f"nerf_synthetic/lego stage-1 -O --bound 1: {f.shape[0]} faces, "
                                                                       f"{v.shape[0]} vertices, ssaa 2, refine error tracking on",
                                                          "covered_pixels_per_view": cov, "coverage": cov / HW,
                                                          "image_head": "fused (n2m_stage1_head)" if tr.fused_head else "torch graph",
                                                          "driver": "engine_stage1.Stage1Engine (fixed launch sequence)" if use_engine else "trainer.Stage1Trainer (torch.autograd)",
                                                          "parallelism": f"views sharded over {world} GPU(s)",
                                                          "parity": "caller pinned to the unchanged reference Python (tests/test_stage1_reference.py); the "
                                                                    "rasterize / interpolate / antialias operators themselves UNPINNED: nvdiffrast is not "
                                                                    "under /root/reference (tests/test_raster_*.py check them against the scalar oracle)"},
                          "roofline": roof, "kernels": kernels, "cpu_baseline": cpu}))
    if world > 1:
        dist.destroy_process_group()


def bench_pipeline(args, device):
    """BASELINE config 5 end to end on one GPU (scripts/runall_syn_sdf.sh:1-2): `--sdf` stage 0 (short schedule: iters0 steps, the ramps of
    nerf/utils.py:651-655 over its first half) -> export_stage0 (density volume -> device marching cubes at the zero level of the sdf,
    nerf/renderer.py:472-545 without the pymeshlab cleaning / decimation: SURVEY section 2 OUT) -> stage 1 WITHOUT --sdf on that mesh (the
    stage-0 weights stay, nerf/utils.py:585-588), step executor -> PSNR of the stage-1 RASTER render against the analytic ground truth.
    One JSON line: wall time and throughput per phase + both PSNRs."""
    import tempfile
    from nerf2mesh_amd import export, synthetic
    from nerf2mesh_amd.engine import Stage0Engine
    from nerf2mesh_amd.engine_stage1 import Stage1Engine
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    from nerf2mesh_amd.trainer import Stage1Trainer
    it0, it1, res = int(args.pipeline_iters0), int(args.pipeline_iters1), int(args.pipeline_resolution)
    torch.manual_seed(0)
    opt = make_options(O=True, bound=1, dt_gamma=0, sdf=True, iters=it0, fused_mlp=True)
    poses = synthetic.make_cameras(100, seed=0)
    model = NeRFNetwork(opt)
    t0 = time.perf_counter()
    model.to(device).init_double_sphere(iters=1024)              # (nerf/utils.py:594; the reference runs 8192 such steps)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    eng = Stage0Engine(model, opt, poses, device, seed=0)
    eng.mark_untrained()
    eng.train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it0 - 1):
        eng.train_step()
    torch.cuda.synchronize()
    t_s0 = time.perf_counter() - t0
    psnr0 = [eng.eval_psnr(cam=c, use_ema=True) for c in (0, 33, 66)]
    phases = {"stage0 (--sdf, step executor)": {"steps": it0, "wall_s": t_s0, "ms_per_step": 1e3 * t_s0 / (it0 - 1), "samples_per_sec": eng.samples_seen / t_s0,
                                               "rays_per_sec": eng.rays_seen / t_s0, "psnr_volume_render_ema_quarter_res": sum(psnr0) / len(psnr0),
                                               "sdf_pretraining_s": t_init}}
    with tempfile.TemporaryDirectory() as tmp:
        with eng.averaged_parameters():                          # the reference exports from the best (EMA) checkpoint
            t0 = time.perf_counter()
            meshes = model.export_stage0(os.path.join(tmp, "mesh_stage0"), resolution=res)
            torch.cuda.synchronize()
            t_ex = time.perf_counter() - t0
        v, f = meshes[0]
        phases["export_stage0 (device marching cubes)"] = {"wall_s": t_ex, "resolution": res, "vertices": int(v.shape[0]), "faces": int(f.shape[0]),
                                                            "note": "raw iso-surface: clean_mesh / decimate_mesh (pymeshlab) are out of scope"}
        rv, rt = export.read_ply(os.path.join(tmp, "mesh_stage0", "mesh_0.ply"))
    if rt.shape[0] == 0:
        print(json.dumps({"metric": "pipeline", "value": None, "error": "the stage-0 iso-surface is empty", "phases": phases}))
        return
    # stage 1 runs WITHOUT --sdf (runall_syn_sdf.sh:2): main.py's plain -O options on the stage-0 weights (nerf/utils.py:585-588 loads them model_only)
    opt = make_options(O=True, bound=1, dt_gamma=0, stage=1, iters=max(it1, 501), fused_mlp=True)
    model.opt = opt
    tr = Stage1Trainer(model, opt, poses, torch.from_numpy(rv), torch.from_numpy(rt), device)
    stepper = Stage1Engine(tr) if Stage1Engine.supported(tr) else tr
    stepper.train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it1 - 1):
        stepper.train_step()
    torch.cuda.synchronize()
    t_s1 = time.perf_counter() - t0
    # PSNR of the RASTER render (what stage 1 ships), white background, full resolution
    model.eval()
    ps = []
    with torch.no_grad():
        for vw in (0, 33, 66):
            rays_o, rays_d, rgba = tr._view(vw)
            with torch.autocast("cuda", dtype=torch.float16):
                out = model.render_stage1(rays_o, rays_d, tr.mvps[vw], tr.H, tr.W, bg_color=1, shading="full")
            gt = rgba[..., :3] * rgba[..., 3:] + (1 - rgba[..., 3:])
            ps.append(float(-10 * torch.log10(torch.mean((out["image"].reshape(-1, 3).float() - gt.reshape(-1, 3)) ** 2))))
    phases["stage1 (raster refinement, step executor)" if stepper is not tr else "stage1 (raster refinement, autograd trainer)"] = {
        "steps": it1, "wall_s": t_s1, "ms_per_step": 1e3 * t_s1 / (it1 - 1), "output_px_per_sec": 800 * 800 * (it1 - 1) / t_s1, "psnr_raster_render_full_res": sum(ps) / len(ps)}
    total = t_init + t_s0 + t_ex + t_s1
    print(json.dumps({"metric": "pipeline_wall_seconds", "value": total, "unit": "s", "n_gpus": 1, "higher_is_better": False, "data": "synthetic",
                      "config": {"workload": f"nerf_synthetic/lego --sdf stage 0 ({it0} steps) -> export_stage0 ({res}^3) -> stage 1 ({it1} steps), -O --bound 1 --dt_gamma 0, "
                                             "800x800 x 100 synthetic views, one GPU (BASELINE config 5's pipeline at a short schedule)"},
                      "phases": phases, "psnr_stage0_volume": sum(psnr0) / len(psnr0), "psnr_stage1_raster": sum(ps) / len(ps)}))


def other_configs(timeout_s=180, budget_s=480):
    """BASELINE configs 3, 4, 5 and the drop-in path, measured by this same script in child processes behind the headline window (rank 0, one
    GPU; ~50 timed steps each after the recipe's own pre-training): driver-visible evidence for what the headline line does not cover.
    Every entry is a summary of the child's own JSON line (or says why there is none); the headline fields are not touched."""
    import subprocess
    me = os.path.abspath(__file__)
    runs = {"sdf (config 5, stage 0, end of the schedule)": ["--recipe", "sdf", "--steps", "48", "--warmup", "8"],
            "garden (config 4's recipe)": ["--recipe", "garden", "--steps", "48", "--warmup", "8"],
            "stage1 (config 3)": ["--stage", "1", "--steps", "50", "--warmup", "10"],
            "dropin (unchanged reference Python over backends/_*.py, config 2's recipe)": ["--dropin", "--steps", "48", "--warmup", "16"],
            "dropin + opt-in fused field (backends.install(fused_mlp=True))": ["--dropin-fused", "--steps", "48", "--warmup", "16"],
            "pipeline (config 5 end to end, short schedule)": ["--pipeline"]}
    out = {}
    t_all = time.perf_counter()
    for name, extra in runs.items():
        cmd = [sys.executable, me] + extra + ["--no-cpu-baseline", "--no-other-configs"]
        t0 = time.perf_counter()
        if t0 - t_all > budget_s:      # the extras share ONE budget: a slow box must not push the headline line past the caller's patience
            out[name] = {"skipped": f"the extras' budget of {budget_s} s was used up by the entries before this one"}
            continue
        timeout_s = min(timeout_s, max(30.0, budget_s - (t0 - t_all)))
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
                continue
            j = json.loads(lines[-1])
            roof = j.get("roofline") or {}
            out[name] = {"ms_per_step": j.get("ms_per_step"), "value": j.get("value"), "unit": j.get("unit"), "steps": j.get("steps"),
                         "rays_per_sec": j.get("rays_per_sec"), "workload": (j.get("config") or {}).get("workload"),
                         "driver": (j.get("config") or {}).get("driver"), "sdf_schedule": (j.get("config") or {}).get("sdf_schedule"),
                         "dominant_kernel": {"kernel": roof.get("kernel"), "avg_us": roof.get("avg_us"), "frac": roof.get("frac")} if roof else None,
                         "what": j.get("what"), "field": j.get("field"), "phases": j.get("phases"), "psnr_stage0_volume": j.get("psnr_stage0_volume"),
                         "psnr_stage1_raster": j.get("psnr_stage1_raster"),
                         "wall_s": round(time.perf_counter() - t0, 1), "command": "python bench.py " + " ".join(extra)}
            out[name] = {k: v for k, v in out[name].items() if v is not None}
        except Exception as e:        # a reported extra: never lose the headline over it
            out[name] = {"error": repr(e)[:400]}
    return out


# ---- a first multi-GPU invocation must end in a LINE, not in a hang: RCCL / IPC set-up has never run on this code (one-GPU leases)
_WATCH = {"deadline": None, "phase": "start", "done": False}


def _error_line(msg):
    return json.dumps({"metric": "train_samples_per_sec", "value": None, "unit": "samples/s", "n_gpus": int(os.environ.get("WORLD_SIZE", "1")),
                       "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic", "error": msg[:1200],
                       "config": {"workload": "nerf_synthetic/lego stage-0 -O --bound 1 --dt_gamma 0", "phase": _WATCH["phase"]}})


def _phase(name, seconds):
    """Names the phase the run is in and the time it may take; the watchdog thread ends the process with an error line when it is exceeded."""
    _WATCH["phase"], _WATCH["deadline"] = name, time.monotonic() + seconds


def _start_watchdog():
    import threading

    def watch():
        while not _WATCH["done"]:
            time.sleep(1.0)
            d = _WATCH["deadline"]
            if d is not None and time.monotonic() > d and not _WATCH["done"]:
                if int(os.environ.get("RANK", "0")) == 0:
                    print(_error_line(f"watchdog: phase '{_WATCH['phase']}' exceeded its time limit (a collective or the IPC mapping hangs?); "
                                      "N2M_BENCH_TIMEOUT raises the limits"), flush=True)
                os._exit(4)
    threading.Thread(target=watch, daemon=True).start()


def main():
    try:
        return _main()
    except SystemExit:
        raise
    except BaseException as e:      # an error line on rank 0 instead of a bare traceback (the driver parses stdout)
        import traceback
        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            print(_error_line(f"{type(e).__name__}: {e}"), flush=True)
        _WATCH["done"] = True
        os._exit(1)
    finally:
        _WATCH["done"] = True


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--pretrain", type=int, default=None, help="untimed iterations before warmup (default 1000 = the reference's diffuse_step: the "
                    "occupancy grid is pruned and the timed steps run with shading='full' like 29 000 of the reference's 30 000 iterations; 300 with --diffuse)")
    ap.add_argument("--diffuse", action="store_true", help="time the reference's first-1000-iterations phase (diffuse shading: no specular branch / loss) "
                    "instead of the steady-state step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not record per-kernel hipEvents in the timed region")
    ap.add_argument("--prof-every", type=int, default=None, help="time every n-th launch of each kernel with hipEvents (1 = all); default "
                    "max(1, steps // 12): at least a dozen samples per kernel whatever --steps is (the driver runs 20).  The events of the step's "
                    "kernels ride on the kernel dispatches themselves (hipExtLaunchKernel start / stop events, N2M_PROF_K in csrc/n2m_common.hpp): no "
                    "marker packets in the queue, the durations are the dispatches' own timestamps like rocprofv3's")
    ap.add_argument("--stage", type=int, default=0, choices=[0, 1], help="0: stage-0 volume rendering (the headline metric); "
                    "1: stage-1 mesh/texture refinement step (BASELINE config 3)")
    ap.add_argument("--autograd", action="store_true", help="A/B: drive the step through torch.autograd (trainer.Stage0Trainer) instead of the step "
                    "executor (engine.Stage0Engine): same kernels, same arguments, more host time")
    ap.add_argument("--recipe", default="lego", choices=["lego", "sdf", "garden"], help="lego: the headline config (BASELINE configs[1]); sdf: "
                    "`--sdf` stage 0 (config 5: NeuS alpha, 7 density evaluations per sample, eikonal loss); garden: config 4's recipe "
                    "(`--bound 16 --enable_cam_near_far --lambda_entropy 1e-3`, 5 cascades, colmap-style AABB, inner/outer TV) on the synthetic yard scene")
    ap.add_argument("--num-points", type=int, default=0, help="measurement aid: override the per-step sample target (2^18 in the recipe); "
                                                               "a tiny value shows every kernel's fixed cost")
    ap.add_argument("--unfused", action="store_true", help="A/B: evaluate the MLPs with nn.Linear calls (the reference graph) instead of the fused MFMA kernels")
    ap.add_argument("--no-other-configs", action="store_true", help="do not append `other_configs` (BASELINE configs 3, 4, 5 and the drop-in path, each "
                    "~50 timed steps in a child process behind the headline window) and `long_run` (192 more steps) to the default line")
    ap.add_argument("--dropin", action="store_true", help="time the drop-in path instead: the reference's unchanged Python over backends/_*.py (dropin_reference_loop)")
    ap.add_argument("--pipeline", action="store_true", help="BASELINE config 5 end to end on one GPU at a short schedule: --sdf stage 0 -> export_stage0 -> stage 1 -> PSNR")
    ap.add_argument("--pipeline-iters0", type=int, default=2000)
    ap.add_argument("--pipeline-iters1", type=int, default=400)
    ap.add_argument("--pipeline-resolution", type=int, default=256)
    ap.add_argument("--dropin-fused", action="store_true", help="--dropin with backends.install(fused_mlp=True): the fused MFMA field behind the unchanged NeRFNetwork")
    args = ap.parse_args()

    from nerf2mesh_amd import _lib, synthetic
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    from nerf2mesh_amd.parallel import init_from_env
    from nerf2mesh_amd.trainer import Stage0Trainer

    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    if args.pretrain is None:
        args.pretrain = 300 if args.diffuse else 1000
        # The occupancy refresh runs every 16th step (nerf/utils.py:1155-1156) and costs more than a whole plain step: a short timed window
        # holds 1/16 of its steps as refresh steps only if it is placed accordingly -- the driver's 20 steps behind 1000 + 5 would hold
        # TWO (10 % instead of the long-run 6.25 %).  The default pre-training length is therefore shifted by up to 15 iterations so that
        # the window holds round(steps / 16) refresh steps; `config.refresh_steps_in_window` says how many it held.
        if args.stage == 0:
            want = int(round(args.steps / 16.0))
            for shift in range(16):
                first = args.pretrain + shift + args.warmup + 1
                if sum(1 for j in range(first, first + args.steps) if (j - 1) % 16 == 0) == want:
                    args.pretrain += shift
                    break
    if args.prof_every is None:
        args.prof_every = max(1, args.steps // 12)
        if args.prof_every > 1 and args.prof_every % 2 == 0:
            args.prof_every += 1       # odd: every 16th launch of a kernel is always the same phase of the 16-step occupancy-refresh cycle
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher -- one rank per GPU under torch.distributed.run, exactly
        # the command the driver uses (the JSON line then reports n_gpus = the rank count that actually ran)
        n_dev = torch.cuda.device_count()
        shared = os.environ.get("N2M_DIST_BACKEND") == "gloo"        # test mode: ranks may share a device over gloo
        if n_dev < args.gpus and not shared:
            sys.exit(f"[bench] --gpus {args.gpus} but only {n_dev} GPU(s) are visible: refusing to measure fewer ranks than asked for")
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    slack = float(os.environ.get("N2M_BENCH_TIMEOUT", "1.0"))       # multiplies every phase limit below
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        _start_watchdog()
        _phase("process group set-up + first collective", 240 * slack)
    rank, world, local = init_from_env()
    if world != args.gpus:
        sys.exit(f"[bench] launched with WORLD_SIZE={world} but --gpus {args.gpus}: the two must agree")
    local = local % torch.cuda.device_count()          # ranks beyond the visible GPUs share devices (gloo test runs only)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    _lib.lib()
    if world > 1:      # pre-flight: one small collective end to end before anything expensive is built on it
        probe = torch.full((1024,), float(rank + 1), device=device)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        assert float(probe[0]) == world * (world + 1) / 2, f"all-reduce probe returned {float(probe[0])}"
        _phase("model + pre-training", 900 * slack)

    if args.pipeline:
        return bench_pipeline(args, device)
    if args.stage == 1:
        return bench_stage1(args, rank, world, device)
    if args.dropin or args.dropin_fused:
        r = dropin_reference_loop(device, steps=args.steps, warmup=args.warmup, pretrain=1000 if args.pretrain is None or args.pretrain >= 1000 else args.pretrain,
                                  fused_field=args.dropin_fused)
        print(json.dumps({"metric": "train_samples_per_sec", "unit": "samples/s", "n_gpus": 1, "higher_is_better": True, "data": "synthetic",
                          "config": {"workload": "nerf_synthetic/lego stage-0 -O --bound 1 --dt_gamma 0 -- the reference's unchanged Python over the drop-in backends"
                                                 + (" + the opt-in fused field (backends.install(fused_mlp=True))" if args.dropin_fused else "")}, **r}))
        return

    torch.manual_seed(0)                                           # seed_everything(0), identical init on every rank
    recipes = {"lego": dict(bound=1, dt_gamma=0),                                    # scripts/runall_syn.sh:1
               "sdf": dict(bound=1, dt_gamma=0, sdf=True),                            # scripts/runall_syn_sdf.sh:1
               # scripts/runall_360_outdoor.sh:2: -O --bound 16 --enable_cam_near_far --lambda_entropy 1e-3 (default dt_gamma 1/256), colmap AABB
               "garden": dict(bound=16, dt_gamma=1 / 256, lambda_entropy=1e-3, enable_cam_near_far=True, scene="garden")}
    # --sdf: the schedules of the recipe (progressive levels 4 -> 16, normal epsilon 1e-1 -> 1e-4, cos anneal 0 -> 1: nerf/utils.py:651-655) run
    # over the first HALF of the iterations; the other half trains in the end state.  iters = 2000 puts the timed steps (1000+) into that
    # end state -- 16 levels, epsilon 1e-4 -- instead of the 4-level start the default 30 000 would show at step 1000
    iters = 2000 if (args.recipe == "sdf" and not args.diffuse) else 30000
    opt = make_options(O=True, iters=iters, fused_mlp=not args.unfused, **recipes[args.recipe])
    if args.num_points > 0:
        opt.num_points = args.num_points
        opt.num_rays = max(64, args.num_points // 16)
    model = NeRFNetwork(opt)
    if args.recipe == "garden":
        model.update_aabb(synthetic.pts_aabb("garden"))       # main.py:234-235: the sparse points give a tighter AABB than the bound
    poses = synthetic.make_cameras(100, seed=0)
    from nerf2mesh_amd.engine import Stage0Engine
    use_engine = not args.autograd and not args.unfused and Stage0Engine.supported(model, opt)
    tr = (Stage0Engine if use_engine else Stage0Trainer)(model, opt, poses, device, rank=rank, world_size=world, seed=0)
    tr.mark_untrained()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.pretrain + args.warmup):
        tr.train_step()
    barrier()
    if world > 1:
        _phase("timed window", 600 * slack)
    first_timed = tr.global_step + 1
    shading_of = lambda it: "diffuse" if (it < opt.diffuse_step or opt.diffuse_only) else "full"       # nerf/utils.py:669-672
    shading = shading_of(first_timed) if shading_of(first_timed) == shading_of(first_timed + args.steps - 1) else "mixed"
    if args.stage == 0 and not args.diffuse and shading != "full":
        print(f"[bench] WARNING: timed steps {first_timed}..{first_timed + args.steps - 1} run with shading={shading} "
              f"(diffuse_step={opt.diffuse_step}); raise --pretrain", file=sys.stderr)
    if not args.no_prof:
        _lib.prof_reset()
        _lib.prof_enable(args.prof_every)       # hipEvent pairs on every n-th launch of each kernel inside the timed region
    s0, r0 = tr.samples_seen, tr.rays_seen
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.train_step()
    barrier()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    samples, rays = tr.samples_seen - s0, tr.rays_seen - r0

    stats = torch.tensor([dt, float(samples), float(rays)], dtype=torch.float64, device=device)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, samples, rays = float(mx[0]), float(sm[1]), float(sm[2])

    # the same step over a window that holds the occupancy refresh at exactly its long-run share (192 = 12 x 16 steps; the headline window of
    # the driver's 20 steps holds one refresh step = 5 % instead of 6.25 %): reported next to the headline, never instead of it
    long_run = None
    if not args.no_other_configs and args.recipe == "lego" and not args.diffuse and not args.autograd:
        _lib.prof_enable(False)
        barrier()
        s1 = tr.samples_seen
        t1 = time.perf_counter()
        for _ in range(192):
            tr.train_step()
        barrier()
        lr_stats = torch.tensor([time.perf_counter() - t1, float(tr.samples_seen - s1)], dtype=torch.float64, device=device)
        if world > 1:
            mx = lr_stats.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = lr_stats.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            lr_stats = torch.stack([mx[0], sm[1]])
        long_run = {"steps": 192, "ms_per_step": 1e3 * float(lr_stats[0]) / 192, "value": float(lr_stats[1]) / float(lr_stats[0]), "unit": "samples/s",
                    "refresh_steps_in_window": 12, "note": "the 192 steps behind the headline window, no per-kernel events"}

    if world > 1:
        _phase("parameter gather + evaluation", 600 * slack)
    # sharded optimizer: every rank owns 1/W of the fp32 table rows -- gather them while all ranks are still here (collective), so that
    # rank 0's PSNR evaluation below runs on complete tables without talking to anybody
    if world > 1 and hasattr(tr, "sync_parameters"):
        tr.sync_parameters()
    barrier()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    kernels = {}
    side_stream = ("march_rays_train_count", "march_rays_train_write", "near_far_from_aabb") if use_engine else ()
    if not args.no_prof:
        for name in ("grid_encode_forward_packed", "grid_encode_forward", "grid_encode_backward", "grad_total_variation", "march_rays_train_count",
                     "march_rays_train_write", "composite_rays_train_forward", "composite_rays_train_backward",
                     "near_far_from_aabb", "packbits", "mlp_forward", "mlp_backward", "adam_step"):
            n, ms, by = _lib.prof_read(name)
            if n:
                seen = _lib.prof_seen(name)              # ALL launches in the timed region (every prof_every-th of them carries events)
                kernels[name] = {"launches": n, "launches_seen": seen, "avg_us": 1e3 * ms / n, "algo_bytes_per_launch": by / n,
                                 "GBps": (by / n) / (ms / n * 1e-3) / 1e9 if ms > 0 else None,
                                 "ms_per_step": (ms / n) * seen / args.steps,                    # mean timed duration x launches per step
                                 "stream": "side (overlaps the main stream: not part of the step's critical path)" if name in side_stream else "main"}
    try:
        peak_measured = measured_stream_peak(device)
    except Exception as e:
        print(f"[bench] stream-copy peak not measured: {e!r}", file=sys.stderr)
        peak_measured = None

    def roofline_of(name):
        k = kernels[name]
        traffic, source = pmc_traffic(name)
        if k["launches"] < 8:
            print(f"[bench] WARNING: {name}: only {k['launches']} timed launches in {args.steps} steps -- lower --prof-every", file=sys.stderr)
        by_traffic = (traffic / (k["avg_us"] * 1e-6) / 1e9) if (traffic and k["avg_us"]) else None
        return {"kernel": name, "bound": "hbm", "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (k["GBps"] / HBM_PEAK_GBS) if k["GBps"] else None, "traffic": traffic, "traffic_source": source,
                # counter bytes next to the convention: traffic_frac < 1 = the design moves fewer bytes than SURVEY 8d counts for the
                # reference's algorithm (frac then overstates how busy the memory system is); frac_of_traffic = counter bytes / time / peak
                "traffic_frac": (traffic / k["algo_bytes_per_launch"]) if (traffic and k["algo_bytes_per_launch"]) else None,
                "achieved_by_traffic": by_traffic, "frac_of_traffic": (by_traffic / HBM_PEAK_GBS) if by_traffic else None,
                "peak_measured": peak_measured, "frac_of_measured": (k["GBps"] / peak_measured) if (k["GBps"] and peak_measured) else None,
                "launches": k["launches"], "avg_us": k["avg_us"], "algo_bytes_per_launch": k["algo_bytes_per_launch"],
                "note": "achieved = algorithmic bytes per launch (SURVEY.md 8d; cache hits count) / mean duration of the entry point's kernels over the "
                        "timed region (hipEvents attached to the dispatches: start of the first kernel to end of the last); peak = nominal HBM3E, "
                        "peak_measured = a 1 GB device-to-device copy (read + write) by the library's own 16-bytes-per-lane kernel (n2m_stream_copy), timed in this "
                        "process after the timed region; traffic = rocprofv3 --pmc bytes of the same command (committed file: counters cannot be "
                        "read from inside the timed process)"}
    roof = roof_lookup = None
    main_ms = sum(k["ms_per_step"] for k in kernels.values() if k["stream"] == "main")
    side_ms = sum(k["ms_per_step"] for k in kernels.values() if k["stream"] != "main")
    if kernels:
        roof = roofline_of(max((k for k in kernels if kernels[k]["stream"] == "main"), key=lambda k: kernels[k]["ms_per_step"]))   # the dominant entry point of the step
        if "grid_encode_forward_packed" in kernels:
            roof_lookup = roofline_of("grid_encode_forward_packed")                          # north_star's hash-grid lookup (training launches only)
        for r in (roof, roof_lookup):       # a broken measurement must not pass as a number (an event pair that closed before its launch once did)
            if r is not None and r["frac"] is not None and not (0.0 < r["frac"] <= 1.0):
                # (flagged, not fatal: other ranks may be waiting in a collective behind this point)
                print(f"[bench] ERROR: implausible roofline entry {r['kernel']}: {r['achieved']:.0f} GB/s of {HBM_PEAK_GBS:.0f} -- the per-kernel "
                      "events do not bracket the launch; the entry is withdrawn", file=sys.stderr)
                r.update(achieved=None, frac=None, note="WITHDRAWN: the measured duration is implausible (events did not bracket the launch)")

    cpu = cpu_ref = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline()
        except Exception as e:   # the baseline is a reported extra; never lose the GPU number over it
            cpu = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        try:
            cpu_ref = cpu_baseline_reference()
        except Exception as e:
            cpu_ref = {"value": None, "unit": "samples/s", "cores": 1, "kind": "reference", "sample": f"failed: {e!r}"}

    try:
        psnr = tr.eval_psnr()
        psnr_ema = tr.eval_psnr(use_ema=True) if getattr(tr, "ema", None) is not None else None
    except Exception as e:
        psnr = psnr_ema = f"failed: {e!r}"

    want_other = (world == 1 and not args.no_other_configs and args.recipe == "lego" and not args.diffuse and not args.autograd and not args.unfused
                  and args.num_points == 0)
    line = {
        "metric": "train_samples_per_sec", "value": samples / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 march/composite, f16 autocast encode+MLP (reference -O recipe)", "data": "synthetic",
        "rays_per_sec": rays / dt,
        "config": {"workload": {"lego": "nerf_synthetic/lego stage-0 -O --bound 1 --dt_gamma 0",
                                "sdf": "nerf_synthetic/lego --sdf stage-0 -O --bound 1 --dt_gamma 0 (NeuS alpha, finite-difference normals, eikonal loss)",
                                "garden": "mip-360 outdoor recipe (scripts/runall_360_outdoor.sh:2) stage-0 -O --bound 16 --enable_cam_near_far "
                                          "--lambda_entropy 1e-3, dt_gamma 1/256 (5 cascades, colmap-style AABB, inner/outer TV) on the synthetic yard scene"}[args.recipe]
                               + ", 800x800 x 100 synthetic views, num_points target 2^18/GPU (adaptive num_rays), occupancy refresh every 16 steps",
                   "sdf_schedule": (f"max_level {model.max_level}, normal epsilon {opt.normal_anneal_epsilon:.2g}, cos_anneal_ratio {opt.cos_anneal_ratio:.2g}"
                                    if args.recipe == "sdf" else None),
                   "shading": shading, "timed_steps": [first_timed, first_timed + args.steps - 1], "diffuse_step": int(opt.diffuse_step),
                   "refresh_steps_in_window": sum(1 for j in range(first_timed, first_timed + args.steps) if (j - 1) % int(opt.update_extra_interval) == 0),
                   "refresh_share_long_run": 1.0 / int(opt.update_extra_interval),
                   "parallelism": (f"dp{world} (rays sharded; peer-store exchange, N2M_PEER_STORE=1: gradient rows stored into their owners' slots by the table "
                                   f"backward, Adam sharded over the ranks, packed rows stored to every rank; no collective in the step)"
                                   if getattr(tr, "peer", None) is not None else
                                   f"dp{world} (rays sharded; table gradients reduce-scattered, Adam sharded over the ranks, packed rows all-gathered; "
                                   f"{dist.get_backend()})" if getattr(tr, "shard", False) else
                                   f"dp{world} (rays sharded, grad all-reduce over {dist.get_backend()})") if world > 1 else "single GPU",
                   "occupancy_refresh": ("sharded over the ranks by Morton range + all-gather of the densities" if getattr(model, "refresh_shard", None) else
                                         "replicated on every rank (N2M_SHARD_REFRESH=0)") if world > 1 else "single GPU",
                   "mlp": "nn.Linear (unfused)" if args.unfused else "fused MFMA field kernels",
                   "driver": "engine.Stage0Engine (fixed launch sequence)" if use_engine else "trainer.Stage0Trainer (torch.autograd)", "pretrain_steps": args.pretrain, "samples_per_step_per_gpu": samples / args.steps / world,
                   "rays_per_step_per_gpu": rays / args.steps / world, "params": 18367240},
        "roofline": roof, "roofline_lookup": roof_lookup, "kernels": kernels,
        "kernels_note": (f"per-kernel hipEvents on every {args.prof_every}-th launch (kernel ids staggered), attached to the kernel dispatches of the "
                         f"step's entry points; ms_per_step = mean timed duration x launches seen / steps. "
                         f"Main-stream entries sum to {main_ms:.3f} ms/step (small launches without events -- bookkeeping -- are not listed); "
                         f"side-stream entries ({side_ms:.3f} ms/step: next-but-one batch's ray generation and march) run beside the main stream's "
                         "Adam / forward and are NOT part of the step time; grid_encode_forward (unpacked) = the occupancy refresh's density query, "
                         "once per 16 steps") if kernels else None,
        "long_run": long_run, "other_configs": None,
        "cpu_baseline": cpu, "cpu_baseline_reference": cpu_ref, "psnr_view0_quarter_res": psnr,
        # (evaluated after pre-training + warm-up + the timed window + the 192-step long_run: `psnr_at_step` training steps in all;
        #  _ema = with the averaged weights the reference evaluates with, nerf/utils.py:1250-1252)
        "psnr_view0_quarter_res_ema": psnr_ema, "psnr_at_step": int(tr.global_step),
        "loss_mean": float(tr.loss_acc / max(tr.global_step, 1)),
    }
    if want_other:
        # the extras run in child processes for minutes: a copy of the finished headline goes to stderr FIRST, so that a caller that gives up on
        # the extras still has the measurement (stdout stays ONE line, printed when everything is in)
        print("[bench] headline, before the extras (safety copy of the line that follows on stdout): " + json.dumps(line), file=sys.stderr, flush=True)
        try:
            line["other_configs"] = other_configs()
        except Exception as e:
            line["other_configs"] = {"error": repr(e)[:400]}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
