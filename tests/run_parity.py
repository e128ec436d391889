"""RUN-LEVEL parity: does a whole training run of the step executor end where the reference's own loop ends?

TEST INFRASTRUCTURE (imports oracle/ref_python: the reference's unchanged Python).  north_star: "final PSNR within 0.1 dB of the reference".
One-step gradient pins (tests/test_reference_engine.py) do not bound the drift of thousands of Adam steps, so this file trains

  (a) REFERENCE LOOP: the unchanged `Trainer.train_step` / `post_train_step` (nerf/utils.py:628-823) in `train_one_epoch`'s order
      (:1152-1214: occupancy refresh every 16 steps, zero_grad, scaler.scale(loss).backward(), TV, scaler.step, scaler.update, LambdaLR step,
      EMA update at the end of every epoch) with main.py:221's torch.optim.Adam(eps=1e-15), GradScaler and main.py:239's schedule, over the
      unchanged nerf/renderer.py + nerf/network.py + autograd wrappers on libn2m_hip.so through nerf2mesh_amd/backends/_*.py, and
  (b) EXECUTOR: nerf2mesh_amd.engine.Stage0Engine (what bench.py times),

from the SAME initial parameters on the same synthetic scene with the same recipe, K seeds x S steps each, and evaluates both end states
on held-out views at full resolution with ONE inference path (nerf2mesh_amd's renderer in eval mode: the reference run's state_dict --
parameters, density grid, bit field -- is loaded into a nerf2mesh_amd NeRFNetwork with strict=True), with the EMA weights as the
reference's evaluate_one_epoch does (nerf/utils.py:1250-1252) and with the raw weights.  torch-ema is un-vendored (requirements.txt:11):
`TorchEma` below restates the library's three torch ops per tensor.

What differs between (a) and (b) by construction: the random draws (the reference loop consumes torch's global generator for the
background colours and a batch generator for the pixels; the executor one [N,6] draw per batch), fp16 rounding points (the fused field) and
summation order.  So the comparison is statistical: mean +- s.e. over seeds and views.

    python tests/run_parity.py --recipe lego --seeds 3 --steps 2000 --views 8        (tools/run_parity.py forwards here)
"""
import argparse
import json
import math
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RECIPES = {
    "lego": dict(bound=1, dt_gamma=0),                                                                       # scripts/runall_syn.sh:1
    "sdf": dict(bound=1, dt_gamma=0, sdf=True),                                                              # scripts/runall_syn_sdf.sh:1
    "garden": dict(bound=16, dt_gamma=1 / 256, lambda_entropy=1e-3, enable_cam_near_far=True, scene="garden"),   # scripts/runall_360_outdoor.sh:2
}
EMA_DECAY = 0.95                    # main.py:241
HELD_OUT_SEED = 4242                # cameras no training run sees


class TorchEma:
    """torch_ema.ExponentialMovingAverage (0.3) as the reference Trainer uses it: shadows = clones of the parameters; update():
    num_updates += 1, decay = min(decay, (1 + n) / (10 + n)), per tensor tmp = s - p; tmp.mul_(1 - decay); s.sub_(tmp)."""

    def __init__(self, parameters, decay):
        self.params = [p for p in parameters if p.requires_grad]
        self.decay, self.num_updates = decay, 0
        self.shadow_params = [p.clone().detach() for p in self.params]

    @torch.no_grad()
    def update(self):
        self.num_updates += 1
        omd = 1.0 - min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        for s, p in zip(self.shadow_params, self.params):
            tmp = s - p
            tmp.mul_(omd)
            s.sub_(tmp)


def make_opt(recipe, steps, fused=True):
    from nerf2mesh_amd.options import make_options
    return make_options(O=True, iters=steps, fused_mlp=fused, **RECIPES[recipe])


def scene_of(recipe):
    return RECIPES[recipe].get("scene", "lego")


def initial_state(recipe, seed, steps, device):
    """Initial model state (reference key names), the same for both runs of a seed: nerf2mesh_amd's constructor under torch.manual_seed(seed)
    (same distributions as the reference's: nerf/network.py, gridencoder/grid.py:reset_parameters), + the SDF pre-training (nerf/utils.py:594)."""
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.network import NeRFNetwork
    torch.manual_seed(seed)
    opt = make_opt(recipe, steps, fused=False)
    model = NeRFNetwork(opt).to(device)
    if recipe == "garden":
        model.update_aabb(synthetic.pts_aabb("garden"))          # main.py:234-235
    if recipe == "sdf":
        model.init_double_sphere(iters=2048)                      # (the reference runs 8192 such steps; both runs start from THIS state either way)
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


# ----------------------------------------------------------------------------------------------------------------- (a) the reference loop
def train_reference(recipe, seed, steps, poses, images, init, device, log=None, draw_seed=None, fused_field=False):
    """Returns (model state_dict, EMA shadow list in model.parameters() order, wall seconds).  draw_seed: seed of everything random about the
    batches (pixels, background colours, march jitter, refresh jitter); default = `seed`."""
    draw_seed = seed if draw_seed is None else draw_seed
    from oracle import ref_python as RP
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd import synthetic
    assert RP.available(), "reference Python not available (neither /root/reference nor oracle/_ref/pyref)"
    ns = RP.load("hip")
    RP.use_backend("hip")
    from nerf2mesh_amd import backends
    # (triangulation aid: the same unchanged loop with the opt-in fused MFMA field behind the unchanged class, backends.fuse_field)
    (backends.fuse_field if fused_field else backends.unfuse_field)(ns.network.NeRFNetwork)
    torch.manual_seed(draw_seed)
    d = dict(vars(RP.reference_opt()))
    d.update(vars(make_opt(recipe, steps, fused=False)))
    for k in ("scene", "fused_mlp", "enable_cam_near_far"):
        d.pop(k, None)
    d.update(bound=float(d["bound"]), data_format="colmap" if recipe == "garden" else "nerf", lambda_depth=0.0)
    opt = types.SimpleNamespace(**d)
    model = ns.network.NeRFNetwork(opt).cuda()
    if recipe == "garden":
        model.update_aabb(synthetic.pts_aabb("garden"))
    missing = model.load_state_dict(init, strict=True)
    scene = scene_of(recipe)
    cnf = synthetic.cam_near_far(poses, scene) if RECIPES[recipe].get("enable_cam_near_far") else None
    if opt.mark_untrained:                                                                               # nerf/utils.py:924-925
        f = synthetic.LEGO_FOCAL
        import numpy as np
        ds = types.SimpleNamespace(poses=poses, intrinsics=np.array([f, f, synthetic.LEGO_HW / 2, synthetic.LEGO_HW / 2], dtype=np.float32))
        if cnf is not None:
            ds.cam_near_far = cnf
        model.mark_untrained_grid(ds)
    optimizer = torch.optim.Adam(model.get_params(opt.lr), eps=1e-15)                                    # main.py:221
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda it: 0.01 + 0.99 * (it / 500) if it <= 500 else 0.1 ** ((it - 500) / (opt.iters - 500)))
    T = ns.utils.Trainer
    me = types.SimpleNamespace(opt=opt, model=model, global_step=0, device=device, criterion=torch.nn.MSELoss(reduction="none"), optimizer=optimizer,
                               scaler=torch.cuda.amp.GradScaler(enabled=True), tmp_xyzs=None)
    ema = TorchEma(model.parameters(), EMA_DECAY)                                                        # nerf/utils.py:544-545
    epoch_len = poses.shape[0]                                                                            # len(train_loader): batch_size 1 over the views
    gen = torch.Generator(device=device).manual_seed(draw_seed)
    H = W = synthetic.LEGO_HW
    model.train()
    t0 = time.perf_counter()
    for it in range(steps):
        if me.global_step % opt.update_extra_interval == 0:                                               # nerf/utils.py:1155-1156
            model.update_extra_state()
        me.global_step += 1
        optimizer.zero_grad()
        N = int(opt.num_rays)
        cam = torch.randint(0, poses.shape[0], (N,), device=device, generator=gen)                        # nerf/provider.py:302-303
        pix = torch.randint(0, H * W, (N,), device=device, generator=gen)
        o, dd, rgba = torch.empty(N, 3, device=device), torch.empty(N, 3, device=device), torch.empty(N, 4, device=device)
        L.call("n2m_get_rays", L.ptr(poses), L.ptr(cam), L.ptr(pix), N, H, W, float(synthetic.LEGO_FOCAL), float(synthetic.LEGO_FOCAL), W / 2, H / 2,
               L.ptr(images), L.ptr(o), L.ptr(dd), L.ptr(rgba), L.stream())
        data = {"rays_o": o, "rays_d": dd, "index": cam, "images": rgba}
        if cnf is not None:
            data["cam_near_far"] = cnf[cam]                                                               # nerf/colmap_provider.py collate
        _, _, loss = T.train_step(me, data)
        me.scaler.scale(loss).backward()                                                                  # :1172
        T.post_train_step(me)                                                                             # :1174 (unscale + in-place TV)
        me.scaler.step(optimizer)
        me.scaler.update()
        scheduler.step()
        lv = loss.item()                                                                                  # :1182
        if (it + 1) % epoch_len == 0:                                                                     # :1213-1214
            ema.update()
        if log is not None and (it + 1) % 500 == 0:
            log(f"    reference loop step {it + 1}: loss {lv:.5f}, num_rays {opt.num_rays}")
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    shadow = {n: s.clone() for n, s in zip(names, ema.shadow_params)}
    # optimizer steps actually TAKEN (GradScaler skips a step whose gradients hold an inf / nan) and the loss scale the run ended at
    st = optimizer.state[model.encoder.embeddings]
    backends.unfuse_field(ns.network.NeRFNetwork)
    DIAG["reference-fused" if fused_field else "reference"] = {"adam_steps_taken": int(st["step"]) if "step" in st else 0, "loss_scale": float(me.scaler.get_scale()), "num_rays": int(opt.num_rays)}
    return sd, shadow, wall


DIAG = {}


# ----------------------------------------------------------------------------------------------------------------------- (b) the executor
def train_engine(recipe, seed, steps, poses, init, device, log=None):
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.engine import Stage0Engine
    from nerf2mesh_amd.network import NeRFNetwork
    torch.manual_seed(seed)
    opt = make_opt(recipe, steps, fused=True)
    model = NeRFNetwork(opt)
    if recipe == "garden":
        model.update_aabb(synthetic.pts_aabb("garden"))
    model = model.to(device)
    model.load_state_dict(init, strict=True)
    eng = Stage0Engine(model, opt, poses, device, seed=seed, ema_decay=EMA_DECAY)
    if os.environ.get("N2M_PARITY_ENGINE_NO_SCALE_GROWTH") == "1":
        # (triangulation aid: the loss scale never grows beyond GradScaler's initial 65536 -- the reference's fp16 graph overflows above ~2^16 and
        #  lives at 2^15-2^16, the executor's fused field keeps intermediate gradients in fp32 and climbs to 2^24-2^27)
        gf, bf, gi = eng.optimizer.growth
        eng.optimizer.growth = (1.0, bf, gi)
    eng.mark_untrained()
    t0 = time.perf_counter()
    for it in range(steps):
        loss = eng.train_step()
        if log is not None and (it + 1) % 500 == 0:
            log(f"    executor step {it + 1}: loss {float(loss):.5f}, num_rays {eng.num_rays}")
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    shadow = {n: s.clone() for n, s in zip(names, eng.ema.shadow_params)}
    o = eng.optimizer
    DIAG["engine"] = {"adam_steps_taken": int(o.steps.max()) if hasattr(o, "steps") else None, "loss_scale": float(o.scale), "num_rays": int(eng.num_rays)}
    return sd, shadow, wall


def train_trainer(recipe, seed, steps, poses, init, device, log=None):
    """(triangulation aid) trainer.Stage0Trainer: the executor's kernels driven through torch.autograd."""
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.trainer import Stage0Trainer
    torch.manual_seed(seed)
    opt = make_opt(recipe, steps, fused=True)
    model = NeRFNetwork(opt)
    if recipe == "garden":
        model.update_aabb(synthetic.pts_aabb("garden"))
    model = model.to(device)
    model.load_state_dict(init, strict=True)
    tr = Stage0Trainer(model, opt, poses, device, seed=seed, ema_decay=EMA_DECAY)
    tr.mark_untrained()
    t0 = time.perf_counter()
    for it in range(steps):
        tr.train_step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    shadow = {n: s.clone() for n, s in zip(names, tr.ema.shadow_params)}
    return sd, shadow, wall


# ------------------------------------------------------------------------------------------------------------- one inference path for both
@torch.no_grad()
def evaluate(recipe, steps, state, shadow, held_out, device):
    """Per-view PSNR of `state` (and of `state` with the parameters replaced by `shadow`) on the held-out cameras: 800 x 800, white
    background, nerf2mesh_amd's renderer in eval mode (march_rays / composite_rays loop), analytic ground truth of the synthetic scene."""
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.network import NeRFNetwork
    opt = make_opt(recipe, steps, fused=True)
    if opt.sdf:        # end of the schedules (nerf/utils.py:651-655 at global_step = iters)
        opt.cos_anneal_ratio, opt.normal_anneal_epsilon = 1.0, 1e-1 * (1 - 0.999)
    model = NeRFNetwork(opt)
    if recipe == "garden":
        model.update_aabb(synthetic.pts_aabb("garden"))
    model = model.to(device)
    scene = scene_of(recipe)
    boxes = synthetic.boxes(device, scene)
    cnf = synthetic.cam_near_far(held_out, scene) if RECIPES[recipe].get("enable_cam_near_far") else None
    HW = synthetic.LEGO_HW
    pix = torch.arange(HW * HW, device=device)
    out = {}
    for kind, sd in (("ema", {**state, **shadow}), ("raw", state)):
        model.load_state_dict(sd, strict=True)
        if opt.sdf:
            model.max_level = 16
        model.eval()
        psnrs = []
        for c in range(held_out.shape[0]):
            o, d = synthetic.rays_from_pixels(held_out, torch.full_like(pix, c), pix)
            rgba = synthetic.render_gt(o, d, boxes)
            gt = rgba[:, :3] * rgba[:, 3:] + (1 - rgba[:, 3:])
            r = model.render(o, d, bg_color=1, perturb=False, shading="full", dt_gamma=opt.dt_gamma, max_steps=opt.max_steps, T_thresh=1e-4,
                             cam_near_far=None if cnf is None else cnf[c:c + 1])
            mse = torch.mean((r["image"] - gt) ** 2)
            psnrs.append(float(-10 * torch.log10(mse)))
        out[kind] = psnrs
    return out


def mean_se(xs):
    n = len(xs)
    m = sum(xs) / n
    if n < 2:
        return m, float("nan")
    return m, math.sqrt(sum((x - m) ** 2 for x in xs) / (n - 1) / n)


def run(recipe="lego", seeds=3, steps=2000, views=8, train_views=100, paths=("reference", "engine"), device=None, log=print, engine_repeat=False,
        reference_redraw=False, first_seed=0):
    """Returns the result table (dict).  engine_repeat: a second executor run per seed from the same state -- its distance from the first
    is the executor's own run-to-run spread (0 when the step is bit-reproducible).  reference_redraw: a second REFERENCE run per seed from
    the same initial state with other random draws (pixels, backgrounds, jitter) -- the distance between the two reference runs is the
    noise floor any comparison of two runs with different draws sits on (the executor necessarily draws differently from the reference)."""
    from nerf2mesh_amd import synthetic
    device = device or torch.device("cuda", 0)
    scene = scene_of(recipe)
    poses = synthetic.make_cameras(train_views, seed=0).to(device).float().contiguous()
    held_out = synthetic.make_cameras(views, seed=HELD_OUT_SEED).to(device).float().contiguous()
    images = synthetic.preload_images(poses, synthetic.boxes(device, scene))
    res = {"recipe": recipe, "steps": steps, "seeds": seeds, "held_out_views": views, "train_views": train_views, "runs": []}
    for s in range(first_seed, first_seed + seeds):
        init = initial_state(recipe, s, steps, device)
        row = {"seed": s}
        for path in paths:
            log(f"  seed {s}: {path} ...")
            if path == "reference":
                sd, sh, wall = train_reference(recipe, s, steps, poses, images, init, device, log)
            elif path == "reference-fused":
                sd, sh, wall = train_reference(recipe, s, steps, poses, images, init, device, log, fused_field=True)
            elif path == "trainer":
                sd, sh, wall = train_trainer(recipe, s, steps, poses, init, device, log)
            else:
                sd, sh, wall = train_engine(recipe, s, steps, poses, init, device, log)
            ev = evaluate(recipe, steps, sd, sh, held_out, device)
            row[path] = {"psnr_ema": ev["ema"], "psnr_raw": ev["raw"], "train_wall_s": wall, "ms_per_step": 1e3 * wall / steps, **DIAG.get(path, {})}
            log(f"  seed {s}: {path}: optimizer steps taken {row[path].get('adam_steps_taken')} of {steps}, final loss scale {row[path].get('loss_scale')}")
            log(f"  seed {s}: {path}: PSNR (EMA weights) {sum(ev['ema']) / views:.3f} dB, (raw) {sum(ev['raw']) / views:.3f} dB, {1e3 * wall / steps:.2f} ms/step")
            if path == "reference" and reference_redraw:
                sd2, sh2, _ = train_reference(recipe, s, steps, poses, images, init, device, draw_seed=s + 1000)
                ev2 = evaluate(recipe, steps, sd2, sh2, held_out, device)
                row["reference_redraw"] = {"psnr_ema": ev2["ema"], "psnr_raw": ev2["raw"]}
                log(f"  seed {s}: reference with other draws: PSNR (EMA weights) {sum(ev2['ema']) / views:.3f} dB, (raw) {sum(ev2['raw']) / views:.3f} dB")
                del sd2, sh2
            if path == "engine" and engine_repeat:
                sd2, sh2, _ = train_engine(recipe, s, steps, poses, init, device)
                same = all(torch.equal(sd[k], sd2[k]) for k in sd) and all(torch.equal(sh[k], sh2[k]) for k in sh)
                ev2 = evaluate(recipe, steps, sd2, sh2, held_out, device)
                row["engine_repeat"] = {"bit_identical_state": bool(same), "psnr_ema": ev2["ema"],
                                        "abs_diff_mean_psnr": abs(sum(ev2["ema"]) - sum(ev["ema"])) / views}
                log(f"  seed {s}: executor repeated: state bit-identical = {same}, |d mean PSNR| = {row['engine_repeat']['abs_diff_mean_psnr']:.4f} dB")
            del sd, sh
            torch.cuda.empty_cache()
        res["runs"].append(row)
    # summary: per path the mean over seeds of the per-run mean-over-views PSNR; the difference PAIRED by seed (same initial state)
    summ = {}
    for path in paths:
        for kind in ("psnr_ema", "psnr_raw"):
            per_run = [sum(r[path][kind]) / views for r in res["runs"]]
            m, se = mean_se(per_run)
            summ[f"{path}_{kind}"] = {"mean": m, "se": se, "per_seed": per_run}
    if "reference" in paths and "engine" in paths:
        for kind in ("psnr_ema", "psnr_raw"):
            dif = [sum(r["engine"][kind]) / views - sum(r["reference"][kind]) / views for r in res["runs"]]
            m, se = mean_se(dif)
            summ[f"delta_{kind}"] = {"mean": m, "se": se, "per_seed": dif}
    for other in paths:
        if other in ("reference", "engine") or "reference" not in paths:
            continue
        for kind in ("psnr_ema", "psnr_raw"):
            dif = [sum(r[other][kind]) / views - sum(r["reference"][kind]) / views for r in res["runs"]]
            m, se = mean_se(dif)
            summ[f"{other}_minus_reference_{kind}"] = {"mean": m, "se": se, "per_seed": dif}
    if reference_redraw and "reference" in paths:
        for kind in ("psnr_ema", "psnr_raw"):
            dif = [sum(r["reference_redraw"][kind]) / views - sum(r["reference"][kind]) / views for r in res["runs"]]
            m, se = mean_se(dif)
            summ[f"redraw_delta_{kind}"] = {"mean": m, "se": se, "per_seed": dif,
                                            "rms": math.sqrt(sum(x * x for x in dif) / len(dif))}
    for k in ("delta_psnr_ema", "delta_psnr_raw"):
        if k in summ:
            summ[k]["rms"] = math.sqrt(sum(x * x for x in summ[k]["per_seed"]) / len(summ[k]["per_seed"]))
    res["summary"] = summ
    return res


def format_table(res):
    v = res["held_out_views"]
    lines = [f"run parity -- recipe {res['recipe']}, {res['seeds']} seed(s) x {res['steps']} steps, {res['train_views']} training views, "
             f"{v} held-out 800x800 views, one inference path (nerf2mesh_amd renderer, eval mode)",
             "PSNR in dB against the analytic ground truth; EMA = averaged weights (decay 0.95, one update per epoch), raw = last iterate", ""]
    paths = [p for p in ("reference", "engine", "reference-fused", "trainer") if p in res["runs"][0]]
    two = paths == ["reference", "engine"]
    lines.append("seed | " + " | ".join(f"{p[-9:]:>9} EMA   raw  ms/step" for p in paths) + (" | delta EMA  delta raw" if two else ""))
    for r in res["runs"]:
        cells = [f"{sum(r[p]['psnr_ema']) / v:13.3f} {sum(r[p]['psnr_raw']) / v:6.3f} {r[p]['ms_per_step']:7.2f}" for p in paths]
        tail = ""
        if two:
            tail = f" | {sum(r['engine']['psnr_ema']) / v - sum(r['reference']['psnr_ema']) / v:+9.3f}  {sum(r['engine']['psnr_raw']) / v - sum(r['reference']['psnr_raw']) / v:+9.3f}"
        lines.append(f"{r['seed']:4d} | " + " | ".join(cells) + tail)
        lines.append("     |   optimizer steps taken (GradScaler skips overflowing ones) / final loss scale: " +
                     ", ".join(f"{p} {r[p].get('adam_steps_taken')} / {r[p].get('loss_scale'):g}" for p in paths if r[p].get("loss_scale") is not None))
        if "reference_redraw" in r:
            e = r["reference_redraw"]
            lines.append(f"     |   reference loop again, same initial state, other draws: EMA {sum(e['psnr_ema']) / v:.3f} raw {sum(e['psnr_raw']) / v:.3f} "
                         f"(delta vs the first reference run {sum(e['psnr_ema']) / v - sum(r['reference']['psnr_ema']) / v:+.3f} / "
                         f"{sum(e['psnr_raw']) / v - sum(r['reference']['psnr_raw']) / v:+.3f})")
        if "engine_repeat" in r:
            e = r["engine_repeat"]
            lines.append(f"     |   executor repeated from the same state: bit-identical end state = {e['bit_identical_state']}, |d mean PSNR| = {e['abs_diff_mean_psnr']:.4f} dB")
    lines.append("")
    for k, s in res["summary"].items():
        lines.append(f"{k:>24}: mean {s['mean']:+.3f} +- {s['se']:.3f} (s.e. over seeds)" + (f", rms {s['rms']:.3f}" if "rms" in s else ""))
    return "\n".join(lines)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--recipe", default="lego", choices=sorted(RECIPES))
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--first-seed", type=int, default=0)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--train-views", type=int, default=100)
    ap.add_argument("--engine-repeat", action="store_true")
    ap.add_argument("--reference-redraw", action="store_true")
    ap.add_argument("--only", default=None, choices=["reference", "engine"])
    ap.add_argument("--paths", default=None, help="comma-separated: reference, engine, reference-fused, trainer")
    ap.add_argument("--out", default=None, help="write the table (.txt) and the raw numbers (.json) under this path prefix")
    a = ap.parse_args(argv)
    paths = ("reference", "engine") if a.only is None else (a.only,)
    if a.paths:
        paths = tuple(a.paths.split(","))
    res = run(a.recipe, a.seeds, a.steps, a.views, a.train_views, paths, engine_repeat=a.engine_repeat, reference_redraw=a.reference_redraw, first_seed=a.first_seed)
    table = format_table(res)
    print(table)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out + ".txt", "w") as f:
            f.write(table + "\n")
        with open(a.out + ".json", "w") as f:
            json.dump(res, f)
    return res


if __name__ == "__main__":
    main()
