"""The step executor itself (engine.Stage0Engine, the path bench.py times) against the UNCHANGED reference Python -- on every recipe it runs.

tests/test_reference_render.py pins the restated renderer / trainer to the reference's own `render`; tests/test_engine.py holds the executor
to the trainer.  Here the executor is driven for ONE training step from the exact state of a whole-render fixture (tests/golden/render_*.npz:
parameters of render_case.make_state, the fixture's occupancy bit field, the 64x64 crop of camera 0) and compared with the reference's own
iteration on the same device:

    nerf/utils.py:640-683 (train_step: background, MSE on rgb + mask, mean), :728-733 (entropy), :735-738 (specular regulariser), :740-743
    (eikonal), :651-655 (SDF schedules), :1187 (scaler.scale(loss).backward()), :802-823 (post_train_step: unscale, in-place TV at the batch's
    samples, inner / outer split for bound > 1)          over          nerf/renderer.py:676-813 (render), nerf/network.py:135-156 (normals)

run by the unchanged reference Python over the HIP `_backend` modules (oracle/_ref/pyref on the GPU box), once under `-O` (fp16 autocast) --
the yardstick -- and once in fp32 -- the truth.  Cases:

    nerf      BASELINE config 1 / 2: bound 1, dt_gamma 0                                   (train_step: one launch sequence)
    garden    BASELINE config 4: bound 16, 5 cascades, dt_gamma 1/256, per-ray near / far clamp, entropy loss (grad_weights), inner / outer TV
    sdf-mid   BASELINE config 5 half way up its ramps: 10 of 16 levels, epsilon 0.05, cos_anneal_ratio 0.5 -- stacked finite-difference pass,
              TV of all sixteen levels inside the batch's backward over zero feature gradients
    sdf-late  ... at the end of the ramps: 16 levels, epsilon 1e-4 -- copies FOLDED into the batch's table backward + the per-level lists call

Sample count and sample positions must be EXACT; every gradient of the executor (fused MFMA field, binned fixed-point table backward, TV
folded in, SDF head kernels) must be no farther from the fp32 truth than the reference's own fp16 graph is (x 1.5 + a floor), the bar
tests/test_reference_render.py uses for the restated renderer under -O.  The SDF cases add the deterministic single-step comparison of the
executor with the autograd trainer from the same state (the late schedule is chaotic over several steps; one step is not)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import render_case as RC   # noqa: E402
from test_reference_render import _reference_on_hip, _state, fixture, relmax   # noqa: E402

MLP = ["sigma_net.net.0.weight", "sigma_net.net.1.weight", "color_net.net.0.weight", "color_net.net.1.weight", "color_net.net.2.weight",
       "specular_net.net.0.weight", "specular_net.net.1.weight"]

# per case: which fixture, recipe switches, the schedule position of step 1 as `iters` (SDF: ratio = 1 / (0.5 iters)), the TV weight (garden:
# the fixture's visible 1e-3 instead of the recipe's 1e-8, so that the inner / outer split is part of what is compared)
CASES = {
    "nerf": dict(fx="nerf", sdf=False, garden=False, iters=30000, lambda_tv=1e-8),
    "garden": dict(fx="garden", sdf=False, garden=True, iters=30000, lambda_tv=RC.GARDEN["lambda_tv"]),
    "sdf-mid": dict(fx="sdf", sdf=True, garden=False, iters=4, lambda_tv=1e-8),
    "sdf-late": dict(fx="sdf", sdf=True, garden=False, iters=2, lambda_tv=1e-8),
}


def _schedule(iters, step=1):
    """nerf/utils.py:651-655 at global_step = step."""
    r = min(1, step / (0.5 * iters))
    return dict(cos_anneal_ratio=r, normal_anneal_epsilon=1e-1 * (1 - min(0.999, step / (0.5 * iters))), max_level=4 + int(12 * r))


def _batch(device, garden):
    from nerf2mesh_amd import synthetic as S
    poses, _ = RC.cameras()
    o, d = S.crop_rays(poses, cam=0, size=RC.CROP)
    g = torch.Generator().manual_seed(5)
    rgba = torch.rand(o.shape[0], 4, generator=g)
    rgba[:, 3] = (rgba[:, 3] > 0.3).float()                       # a mask with both values
    cnf = None
    if garden:                                                    # the fixture's per-ray (near, far) pairs (render_case.run_case)
        g = torch.Generator().manual_seed(32)
        cnf = (RC.cam_near_far()[0].unsqueeze(0) + (torch.rand(o.shape[0], 2, generator=g) - 0.5) * torch.tensor([0.4, 1.0])).to(device).contiguous()
    return o.to(device).contiguous(), d.to(device).contiguous(), rgba.to(device).contiguous(), cnf


def _reference_step(case, fp16, bits, o, d, rgba, cnf, scale):
    """One iteration of the reference's train loop on its own model: returns M, sample positions, loss, unscaled gradients."""
    c = CASES[case]
    model = _reference_on_hip(c["sdf"], fp16=fp16, garden=c["garden"])
    opt = model.opt
    model.train()
    dt_gamma = 0
    if c["garden"]:
        model.update_aabb(np.asarray(RC.GARDEN["aabb"], dtype=np.float32))                          # main.py:234-235
        dt_gamma = RC.GARDEN["dt_gamma"]
    if c["sdf"]:
        sch = _schedule(c["iters"])
        opt.cos_anneal_ratio, opt.normal_anneal_epsilon, model.max_level = sch["cos_anneal_ratio"], sch["normal_anneal_epsilon"], sch["max_level"]
    model.density_bitfield.copy_(torch.from_numpy(bits).to(model.density_bitfield.device))
    gt_mask = rgba[:, 3:]
    gt_rgb = rgba[:, :3] * gt_mask + 1 * (1 - gt_mask)                                           # nerf/utils.py:663-666, white background
    with torch.autocast("cuda", dtype=torch.float16, enabled=fp16):                              # nerf/utils.py:1183
        out = model.render(o, d, bg_color=1, perturb=False, max_steps=1024, shading="full", dt_gamma=dt_gamma, cam_near_far=cnf)
        loss = 1.0 * torch.nn.functional.mse_loss(out["image"], gt_rgb, reduction="none").mean(-1)          # :676 (lambda_rgb 1, main.py)
        loss = loss + 0.1 * torch.nn.functional.mse_loss(out["weights_sum"], gt_mask.squeeze(1), reduction="none")      # :678-680 (lambda_mask 0.1)
        loss = loss.mean()                                                                        # :780
        if c["garden"]:                                                                           # :728-733 (lambda_entropy 1e-3, runall_360_outdoor.sh)
            w = out["weights"].clamp(1e-5, 1 - 1e-5)
            w2 = out["weights_sum"].clamp(1e-5, 1 - 1e-5)
            ent = lambda p: (-p * torch.log2(p) - (1 - p) * torch.log2(1 - p)).mean()
            loss = loss + RC.GARDEN["lambda_entropy"] * (ent(w) + ent(w2))
        loss = loss + 1e-5 * (out["speculars"] ** 2).sum(-1).mean()                               # :735-738 (lambda_specular 1e-5)
        if c["sdf"]:                                                                              # :740-743 (lambda_eikonal 0.1)
            loss = loss + 0.1 * ((torch.linalg.norm(out["normal"], ord=2, dim=-1) - 1) ** 2).mean()
    for p in model.parameters():
        p.grad = None
    (loss * scale).backward()                                                                        # :1187 scaler.scale(loss).backward()
    grads = {n: p.grad.detach().float() / scale for n, p in model.named_parameters() if p.grad is not None}     # :812 scaler.unscale_
    enc = model.encoder
    saved, enc.embeddings.grad = enc.embeddings.grad, grads["encoder.embeddings"].clone()
    xyzs = out["xyzs"].detach()
    if c["garden"]:                                                                               # :815-821
        inner = xyzs.abs().amax(dim=-1) <= 1
        enc.grad_total_variation(c["lambda_tv"], xyzs[inner].contiguous(), model.bound)
        enc.grad_total_variation(c["lambda_tv"] * 10, xyzs[~inner].contiguous(), model.bound)
    else:
        enc.grad_total_variation(c["lambda_tv"], xyzs, model.bound)                               # :823
    grads["encoder.embeddings"] = enc.embeddings.grad.clone()
    enc.embeddings.grad = saved
    return int(out["num_points"]), xyzs.float().cpu().numpy(), float(loss), {k: v.cpu().numpy() for k, v in grads.items()}


def _options(case):
    from nerf2mesh_amd.options import make_options
    c = CASES[case]
    kw = dict(O=True, bound=1, dt_gamma=0, iters=c["iters"], fused_mlp=True, diffuse_step=0, background="white", lambda_tv=c["lambda_tv"], sdf=c["sdf"])
    if c["garden"]:
        kw.update(bound=RC.GARDEN["bound"], dt_gamma=RC.GARDEN["dt_gamma"], lambda_entropy=RC.GARDEN["lambda_entropy"], enable_cam_near_far=True,
                  scene="garden")
    return make_options(**kw)


class _FixtureBatch:
    """synthetic.batch_from_uniforms replaced for the duration: the driver's own batch kernel runs (counters, buffer set), then the fixture's
    rays, ground truth, jitter 0 (= perturb off) and near / far (+ the per-ray clamp of nerf/renderer.py:689-691) overwrite its first N rows."""

    def __init__(self, o, d, rgba, cnf):
        self.o, self.d, self.rgba, self.cnf = o, d, rgba, cnf

    def __enter__(self):
        from nerf2mesh_amd import raymarching, synthetic
        self.synthetic, self.orig = synthetic, synthetic.batch_from_uniforms
        N = self.o.shape[0]

        def fixture_batch(poses, images, u, aabb, min_near, out=None, counter=None, cam_near_far=None, **kw):
            r = self.orig(poses, images, u, aabb, min_near, out=out, counter=counter, cam_near_far=cam_near_far, **kw)
            bo, bd, brgba, nears, fars, noises, bg = r if out is None else out
            if bo.shape[0] < N:                                      # a LATER batch of the autograd trainer (adaptive num_rays): not the step compared
                return r
            bo[:N].copy_(self.o); bd[:N].copy_(self.d); brgba[:N].copy_(self.rgba); noises[:N].zero_()
            n2, f2 = raymarching.near_far_from_aabb(self.o, self.d, aabb, min_near)
            if self.cnf is not None:
                n2, f2 = torch.maximum(n2, self.cnf[:, 0]), torch.minimum(f2, self.cnf[:, 1])
            nears[:N].copy_(n2); fars[:N].copy_(f2)
            return r
        synthetic.batch_from_uniforms = fixture_batch
        return self

    def __exit__(self, *a):
        self.synthetic.batch_from_uniforms = self.orig


def _engine_step(case, bits, o, d, rgba, cnf, scale):
    """Stage0Engine for one step from the fixture's state; returns M, positions, loss, the gradients in front of the optimizer (unscaled)."""
    from nerf2mesh_amd.engine import Stage0Engine
    from nerf2mesh_amd.network import NeRFNetwork
    c = CASES[case]
    dev = o.device
    torch.manual_seed(0)
    opt = _options(case)
    model = NeRFNetwork(opt)
    model.load_state_dict(_state(c["sdf"], c["garden"]), strict=False)
    if c["garden"]:
        model.update_aabb(np.asarray(RC.GARDEN["aabb"], dtype=np.float32))
    eng = Stage0Engine(model, opt, RC.cameras()[0], dev, seed=0)
    eng.model.density_bitfield.copy_(torch.from_numpy(bits).to(dev))
    eng._refresh = lambda: None                                      # (the step in front of batch 1 would refresh the grid: the fixture's stays)
    N = o.shape[0]
    eng.num_rays = N
    eng.optimizer.scale.fill_(scale)
    SCALE = scale
    cap = {}
    orig_finish, orig_lr = eng._finish, eng._lr_step

    def finish(b):
        M = orig_finish(b)
        cap.setdefault("b", (b, M))
        return M

    def lr_step(*a, **k):
        if "g1" not in cap:                                          # the gradients of step 1, before the optimizer consumes them
            torch.cuda.synchronize()
            cap["g1"], cap["g2"] = eng.g1.detach().clone(), eng.g2.detach().clone()
            cap["dw"] = [v.detach().clone() for v in eng.dw_views]
            if c["sdf"]:
                cap["dvar"] = eng._sdf_buf()["d_var"].detach().clone()
        return orig_lr(*a, **k)
    eng._finish, eng._lr_step = finish, lr_step
    with _FixtureBatch(o, d, rgba, cnf):
        loss = float(eng.train_step())
    torch.cuda.synchronize()
    b, M = cap["b"]
    eng.overflowed = float(eng.optimizer.scale) < scale          # (the bookkeeping launch has backed the scale off and cleared found_inf)
    grads = {"encoder.embeddings": cap["g1"].float().cpu().numpy() / SCALE, "encoder_color.embeddings": cap["g2"].float().cpu().numpy() / SCALE}
    for name, g in zip(MLP, cap["dw"]):
        grads[name] = g.float().cpu().numpy() / SCALE
    if c["sdf"]:
        grads["variance"] = cap["dvar"].float().cpu().numpy() / SCALE
    return eng, M, b.samples[:3 * M].view(M, 3).cpu().numpy(), loss, grads


def _trainer_step(case, bits, o, d, rgba, cnf, scale):
    """trainer.Stage0Trainer (the same kernels through torch.autograd) for one step from the same state: gradients as its optimizer sees them."""
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.trainer import Stage0Trainer
    c = CASES[case]
    dev = o.device
    torch.manual_seed(0)
    opt = _options(case)
    model = NeRFNetwork(opt)
    model.load_state_dict(_state(c["sdf"], c["garden"]), strict=False)
    if c["garden"]:
        model.update_aabb(np.asarray(RC.GARDEN["aabb"], dtype=np.float32))
    tr = Stage0Trainer(model, opt, RC.cameras()[0], dev, seed=0)
    tr.model.density_bitfield.copy_(torch.from_numpy(bits).to(dev))
    tr.model.update_extra_state = lambda *a, **k: None
    tr.num_rays = o.shape[0]
    tr.optimizer.scale.fill_(scale)
    SCALE = scale
    cap = {}
    opt_ = tr.optimizer
    orig_step = opt_.step

    def step(flagged=()):
        if not cap:
            torch.cuda.synchronize()
            for name, p in tr.model.named_parameters():
                g = None
                for table in (opt_.half_grads, opt_.ext_grads):
                    if g is None and p in table:
                        g = table[p]()
                g = p.grad if g is None else g
                if g is not None:
                    cap[name] = g.detach().float().cpu().numpy().copy() / SCALE
        return orig_step(flagged=flagged)
    opt_.step = step
    with _FixtureBatch(o, d, rgba, cnf):
        tr.train_step()
    torch.cuda.synchronize()
    return tr, cap


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_step_executor_against_the_unchanged_reference_iteration(case):
    c = CASES[case]
    dev = torch.device("cuda", 0)
    fx = fixture(c["fx"])
    bits = fx["density_bitfield"]
    o, d, rgba, cnf = _batch(dev, c["garden"])
    # the loss scale of the step: GradScaler's (nerf/utils.py:506: 65536, halved after every step whose gradients overflow) -- the largest
    # one at which neither the reference's -O graph nor the executor overflows (the SDF recipe's 1 / epsilon gradients on an fp16 path settle
    # it lower within a few steps of a real run); the same on all sides
    scale = 65536.0
    for _ in range(12):
        m16, x16, l16, g16 = _reference_step(case, True, bits, o, d, rgba, cnf, scale)
        finite = all(np.isfinite(v).all() for v in g16.values())
        eng, M, xyz, loss, mine = _engine_step(case, bits, o, d, rgba, cnf, scale) if finite else (None,) * 5
        if finite and not eng.overflowed:
            break
        scale *= 0.5
    else:
        raise AssertionError("no loss scale down to 16 at which the step does not overflow")
    m32, x32, l32, g32 = _reference_step(case, False, bits, o, d, rgba, cnf, scale)
    assert m32 == m16 == int(fx["num_points"]), "same bit field, same rays: the fixture's sample count"
    assert M == m32, f"executor marched {M} samples, the reference {m32}"
    assert np.array_equal(xyz, x32), "sample positions differ from the reference's"
    if c["sdf"]:
        sch = _schedule(c["iters"])
        assert eng.model.max_level == sch["max_level"] and abs(eng.opt.normal_anneal_epsilon - sch["normal_anneal_epsilon"]) < 1e-12
        if case == "sdf-late":      # the folded copies + the per-level lists call, not the stacked pass
            assert 0 < eng.last_fold_left < 0.1 * 16 * 6 * M
        else:
            assert not hasattr(eng, "last_fold_left")

    rows, bad = [], []
    names = ["encoder.embeddings", "encoder_color.embeddings"] + MLP + (["variance"] if c["sdf"] else [])
    for name in names:
        truth, ref16 = g32[name], g16[name]
        got = mine[name].reshape(truth.shape)
        if "embeddings" in name:                                     # the dense head like the fixtures (levels 0-2 and the start of 3) and the whole table
            e16, e = max(relmax(ref16[:65536], truth[:65536]), relmax(ref16, truth)), max(relmax(got[:65536], truth[:65536]), relmax(got, truth))
            # rows the executor leaves at zero: no more than the reference's own fp16 graph loses to underflow (x 2)
            assert (np.abs(got).sum(-1) != 0).sum() > 0.5 * (np.abs(ref16).sum(-1) != 0).sum()
        elif name == "variance":                                     # a scalar: relative error
            den = max(abs(float(truth)), 1e-30)
            e16, e = abs(float(ref16) - float(truth)) / den, abs(float(got) - float(truth)) / den
        else:
            e16, e = relmax(ref16, truth), relmax(got, truth)
        rows.append(f"  {name:28s} vs fp32 reference: reference -O {e16:.3g}, executor {e:.3g}")
        if not np.isfinite(e16):                                     # the reference's own fp16 graph overflowed: the executor must simply be finite and near
            e16 = 0.25
        if not e <= 1.5 * e16 + 2e-3:
            bad.append(rows[-1])
    rows.append(f"  loss: fp32 {l32:.6g}, reference -O {l16:.6g}, executor {loss:.6g}")
    print(f"\n[{case}] loss scale {scale:g}; step executor vs the unchanged reference iteration (gradients, relative to the largest entry):\n" + "\n".join(rows))
    assert not bad, "\n".join(bad)
    if loss != 0.0:                                                  # (the executor reports a step's loss one bookkeeping launch later)
        assert abs(loss - l32) <= 5e-3 * abs(l32) + abs(l16 - l32) * 1.5


# engine vs trainer, one step, same state: fp32 / fp16 association only (the folded copies sum a +eps / -eps pair inside the batch's backward,
# the trainer in a second call onto the same rows; the weight gradients are summed per workgroup in both).  Relative to the largest entry.
ONE_STEP_TOL = {"encoder.embeddings": 1e-4, "encoder_color.embeddings": 1e-4, "variance": 1e-4, "mlp": 1e-4}      # measured: <= 1.8e-5 (SDF), 0 (garden)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["sdf-mid", "sdf-late", "garden"])
def test_step_executor_equals_the_autograd_trainer_on_one_step(case):
    """The deterministic single-step form of tests/test_engine.py's multi-step comparisons: the SDF recipe late in its schedule (epsilon 1e-4,
    folded copies + lists path), half way (stacked pass) and the outdoor recipe, executor against trainer from an identical state -- gradients
    in front of the optimizer.  Several steps of the late SDF schedule amplify rounding by orders of magnitude (test_engine.py); one step does not."""
    c = CASES[case]
    dev = torch.device("cuda", 0)
    bits = fixture(c["fx"])["density_bitfield"]
    o, d, rgba, cnf = _batch(dev, c["garden"])
    scale = 65536.0
    for _ in range(12):
        eng, M, xyz, loss, mine = _engine_step(case, bits, o, d, rgba, cnf, scale)
        if not eng.overflowed:
            break
        scale *= 0.5
    assert not eng.overflowed
    tr, theirs = _trainer_step(case, bits, o, d, rgba, cnf, scale)
    assert float(tr.optimizer.scale) >= scale, "the trainer overflows at a loss scale the executor does not"
    assert tr.last_num_points == M
    rows, bad = [], []
    for name in ["encoder.embeddings", "encoder_color.embeddings"] + MLP + (["variance"] if c["sdf"] else []):
        a, b = mine[name].reshape(-1), theirs[name].reshape(-1)
        e = relmax(a, b)
        tol = ONE_STEP_TOL.get(name, ONE_STEP_TOL["mlp"])
        rows.append(f"  {name:28s} executor vs trainer {e:.3g} (limit {tol:.3g})")
        if not e <= tol:
            bad.append(rows[-1])
    print(f"\n[{case}] loss scale {scale:g}; one step, executor vs autograd trainer (gradients, relative to the largest entry):\n" + "\n".join(rows))
    assert not bad, "\n".join(bad)
