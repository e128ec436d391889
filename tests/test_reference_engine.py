"""The step executor itself (engine.Stage0Engine, the path bench.py times) against the UNCHANGED reference Python.

tests/test_reference_render.py pins the restated renderer / trainer to the reference's own `render`; tests/test_engine.py holds the executor
to the trainer.  Here the executor is driven for ONE training step from the exact state of BASELINE config 1's fixture
(tests/golden/render_nerf.npz: parameters of render_case.make_state, the fixture's occupancy bit field, the 64x64 crop of camera 0) and
compared with the reference's own iteration on the same device:

    nerf/utils.py:640-683 (train_step: background, MSE on rgb + mask, mean), :735-738 (specular regulariser), :1187 (scaler.scale(loss).backward()),
    :802-823 (post_train_step: unscale, in-place TV at the batch's samples)          over          nerf/renderer.py:676-813 (render)

run by the unchanged reference Python over the HIP `_backend` modules (oracle/_ref/pyref on the GPU box), once under `-O` (fp16 autocast) --
the yardstick -- and once in fp32 -- the truth.  Sample count and sample positions must be EXACT; every gradient of the executor (fused MFMA
field, binned fixed-point table backward, TV folded in) must be no farther from the fp32 truth than the reference's own fp16 graph is
(x 1.5 + a floor), the bar tests/test_reference_render.py uses for the restated renderer under -O."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import render_case as RC   # noqa: E402
from test_reference_render import _reference_on_hip, fixture, relmax   # noqa: E402

SCALE = 65536.0
MLP = ["sigma_net.net.0.weight", "sigma_net.net.1.weight", "color_net.net.0.weight", "color_net.net.1.weight", "color_net.net.2.weight",
       "specular_net.net.0.weight", "specular_net.net.1.weight"]


def _batch(device):
    from nerf2mesh_amd import synthetic as S
    poses, _ = RC.cameras()
    o, d = S.crop_rays(poses, cam=0, size=RC.CROP)
    g = torch.Generator().manual_seed(5)
    rgba = torch.rand(o.shape[0], 4, generator=g)
    rgba[:, 3] = (rgba[:, 3] > 0.3).float()                       # a mask with both values
    return o.to(device).contiguous(), d.to(device).contiguous(), rgba.to(device).contiguous()


def _reference_step(fp16, bits, o, d, rgba):
    """One iteration of the reference's train loop on its own model: returns M, sample positions, loss, unscaled gradients."""
    model = _reference_on_hip(False, fp16=fp16)
    opt = model.opt
    model.train()
    model.density_bitfield.copy_(torch.from_numpy(bits).to(model.density_bitfield.device))
    gt_mask = rgba[:, 3:]
    gt_rgb = rgba[:, :3] * gt_mask + 1 * (1 - gt_mask)                                           # nerf/utils.py:663-666, white background
    with torch.autocast("cuda", dtype=torch.float16, enabled=fp16):                              # nerf/utils.py:1183
        out = model.render(o, d, bg_color=1, perturb=False, max_steps=1024, shading="full", dt_gamma=0)
        loss = 1.0 * torch.nn.functional.mse_loss(out["image"], gt_rgb, reduction="none").mean(-1)          # :676 (lambda_rgb 1, main.py)
        loss = loss + 0.1 * torch.nn.functional.mse_loss(out["weights_sum"], gt_mask.squeeze(1), reduction="none")      # :678-680 (lambda_mask 0.1)
        loss = loss.mean()                                                                        # :780
        loss = loss + 1e-5 * (out["speculars"] ** 2).sum(-1).mean()                               # :735-738 (lambda_specular 1e-5)
    for p in model.parameters():
        p.grad = None
    (loss * SCALE).backward()                                                                     # :1187 scaler.scale(loss).backward()
    grads = {n: p.grad.detach().float() / SCALE for n, p in model.named_parameters() if p.grad is not None}     # :812 scaler.unscale_
    enc = model.encoder
    saved, enc.embeddings.grad = enc.embeddings.grad, grads["encoder.embeddings"].clone()
    enc.grad_total_variation(1e-8, out["xyzs"].detach(), model.bound)                             # :823 (lambda_tv 1e-8)
    grads["encoder.embeddings"] = enc.embeddings.grad.clone()
    enc.embeddings.grad = saved
    return int(out["num_points"]), out["xyzs"].detach().float().cpu().numpy(), float(loss), {k: v.cpu().numpy() for k, v in grads.items()}


@pytest.mark.gpu
def test_step_executor_against_the_unchanged_reference_iteration():
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.engine import Stage0Engine
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    dev = torch.device("cuda", 0)
    fx = fixture("nerf")
    bits = fx["density_bitfield"]
    o, d, rgba = _batch(dev)
    N = o.shape[0]
    m32, x32, l32, g32 = _reference_step(False, bits, o, d, rgba)
    m16, x16, l16, g16 = _reference_step(True, bits, o, d, rgba)
    assert m32 == m16 == int(fx["num_points"]), "same bit field, same rays: the fixture's sample count"

    # ---- the executor: same parameters, same bit field, same batch (its own batch kernel replaced by the fixture's rays, jitter 0 = perturb off)
    torch.manual_seed(0)
    opt = make_options(O=True, bound=1, dt_gamma=0, iters=30000, fused_mlp=True, diffuse_step=0, background="white")
    model = NeRFNetwork(opt)
    model.load_state_dict(RC.make_state(False), strict=False)
    eng = Stage0Engine(model, opt, RC.cameras()[0], dev, seed=0)
    eng.model.density_bitfield.copy_(torch.from_numpy(bits).to(dev))
    eng._refresh = lambda: None                                      # (the step in front of batch 1 would refresh the grid: the fixture's stays)
    eng.num_rays = N
    cap = {}
    orig_batch = synthetic.batch_from_uniforms

    def fixture_batch(poses, images, u, aabb, min_near, out=None, counter=None, cam_near_far=None):
        r = orig_batch(poses, images, u, aabb, min_near, out=out, counter=counter, cam_near_far=cam_near_far)     # near / far, counters: the kernel's own
        bo, bd, brgba, nears, fars, noises, bg = out
        assert bo.shape[0] >= N
        bo[:N].copy_(o); bd[:N].copy_(d); brgba[:N].copy_(rgba); noises[:N].zero_()
        from nerf2mesh_amd import raymarching
        n2, f2 = raymarching.near_far_from_aabb(o, d, aabb, min_near)
        nears[:N].copy_(n2); fars[:N].copy_(f2)
        return r
    synthetic.batch_from_uniforms = fixture_batch
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
        return orig_lr(*a, **k)
    eng._finish, eng._lr_step = finish, lr_step
    try:
        loss = float(eng.train_step())
    finally:
        synthetic.batch_from_uniforms = orig_batch
    torch.cuda.synchronize()
    b, M = cap["b"]
    assert M == m32, f"executor marched {M} samples, the reference {m32}"
    xyz = b.samples[:3 * M].view(M, 3).cpu().numpy()
    assert np.array_equal(xyz, x32), "sample positions differ from the reference's"
    scale = float(eng.optimizer.scale) if float(eng.optimizer.found_inf) == 0 else SCALE
    assert float(eng.optimizer.found_inf) == 0

    rows, bad = [], []
    mine = {"encoder.embeddings": cap["g1"].float().cpu().numpy() / SCALE, "encoder_color.embeddings": cap["g2"].float().cpu().numpy() / SCALE}
    for name, g in zip(MLP, cap["dw"]):
        mine[name] = g.float().cpu().numpy().reshape(g32[name].shape) / SCALE
    for name in ["encoder.embeddings", "encoder_color.embeddings"] + MLP:
        truth, ref16, got = g32[name], g16[name], mine[name].reshape(g32[name].shape)
        if "embeddings" in name:                                     # the dense head like the fixtures (levels 0-2 and the start of 3) and the whole table
            e16, e = max(relmax(ref16[:65536], truth[:65536]), relmax(ref16, truth)), max(relmax(got[:65536], truth[:65536]), relmax(got, truth))
            assert (np.abs(got).sum(-1) != 0).sum() > 0.5 * (np.abs(truth).sum(-1) != 0).sum()
        else:
            e16, e = relmax(ref16, truth), relmax(got, truth)
        rows.append(f"  {name:28s} vs fp32 reference: reference -O {e16:.3g}, executor {e:.3g}")
        if not e <= 1.5 * e16 + 2e-3:
            bad.append(rows[-1])
    rows.append(f"  loss: fp32 {l32:.6g}, reference -O {l16:.6g}, executor {loss:.6g}")
    print("\nstep executor vs the unchanged reference iteration (gradients, relative to the largest entry):\n" + "\n".join(rows))
    assert not bad, "\n".join(bad)
    if loss != 0.0:                                                  # (the executor reports a step's loss one bookkeeping launch later)
        assert abs(loss - l32) <= 5e-3 * abs(l32) + abs(l16 - l32) * 1.5
