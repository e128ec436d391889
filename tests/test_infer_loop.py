"""The inference loop with the ray count on the device (renderer._infer_loop_device over n2m_march_rays_dev / n2m_composite_rays_dev /
n2m_compact_alive_dev) against the reference's host-paced loop (nerf/renderer.py:764-802, restated in renderer._infer_loop_host over the
kernels that are checked against the oracle in tests/test_hip_parity.py): same image, depth and opacity BIT FOR BIT with the fused field
(per-sample results independent of the batch size), same round count, no host read of n_alive on the critical path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _trained(steps=80, **over):
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.engine import Stage0Engine
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    torch.manual_seed(0)
    kw = dict(O=True, bound=1, dt_gamma=0, iters=30000, fused_mlp=True)
    kw.update(over)
    opt = make_options(**kw)
    model = NeRFNetwork(opt)
    if opt.scene == "garden":
        model.update_aabb(synthetic.pts_aabb("garden"))
    eng = Stage0Engine(model, opt, synthetic.make_cameras(8, seed=0), torch.device("cuda", 0), seed=0)
    eng.mark_untrained()
    for _ in range(steps):
        eng.train_step()
    torch.cuda.synchronize()
    return eng


def _view(eng, cam, res=160):
    from nerf2mesh_amd import synthetic
    dev = eng.device
    ds = synthetic.LEGO_HW // res
    jj, ii = torch.meshgrid(torch.arange(res, device=dev), torch.arange(res, device=dev), indexing="ij")
    pix = (jj * ds * synthetic.LEGO_HW + ii * ds).reshape(-1)
    return synthetic.rays_from_pixels(eng.poses, torch.full_like(pix, cam), pix)


@pytest.mark.parametrize("recipe", ["lego", "garden"])
def test_device_count_loop_renders_the_same_image(recipe):
    from nerf2mesh_amd import renderer as R
    over = {} if recipe == "lego" else dict(bound=16, dt_gamma=1 / 256, lambda_entropy=1e-3, enable_cam_near_far=True, scene="garden")
    eng = _trained(**over)
    model = eng.model.eval()
    o, d = _view(eng, 2)
    outs = {}
    for name, host in (("host", True), ("device", False)):
        R._HOST_INFER_LOOP = host
        try:
            with torch.no_grad():
                res = model.render(o, d, bg_color=1, perturb=False, shading="full", dt_gamma=eng.opt.dt_gamma, max_steps=eng.opt.max_steps,
                                   T_thresh=1e-4)
        finally:
            R._HOST_INFER_LOOP = False
        outs[name] = {k: res[k].detach().cpu().numpy() for k in ("image", "depth")}
    assert outs["host"]["image"].std() > 0.02, "nothing was rendered"
    for k in ("image", "depth"):
        assert np.array_equal(outs["host"][k], outs["device"][k]), f"{k}: {np.abs(outs['host'][k] - outs['device'][k]).max()}"
    assert 2 <= model.last_infer_rounds <= eng.opt.max_steps + 4


def test_device_count_loop_handles_rays_that_see_nothing():
    """Every ray misses the box: n_alive drops to zero in the first round, the loop ends after the run-ahead rounds, outputs stay zero
    (white after the background blend)."""
    eng = _trained(steps=20)
    model = eng.model.eval()
    o, d = _view(eng, 0, res=40)
    with torch.no_grad():
        res = model.render(o + 100.0, d, bg_color=1, perturb=False, shading="full", dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    assert torch.equal(res["image"], torch.ones_like(res["image"])) and float(res["depth"].abs().max()) == 0.0
    assert model.last_infer_rounds <= 5
