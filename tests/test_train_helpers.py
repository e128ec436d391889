"""GPU parity of the fused training-step helpers against the torch graph the reference builds (nerf/utils.py:658-683,
nerf/renderer.py:747).  fp32 tolerance: the kernel sums in a different (fixed) order than torch's reductions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [1, 255, 16588])
@pytest.mark.parametrize("bg_kind", ["white", "random"])
def test_photo_loss_matches_torch_graph(N, bg_kind):
    import torch
    import torch.nn.functional as F
    from nerf2mesh_amd.losses import photo_loss
    g = torch.Generator(device="cuda").manual_seed(N)
    image = torch.rand(N, 3, device="cuda", generator=g).requires_grad_()
    ws = torch.rand(N, device="cuda", generator=g).requires_grad_()
    gt = torch.rand(N, 4, device="cuda", generator=g)
    gt[: N // 3, 3] = 0
    gt[N // 3: N // 2, 3] = 1
    bg = 1 if bg_kind == "white" else torch.rand(N, 3, device="cuda", generator=g)
    lam_rgb, lam_mask = 1.0, 0.1

    pred = image + (1 - ws).unsqueeze(-1) * bg
    mask = gt[..., 3:]
    target = gt[..., :3] * mask + bg * (1 - mask)
    ref = (lam_rgb * F.mse_loss(pred, target, reduction="none").mean(-1) + lam_mask * F.mse_loss(ws, mask.squeeze(1), reduction="none")).mean()
    (ref * 1024.0).backward()
    gi, gw = image.grad.clone(), ws.grad.clone()
    image.grad = ws.grad = None

    got = photo_loss(image, ws, gt, bg, lam_rgb, lam_mask)
    (got * 1024.0).backward()
    np.testing.assert_allclose(got.item(), ref.item(), rtol=2e-6)
    np.testing.assert_allclose(image.grad.cpu().numpy(), gi.cpu().numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(ws.grad.cpu().numpy(), gw.cpu().numpy(), rtol=1e-5, atol=1e-7)
    again = photo_loss(image.detach(), ws.detach(), gt, bg, lam_rgb, lam_mask)
    assert again.item() == got.item(), "fixed summation order: bit-reproducible"


def test_get_rays_matches_torch_statement():
    """n2m_get_rays == synthetic.rays_from_pixels (get_rays of nerf/utils.py:242-290 in torch ops) + images[cam, pix]."""
    import torch
    from nerf2mesh_amd import synthetic as S
    poses = S.make_cameras(7, seed=3).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    H = W = 96
    images = torch.rand(7, H * W, 4, device="cuda", generator=g)
    N = 5003
    g2 = torch.Generator(device="cuda").manual_seed(5)
    o, d, rgba = S.random_batch(poses, images, N, g2, H, W, 111.5)
    g2.manual_seed(5)
    cam = torch.randint(0, 7, (N,), device="cuda", generator=g2)
    pix = torch.randint(0, H * W, (N,), device="cuda", generator=g2)
    ro, rd = S.rays_from_pixels(poses, cam, pix, H, W, 111.5)
    assert torch.equal(o, ro)
    np.testing.assert_allclose(d.cpu().numpy(), rd.cpu().numpy(), rtol=2e-6, atol=1e-7)     # same products, fp32 sum order may differ
    assert torch.equal(rgba, images[cam, pix])
