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


@pytest.mark.parametrize("bg_kind", ["white", "random"])
def test_fused_composite_loss_equals_the_four_kernel_chain(bg_kind):
    """n2m_composite_loss_train == composite forward -> photo loss forward/backward -> composite backward (the chain whose parts are
    checked against the oracle / the torch graph elsewhere): gradients, opacities and colours bit for bit; the loss value up to its
    summation order.  Rays: the marcher's own output on the synthetic scene (ranges tile [0, M)), plus empty rays and early stops."""
    import torch
    from nerf2mesh_amd import _lib as L, raymarching, synthetic as S
    from nerf2mesh_amd.losses import photo_loss
    dev = torch.device("cuda")
    poses = S.make_cameras(16, seed=2).to(dev)
    bits = raymarching.packbits(S.scene_density_grid(H=128, device=dev), 10.0)
    g = torch.Generator(device=dev).manual_seed(9)
    o, d = S.random_rays(poses, 6000, g)
    nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.05)
    xyzs, dirs, ts, rays = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, 0.0, 1024)
    M, N = xyzs.shape[0], o.shape[0]
    assert M > 20000 and int((rays[:, 1] == 0).sum()) > 0
    sig = (torch.rand(M, device=dev, generator=g) * 60).requires_grad_()          # dense enough for early stops
    rgb = torch.rand(M, 3, device=dev, generator=g).requires_grad_()
    gt = torch.rand(N, 4, device=dev, generator=g)
    gt[: N // 3, 3] = 0
    bg = 1 if bg_kind == "white" else torch.rand(N, 3, device=dev, generator=g)
    scale = torch.tensor(512.0, device=dev)
    w, ws, dp, im = raymarching.composite_rays_train(sig, rgb, ts, rays, 1e-4, False, rays_tile_samples=True)
    loss = photo_loss(im, ws, gt, bg, 1.0, 0.1)
    loss.backward(gradient=scale)
    d_sr = torch.empty(4 * M, device=dev)
    out_ws, out_im = torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
    partial = torch.empty((N + 3) // 4, device=dev)
    ticket = torch.zeros(1, dtype=torch.int32, device=dev)
    lv, lsum = torch.zeros(1, device=dev), torch.full((1,), 2.0, device=dev)
    bg_t = bg if torch.is_tensor(bg) else None
    # + the optional live counts (round 5, n2m_composite_live_counts): per ray the samples up to and including the early stop
    live = torch.full((N,), -7, dtype=torch.int32, device=dev)
    block_live = torch.full(((N + 15) // 16,), -7, dtype=torch.int32, device=dev)
    L.call("n2m_composite_live_counts", L.ptr(live), L.ptr(block_live))
    try:
        L.call("n2m_composite_loss_train", L.ptr(sig.detach()), L.ptr(rgb.detach()), L.ptr(ts), L.ptr(rays), M, N, 1e-4, L.ptr(gt), L.ptr(bg_t),
               1.0 if bg_t is None else 0.0, 1.0, 0.1, L.ptr(scale), L.ptr(out_ws), L.ptr(out_im), L.ptr(d_sr[:M]), L.ptr(d_sr[M:]), L.ptr(partial),
               L.ptr(ticket), L.ptr(lv), L.ptr(lsum), L.stream())
    finally:
        L.call("n2m_composite_live_counts", None, None)
    assert torch.equal(out_ws, ws.detach()) and torch.equal(out_im, im.detach())
    assert torch.equal(d_sr[:M], sig.grad) and torch.equal(d_sr[M:].view(M, 3), rgb.grad)
    np.testing.assert_allclose(lv.item(), loss.item(), rtol=5e-6)
    assert abs(lsum.item() - (2.0 + lv.item())) < 1e-6 and int(ticket) == 0
    # the stop position is where the forward's weights end: behind it composite_rays_train leaves zeros (raymarching.cu:553), and the
    # sample the stop fell on still has a weight (sigma > 0 everywhere here)
    rays_np, w_np, live_np = rays.cpu().numpy(), w.detach().cpu().numpy(), live.cpu().numpy()
    gs, gr = d_sr[:M].cpu().numpy(), d_sr[M:].view(M, 3).cpu().numpy()
    stopped = 0
    for r in range(N):
        off, cnt = int(rays_np[r, 0]), int(rays_np[r, 1])
        nz = np.flatnonzero(w_np[off:off + cnt])
        want = (int(nz[-1]) + 1) if nz.size else 0
        if cnt and want < cnt:
            stopped += 1
        assert live_np[r] == (want if cnt else 0), (r, live_np[r], want, cnt)
        assert not gs[off + live_np[r]:off + cnt].any() and not gr[off + live_np[r]:off + cnt].any()
    assert stopped > 200
    assert np.array_equal(block_live.cpu().numpy(), np.add.reduceat(live_np, np.arange(0, N, 16)))


def test_fused_composite_loss_with_the_entropy_regulariser():
    """n2m_composite_loss_train_ent == the autograd statement of config 4's loss (nerf/utils.py:728-733 on top of the rgb + mask loss):
    the entropy of the sample weights reaches composite_rays_train's backward as grad_weights (the per-sample factor the reference kernel
    applies at raymarching.cu:676), the entropy of weights_sum as an extra grad_weights_sum.  Gradients to 2e-5 of their maximum (log2
    spelled once instead of twice, fp32), loss value to 1e-5; lambda = 0 reproduces the plain kernel bit for bit."""
    import torch
    from nerf2mesh_amd import _lib as L, raymarching, synthetic as S
    from nerf2mesh_amd.losses import photo_loss
    dev = torch.device("cuda")
    poses = S.make_cameras(16, seed=2).to(dev)
    bits = raymarching.packbits(S.scene_density_grid(H=128, device=dev), 10.0)
    g = torch.Generator(device=dev).manual_seed(19)
    o, d = S.random_rays(poses, 6000, g)
    nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.05)
    xyzs, dirs, ts, rays = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, 0.0, 1024)
    M, N = xyzs.shape[0], o.shape[0]
    sig = (torch.rand(M, device=dev, generator=g) * 40).requires_grad_()          # early stops on most rays that hit
    rgb = torch.rand(M, 3, device=dev, generator=g).requires_grad_()
    gt = torch.rand(N, 4, device=dev, generator=g)
    bg = torch.rand(N, 3, device=dev, generator=g)
    scale = torch.tensor(256.0, device=dev)
    lam = 0.05                                                                    # (the recipe's 1e-3 would hide in the fp32 noise of the rgb term)
    w, ws, dp, im = raymarching.composite_rays_train(sig, rgb, ts, rays, 1e-4, False, rays_tile_samples=True)
    loss = photo_loss(im, ws, gt, bg, 1.0, 0.1)
    ent = lambda p: (-p * torch.log2(p) - (1 - p) * torch.log2(1 - p)).mean()
    loss = loss + lam * (ent(w.clamp(1e-5, 1 - 1e-5)) + ent(ws.clamp(1e-5, 1 - 1e-5)))
    loss.backward(gradient=scale)

    def fused(lam_):
        d_sr = torch.empty(4 * M, device=dev)
        partial = torch.empty((N + 15) // 16, device=dev)
        L.call("n2m_composite_loss_train_ent", L.ptr(sig.detach()), L.ptr(rgb.detach()), L.ptr(ts), L.ptr(rays), M, N, 1e-4, L.ptr(gt), L.ptr(bg),
               0.0, 1.0, 0.1, L.ptr(scale), None, None, L.ptr(d_sr[:M]), L.ptr(d_sr[M:]), L.ptr(partial), None, None, None, float(lam_), L.stream())
        return d_sr[:M].clone(), d_sr[M:].view(M, 3).clone(), float(partial.double().sum() / N)
    gs, gr, lv = fused(lam)
    assert torch.equal(gr, rgb.grad)                                               # the colour gradient does not see the term
    err = (gs - sig.grad).abs().max().item() / sig.grad.abs().max().item()
    assert err <= 2e-5, err
    assert abs(lv - loss.item()) <= 1e-5 * abs(loss.item())
    # the term matters at this weight, and vanishes exactly at lambda = 0
    gs0, gr0, lv0 = fused(0.0)
    assert (gs0 - gs).abs().max().item() > 1e-3 * gs.abs().max().item()
    d_sr = torch.empty(4 * M, device=dev)
    partial = torch.empty((N + 15) // 16, device=dev)
    L.call("n2m_composite_loss_train", L.ptr(sig.detach()), L.ptr(rgb.detach()), L.ptr(ts), L.ptr(rays), M, N, 1e-4, L.ptr(gt), L.ptr(bg),
           0.0, 1.0, 0.1, L.ptr(scale), None, None, L.ptr(d_sr[:M]), L.ptr(d_sr[M:]), L.ptr(partial), None, None, None, L.stream())
    assert torch.equal(d_sr[:M], gs0) and torch.equal(d_sr[M:].view(M, 3), gr0)


def test_sdf_head_kernels_against_the_torch_statement():
    """n2m_sdf_offsets / n2m_sdf_alpha_forward / n2m_sdf_alpha_backward against the torch statement of the reference's SDF branch that
    renderer.render / network.normal spell out (nerf/renderer.py:724-739, nerf/network.py:143-154, eikonal term nerf/utils.py:740-743,
    pinned to the unchanged reference Python by tests/test_reference_render.py[sdf]): offsets bit for bit, alpha / normal to fp32 rounding,
    gradients w.r.t. the sdf, the six finite-difference values and the variance to 2e-5 of their maximum."""
    import torch
    import torch.nn.functional as F
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd.renderer import safe_normalize
    p = L.ptr
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(5)
    M, eps, car, bound = 70001, 3e-2, 0.3, 1.0
    xyz = (torch.rand(M, 3, device=dev, generator=g) * 2 - 1) * 1.01
    xyz.clamp_(-1, 1)
    pts = torch.empty(M, 6, 3, device=dev); pts01 = torch.empty(M, 6, 3, device=dev)         # sample-major: the six copies adjacent
    L.call("n2m_sdf_offsets", p(xyz), M, eps, bound, p(pts), p(pts01), L.stream())
    pts, pts01 = pts.permute(1, 0, 2), pts01.permute(1, 0, 2)                                  # -> the torch statement's [6, M, 3]
    off = torch.zeros(6, 1, 3, device=dev)
    for axis in range(3):
        off[2 * axis, 0, axis] = eps
        off[2 * axis + 1, 0, axis] = -eps
    want = (xyz.unsqueeze(0) + off).clamp(-bound, bound)
    assert torch.equal(pts, want) and torch.equal(pts01, (want + bound) / (2 * bound))

    sdf = (torch.randn(M, device=dev, generator=g) * 0.05).requires_grad_()
    s6 = (sdf.detach().unsqueeze(0) + torch.randn(6, M, device=dev, generator=g) * eps * 1.2).requires_grad_()
    s6.data[:, :7] = sdf.detach()[:7]                                  # a few exactly flat samples: zero normal (the clamps of safe_normalize / norm)
    dirs = torch.randn(M, 3, device=dev, generator=g) * 2.0
    ts = torch.rand(M, 2, device=dev, generator=g) * 0.01 + 0.002
    var = torch.tensor(0.35, device=dev, requires_grad=True)
    lam_eik, seed = 0.1, torch.tensor(64.0, device=dev)
    # torch statement
    normal = torch.stack([0.5 * (s6[0] - s6[1]) / eps, 0.5 * (s6[2] - s6[3]) / eps, 0.5 * (s6[4] - s6[5]) / eps], dim=-1)
    true_cos = (safe_normalize(dirs) * safe_normalize(normal)).sum(-1)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - car) + F.relu(-true_cos) * car)
    inv_s = torch.exp(var * 10.0).clip(1e-6, 1e6)
    prev, nxt = torch.sigmoid((sdf - iter_cos * ts[:, 1] * 0.5) * inv_s), torch.sigmoid((sdf + iter_cos * ts[:, 1] * 0.5) * inv_s)
    alpha = ((prev - nxt + 1e-5) / (prev + 1e-5)).view(-1).clip(0, 1)
    eik = ((torch.linalg.norm(normal, ord=2, dim=-1) - 1) ** 2).mean()
    w = torch.randn(M, device=dev, generator=g)
    loss = (alpha * w).sum() + lam_eik * eik * seed          # the executor's d_alpha already carries the seed; the eikonal term takes it from `seed`
    loss.backward()
    # kernels
    a2, n2 = torch.empty(M, device=dev), torch.empty(M, 3, device=dev)
    nb = (M + 255) // 256
    eik_part = torch.empty(nb, device=dev)
    s6k = s6.detach().t().contiguous()                                                       # [M, 6] as the kernels take it
    L.call("n2m_sdf_alpha_forward", p(sdf.detach()), p(s6k), p(dirs), p(ts), M, p(var.detach()), eps, car, p(a2), p(n2), p(eik_part), L.stream())
    assert float((a2 - alpha.detach()).abs().max()) <= 2e-6
    assert float((n2 - normal.detach()).abs().max()) <= 1e-6 * float(normal.detach().abs().max())
    assert abs(float(eik_part.double().sum() / M) - float(eik)) <= 1e-5 * float(eik)
    d_sdf, d_s6 = torch.empty(M, device=dev), torch.empty(M, 6, device=dev)
    var_part, d_var, finf = torch.empty(nb, device=dev), torch.empty(1, device=dev), torch.zeros((), device=dev)
    L.call("n2m_sdf_alpha_backward", p(w), p(sdf.detach()), p(s6k), p(dirs), p(ts), M, p(var.detach()), eps, car, p(seed),
           float(lam_eik * 2.0 / M), p(d_sdf), p(d_s6), p(var_part), p(d_var), p(finf), L.stream())
    rel = lambda a, b: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
    assert rel(d_sdf, sdf.grad) <= 2e-5, rel(d_sdf, sdf.grad)
    assert rel(d_s6.t(), s6.grad) <= 2e-5, rel(d_s6.t(), s6.grad)
    assert abs(float(d_var) - float(var.grad)) <= 2e-4 * abs(float(var.grad)), (float(d_var), float(var.grad))
    assert float(finf) == 0.0


@pytest.mark.parametrize("cascade,bound", [(1, 1.0), (3, 4.0)])
def test_occupancy_refresh_kernels_against_the_torch_statement(cascade, bound):
    """n2m_occupancy_points / n2m_occupancy_update against the reference's expressions (nerf/renderer.py:1096-1100, 1133-1140) written in
    torch: points and the updated grid bit for bit (same operations, same rounding points), mean to fp32 accuracy (the kernel sums
    per-workgroup partials in double, torch.mean in fp32), threshold = min(mean, density_thresh), packbits from the device threshold."""
    import torch
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd import raymarching
    p = L.ptr
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(cascade)
    H = 128
    n_cells = H ** 3
    coords = raymarching.morton3D_invert(torch.arange(n_cells, dtype=torch.int32, device=dev))
    cells = 2 * coords.float() / (H - 1) - 1
    grid = torch.rand(cascade, n_cells, device=dev, generator=g) * 20 - 2        # some cells < 0: -1 marks untrained cells
    grid[:, ::7] = -1.0
    tmp_all = torch.rand(cascade, n_cells, device=dev, generator=g) * 30
    tmp_all[:, ::11] = -0.5                                                      # a negative sample leaves its cell alone
    tmp_all[0, 5] = float("nan")
    for cas in range(cascade):
        b = min(2 ** cas, bound)
        hgs = b / H
        u = torch.rand(n_cells, 3, device=dev, generator=g)
        want = cells * (b - hgs) + (u * 2 - 1) * hgs
        got = torch.empty_like(cells)
        L.call("n2m_occupancy_points", p(cells), p(u), None, float(b - hgs), float(hgs), p(got), n_cells, L.stream())
        assert torch.equal(got, want), f"cascade {cas}"
        # only the cells whose grid value is >= 0: the listed cells get the points they would have got
        idx = torch.nonzero(grid[cas] >= 0).reshape(-1)
        sub = torch.full_like(cells, 123.0)
        L.call("n2m_occupancy_points", p(cells), p(u), p(idx.to(torch.int32)), float(b - hgs), float(hgs), p(sub), idx.numel(), L.stream())
        assert torch.equal(sub[:idx.numel()], want[idx]) and bool((sub[idx.numel():] == 123.0).all())
    decay, thresh0 = 0.95, 10.0
    valid = (grid >= 0) & (tmp_all >= 0)
    want_grid = torch.where(valid, torch.maximum(grid * decay, tmp_all), grid)
    want_mean = torch.mean(want_grid.clamp(min=0).double()).item()
    n_part = int(L.lib().n2m_occupancy_update_partials(grid.numel()))
    partials = torch.empty(n_part, device=dev)
    ticket = torch.zeros(1, dtype=torch.int32, device=dev)
    mean, thresh = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    for rep in range(2):                                                          # the ticket resets itself
        got_grid = grid.clone()
        L.call("n2m_occupancy_update", p(got_grid), p(tmp_all), decay, grid.numel(), thresh0, p(partials), p(ticket), p(mean), p(thresh), L.stream())
        torch.cuda.synchronize()
        assert torch.equal(got_grid, want_grid)
        assert abs(mean.item() - want_mean) <= 2e-7 * want_mean
        assert thresh.item() == min(mean.item(), thresh0) and int(ticket.item()) == 0
    L.call("n2m_occupancy_update", p(got_grid), p(tmp_all), decay, grid.numel(), 1e9, p(partials), p(ticket), p(mean), p(thresh), L.stream())
    assert thresh.item() == mean.item()
    bits_dev = raymarching.packbits(got_grid, thresh)
    bits_host = raymarching.packbits(got_grid, thresh.item())
    assert torch.equal(bits_dev, bits_host)


def test_density_query_from_the_packed_rows_equals_the_plain_table():
    """The occupancy refresh's density query reads the density column of the packed rows (n2m_grid_encode_forward_packed with outputs2 = NULL):
    same features, bit for bit, as n2m_grid_encode_forward on the fp32 table -- all 16 levels and with a level cap."""
    import torch
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    from nerf2mesh_amd import fused
    torch.manual_seed(3)
    dev = torch.device("cuda", 0)
    net = NeRFNetwork(make_options(O=True, bound=1, fused_mlp=True)).to(dev)
    with torch.no_grad():
        net.encoder.embeddings.uniform_(-1, 1)
        net.encoder_color.embeddings.uniform_(-1, 1)
    x = torch.rand(100_003, 3, device=dev) * 2.2 - 1.1                           # some points outside the cube
    pk = net.packed_tables()
    aff = fused._affine(1.0)
    for ml in (16, 5):
        h_packed = fused._encode_lm_packed(x, pk, net, ml, aff, density_only=True)[0]
        h_plain = fused._encode_lm((x + 1) / 2, net.encoder.embeddings.detach(), net.encoder, ml)
        assert torch.equal(h_packed.view(-1), h_plain.view(-1)), f"max_level {ml}"
    with torch.no_grad():
        s1 = net.density(x)["sigma"]
    assert torch.isfinite(s1).all()


def test_occupancy_refresh_skips_untrained_cells_without_changing_the_result():
    """Cells marked -1 are never updated (nerf/renderer.py:1131-1134); update_extra_state therefore queries the density of the OTHER cells
    only.  Against a torch statement of the reference's full update fed with the same random draws: same grid, same bit field."""
    import torch
    from nerf2mesh_amd import raymarching
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    net = NeRFNetwork(make_options(O=True, bound=4, fused_mlp=True)).to(dev)
    assert net.cascade == 3
    with torch.no_grad():
        net.encoder.embeddings.uniform_(-1, 1)
    g = torch.Generator(device=dev).manual_seed(1)
    net.density_grid.copy_(torch.rand(net.density_grid.shape, device=dev, generator=g) * 5)
    net.density_grid[0, ::3] = -1
    net.density_grid[1, 100_000:] = -1
    net.density_grid[2] = -1                                    # a cascade without a single valid cell
    before = net.density_grid.clone()
    state = torch.cuda.get_rng_state(dev)
    net.update_extra_state()
    got_grid, got_bits, got_mean = net.density_grid.clone(), net.density_bitfield.clone(), net.mean_density
    # the reference's full update with the same draws
    torch.cuda.set_rng_state(state, dev)
    cells = net._cells()
    tmp = torch.empty_like(before)
    with torch.no_grad():
        for cas in range(net.cascade):
            bound = min(2 ** cas, net.bound)
            hgs = bound / net.grid_size
            xyzs = cells * (bound - hgs) + (torch.rand_like(cells) * 2 - 1) * hgs
            with torch.autocast(device_type="cuda", dtype=torch.float16):
                tmp[cas] = net.density(xyzs)["sigma"].reshape(-1).float()
    valid = (before >= 0) & (tmp >= 0)
    want = torch.where(valid, torch.maximum(before * 0.95, tmp), before)
    assert torch.equal(got_grid, want)
    assert bool((got_grid[2] == -1).all()) and bool((got_grid[0, ::3] == -1).all())
    want_mean = float(want.clamp(min=0).double().mean())
    assert abs(got_mean - want_mean) <= 2e-7 * want_mean
    assert torch.equal(got_bits, raymarching.packbits(want, min(got_mean, net.density_thresh)))
    # a second refresh reuses the cached lists (the grid was only touched through the kernels) and still agrees
    net.update_extra_state()
    assert bool((net.density_grid[2] == -1).all())
    # marking through torch invalidates them
    net.density_grid[0, 1::3] = -1
    net.update_extra_state()
    assert bool((net.density_grid[0, 1::3] == -1).all())
