"""GPU parity: libn2m_hip.so (through the C ABI) against the CPU oracle on identical seeded inputs.

Bars (north_star): bit-exact for ray/sample indexing, occupancy counts and every integer/byte output;
fp32 tolerance stated per test for floating-point outputs.  Many fp32 outputs are in fact bit-identical to the
oracle because the HIP sources are compiled with -ffp-contract=off and follow the oracle's operation order;
where the kernel re-associates (wave-parallel compositing, atomics) the tolerance is written down.
"""
import numpy as np
import pytest

from conftest import lego_offsets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    import torch
    assert torch.cuda.is_available()
    from nerf2mesh_amd import backends, _lib
    _lib.lib()                       # raises if the HIP library is missing: no silent fallback
    backends.install()
    import _raymarching_mob, _gridencoder, _shencoder
    return {"rm": _raymarching_mob, "ge": _gridencoder, "sh": _shencoder, "torch": torch, "L": _lib}


def dev(be, a):
    return be["torch"].from_numpy(np.ascontiguousarray(a)).cuda()


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    v = {2: np.uint16, 4: np.uint32, 1: np.uint8}[a.dtype.itemsize]
    return np.array_equal(a.view(v), b.view(v))


def make_rays(scene, n, seed=0):
    torch, S = scene["torch"], scene["S"]
    g = torch.Generator().manual_seed(seed)
    o, d = S.random_rays(scene["poses"], n, g)
    return o.numpy(), d.numpy()


def test_morton_packbits_flatten(be, oracle):
    torch, rm = be["torch"], be["rm"]
    rng = np.random.default_rng(0)
    c = rng.integers(0, 1024, (100003, 3)).astype(np.int32)
    out = torch.empty(c.shape[0], dtype=torch.int32, device="cuda")
    rm.morton3D(dev(be, c), c.shape[0], out)
    assert np.array_equal(out.cpu().numpy(), oracle.morton3D(c))
    back = torch.empty(c.shape[0], 3, dtype=torch.int32, device="cuda")
    rm.morton3D_invert(out, c.shape[0], back)
    assert np.array_equal(back.cpu().numpy(), c)
    # full-size occupancy grid (C=1, H=128) and a 5-cascade one with an unaligned tail
    for n_floats in (128 ** 3, 5 * 128 ** 3, 8 * 1237):
        grid = rng.normal(size=n_floats).astype(np.float32)
        grid[:8] = [0.5, np.nextafter(np.float32(0.5), np.float32(1)), 0.4999999, -1, np.nan, np.inf, 0.5, 1]
        bf = torch.zeros(n_floats // 8, dtype=torch.uint8, device="cuda")
        rm.packbits(dev(be, grid), n_floats // 8, 0.5, bf)
        assert np.array_equal(bf.cpu().numpy(), oracle.packbits(grid, 0.5))
        from nerf2mesh_amd import raymarching as R                      # threshold from device memory: same bits
        bf2 = R.packbits(dev(be, grid), torch.tensor(0.5, device="cuda"))
        assert torch.equal(bf2, bf)
    rays = np.array([[0, 3], [3, 0], [3, 70], [73, 1]], np.int32)
    res = torch.zeros(74, dtype=torch.int32, device="cuda")
    rm.flatten_rays(dev(be, rays), 4, 74, res)
    assert np.array_equal(res.cpu().numpy(), oracle.flatten_rays(rays, 74))


def test_near_far(be, oracle, scene):
    torch, rm = be["torch"], be["rm"]
    N = 50000
    o, d = make_rays(scene, N)
    d[:50, 0] = 0.0
    d[50:60] = 0.0
    o[60:80] *= 0.1
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears = torch.empty(N, device="cuda")
    fars = torch.empty(N, device="cuda")
    rm.near_far_from_aabb(dev(be, o), dev(be, d), dev(be, aabb), N, 0.05, nears, fars)
    on, of = oracle.near_far_from_aabb(o, d, aabb, 0.05)
    assert bits_equal(nears.cpu().numpy(), on) and bits_equal(fars.cpu().numpy(), of)


def test_sph_from_ray(be, oracle, scene):
    torch, rm = be["torch"], be["rm"]
    o, d = make_rays(scene, 1000)
    c = torch.empty(1000, 2, device="cuda")
    rm.sph_from_ray(dev(be, o), dev(be, d), 5.0, 1000, c)
    np.testing.assert_allclose(c.cpu().numpy(), oracle.sph_from_ray(o, d, 5.0), rtol=0, atol=5e-6)   # atan2/sqrt libm vs ocml


def _hip_march_train(be, o, d, bits, bound, contract, dt_gamma, max_steps, C, H, nears, fars, noises):
    torch, rm = be["torch"], be["rm"]
    N = o.shape[0]
    rays = torch.empty(N, 2, dtype=torch.int32, device="cuda")
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")
    args = (dev(be, o), dev(be, d), dev(be, bits), bound, contract, dt_gamma, max_steps, N, C, H, dev(be, nears), dev(be, fars))
    nz = dev(be, noises)
    rm.march_rays_train(*args, None, None, None, rays, counter, nz)
    M = int(counter.item())
    xyzs = torch.zeros(M, 3, device="cuda"); dirs = torch.zeros(M, 3, device="cuda"); ts = torch.zeros(M, 2, device="cuda")
    rm.march_rays_train(*args, xyzs, dirs, ts, rays, counter, nz)
    return xyzs.cpu().numpy(), dirs.cpu().numpy(), ts.cpu().numpy(), rays.cpu().numpy()


@pytest.mark.parametrize("cfg", [
    dict(bound=1.0, contract=False, dt_gamma=0.0, C=1, N=4096),
    dict(bound=1.0, contract=False, dt_gamma=1 / 256, C=1, N=4096),
    dict(bound=4.0, contract=False, dt_gamma=1 / 256, C=3, N=3000),
    dict(bound=4.0, contract=True, dt_gamma=0.0, C=2, N=3000),
    dict(bound=1.0, contract=False, dt_gamma=0.0, C=1, N=2000, max_steps=64, dense=True),
    dict(bound=1.0, contract=False, dt_gamma=0.0, C=1, N=200000),          # > 131072 rays: 3-phase scan path
    dict(bound=1.0, contract=False, dt_gamma=0.0, C=1, N=1000, empty=True),  # all-empty grid: M = 0
])
def test_march_rays_train_bit_exact(be, oracle, scene, cfg):
    S = scene["S"]
    C, H = cfg["C"], 128 if cfg["C"] == 1 else 64
    rng = np.random.default_rng(5)
    if cfg.get("dense"):
        bits = np.full(C * H ** 3 // 8, 255, np.uint8)
    elif cfg.get("empty"):
        bits = np.zeros(C * H ** 3 // 8, np.uint8)
    elif C == 1:
        bits = scene["bits"]
    else:
        grid = S.scene_density_grid(H=H, cascade=C, bound=cfg["bound"] if not cfg["contract"] else 2.0).numpy()
        grid += (rng.random(grid.shape) < 0.02).astype(np.float32) * 50
        bits = oracle.packbits(grid, 10.0)
    N = cfg["N"]
    o, d = make_rays(scene, N, seed=9)
    if cfg["bound"] > 1:
        o = o * 1.1
    b = cfg["bound"]
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-b, -b, -b, b, b, b], np.float32), 0.05)
    noises = rng.random(N).astype(np.float32)
    ms = cfg.get("max_steps", 1024)
    hx, hd, ht, hr = _hip_march_train(be, o, d, bits, b, cfg["contract"], cfg["dt_gamma"], ms, C, H, nears, fars, noises)
    ox, od, ot, orr = oracle.march_rays_train(o, d, b, cfg["contract"], bits, C, H, nears, fars, noises, cfg["dt_gamma"], ms)
    assert np.array_equal(hr, orr), "per-ray (offset, count) must match the oracle exactly"
    if cfg.get("empty"):
        assert hx.shape[0] == 0
        return
    assert orr[:, 1].sum() > 1000
    assert bits_equal(hx, ox) and bits_equal(hd, od) and bits_equal(ht, ot)


@pytest.mark.parametrize("cfg", [
    dict(bound=1.0, contract=False, dt_gamma=0.0, C=1, N=4099),
    dict(bound=1.0, contract=False, dt_gamma=1 / 256, C=1, N=4096),
    dict(bound=4.0, contract=False, dt_gamma=1 / 256, C=3, N=3001),
    dict(bound=4.0, contract=True, dt_gamma=0.0, C=2, N=3000),
    dict(bound=1.0, contract=False, dt_gamma=0.0, C=1, N=2000, max_steps=64, dense=True),
    dict(bound=1.0, contract=False, dt_gamma=0.0, C=1, N=70001),           # 274 groups of 256 rays
    dict(bound=1.0, contract=False, dt_gamma=0.0, C=1, N=1000, empty=True),
    dict(bound=4.0, contract=False, dt_gamma=0.0, C=3, N=3000, random=0.3),   # > 24 sample-bearing chunks per ray: re-march fallback
    dict(bound=1.0, contract=False, dt_gamma=0.0, C=1, N=3, ),
])
def test_march_single_pass_equals_two_pass_protocol(be, oracle, scene, cfg):
    """n2m_march_rays_train_fused (one march: count + recorded chunks, then replay) == the oracle's two-pass march: (offset, count) per
    ray, the sample count and every sample bit for bit; with fewer rows than samples the rays that do not fit are left unwritten."""
    torch, S = be["torch"], scene["S"]
    from nerf2mesh_amd import raymarching as R
    C, H = cfg["C"], 128 if cfg["C"] == 1 else 64
    rng = np.random.default_rng(5)
    if cfg.get("dense"):
        bits = np.full(C * H ** 3 // 8, 255, np.uint8)
    elif cfg.get("empty"):
        bits = np.zeros(C * H ** 3 // 8, np.uint8)
    elif cfg.get("random"):
        bits = oracle.packbits((rng.random((C, H ** 3)) < cfg["random"]).astype(np.float32) * 50, 10.0)
    elif C == 1:
        bits = scene["bits"]
    else:
        grid = S.scene_density_grid(H=H, cascade=C, bound=cfg["bound"] if not cfg["contract"] else 2.0).numpy()
        grid += (rng.random(grid.shape) < 0.02).astype(np.float32) * 50
        bits = oracle.packbits(grid, 10.0)
    N, b = cfg["N"], cfg["bound"]
    o, d = make_rays(scene, N, seed=9)
    if b > 1:
        o = o * 1.1
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-b, -b, -b, b, b, b], np.float32), 0.05)
    noises = rng.random(N).astype(np.float32)
    ms = cfg.get("max_steps", 1024)
    ox, od, ot, orr = oracle.march_rays_train(o, d, b, cfg["contract"], bits, C, H, nears, fars, noises, cfg["dt_gamma"], ms)
    M = int(orr[:, 1].sum())
    a = (dev(be, o), dev(be, d), b, cfg["contract"], dev(be, bits), C, H, dev(be, nears), dev(be, fars), dev(be, noises), cfg["dt_gamma"], ms)
    for cap in (M + 37, M, max(M // 2 + 5, 0)):
        buf = torch.full((max(cap, 1) * 8,), -7.0, device="cuda")
        out = (buf[:3 * cap].view(-1, 3), buf[3 * cap:6 * cap].view(-1, 3), buf[6 * cap:8 * cap].view(-1, 2))
        x, dd, t, rays, counter = R.march_rays_train_fused(*a, max_points=cap, out=out)
        assert int(counter.item()) == M
        assert np.array_equal(rays.cpu().numpy(), orr), "per-ray (offset, count) must match the oracle exactly"
        if M == 0:
            continue
        hx, ht, hd = x.cpu().numpy(), t.cpu().numpy(), dd.cpu().numpy()
        fits = (orr[:, 0].astype(np.int64) + orr[:, 1]) <= cap
        if cap >= M:
            assert fits.all() and bits_equal(hx[:M], ox) and bits_equal(hd[:M], od) and bits_equal(ht[:M], ot)
        else:
            keep = np.zeros(cap, bool)
            for off, cnt in orr[fits & (orr[:, 1] > 0)]:
                keep[off:off + cnt] = True
            assert keep.any()
            assert bits_equal(hx[keep], ox[:cap][keep]) and bits_equal(ht[keep], ot[:cap][keep])
            assert (hx[~keep] == -7.0).all() and (ht[~keep] == -7.0).all()       # rays that do not fit are not written


def test_march_parallel_resolution_falls_back_exactly(be, oracle, scene):
    """Random 30 % occupancy = hundreds of empty->occupied crossings per ray: some exit times round past a candidate of the next,
    occupied voxel, the prefix-maximum check fails and the ray is re-marched serially.  Results stay bit-exact and the fallback is taken."""
    import ctypes
    rng = np.random.default_rng(11)
    H = 128
    bits = oracle.packbits((rng.random((1, H ** 3)) < 0.3).astype(np.float32) * 50, 10.0)
    N = 60000
    o, d = make_rays(scene, N, seed=21)
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.05)
    noises = rng.random(N).astype(np.float32)
    cnt = ctypes.c_uint32(0)
    be["L"].call("n2m_march_fallback_count", ctypes.addressof(cnt))       # reset
    for dt_gamma in (0.0, 1 / 256):
        hx, hd, ht, hr = _hip_march_train(be, o, d, bits, 1.0, False, dt_gamma, 1024, 1, H, nears, fars, noises)
        ox, od, ot, orr = oracle.march_rays_train(o, d, 1.0, False, bits, 1, H, nears, fars, noises, dt_gamma, 1024)
        assert np.array_equal(hr, orr)
        assert bits_equal(hx, ox) and bits_equal(ht, ot)
        be["L"].call("n2m_march_fallback_count", ctypes.addressof(cnt))
        print(f"dt_gamma {dt_gamma}: {cnt.value} of {N} rays took the serial fallback, {int(orr[:, 1].sum())} samples")
        assert 0 < cnt.value < N // 4


def test_march_split_form_equals_single_call(be, oracle, scene):
    """march_rays_train_begin/finish (pass 1 issued ahead, count read through a pinned copy + event) == march_rays_train."""
    torch = be["torch"]
    from nerf2mesh_amd import raymarching as R
    N = 5000
    o, d = make_rays(scene, N, seed=33)
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.05)
    noises = np.random.default_rng(2).random(N).astype(np.float32)
    a = (dev(be, o), dev(be, d), 1.0, False, dev(be, scene["bits"]), 1, 128, dev(be, nears), dev(be, fars))
    ref = R.march_rays_train(*a, True, 1 / 256, 1024, dev(be, noises))
    ticket = R.march_rays_train_begin(*a, True, 1 / 256, 1024, dev(be, noises))
    junk = torch.randn(1 << 22, device="cuda").sin_().sum()      # unrelated work queued between the two halves
    got = R.march_rays_train_finish(ticket)
    assert junk.isfinite()
    assert ref[0].shape[0] > 1000
    for x, y in zip(ref, got):
        assert torch.equal(x, y)
    # speculative write pass (n2m_march_rays_train_write into buffers of expect_points rows, queued before the count is known):
    # large enough -> finish() slices them; too small -> rays that do not fit are skipped by the kernel and finish() re-marches
    M = ref[0].shape[0]
    for cap, spec_used in ((M + 100, True), (M, True), (M - 1, False), (64, False)):
        ticket = R.march_rays_train_begin(*a, True, 1 / 256, 1024, dev(be, noises), expect_points=cap)
        canary = ticket.spec[0].new_full((1,), 0)            # keeps the buffers alive; over-capacity rays must not be written
        got = R.march_rays_train_finish(ticket)
        assert (got[0].data_ptr() == ticket.spec[0].data_ptr()) == spec_used
        for x, y in zip(ref, got):
            assert torch.equal(x, y)
        del canary


def test_march_counter_base_and_repeatability(be, oracle, scene):
    """offsets start at the counter's entry value; two runs give identical packing (deterministic scan)."""
    torch, rm = be["torch"], be["rm"]
    N = 3000
    o, d = make_rays(scene, N, seed=21)
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.05)
    noises = np.zeros(N, np.float32)
    args = (dev(be, o), dev(be, d), dev(be, scene["bits"]), 1.0, False, 0.0, 1024, N, 1, 128, dev(be, nears), dev(be, fars))
    outs = []
    for base in (0, 0, 777):
        rays = torch.empty(N, 2, dtype=torch.int32, device="cuda")
        counter = torch.full((1,), base, dtype=torch.int32, device="cuda")
        rm.march_rays_train(*args, None, None, None, rays, counter, dev(be, noises))
        outs.append((rays.cpu().numpy(), int(counter.item())))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[2][0][:, 0], outs[0][0][:, 0] + 777) and outs[2][1] == outs[0][1] + 777
    _, _, _, orr = oracle.march_rays_train(o, d, 1.0, False, scene["bits"], 1, 128, nears, fars, noises, 0.0, 1024, counter0=777)
    assert np.array_equal(outs[2][0], orr)


@pytest.mark.parametrize("alpha_mode", [False, True])
def test_composite_train(be, oracle, alpha_mode):
    torch, rm = be["torch"], be["rm"]
    rng = np.random.default_rng(3)
    N = 5000
    counts = rng.integers(0, 200, N).astype(np.int32)
    counts[:5] = 0
    counts[10] = 1024
    counts[-1] = 37
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    rays = np.stack([offs, counts], 1).astype(np.int32)
    M = int(counts.sum()) - 3                  # the last ray overflows the sample buffer -> zeros (guard raymarching.cu:521)
    sig = (rng.random(M) * (1.0 if alpha_mode else 60.0)).astype(np.float32)
    sig[offs[10]:offs[10] + 1024] *= 0.002      # a long, thin ray: many chunks before the early stop
    if alpha_mode:
        sig = np.clip(sig, 0, 0.98)
    rgb = rng.random((M, 3)).astype(np.float32)
    ts = np.stack([np.cumsum(rng.random(M)).astype(np.float32) * 0.01 + 2, np.full(M, 0.0034, np.float32)], 1).astype(np.float32)
    w = torch.zeros(M, device="cuda"); ws = torch.empty(N, device="cuda"); dp = torch.empty(N, device="cuda"); im = torch.empty(N, 3, device="cuda")
    rm.composite_rays_train_forward(dev(be, sig), dev(be, rgb), dev(be, ts), dev(be, rays), M, N, 1e-4, alpha_mode, w, ws, dp, im)
    ow, ows, odp, oim = oracle.composite_rays_train_forward(sig, rgb, ts, rays, 1e-4, alpha_mode)
    # wave-parallel prefix product / tree sums vs the oracle's serial recurrence: fp32 re-association only.
    tol = dict(rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(w.cpu().numpy(), ow, **tol)
    np.testing.assert_allclose(ws.cpu().numpy(), ows, **tol)
    # depth sums w*t with t ~ 2..12 over up to 1024 samples: the oracle's SERIAL fp32 sum carries ~n*eps/2 = 6e-5
    # relative error itself, the wave tree sum less; 2e-4 covers the worst (1024-sample) ray
    np.testing.assert_allclose(dp.cpu().numpy(), odp, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(im.cpu().numpy(), oim, **tol)
    # samples after the early stop are exact zeros in both, up to threshold ties
    assert np.mean((w.cpu().numpy() == 0) != (ow == 0)) < 1e-4
    # the kernel writes every sample of every ray range (these rays tile [0, M)): no zero-fill needed beforehand
    w2 = torch.full((M,), 7.0, device="cuda")
    rm.composite_rays_train_forward(dev(be, sig), dev(be, rgb), dev(be, ts), dev(be, rays), M, N, 1e-4, alpha_mode, w2, ws, dp, im)
    assert torch.equal(w, w2)
    gw, gws, gd, gi = (rng.normal(size=M).astype(np.float32), rng.normal(size=N).astype(np.float32),
                       rng.normal(size=N).astype(np.float32), rng.normal(size=(N, 3)).astype(np.float32))
    gs = torch.zeros(M, device="cuda"); gr = torch.zeros(M, 3, device="cuda")
    rm.composite_rays_train_backward(dev(be, gw), dev(be, gws), dev(be, gd), dev(be, gi), dev(be, sig), dev(be, rgb), dev(be, ts),
                                     dev(be, rays), dev(be, ows), dev(be, odp), dev(be, oim), M, N, 1e-4, alpha_mode, gs, gr)
    ogs, ogr = oracle.composite_rays_train_backward(gw, gws, gd, gi, sig, rgb, ts, rays, ows, odp, oim, 1e-4, alpha_mode)
    gs2 = torch.full((M,), -3.0, device="cuda"); gr2 = torch.full((M, 3), 5.0, device="cuda")
    rm.composite_rays_train_backward(dev(be, gw), dev(be, gws), dev(be, gd), dev(be, gi), dev(be, sig), dev(be, rgb), dev(be, ts),
                                     dev(be, rays), dev(be, ows), dev(be, odp), dev(be, oim), M, N, 1e-4, alpha_mode, gs2, gr2)
    assert torch.equal(gs, gs2) and torch.equal(gr, gr2)
    np.testing.assert_allclose(gr.cpu().numpy(), ogr, rtol=2e-5, atol=2e-6)
    # grad_sigma subtracts nearly equal suffix sums: absolute tolerance scaled to the magnitudes involved
    scale = np.abs(ogs).max()
    np.testing.assert_allclose(gs.cpu().numpy(), ogs, rtol=1e-3, atol=2e-5 * max(scale, 1.0))


def test_composite_train_matches_autograd(be):
    """Backward kernel == autograd of the same recurrence in float64, for the image, alpha and
    depth outputs.  (The `weights` output is left out on purpose: the reference multiplies the whole suffix by
    grad_weights[i] instead of the per-sample grad_weights[j], raymarching.cu:676 -- exact only for a constant
    grad_weights; that behaviour is reproduced and pinned by the oracle comparison above, not by autograd.)"""
    torch = be["torch"]
    from nerf2mesh_amd import raymarching
    rng = np.random.default_rng(12)
    N = 64
    counts = rng.integers(1, 90, N)
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
    rays = torch.tensor(np.stack([offs, counts], 1), dtype=torch.int32, device="cuda")
    M = int(counts.sum())
    sig = torch.tensor(rng.random(M) * 20, dtype=torch.float32, device="cuda", requires_grad=True)
    rgb = torch.tensor(rng.random((M, 3)), dtype=torch.float32, device="cuda", requires_grad=True)
    ts = torch.tensor(np.stack([np.cumsum(rng.random(M)) * 0.01 + 2, np.full(M, 0.01)], 1), dtype=torch.float32, device="cuda")
    w, ws, dp, im = raymarching.composite_rays_train(sig, rgb, ts, rays, 0.0, False)     # T_thresh 0: no early stop
    (im.sum() * 0.7 + (ws * 1.3).sum() + (dp * 0.2).sum()).backward()
    s64 = sig.detach().double().cpu().requires_grad_(True)
    r64 = rgb.detach().double().cpu().requires_grad_(True)
    t64 = ts.double().cpu()
    tot = 0
    for n in range(N):
        sl = slice(int(offs[n]), int(offs[n] + counts[n]))
        a = 1 - torch.exp(-s64[sl] * t64[sl, 1])
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.float64), 1 - a[:-1]]), 0)
        ww = a * T
        tot = tot + (ww[:, None] * r64[sl]).sum() * 0.7 + ww.sum() * 1.3 + (ww * t64[sl, 0]).sum() * 0.2
    tot.backward()
    np.testing.assert_allclose(sig.grad.cpu().numpy(), s64.grad.numpy(), rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(rgb.grad.cpu().numpy(), r64.grad.numpy(), rtol=1e-4, atol=1e-6)


def test_inference_loop(be, oracle, scene):
    """march_rays / composite_rays / compact_alive over several rounds, state carried on both sides."""
    torch, rm = be["torch"], be["rm"]
    from nerf2mesh_amd import raymarching
    N = 20000
    o, d = make_rays(scene, N, seed=11)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.05)
    rng = np.random.default_rng(2)
    alive_o = np.arange(N, dtype=np.int32)
    rays_t_o = nears.copy()
    ws_o, dp_o, im_o = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    alive_h = dev(be, alive_o); rays_t_h = dev(be, rays_t_o)
    ws_h, dp_h, im_h = dev(be, ws_o), dev(be, dp_o), dev(be, im_o)
    O, D, B = dev(be, o), dev(be, d), dev(be, scene["bits"])
    NE, FA = dev(be, nears), dev(be, fars)
    total = 0
    for it in range(8):
        n_alive = alive_o.shape[0]
        if n_alive == 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        M = n_alive * n_step
        noises = np.zeros(n_alive, np.float32)
        xyzs = torch.zeros(M, 3, device="cuda"); dirs = torch.zeros(M, 3, device="cuda"); ts = torch.zeros(M, 2, device="cuda")
        rm.march_rays(n_alive, n_step, alive_h, rays_t_h, O, D, 1.0, False, 0.0, 1024, 1, 128, B, NE, FA, xyzs, dirs, ts, dev(be, noises))
        ox, od, ot = oracle.march_rays(n_alive, n_step, alive_o, rays_t_o, o, d, 1.0, False, scene["bits"], 1, 128, nears, fars, noises)
        assert bits_equal(xyzs.cpu().numpy(), ox) and bits_equal(dirs.cpu().numpy(), od) and bits_equal(ts.cpu().numpy(), ot)
        total += int((ot[:, 0] > 0).sum())
        sig = (rng.random(M) * 40).astype(np.float32)
        rgb = rng.random((M, 3)).astype(np.float32)
        rm.composite_rays(n_alive, n_step, 1e-2, False, alive_h, rays_t_h, dev(be, sig), dev(be, rgb), ts, ws_h, dp_h, im_h)
        oracle.composite_rays(n_alive, n_step, alive_o, rays_t_o, sig, rgb, ot, ws_o, dp_o, im_o, 1e-2, False)
        assert np.array_equal(alive_h.cpu().numpy(), alive_o)             # same rays end, exactly
        assert bits_equal(rays_t_h.cpu().numpy(), rays_t_o)
        # same serial arithmetic per ray on both sides (no re-association): bit-identical unless expf differs
        np.testing.assert_allclose(ws_h.cpu().numpy(), ws_o, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(im_h.cpu().numpy(), im_o, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(dp_h.cpu().numpy(), dp_o, rtol=1e-6, atol=1e-6)
        alive_h = raymarching.compact_alive(alive_h)
        alive_o = oracle.compact_alive(alive_o).copy()
        assert np.array_equal(alive_h.cpu().numpy(), alive_o)
        # hand the oracle's accumulators to the device so rounding differences cannot compound across rounds
        ws_h, dp_h, im_h = dev(be, ws_o), dev(be, dp_o), dev(be, im_o)
    assert total > 5000


def test_inference_march_lane_and_wave_forms(be, oracle, scene):
    """n2m_march_rays picks its kernel by the number of rays alive (raymarching.hip: one lane per ray for a whole frame, one wave per ray at
    or below 131 072): both against the oracle, bit for bit -- a first round of 140 000 rays (lane form), then the survivors' second round
    with n_step = 2 (wave form, starting mid-ray from the t the first round's samples ended at)."""
    torch, rm = be["torch"], be["rm"]
    N = 140000
    o, d = make_rays(scene, N, seed=21)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.05)
    O, D, B, NE, FA = dev(be, o), dev(be, d), dev(be, scene["bits"]), dev(be, nears), dev(be, fars)
    alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    for n_step in (1, 2):
        n_alive = alive.shape[0]
        M = n_alive * n_step
        noises = np.zeros(n_alive, np.float32)
        xyzs = torch.zeros(M, 3, device="cuda"); dirs = torch.zeros(M, 3, device="cuda"); ts = torch.zeros(M, 2, device="cuda")
        rm.march_rays(n_alive, n_step, dev(be, alive), dev(be, rays_t), O, D, 1.0, False, 0.0, 1024, 1, 128, B, NE, FA, xyzs, dirs, ts, dev(be, noises))
        ox, od, ot = oracle.march_rays(n_alive, n_step, alive, rays_t, o, d, 1.0, False, scene["bits"], 1, 128, nears, fars, noises)
        assert bits_equal(xyzs.cpu().numpy(), ox) and bits_equal(dirs.cpu().numpy(), od) and bits_equal(ts.cpu().numpy(), ot)
        # next round: the rays that found a sample go on from where it ended (what composite_rays would leave in rays_t for a transparent sample)
        hit = ot.reshape(n_alive, n_step, 2)[:, -1, 0] > 0
        rays_t = rays_t.copy()
        rays_t[alive[hit]] = ot.reshape(n_alive, n_step, 2)[hit, -1, 0]
        alive = alive[hit]
        assert 1000 < alive.shape[0] <= 131072 or n_step == 2


@pytest.mark.parametrize("cfg", [
    dict(bound=1.0, contract=False, dt_gamma=1 / 256, C=1),               # the serial per-lane time chain instead of the closed form
    dict(bound=4.0, contract=False, dt_gamma=1 / 256, C=3),               # cascades: mip from position and from dt
    dict(bound=4.0, contract=True, dt_gamma=0.0, C=2),                    # L-inf contraction, samples outside the unit box always kept
    dict(bound=16.0, contract=False, dt_gamma=1 / 256, C=5, noise=True),  # BASELINE config 4's shape, first round jittered
])
def test_inference_march_forms_on_every_march_config(be, oracle, scene, cfg):
    """The lane form (> 131 072 rays alive) and the wave form (raymarching.hip march_infer_wave_*) of n2m_march_rays on the configurations
    test_march_rays_train_bit_exact holds the training marcher to (raymarching.cu:712-838: dt_gamma > 0, cascades, contraction): xyzs, dirs,
    ts bit for bit against the oracle, three rounds (n_step 1 lane form, then n_step 2 and 4 wave form, each starting mid-ray)."""
    torch, rm, S = be["torch"], be["rm"], scene["S"]
    C, b = cfg["C"], cfg["bound"]
    H = 128 if C == 1 else 64
    rng = np.random.default_rng(17)
    if C == 1:
        bits = scene["bits"]
    else:
        grid = S.scene_density_grid(H=H, cascade=C, bound=b if not cfg["contract"] else 2.0).numpy()
        grid += (rng.random(grid.shape) < 0.02).astype(np.float32) * 50
        bits = oracle.packbits(grid, 10.0)
    N = 140000
    o, d = make_rays(scene, N, seed=23)
    if b > 1:
        o = o * 1.1
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-b, -b, -b, b, b, b], np.float32), 0.05)
    O, D, B, NE, FA = dev(be, o), dev(be, d), dev(be, bits), dev(be, nears), dev(be, fars)
    alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    found = 0
    for rnd, n_step in enumerate((1, 2, 4)):
        n_alive = alive.shape[0]
        M = n_alive * n_step
        noises = rng.random(n_alive).astype(np.float32) if (cfg.get("noise") and rnd == 0) else np.zeros(n_alive, np.float32)
        xyzs = torch.zeros(M, 3, device="cuda"); dirs = torch.zeros(M, 3, device="cuda"); ts = torch.zeros(M, 2, device="cuda")
        rm.march_rays(n_alive, n_step, dev(be, alive), dev(be, rays_t), O, D, b, cfg["contract"], cfg["dt_gamma"], 1024, C, H, B, NE, FA,
                      xyzs, dirs, ts, dev(be, noises))
        ox, od, ot = oracle.march_rays(n_alive, n_step, alive, rays_t, o, d, b, cfg["contract"], bits, C, H, nears, fars, noises,
                                       cfg["dt_gamma"], 1024)
        assert bits_equal(ts.cpu().numpy(), ot), f"round {rnd}: ts"
        assert bits_equal(xyzs.cpu().numpy(), ox) and bits_equal(dirs.cpu().numpy(), od), f"round {rnd}: positions"
        found += int((ot[:, 0] > 0).sum())
        last = ot.reshape(n_alive, n_step, 2)[:, -1, 0]
        hit = last > 0
        rays_t = rays_t.copy()
        rays_t[alive[hit]] = last[hit]
        alive = alive[hit]
        if rnd == 0:
            alive = alive[:131072]                         # the survivors' rounds take the wave form
        assert alive.shape[0] > 1000, "the configuration must keep rays alive for the wave-form rounds"
    assert found > 50000


def test_compact_alive_large(be, oracle):
    torch = be["torch"]
    from nerf2mesh_amd import raymarching
    rng = np.random.default_rng(4)
    for n in (0, 1, 63, 1024, 131072, 131073, 640000):
        a = rng.integers(-1, 5, n).astype(np.int32)
        a[a >= 0] = np.arange((a >= 0).sum(), dtype=np.int32)
        got = raymarching.compact_alive(dev(be, a)).cpu().numpy() if n else np.zeros(0, np.int32)
        assert np.array_equal(got, oracle.compact_alive(a))


@pytest.mark.parametrize("stride", [1, 4])
def test_select_positive_is_torch_nonzero(be, stride):
    """n2m_select_positive (the covered pixels of a stage-1 frame, nerf/renderer.py:862-863): the indices of torch.nonzero(v > 0), in its order,
    and their number -- single-block and three-phase scan paths, strided column, NaN / -0 / denormal / inf values, nothing and everything selected."""
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    rng = np.random.default_rng(11)
    for n in (0, 1, 63, 1025, 131072, 131073, 2560000):
        buf = np.zeros((max(n, 1), stride), np.float32)
        v = rng.standard_normal(n).astype(np.float32)
        v[rng.random(n) < 0.6] = 0.0
        if n > 8:
            v[:8] = [np.nan, -0.0, 1e-42, np.inf, -np.inf, 0.0, 1.0, -1.0]
        for fill in ("mixed", "none", "all"):
            w = {"mixed": v, "none": -np.abs(v), "all": np.abs(v) + 1.0}[fill]
            buf[:n, stride - 1] = w
            t = dev(be, buf)
            out = torch.full((max(n, 1),), -7, dtype=torch.int64, device="cuda")
            k = torch.full((1,), -1, dtype=torch.int32, device="cuda")
            L.call("n2m_select_positive", t.data_ptr() + 4 * (stride - 1), n, stride, L.ptr(out), L.ptr(k), L.stream())
            torch.cuda.synchronize()
            want = torch.nonzero(t[:n, stride - 1] > 0, as_tuple=False).squeeze(1)
            assert int(k) == want.numel(), (n, fill)
            assert torch.equal(out[:want.numel()], want), (n, fill)
            assert bool((out[want.numel():] == -7).all()), "nothing is written behind the selection"


GRID_CASES = [
    (3, 1, False, 0, False, 0),
    (3, 2, True, 0, False, 0),
    (3, 2, False, 0, False, 0),
    (3, 4, True, 1, False, 1),
    (2, 8, False, 0, True, 0),
    (3, 8, True, 0, False, 1),
    (4, 2, False, 0, False, 0),
    (5, 1, False, 0, False, 0),
    (2, 4, False, 1, True, 1),
]


@pytest.mark.parametrize("D,C,half,gridtype,align,interp", GRID_CASES)
def test_grid_encode_forward_exact(be, oracle, D, C, half, gridtype, align, interp):
    """Forward (+dy_dx), both layouts: BIT-identical to the oracle in fp32 and fp16."""
    torch, ge = be["torch"], be["ge"]
    from nerf2mesh_amd import _lib as L
    rng = np.random.default_rng(4)
    Lv, H = (16, 16) if D == 3 else (8, 8)
    pls = 1.3819129 if D == 3 else 1.5
    offs = oracle.level_offsets(D, Lv, pls, H, 19 if D == 3 else 14, align)
    S = float(np.log2(pls))
    emb = (rng.random((int(offs[-1]), C), dtype=np.float32) * 2 - 1) * (0.5 if half else 1e-1)
    B = 20011
    x = rng.random((B, D), dtype=np.float32)
    x[0] = 0.0; x[1] = 1.0; x[2, 0] = -0.001; x[3, 1] = 1.001
    embt = emb.astype(np.float16) if half else emb
    tdt = torch.float16 if half else torch.float32
    X, E, O = dev(be, x), dev(be, embt), dev(be, offs)
    for max_level in (Lv, 5):
        out = torch.zeros(Lv, B, C, dtype=tdt, device="cuda")
        dy = torch.zeros(B, Lv * D * C, dtype=tdt, device="cuda")
        ge.grid_encode_forward(X, E, O, out, B, D, C, Lv, max_level, S, H, dy, gridtype, align, interp)
        oo, ody = oracle.grid_encode_forward(x, embt, offs, S, H, max_level, True, gridtype, align, interp)
        assert bits_equal(out.cpu().numpy(), oo)
        assert bits_equal(dy.cpu().numpy(), ody)
        bm = torch.full((B, Lv * C), 7.0, dtype=tdt, device="cuda")
        L.call("n2m_grid_encode_forward_bm", X.data_ptr(), E.data_ptr(), O.data_ptr(), bm.data_ptr(), B, D, C, Lv, max_level, S, H,
               gridtype, int(align), interp, L.F16 if half else L.F32, L.stream())
        assert bits_equal(bm.cpu().numpy(), oo.transpose(1, 0, 2).reshape(B, Lv * C))


@pytest.mark.parametrize("D,C,half,gridtype,align,interp", [c for c in GRID_CASES if not (c[2] and c[1] % 2)])
def test_grid_encode_backward(be, oracle, D, C, half, gridtype, align, interp):
    """Backward: atomics make the summation order free, so fp tolerance (fp32: 1e-5 rel; fp16: half ulps of the sums)."""
    torch, ge = be["torch"], be["ge"]
    from nerf2mesh_amd import _lib as L
    rng = np.random.default_rng(14)
    Lv, H = (16, 16) if D == 3 else (8, 8)
    pls = 1.3819129 if D == 3 else 1.5
    offs = oracle.level_offsets(D, Lv, pls, H, 19 if D == 3 else 14, align)
    S = float(np.log2(pls))
    emb = (rng.random((int(offs[-1]), C), dtype=np.float32) * 2 - 1) * 0.1
    B = 6007
    x = rng.random((B, D), dtype=np.float32)
    x[2, 0] = -0.001
    embt = emb.astype(np.float16) if half else emb
    tdt = torch.float16 if half else torch.float32
    X, E, O = dev(be, x), dev(be, embt), dev(be, offs)
    g = rng.normal(size=(Lv, B, C)).astype(embt.dtype)
    _, ody = oracle.grid_encode_forward(x, embt, offs, S, H, Lv, True, gridtype, align, interp)
    for max_level in (Lv, 5):
        ge_out = torch.zeros(int(offs[-1]), C, dtype=tdt, device="cuda")
        gin = torch.zeros(B, D, dtype=tdt, device="cuda")
        ge.grid_encode_backward(dev(be, g), X, E, O, ge_out, B, D, C, Lv, max_level, S, H, dev(be, ody), gin, gridtype, align, interp)
        og, ogi = oracle.grid_encode_backward(g, x, embt, offs, S, H, max_level, ody, gridtype, align, interp)
        got = ge_out.cpu().numpy().astype(np.float32)
        ref = og.astype(np.float32)
        if half:
            # coarse rows sum hundreds of half-rounded terms: compare against an fp64 accumulation bound
            np.testing.assert_allclose(got, ref, rtol=3e-2, atol=3e-2 * np.abs(ref).max())
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
        assert bits_equal(gin.cpu().numpy(), ogi)           # input gradient is a serial per-thread sum: exact
        gbm = torch.zeros(int(offs[-1]), C, dtype=tdt, device="cuda")
        gsm = dev(be, np.ascontiguousarray(g.transpose(1, 0, 2)).reshape(B, Lv * C))
        L.call("n2m_grid_encode_backward_bm", gsm.data_ptr(), X.data_ptr(), E.data_ptr(), O.data_ptr(), gbm.data_ptr(), B, D, C, Lv,
               max_level, S, H, gridtype, int(align), interp, L.F16 if half else L.F32, L.stream())
        got2 = gbm.cpu().numpy().astype(np.float32)
        if half:
            np.testing.assert_allclose(got2, ref, rtol=3e-2, atol=3e-2 * np.abs(ref).max())
        else:
            np.testing.assert_allclose(got2, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())


def test_grid_encode_forward_pair_is_bit_identical_to_two_calls(be):
    """n2m_grid_encode_forward_pair == two n2m_grid_encode_forward calls (fp32 C=1 + fp16 C=2 tables of identical geometry)."""
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd.gridencoder import GridEncoder
    for gridtype, align, interp in (("hash", False, "linear"), ("tiled", True, "smoothstep")):
        e1 = GridEncoder(level_dim=1, desired_resolution=2048, gridtype=gridtype, align_corners=align, interpolation=interp).cuda()
        e2 = GridEncoder(level_dim=2, desired_resolution=2048, gridtype=gridtype, align_corners=align, interpolation=interp).cuda()
        with torch.no_grad():
            e1.embeddings.uniform_(-1, 1); e2.embeddings.uniform_(-1, 1)
        emb1, emb2 = e1.embeddings.detach().contiguous(), e2.embeddings.detach().half().contiguous()
        B = 70001
        x = torch.rand(B, 3, device="cuda")
        x[0] = 0.0; x[1] = 1.0; x[2, 1] = 1.001
        S = float(np.log2(e1.per_level_scale))
        for ml in (16, 7):
            h1 = torch.full((16, B, 1), 3.0, device="cuda"); h2 = torch.full((16, B, 2), 3.0, device="cuda", dtype=torch.float16)
            L.call("n2m_grid_encode_forward_pair", x.data_ptr(), emb1.data_ptr(), emb2.data_ptr(), e1.offsets.data_ptr(), h1.data_ptr(), h2.data_ptr(),
                   B, 16, ml, S, 16, e1.gridtype_id, int(align), e1.interp_id, 1.0, 0.0, L.stream())
            r1 = torch.full_like(h1, 3.0); r2 = torch.full_like(h2, 3.0)
            for emb, out, C, dt in ((emb1, r1, 1, L.F32), (emb2, r2, 2, L.F16)):
                L.call("n2m_grid_encode_forward", x.data_ptr(), emb.data_ptr(), e1.offsets.data_ptr(), out.data_ptr(), B, 3, C, 16, ml, S, 16, None,
                       e1.gridtype_id, int(align), e1.interp_id, dt, L.stream())
            assert torch.equal(h1, r1) and torch.equal(h2.view(torch.int16), r2.view(torch.int16))
            # packed copy of the two tables (8-byte rows): same outputs
            pk = torch.empty(emb1.shape[0], 2, device="cuda")
            pk[:, 0] = emb1[:, 0]
            pk.view(torch.float16)[:, 2:] = emb2
            p1 = torch.full_like(h1, 3.0); p2 = torch.full_like(h2, 3.0)
            L.call("n2m_grid_encode_forward_packed", x.data_ptr(), pk.data_ptr(), e1.offsets.data_ptr(), p1.data_ptr(), p2.data_ptr(), B, 16, ml, S,
                   16, e1.gridtype_id, int(align), e1.interp_id, 1.0, 0.0, L.stream())
            assert torch.equal(p1, r1) and torch.equal(p2.view(torch.int16), r2.view(torch.int16))
            # raw points in [-bound, bound] + in-kernel (x * 1/(2 bound) + 0.5) == torch's (x + bound) / (2 bound), bit for bit
            for bound in (1.0, 2.0, 16.0):
                raw = (x * 2 - 1) * bound
                x01t = (raw + bound) / (2 * bound)
                a1 = torch.full_like(h1, 3.0); a2 = torch.full_like(h2, 3.0); b1 = torch.full_like(h1, 3.0); b2 = torch.full_like(h2, 3.0)
                L.call("n2m_grid_encode_forward_pair", raw.data_ptr(), emb1.data_ptr(), emb2.data_ptr(), e1.offsets.data_ptr(), a1.data_ptr(),
                       a2.data_ptr(), B, 16, ml, S, 16, e1.gridtype_id, int(align), e1.interp_id, 1.0 / (2 * bound), 0.5, L.stream())
                L.call("n2m_grid_encode_forward_pair", x01t.data_ptr(), emb1.data_ptr(), emb2.data_ptr(), e1.offsets.data_ptr(), b1.data_ptr(),
                       b2.data_ptr(), B, 16, ml, S, 16, e1.gridtype_id, int(align), e1.interp_id, 1.0, 0.0, L.stream())
                assert torch.equal(a1, b1) and torch.equal(a2.view(torch.int16), b2.view(torch.int16))


def test_packed_forward_in_level_halves_is_the_full_call(be):
    """n2m_grid_encode_forward_packed_levels (the lookup of a sharded step whose packed rows arrive in two level chunks): levels [0, 8) and
    [8, 16) written by two calls == the one full call, bit for bit, on both outputs; a call leaves the other levels' rows alone."""
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd.gridencoder import GridEncoder
    p = L.ptr
    B = 100003
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.rand(B, 3, device="cuda", generator=g) * 2.2 - 1.1).contiguous()           # some points outside [-1, 1]
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    e2 = GridEncoder(level_dim=2, desired_resolution=2048).cuda()
    rows = e1.embeddings.shape[0]
    packed = torch.empty(rows, 2, dtype=torch.float32, device="cuda")
    packed[:, 0] = e1.embeddings.detach()[:, 0]
    packed[:, 1] = e2.embeddings.detach().half().view(torch.float32)[:, 0]
    geo = (16, 16, float(np.log2(e1.per_level_scale)), int(e1.base_resolution), e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id, 0.5, 0.5)
    a1 = torch.empty(16, B, device="cuda"); a2 = torch.empty(16, B, 2, device="cuda", dtype=torch.float16)
    L.call("n2m_grid_encode_forward_packed", p(x), p(packed), p(e1.offsets), p(a1), p(a2), B, *geo, L.stream())
    b1 = torch.full_like(a1, 7.0); b2 = torch.full_like(a2, 7.0)
    L.call("n2m_grid_encode_forward_packed_levels", p(x), p(packed), p(e1.offsets), p(b1), p(b2), B, *geo, 0, 8, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(b1[:8], a1[:8]) and torch.equal(b2[:8], a2[:8]) and bool((b1[8:] == 7.0).all()) and bool((b2[8:] == 7.0).all())
    L.call("n2m_grid_encode_forward_packed_levels", p(x), p(packed), p(e1.offsets), p(b1), p(b2), B, *geo, 8, 8, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(b1, a1) and torch.equal(b2, a2)
    with pytest.raises(RuntimeError):
        L.call("n2m_grid_encode_forward_packed_levels", p(x), p(packed), p(e1.offsets), p(b1), p(b2), B, *geo, 12, 8, L.stream())


def test_grid_backward_linearity_full_size(be):
    """Size-independent property at the BASELINE size (B = 2^18, lego tables): backward is linear in grad and
    sum(grad_embeddings) == sum_b sum_l grad[b,l] (the 8 interpolation weights of a sample sum to 1)."""
    torch = be["torch"]
    from nerf2mesh_amd.gridencoder import GridEncoder
    enc = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    B = 2 ** 18
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(B, 3, device="cuda", generator=g) * 2 - 1
    out = enc(x, bound=1)
    assert out.shape == (B, 16)
    w = torch.randn(B, 16, device="cuda", generator=g)
    (out * w).sum().backward()
    g1 = enc.embeddings.grad.clone()
    enc.embeddings.grad = None
    (enc(x, bound=1) * (2.5 * w)).sum().backward()
    np.testing.assert_allclose(enc.embeddings.grad.sum().item(), 2.5 * g1.sum().item(), rtol=1e-3)
    np.testing.assert_allclose(g1.double().sum().item(), w.double().sum().item(), rtol=1e-3, atol=1.0)
    # per-level mass conservation
    offs = enc.host_offsets
    for l in (0, 4, 5, 15):
        np.testing.assert_allclose(g1[offs[l]:offs[l + 1]].double().sum().item(), w[:, l].double().sum().item(), rtol=1e-3, atol=0.5)


def test_grad_total_variation(be, oracle):
    torch, ge = be["torch"], be["ge"]
    rng = np.random.default_rng(6)
    offs, S = lego_offsets(1.0)
    emb = (rng.random((int(offs[-1]), 1), dtype=np.float32) * 2 - 1) * 1e-2
    x = rng.random((50000, 3), dtype=np.float32)
    x[:3] = [[0, 0, 0], [1, 1, 1], [1.2, 0.5, 0.5]]
    g0 = rng.normal(size=emb.shape).astype(np.float32) * 1e-6
    g_h = dev(be, g0)
    g_o = g0.copy()
    ge.grad_total_variation(dev(be, x), dev(be, emb), g_h, dev(be, offs), 1e-3, 50000, 3, 1, 16, S, 16, 0, False)
    oracle.grad_total_variation(x, emb, g_o, offs, 1e-3, S, 16, 0, False)
    np.testing.assert_allclose(g_h.cpu().numpy(), g_o, rtol=1e-4, atol=1e-8)    # atomic order only


def _binned_backward(be, g, X, offs, ge_out, C, Lv, max_level, S, H, gridtype, align, interp, half):
    from nerf2mesh_amd import _lib as L
    ho = np.ascontiguousarray(offs, dtype=np.int32)
    B = X.shape[0]
    dt = L.F16 if half else L.F32
    need = L.lib().n2m_grid_binned_workspace_bytes(B, 3, C, max_level, ho.ctypes.data, dt, 0)
    assert need > 0
    ws = be["torch"].empty(need, dtype=be["torch"].uint8, device="cuda")
    L.call("n2m_grid_encode_backward_binned", g.data_ptr(), X.data_ptr(), ho.ctypes.data, ge_out.data_ptr(), B, 3, C, Lv, max_level, S, H,
           gridtype, int(align), interp, dt, None, 0.0, 0.0, 1.0, None, None, ws.data_ptr(), need, L.stream())


@pytest.mark.parametrize("C,half,gridtype,align,interp,log2", [
    (1, False, 0, False, 0, 19), (2, True, 0, False, 0, 19), (1, False, 1, False, 1, 19), (2, True, 0, True, 0, 19),
    (1, False, 0, False, 0, 14), (2, True, 1, True, 1, 16)])
def test_grid_encode_backward_binned(be, oracle, C, half, gridtype, align, interp, log2):
    """Binned fixed-point backward == oracle (same bars as the generic kernel), accumulates into a pre-filled table,
    is bit-reproducible run to run, and lets inf/nan through untouched."""
    torch = be["torch"]
    rng = np.random.default_rng(15)
    Lv, H, pls = 16, 16, 1.3819129
    offs = oracle.level_offsets(3, Lv, pls, H, log2, align)
    S = float(np.log2(pls))
    B = 6007
    x = rng.random((B, 3), dtype=np.float32)
    x[2, 0] = -0.001
    x[5] = 1.0
    tdt = torch.float16 if half else torch.float32
    ndt = np.float16 if half else np.float32
    emb = np.zeros((int(offs[-1]), C), ndt)
    g = rng.normal(size=(Lv, B, C)) * np.exp(rng.normal(size=(Lv, B, 1)) * (1.5 if half else 3))      # wide dynamic range
    g = np.clip(g, -3e4, 3e4).astype(ndt)
    g[:, 7] = 0
    X, G = dev(be, x), dev(be, g)
    for max_level in (Lv, 5):
        pre = (rng.normal(size=emb.shape) * 0.01).astype(ndt)
        out = dev(be, pre.copy())
        _binned_backward(be, G, X, offs, out, C, Lv, max_level, S, H, gridtype, align, interp, half)
        og = oracle.grid_encode_backward(g, x, emb, offs, S, H, max_level, None, gridtype, align, interp)
        got = out.cpu().numpy().astype(np.float64) - pre.astype(np.float64)
        ref = og.astype(np.float64)
        if half:
            # the reference's own result is one ORDER of serial half adds (gridencoder.cu:324-334): only a loose bar holds against it ...
            np.testing.assert_allclose(got, ref, rtol=3e-2, atol=3e-2 * np.abs(ref).max())
            # ... the tight one is against what this kernel claims to compute: half(pre + EXACT sum of the reference's half-rounded terms).
            # Allowed: the final rounding (half an ulp of the result; 1.002 for float-then-half double rounding), the 64-bit fixed-point
            # unit (2^-38 of the level's largest term, per term) and the two fp32 roundings of the flush (sum -> float, float + pre)
            ex, _, cnt = oracle.grid_encode_backward_exact(g, x, offs, S, H, C, True, max_level, gridtype, align, interp)
            want = pre.astype(np.float64) + ex
            have = out.cpu().numpy().astype(np.float64)
            level_of = np.repeat(np.arange(Lv), np.diff(offs))
            lmax = np.array([np.abs(g[l].astype(np.float32)).max() for l in range(Lv)], np.float64)[level_of][:, None]
            bound = 0.5 * _half_ulp(np.maximum(np.abs(want), np.abs(have))) * 1.002 + cnt * 2.0 ** -38 * lmax + 2.0 ** -22 * (np.abs(ex) + np.abs(want))
            worst = (np.abs(have - want) / bound)[cnt > 0].max()
            print(f"binned fp16 C={C} max_level={max_level}: worst err / exact-sum bound {worst:.3f} over {int((cnt > 0).sum())} entries")
            assert worst <= 1.0, f"binned fp16 gradient is not the exact sum of the reference's terms (err/bound {worst:.3f})"
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
        assert np.array_equal(out.cpu().numpy()[offs[max_level]:], pre[offs[max_level]:])     # levels >= max_level untouched
        out2 = dev(be, pre.copy())
        _binned_backward(be, G, X, offs, out2, C, Lv, max_level, S, H, gridtype, align, interp, half)
        assert torch.equal(out, out2), "integer accumulation + single-owner flush must be bit-reproducible"
    # non-finite gradients propagate to exactly the rows they touch
    g2 = g.copy()
    g2[3, 11] = np.inf
    out = torch.zeros(int(offs[-1]), C, dtype=tdt, device="cuda")
    _binned_backward(be, dev(be, g2), X, offs, out, C, Lv, Lv, S, H, gridtype, align, interp, half)
    clean = torch.zeros(int(offs[-1]), C, dtype=tdt, device="cuda")
    _binned_backward(be, G, X, offs, clean, C, Lv, Lv, S, H, gridtype, align, interp, half)
    bad = ~torch.isfinite(out).all(dim=1)
    assert 1 <= int(bad.sum()) <= 8
    lo, hi = int(offs[3]), int(offs[4])
    assert bool(bad[lo:hi].any()) and not bool(bad[:lo].any()) and not bool(bad[hi:].any())
    assert torch.equal(out[~bad], clean[~bad])


def test_grid_backward_binned_full_size(be):
    """BASELINE size (B = 2^18, lego tables, both table formats): per-level mass conservation, agreement with the
    partition kernel, reproducibility of the single-owner (hashed) levels."""
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward
    B = 2 ** 18
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(B, 3, device="cuda", generator=gen)
    for C, tdt in ((1, torch.float32), (2, torch.float16)):
        enc = GridEncoder(level_dim=C, desired_resolution=2048).cuda()
        offs = enc.host_offsets
        grad = (torch.randn(16, B, C, device="cuda", generator=gen) * 0.05).to(tdt)
        a = torch.zeros(offs[-1], C, device="cuda", dtype=tdt)
        assert binned_backward(enc, grad, x, a, 16)
        b = torch.zeros_like(a)
        assert binned_backward(enc, grad, x, b, 16)
        ref = torch.zeros_like(a)
        emb = enc.embeddings.detach().to(tdt)
        L.call("n2m_grid_encode_backward", grad.data_ptr(), x.data_ptr(), emb.data_ptr(), enc.offsets.data_ptr(), ref.data_ptr(), B, 3, C, 16, 16,
               float(np.log2(enc.per_level_scale)), 16, None, None, 0, 0, 0, L.F16 if C == 2 else L.F32, L.stream())
        tol = 2e-2 if C == 2 else 1e-4
        np.testing.assert_allclose(a.float().cpu().numpy(), ref.float().cpu().numpy(), rtol=tol, atol=tol * float(ref.float().abs().max()))
        for l in range(16):
            sl = slice(offs[l], offs[l + 1])
            np.testing.assert_allclose(a[sl].double().sum().item(), grad[l].double().sum().item(), rtol=2e-3, atol=0.5 if C == 2 else 0.05)
            if offs[l + 1] - offs[l] == 2 ** 19:
                assert torch.equal(a[sl], b[sl])


def test_grid_backward_binned_multi_pass(be):
    """B > 2^20 runs in passes over one workspace: same result as the partition kernel."""
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward
    B = 2 ** 20 + 70001
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand(B, 3, device="cuda", generator=gen)
    enc = GridEncoder(level_dim=2, desired_resolution=2048).cuda()
    offs = enc.host_offsets
    grad = (torch.randn(16, B, 2, device="cuda", generator=gen) * 0.05).half()
    a = torch.zeros(offs[-1], 2, device="cuda", dtype=torch.float16)
    assert binned_backward(enc, grad, x, a, 16)
    ref = torch.zeros_like(a)
    emb = enc.embeddings.detach().half()
    L.call("n2m_grid_encode_backward", grad.data_ptr(), x.data_ptr(), emb.data_ptr(), enc.offsets.data_ptr(), ref.data_ptr(), B, 3, 2, 16, 16,
           float(np.log2(enc.per_level_scale)), 16, None, None, 0, 0, 0, L.F16, L.stream())
    np.testing.assert_allclose(a.float().cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-2, atol=2e-2 * float(ref.float().abs().max()))
    for l in (0, 7, 15):
        np.testing.assert_allclose(a[offs[l]:offs[l + 1]].double().sum().item(), grad[l].double().sum().item(), rtol=2e-3, atol=1.0)


@pytest.mark.parametrize("with_tv", [False, True])
def test_grid_backward_binned_pair_equals_two_calls(be, with_tv):
    """One shared fill for the density (fp32 C=1) and colour (fp16 C=2) tables == the two single-table calls: the sums are exact
    fixed-point sums of the same products, so hashed levels agree bit for bit and the split dense levels to atomic-order noise."""
    torch = be["torch"]
    from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward, binned_backward_pair
    B = 100003
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = torch.rand(B, 3, device="cuda", generator=gen)
    x[5] = 1.0; x[6, 1] = -0.01
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    e2 = GridEncoder(level_dim=2, desired_resolution=2048).cuda()
    offs = e1.host_offsets
    d1 = torch.randn(16, B, 1, device="cuda", generator=gen) * (torch.rand(1, B, 1, device="cuda", generator=gen) < 0.7)
    d2 = (torch.randn(16, B, 2, device="cuda", generator=gen) * (torch.rand(1, B, 1, device="cuda", generator=gen) < 0.7)).half()
    emb1 = e1.embeddings.detach().float().contiguous()
    scale = torch.tensor(128.0, device="cuda")
    tv = (emb1, 1e-3, 1e-2, 0.3, scale) if with_tv else None
    a1 = torch.zeros(offs[-1], 1, device="cuda"); a2 = torch.zeros(offs[-1], 2, device="cuda", dtype=torch.float16)
    flag = torch.zeros((), device="cuda")
    assert binned_backward_pair(e1, e2, d1, d2, x, a1, a2, 16, tv=tv, found_inf=flag) and float(flag) == 0
    b1 = torch.zeros_like(a1); b2 = torch.zeros_like(a2)
    assert binned_backward(e1, d1, x, b1, 16, tv=tv) and binned_backward(e2, d2, x, b2, 16)
    np.testing.assert_allclose(a1.cpu().numpy(), b1.cpu().numpy(), rtol=1e-5, atol=1e-6 * float(b1.abs().max()))
    np.testing.assert_allclose(a2.float().cpu().numpy(), b2.float().cpu().numpy(), rtol=2e-3, atol=2e-3 * float(b2.float().abs().max()))
    for l in range(9, 16):                           # below level 9 the shared fill merges same-cell runs first (fp32 partial sums)
        if offs[l + 1] - offs[l] == 2 ** 19:
            sl = slice(offs[l], offs[l + 1])
            assert torch.equal(a1[sl], b1[sl]) and torch.equal(a2[sl], b2[sl])
    d2n = d2.clone(); d2n[3, 17, 0] = float("inf")
    assert binned_backward_pair(e1, e2, d1, d2n, x, torch.zeros_like(a1), torch.zeros_like(a2), 16, found_inf=flag) and float(flag) == 1


@pytest.mark.parametrize("B,max_level", [(5, 16), (1025, 7), (40000, 1)])
def test_grid_backward_binned_pair_edge_cases(be, oracle, B, max_level):
    """Shared-fill backward vs the oracle on tiny / ragged batches and truncated level ranges; all-zero gradients leave the tables
    untouched; levels >= max_level are never written."""
    torch = be["torch"]
    from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward_pair
    rng = np.random.default_rng(B)
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    e2 = GridEncoder(level_dim=2, desired_resolution=2048).cuda()
    offs = np.asarray(e1.host_offsets, np.int32)
    S = float(np.log2(e1.per_level_scale))
    x = rng.random((B, 3), dtype=np.float32)
    x[0] = [1.0, 0.0, 0.5]
    d1 = rng.normal(size=(16, B, 1)).astype(np.float32)
    d2 = rng.normal(size=(16, B, 2)).astype(np.float16)
    pre1 = (rng.normal(size=(int(offs[-1]), 1)) * 0.01).astype(np.float32)
    pre2 = (rng.normal(size=(int(offs[-1]), 2)) * 0.01).astype(np.float16)
    a1, a2 = dev(be, pre1.copy()), dev(be, pre2.copy())
    assert binned_backward_pair(e1, e2, dev(be, d1), dev(be, d2), dev(be, x), a1, a2, max_level)
    o1 = oracle.grid_encode_backward(d1, x, np.zeros_like(pre1), offs, S, 16, max_level)
    o2 = oracle.grid_encode_backward(d2, x, np.zeros_like(pre2), offs, S, 16, max_level)
    np.testing.assert_allclose(a1.cpu().numpy() - pre1, o1, rtol=1e-4, atol=1e-5 * max(np.abs(o1).max(), 1e-6))
    got2 = a2.cpu().numpy().astype(np.float32) - pre2.astype(np.float32)
    np.testing.assert_allclose(got2, o2.astype(np.float32), rtol=3e-2, atol=3e-2 * max(np.abs(o2).max(), 1e-3))
    top = int(offs[max_level])
    assert np.array_equal(a1.cpu().numpy()[top:], pre1[top:]) and np.array_equal(a2.cpu().numpy()[top:], pre2[top:])
    z1, z2 = dev(be, pre1.copy()), dev(be, pre2.copy())
    assert binned_backward_pair(e1, e2, torch.zeros(16, B, 1, device="cuda"), torch.zeros(16, B, 2, device="cuda", dtype=torch.float16), dev(be, x),
                                z1, z2, max_level)
    assert np.array_equal(z1.cpu().numpy(), pre1) and np.array_equal(z2.cpu().numpy(), pre2)


@pytest.mark.parametrize("B,max_level,with_tv", [(100003, 16, True), (1025, 7, False), (0, 16, False), (2 ** 20 + 70001, 16, False)])
def test_grid_backward_binned_pair_overwrite_mode(be, B, max_level, with_tv):
    """overwrite=1 on garbage-filled tables == overwrite=0 on zero-filled ones: bit for bit where one workgroup owns a partition
    (the 2^19-row levels), to atomic-order noise on the split dense levels; rows of levels >= max_level come back as zeros; a
    non-finite gradient reaches the same rows in both modes (second walk after the stores) and raises found_inf."""
    torch = be["torch"]
    from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward_pair
    gen = torch.Generator(device="cuda").manual_seed(B + max_level)
    x = torch.rand(B, 3, device="cuda", generator=gen)
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    e2 = GridEncoder(level_dim=2, desired_resolution=2048).cuda()
    offs = e1.host_offsets
    d1 = torch.randn(16, B, 1, device="cuda", generator=gen) * (torch.rand(1, B, 1, device="cuda", generator=gen) < 0.7)
    d2 = (torch.randn(16, B, 2, device="cuda", generator=gen) * (torch.rand(1, B, 1, device="cuda", generator=gen) < 0.7)).half()
    if B > 100:
        d1[2, 40, 0] = float("inf"); d2[5, 77, 1] = float("nan")
    tv = (e1.embeddings.detach().float().contiguous(), 1e-3, 1e-2, 0.3, torch.tensor(128.0, device="cuda")) if with_tv else None
    z1 = torch.zeros(offs[-1], 1, device="cuda"); z2 = torch.zeros(offs[-1], 2, device="cuda", dtype=torch.float16)
    w1 = torch.full_like(z1, 123.0); w2 = torch.full_like(z2, -7.0)
    f0, f1 = torch.zeros((), device="cuda"), torch.zeros((), device="cuda")
    assert binned_backward_pair(e1, e2, d1, d2, x, z1, z2, max_level, tv=tv, found_inf=f0)
    assert binned_backward_pair(e1, e2, d1, d2, x, w1, w2, max_level, tv=tv, found_inf=f1, overwrite=True)
    assert float(f0) == float(f1) == (1.0 if B > 100 else 0.0)
    a1, b1 = z1.cpu().numpy(), w1.cpu().numpy()
    a2, b2 = z2.float().cpu().numpy(), w2.float().cpu().numpy()
    assert np.array_equal(np.isfinite(a1), np.isfinite(b1)) and np.array_equal(np.isfinite(a2), np.isfinite(b2))
    if B > 100:
        assert not np.isfinite(a1).all() and not np.isfinite(a2).all()
    fin1, fin2 = np.isfinite(a1), np.isfinite(a2)
    np.testing.assert_allclose(b1[fin1], a1[fin1], rtol=1e-5, atol=1e-6 * max(float(np.abs(a1[fin1]).max()), 1e-30))
    # fp16 table, levels split over tile groups: up to 64 groups x passes partial sums reach a row as fp16 atomics, each rounding to
    # half an fp16 ulp (2^-11 relative) in arrival order -- both runs carry that noise: |diff| up to ~sqrt(128) * 2^-11 * |row| typical,
    # 128 * 2^-11 worst.  (2e-3 * max, the earlier bar, is ~3 sigma of that for the 2^20-sample case and failed 4 runs in 12.)
    np.testing.assert_allclose(b2[fin2], a2[fin2], rtol=2e-3, atol=1e-2 * max(float(np.abs(a2[fin2]).max()), 1e-30))
    for l in range(max_level if B else 0):
        if offs[l + 1] - offs[l] == 2 ** 19 and B <= 2 ** 20:
            sl = slice(offs[l], offs[l + 1])
            ok = fin1[sl].all(-1) & fin2[sl].all(-1)        # a row hit by an inf/nan also takes its finite part in atomic order
            assert np.array_equal(a1[sl][ok], b1[sl][ok]) and np.array_equal(a2[sl][ok], b2[sl][ok])
    top = offs[max_level] if B else 0
    assert not b1[top:].any() and not b2[top:].any()


@pytest.mark.parametrize("with_tv", [False, True])
def test_grid_backward_binned_pair_merges_runs_along_rays(be, oracle, with_tv):
    """Consecutive samples of a ray share their cell on the coarse levels; the shared fill reduces such runs of lanes to one entry per
    vertex (segmented DPP scan) before sorting.  Inputs: points marching along rays, 5..40 per ray, so that runs start and end
    anywhere in the 16-lane rows.  Checked against the oracle (no merging) for both tables, and for run-to-run bit-reproducibility."""
    torch = be["torch"]
    from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward_pair
    rng = np.random.default_rng(21)
    pts = []
    while sum(len(p) for p in pts) < 30000:
        o = rng.random(3) * 0.6 + 0.2
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        n = int(rng.integers(5, 41))
        t = np.arange(n)[:, None] * 0.0017 * rng.uniform(0.5, 2.0)
        pts.append((o + t * d).astype(np.float32))
    x = np.clip(np.concatenate(pts), -0.05, 1.05).astype(np.float32)          # a few leave the cube: they break runs
    B = x.shape[0]
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    e2 = GridEncoder(level_dim=2, desired_resolution=2048).cuda()
    offs = np.asarray(e1.host_offsets, np.int32)
    S = float(np.log2(e1.per_level_scale))
    d1 = rng.normal(size=(16, B, 1)).astype(np.float32)
    d2 = rng.normal(size=(16, B, 2)).astype(np.float16)
    emb1 = ((rng.random((int(offs[-1]), 1), dtype=np.float32) * 2 - 1) * 1e-2)
    tv = (dev(be, emb1), 1e-3, 1e-3, 1.0, None) if with_tv else None
    outs = []
    for _ in range(2):
        a1 = torch.full((int(offs[-1]), 1), 9.0, device="cuda"); a2 = torch.full((int(offs[-1]), 2), 9.0, device="cuda", dtype=torch.float16)
        assert binned_backward_pair(e1, e2, dev(be, d1), dev(be, d2), dev(be, x), a1, a2, 16, tv=tv, overwrite=True)
        outs.append((a1, a2))
    for l in range(16):                                  # single-owner partitions: identical bits run to run
        if offs[l + 1] - offs[l] == 2 ** 19:
            sl = slice(int(offs[l]), int(offs[l + 1]))
            assert torch.equal(outs[0][0][sl], outs[1][0][sl]) and torch.equal(outs[0][1][sl], outs[1][1][sl])
    o1 = oracle.grid_encode_backward(d1, x, np.zeros((int(offs[-1]), 1), np.float32), offs, S, 16, 16)
    if with_tv:
        oracle.grad_total_variation(x, emb1, o1, offs, 1e-3, S, 16, 0, False)
    o2 = oracle.grid_encode_backward(d2, x, np.zeros((int(offs[-1]), 2), np.float16), offs, S, 16, 16)
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), o1, rtol=1e-4, atol=1e-5 * np.abs(o1).max())
    np.testing.assert_allclose(outs[0][1].float().cpu().numpy(), o2.astype(np.float32), rtol=3e-2, atol=3e-2 * np.abs(o2).max())
    for l in (0, 3, 8, 15):                              # column sums: nothing lost or counted twice by the merge
        sl = slice(int(offs[l]), int(offs[l + 1]))
        np.testing.assert_allclose(outs[0][0][sl].double().sum().item(), o1[sl].astype(np.float64).sum(), rtol=1e-4, atol=1e-4 * np.abs(d1[l]).sum())


def test_grad_total_variation_binned(be, oracle):
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    rng = np.random.default_rng(7)
    offs, S = lego_offsets(1.0)
    emb = (rng.random((int(offs[-1]), 1), dtype=np.float32) * 2 - 1) * 1e-2
    B = 50000
    x = rng.random((B, 3), dtype=np.float32)
    x[:3] = [[0, 0, 0], [1, 1, 1], [1.2, 0.5, 0.5]]
    g0 = rng.normal(size=emb.shape).astype(np.float32) * 1e-6
    g_h, g_o = dev(be, g0), g0.copy()
    ho = np.ascontiguousarray(offs, dtype=np.int32)
    need = L.lib().n2m_grid_binned_workspace_bytes(B, 3, 1, 16, ho.ctypes.data, L.F32, 1)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    L.call("n2m_grad_total_variation_binned", dev(be, x).data_ptr(), dev(be, emb).data_ptr(), g_h.data_ptr(), ho.ctypes.data, 1e-3, 1e-3, 1.0, None,
           B, 3, 1, 16, S, 16, 0, 0, ws.data_ptr(), need, L.stream())
    oracle.grad_total_variation(x, emb, g_o, offs, 1e-3, S, 16, 0, False)
    np.testing.assert_allclose(g_h.cpu().numpy(), g_o, rtol=1e-4, atol=1e-8)
    # inner/outer weighting (nerf/utils.py:815-821) and the device-side scale factor == two oracle passes with scaled weights
    scale = torch.tensor([512.0], device="cuda")
    g_h2, g_o2 = dev(be, g0), g0.copy()
    L.call("n2m_grad_total_variation_binned", dev(be, x).data_ptr(), dev(be, emb).data_ptr(), g_h2.data_ptr(), ho.ctypes.data, 1e-3, 1e-2, 0.25,
           scale.data_ptr(), B, 3, 1, 16, S, 16, 0, 0, ws.data_ptr(), need, L.stream())
    inner = np.abs(x - 0.5).max(-1) <= 0.25
    oracle.grad_total_variation(x[inner], emb, g_o2, offs, 1e-3 * 512, S, 16, 0, False)
    oracle.grad_total_variation(x[~inner], emb, g_o2, offs, 1e-2 * 512, S, 16, 0, False)
    np.testing.assert_allclose(g_h2.cpu().numpy(), g_o2, rtol=1e-4, atol=1e-7)


def test_grid_backward_binned_with_fused_tv(be, oracle):
    """tv_embeddings != NULL: one call == oracle backward + oracle TV over the same inputs (TV rides on vertex 000's update)."""
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    rng = np.random.default_rng(8)
    offs, S = lego_offsets(1.0)
    emb = (rng.random((int(offs[-1]), 1), dtype=np.float32) * 2 - 1) * 1e-2
    B = 30011
    x = rng.random((B, 3), dtype=np.float32)
    x[:3] = [[0, 0, 0], [1, 1, 1], [1.2, 0.5, 0.5]]
    g = (rng.normal(size=(16, B, 1)) * 1e-3).astype(np.float32)
    g[:, 100:200] = 0                                             # zero gradient, TV term alone keeps the entry alive
    ho = np.ascontiguousarray(offs, dtype=np.int32)
    need = L.lib().n2m_grid_binned_workspace_bytes(B, 3, 1, 16, ho.ctypes.data, L.F32, 0)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    out = torch.zeros(int(offs[-1]), 1, device="cuda")
    scale = torch.tensor([64.0], device="cuda")
    L.call("n2m_grid_encode_backward_binned", dev(be, g).data_ptr(), dev(be, x).data_ptr(), ho.ctypes.data, out.data_ptr(), B, 3, 1, 16, 16, S, 16,
           0, 0, 0, L.F32, dev(be, emb).data_ptr(), 1e-4, 1e-3, 0.3, scale.data_ptr(), None, ws.data_ptr(), need, L.stream())
    ref = oracle.grid_encode_backward(g, x, emb, offs, S, 16, 16, None, 0, False, 0)
    inner = np.abs(x - 0.5).max(-1) <= 0.3
    oracle.grad_total_variation(x[inner], emb, ref, offs, 1e-4 * 64, S, 16, 0, False)
    oracle.grad_total_variation(x[~inner], emb, ref, offs, 1e-3 * 64, S, 16, 0, False)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
    # and the TV part is really there: without it the result differs
    plain = oracle.grid_encode_backward(g, x, emb, offs, S, 16, 16, None, 0, False, 0)
    assert np.abs(ref - plain).max() > 1e-3 * np.abs(ref).max()


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_encode(be, oracle, degree):
    torch, sh = be["torch"], be["sh"]
    rng = np.random.default_rng(7)
    B = 10007
    v = rng.normal(size=(B, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[0] = [0, 0, 1]; v[1] = [1, 0, 0]; v[2] = [0, -1, 0]
    out = torch.empty(B, degree ** 2, device="cuda")
    dy = torch.empty(B, 3 * degree ** 2, device="cuda")
    sh.sh_encode_forward(dev(be, v), out, B, 3, degree, dy)
    oo, ody = oracle.sh_encode_forward(v, degree, True)
    # fp32 recurrences vs the oracle's double evaluation: a few ulp of O(1) values, derivative entries up to O(100)
    np.testing.assert_allclose(out.cpu().numpy(), oo, rtol=0, atol=4e-6)
    np.testing.assert_allclose(dy.cpu().numpy(), ody, rtol=4e-6, atol=1e-4)
    out2 = torch.empty(B, degree ** 2, device="cuda")
    sh.sh_encode_forward(dev(be, v), out2, B, 3, degree, None)
    assert bits_equal(out2.cpu().numpy(), out.cpu().numpy())
    g = rng.normal(size=(B, degree ** 2)).astype(np.float32)
    gi = torch.zeros(B, 3, device="cuda")
    sh.sh_encode_backward(dev(be, g), dev(be, v), B, 3, degree, dev(be, ody), gi)
    assert bits_equal(gi.cpu().numpy(), oracle.sh_encode_backward(g, v, degree, ody))


def test_freq_encode_operator(be, oracle):
    """freq_encode (host operator + HIP kernels) against the oracle and against autograd through torch.sin / torch.cos."""
    torch = be["torch"]
    from nerf2mesh_amd.freqencoder import FreqEncoder
    from nerf2mesh_amd.encoding import get_encoder
    enc, dim = get_encoder("frequency", input_dim=3, multires=6)
    assert isinstance(enc, FreqEncoder) and dim == 3 + 2 * 6 * 3
    x = (torch.randn(5003, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4)) * 2).requires_grad_()
    out = enc(x)
    np.testing.assert_allclose(out.detach().cpu().numpy(), oracle.freq_encode_forward(x.detach().cpu().numpy(), 6), rtol=0, atol=3e-7)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    x2 = x.detach().clone().requires_grad_()
    parts = [x2]
    for f in range(6):
        parts += [torch.sin(x2 * 2.0 ** f), torch.cos(x2 * 2.0 ** f)]
    (torch.cat(parts, -1) * w).sum().backward()
    # the reference's cosine is sin(y + pi/2) in fp32 (freqencoder.cu:58-60): at y = 2^5 x ~ 200 rad the phase add alone costs
    # 1.5e-5, times the 2^f chain factor -> compare against exact sin/cos at that level
    ref_g = x2.grad.cpu().numpy()
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref_g, rtol=2e-3, atol=2e-3 * np.abs(ref_g).max())
    with pytest.raises(RuntimeError, match="C must be D"):
        from nerf2mesh_amd import _lib as L
        L.call("n2m_freq_encode_forward", x.data_ptr(), 8, 3, 4, 20, out.data_ptr(), L.stream())


def test_error_behaviour(be):
    """Same error surface as the reference: RuntimeError for unsupported C/D and bad tensors."""
    torch, ge = be["torch"], be["ge"]
    x = torch.rand(8, 3, device="cuda")
    emb = torch.rand(64, 3, device="cuda")
    offs = torch.tensor([0, 64], dtype=torch.int32, device="cuda")
    out = torch.empty(1, 8, 3, device="cuda")
    with pytest.raises(RuntimeError, match="C must be 1, 2, 4, or 8"):
        ge.grid_encode_forward(x, emb, offs, out, 8, 3, 3, 1, 1, 0.5, 16, None, 0, False, 0)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        ge.grid_encode_forward(x.cpu(), emb, offs, out, 8, 3, 3, 1, 1, 0.5, 16, None, 0, False, 0)
    with pytest.raises(RuntimeError, match="contiguous"):
        ge.grid_encode_forward(x.t().contiguous().t(), emb, offs, out, 8, 3, 3, 1, 1, 0.5, 16, None, 0, False, 0)


def _half_ulp(v):
    """Spacing of fp16 at |v| (subnormal spacing 2^-24 below 2^-14)."""
    a = np.abs(v)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -14)))
    return np.exp2(e - 10)


def test_live_first_sample_order_and_the_fill_path_of_dead_waves(be, oracle, scene):
    """Round 5: n2m_composite_live_counts -> n2m_sample_order_live_first -> n2m_grid_backward_sample_order.
    (a) The permutation is exactly: every ray's live prefix (ray order), then every ray's remaining samples (ray order).
    (b) The table backward in that order against the same call in index order: the unmerged levels (9..15) are BIT-identical (fixed-point
        sums do not depend on the order), the merged levels (0..8) agree to the rounding of the run merge (whose runs the order regroups).
    (c) The TV-only path whole waves of dead samples take is the full path's result bit for bit (measurement switch 128 turns it off)."""
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L, raymarching, synthetic as S
    from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward_pair
    poses = S.make_cameras(64, seed=1).cuda()
    bits = raymarching.packbits(S.scene_density_grid(H=128, device="cuda"), 10.0)
    g = torch.Generator(device="cuda").manual_seed(4)
    o, d = S.random_rays(poses, 20011, g)
    nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device="cuda"), 0.05)
    xyzs, _, _, rays = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, 0.0, 1024)
    M, N = xyzs.shape[0], rays.shape[0]
    assert M > 150000
    x_t = ((xyzs + 1) / 2).contiguous()
    rays_np = rays.cpu().numpy()
    rng = np.random.default_rng(12)
    # per ray: an early stop somewhere (a third of the rays: none, all of their samples are live; a few: at the first sample)
    cnt = rays_np[:, 1].astype(np.int64)
    live = np.minimum(cnt, np.where(rng.random(N) < 0.33, cnt, (rng.random(N) * 0.6 * cnt).astype(np.int64) + (rng.random(N) < 0.9)))
    live_t = torch.from_numpy(live.astype(np.int32)).cuda()
    nblk = (N + 15) // 16
    blk = np.add.reduceat(live, np.arange(0, N, 16)).astype(np.uint32)
    perm = torch.full((M,), -1, dtype=torch.int32, device="cuda")
    L.call("n2m_sample_order_live_first", L.ptr(rays), L.ptr(live_t), L.ptr(torch.from_numpy(blk.view(np.int32)).cuda()), N, M, L.ptr(perm), L.stream())
    # (a) numpy statement
    off = rays_np[:, 0].astype(np.int64)
    idx_live = np.concatenate([off[r] + np.arange(live[r]) for r in range(N)])
    idx_dead = np.concatenate([off[r] + np.arange(live[r], cnt[r]) for r in range(N)])
    want = np.concatenate([idx_live, idx_dead])
    got = perm.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(got), np.arange(M)), "not a permutation"
    assert np.array_equal(got, want)
    # gradients: zero on every level behind the stop, like composite_rays_train's backward leaves them
    alive = np.zeros(M, bool)
    alive[idx_live] = True
    d1 = (rng.standard_normal((16, M, 1)) * np.exp(rng.normal(-6, 2, (1, M, 1))) * alive[None, :, None]).astype(np.float32)
    d2 = (rng.standard_normal((16, M, 2)) * np.exp(rng.normal(-6, 2, (1, M, 1))) * alive[None, :, None] * 128).astype(np.float16)
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    e2 = GridEncoder(level_dim=2, desired_resolution=2048).cuda()
    offs = np.asarray(e1.host_offsets, np.int64)
    rows = int(offs[-1])
    emb1 = dev(be, ((rng.random((rows, 1), dtype=np.float32) * 2 - 1) * 1e-2))
    D1, D2 = dev(be, d1), dev(be, d2)

    def run(order, mode=0, tv=True):
        a1 = torch.zeros(rows, 1, device="cuda"); a2 = torch.zeros(rows, 2, device="cuda", dtype=torch.float16)
        L.call("n2m_debug_fill_times", mode, None)
        L.call("n2m_grid_backward_sample_order", L.ptr(perm) if order else None)
        try:
            assert binned_backward_pair(e1, e2, D1, D2, x_t, a1, a2, 16, tv=(emb1, 1e-4, 1e-4, 1.0, None) if tv else None)
        finally:
            L.call("n2m_grid_backward_sample_order", None)
            L.call("n2m_debug_fill_times", 0, None)
        torch.cuda.synchronize()
        return a1.cpu().numpy(), a2.cpu().numpy()
    for tv in (True, False):
        p1, p2 = run(False, tv=tv)
        q1, q2 = run(True, tv=tv)
        f1, f2 = run(True, mode=128, tv=tv)
        assert bits_equal(q1, f1) and bits_equal(q2, f2), "the TV-only path of dead waves differs from the full path"      # (c)
        lo = int(offs[9])
        assert bits_equal(p1[lo:], q1[lo:]) and bits_equal(p2[lo:], q2[lo:]), "unmerged levels must not depend on the sample order"      # (b)
        for l in range(9):
            a, b = slice(int(offs[l]), int(offs[l + 1])), None
            m1, m2 = np.abs(p1[a]).max(), np.abs(p2[a].astype(np.float32)).max()
            assert np.abs(p1[a] - q1[a]).max() <= 1e-5 * m1 + 1e-30, f"level {l} fp32"
            assert np.abs(p2[a].astype(np.float32) - q2[a].astype(np.float32)).max() <= 2e-3 * m2 + 1e-30, f"level {l} fp16"
        assert np.abs(q1).max() > 0 and np.abs(q2.astype(np.float32)).max() > 0
    # and bit-reproducible run to run in the live-first order
    r1, r2 = run(True)
    s1, s2 = run(True)
    assert bits_equal(r1, s1) and bits_equal(r2, s2)


@pytest.mark.parametrize("with_tv", [False, True])
def test_binned_pair_backward_is_the_exact_sum_at_full_batch(be, oracle, scene, with_tv):
    """B = 2^18 samples of the marcher (ray-ordered, so the run merge is active), lego tables, both table gradients from the shared
    binned backward.  The kernel claims: every term is rounded exactly like the reference rounds it ((half)(w * g) for the fp16 table,
    gridencoder.cu:326), then summed EXACTLY (64-bit fixed point), then rounded once.  So each entry must equal the double-precision
    sum of the reference's terms (oracle.grid_encode_backward_exact) up to
      * the final rounding: half an ulp of the result (fp16 table) / 2^-24 relative (fp32 table),
      * on merged levels (0..8): one rounding of every merged run's partial sum to the entry type: <= 2^-11 (fp16) / 2^-23 (fp32) of the sum
        of the magnitudes of the row's terms,
      * the fixed-point unit: 2^-38 of the level's largest term, half a unit per term.
    The reference's own order-dependent result (serial half adds) is farther from that sum than this bound -- asserted at the end."""
    torch = be["torch"]
    from nerf2mesh_amd import raymarching, synthetic as S
    from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward_pair
    B = 2 ** 18
    poses = S.make_cameras(64, seed=1).cuda()
    bits = raymarching.packbits(S.scene_density_grid(H=128, device="cuda"), 10.0)
    g = torch.Generator(device="cuda").manual_seed(4)
    xs, n = [], 0
    while n < B:
        o, d = S.random_rays(poses, 32768, g)
        nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device="cuda"), 0.05)
        xyzs = raymarching.march_rays_train(o, d, 1.0, False, bits, 1, 128, nears, fars, True, 0.0, 1024)[0]
        xs.append(xyzs); n += xyzs.shape[0]
    x_t = ((torch.cat(xs)[:B] + 1) / 2).contiguous()
    x = x_t.cpu().numpy()
    rng = np.random.default_rng(8)
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    e2 = GridEncoder(level_dim=2, desired_resolution=2048).cuda()
    offs = np.asarray(e1.host_offsets, np.int32)
    Sl = float(np.log2(e1.per_level_scale))
    rows = int(offs[-1])
    # gradient magnitudes like a training step's: a few large, most small, some exactly zero (early-stopped samples)
    d1 = (rng.standard_normal((16, B, 1)) * np.exp(rng.normal(-6, 2, (1, B, 1))) * (rng.random((1, B, 1)) < 0.8)).astype(np.float32)
    d2 = (rng.standard_normal((16, B, 2)) * np.exp(rng.normal(-6, 2, (1, B, 1))) * (rng.random((1, B, 1)) < 0.8) * 128).astype(np.float16)
    emb1 = ((rng.random((rows, 1), dtype=np.float32) * 2 - 1) * 1e-2)
    tv_w = 1e-4
    tv = (dev(be, emb1), tv_w, tv_w, 1.0, None) if with_tv else None
    a1 = torch.zeros(rows, 1, device="cuda"); a2 = torch.zeros(rows, 2, device="cuda", dtype=torch.float16)
    assert binned_backward_pair(e1, e2, dev(be, d1), dev(be, d2), x_t, a1, a2, 16, tv=tv)
    h1, h2 = a1.cpu().numpy().astype(np.float64), a2.float().cpu().numpy().astype(np.float64)

    ex1, mag1, cnt1 = oracle.grid_encode_backward_exact(d1, x, offs, Sl, 16, 1, False)
    ex2, mag2, cnt2 = oracle.grid_encode_backward_exact(d2, x, offs, Sl, 16, 2, True)
    tv_ref = np.zeros((rows, 1), np.float32)
    if with_tv:
        oracle.grad_total_variation(x, emb1, tv_ref, offs, tv_w, Sl, 16, 0, False)
        ex1 = ex1 + tv_ref.astype(np.float64)
    level_of = np.repeat(np.arange(16), np.diff(offs))
    lmax1 = np.array([np.abs(d1[l]).max() for l in range(16)], np.float64)[level_of][:, None] + (np.abs(tv_ref).max() if with_tv else 0.0)
    lmax2 = np.array([np.abs(d2[l].astype(np.float32)).max() for l in range(16)], np.float64)[level_of][:, None]
    merged = (level_of < 9)[:, None].astype(np.float64)
    # small dense levels are split over G tile groups (make_bin_plan: ceil(8 B / partitions / 65536)); each group's partial sum is rounded
    # to the table's type and ADDED with a float atomic, which rounds the running sum again: (G + 1) roundings of at most the magnitude sum
    sizes = np.diff(offs).astype(np.int64)
    parts = -(-((sizes + 15) >> 4) // (4096 // 16))
    G = np.clip(-(-(8 * B // parts) // 65536), 1, 64)
    split = np.where(G > 1, G + 1, 0).astype(np.float64)[level_of][:, None]

    bound2 = 0.5 * _half_ulp(ex2) * 1.002 + (merged + split) * 2.0 ** -11 * mag2 + cnt2 * 2.0 ** -38 * lmax2
    err2 = np.abs(h2 - ex2)
    worst2 = (err2 / np.maximum(bound2, 1e-300))[cnt2 > 0].max()
    # fp32 table.  Final conversion of the fixed-point sum: 2^-24 relative; merged runs: up to 16 fp32 adds per entry; TV: the term
    # rides on vertex 000's product (one more fp32 add per sample) and is an fp32 expression whose association differs from the
    # oracle's ((w * sum) * r vs w * (sum * r)); every sample's term is at most sqrt(6)/6 * weight (Cauchy-Schwarz on the six
    # differences), and at most `pairs` samples reach a row
    bound1 = 2.0 ** -23 * np.abs(ex1) + (merged * 16 + split) * 2.0 ** -24 * mag1 + (cnt1 + 1) * 2.0 ** -38 * lmax1
    if with_tv:
        pairs = oracle.grid_encode_backward_exact(np.ones((16, B, 1), np.float32), x, offs, Sl, 16, 1, False)[2].astype(np.float64)
        tv_each = tv_w * np.sqrt(6.0) / 6.0
        bound1 = bound1 + pairs * tv_each * (8 + 16 * merged) * 2.0 ** -24 + 2.0 ** -23 * mag1 + pairs * 2.0 ** -38 * (lmax1 + tv_each)
    err1 = np.abs(h1 - ex1)
    touched1 = (cnt1 > 0) | (np.abs(tv_ref) > 0)
    worst1 = (err1 / np.maximum(bound1, 1e-300))[touched1].max()
    per_level2 = [float((err2 / np.maximum(bound2, 1e-300))[(level_of == l)[:, None] & (cnt2 > 0)].max()) for l in range(16)]
    per_level1 = [float((err1 / np.maximum(bound1, 1e-300))[(level_of == l)[:, None] & touched1].max()) for l in range(16)]
    print(f"tv={with_tv}: fp16 table worst err/bound {worst2:.3f} ({int((cnt2 > 0).sum())} entries), fp32 table {worst1:.3f} ({int(touched1.sum())} entries)")
    print("   per level fp16:", " ".join(f"{v:.2f}" for v in per_level2))
    print("   per level fp32:", " ".join(f"{v:.2f}" for v in per_level1))
    assert worst2 <= 1.0, f"fp16 table gradient is not the exact sum of the reference's half-rounded terms (err/bound {worst2:.3f})"
    assert worst1 <= 1.0, f"fp32 table gradient off its bound (err/bound {worst1:.3f})"
    assert not h2[cnt2 == 0].any() and not h1[~touched1].any()
    # and the reference's own summation order is farther from the exact sum than we are: its result is one draw of that rounding noise
    ref2 = oracle.grid_encode_backward(d2, x, np.zeros((rows, 2), np.float16), offs, Sl, 16, 16).astype(np.float64)
    sel = cnt2 > 8
    assert np.abs(h2 - ex2)[sel].sum() < np.abs(ref2 - ex2)[sel].sum()


def test_binned_pair_backward_in_two_level_halves_equals_the_full_call(be):
    """n2m_grid_encode_backward_binned_pair_half: levels 8..15 then 0..7 == the full call (levels are independent): bit for bit on
    the single-owner (2^19-row) levels and the TV term included; the split dense levels 0..3 carry float-atomic order noise in both."""
    import ctypes
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd.gridencoder import GridEncoder, _host_offsets
    B = 70001
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand(B, 3, device="cuda", generator=g) * 2 - 1
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    ho = _host_offsets(e1)
    rows = int(ho[-1])
    d1 = torch.randn(16, B, device="cuda", generator=g)
    d2 = torch.randn(16, B, 2, device="cuda", generator=g).half()
    emb = (torch.rand(rows, 1, device="cuda", generator=g) - 0.5) * 1e-2
    scale = torch.tensor(64.0, device="cuda")
    need = L.lib().n2m_grid_binned_pair_workspace_bytes(B, 16, ho.ctypes.data)
    ws = L.workspace(torch.device("cuda"), need)
    outs = []
    for mode in ("full", "halves"):
        g1 = torch.full((rows, 1), 5.0, device="cuda"); g2 = torch.full((rows, 2), 5.0, device="cuda", dtype=torch.float16)
        finf = torch.zeros((), device="cuda")
        args = (L.ptr(d1), L.ptr(d2), L.ptr(x), ho.ctypes.data, L.ptr(g1), L.ptr(g2), B, 16, 16, float(np.log2(e1.per_level_scale)), 16, 0, 0, 0,
                L.ptr(emb), 1e-4, 1e-4, 1.0, L.ptr(scale), L.ptr(finf), 0.5, 0.5, 1, L.ptr(ws), ws.numel(), L.stream())
        if mode == "full":
            L.call("n2m_grid_encode_backward_binned_pair", *args)
        else:
            L.call("n2m_grid_encode_backward_binned_pair_half", *args, 1)
            hi = (g1[int(ho[8]):].clone(), g2[int(ho[8]):].clone())            # final after the first call
            assert bool((g1[:int(ho[8])] == 5.0).all())                         # the other half is untouched so far
            L.call("n2m_grid_encode_backward_binned_pair_half", *args, 2)
            assert torch.equal(hi[0], g1[int(ho[8]):]) and torch.equal(hi[1], g2[int(ho[8]):])
        outs.append((g1, g2))
    (a1, a2), (b1, b2) = outs
    for l in range(16):
        sl = slice(int(ho[l]), int(ho[l + 1]))
        if ho[l + 1] - ho[l] == 2 ** 19 or l >= 4:
            assert torch.equal(a1[sl], b1[sl]) and torch.equal(a2[sl], b2[sl]), f"level {l}"
        else:
            np.testing.assert_allclose(b1[sl].cpu().numpy(), a1[sl].cpu().numpy(), rtol=1e-5, atol=1e-6 * float(a1[sl].abs().max()))
            np.testing.assert_allclose(b2[sl].float().cpu().numpy(), a2[sl].float().cpu().numpy(), rtol=2e-3, atol=1e-2 * float(a2[sl].float().abs().max()))
    with pytest.raises(RuntimeError):
        L.call("n2m_grid_encode_backward_binned_pair_half", *args, 3)


def test_colour_only_pair_backward_equals_the_pair_call(be):
    """n2m_grid_encode_backward_binned_pair with grad1 = NULL (stage 1: the colour table alone through the shared-fill kernels) writes the
    SAME colour-table gradient as the full pair call on the same inputs (bit for bit wherever the pair call itself is reproducible) -- the fp32 table's entries never enter the fp16
    sums -- and leaves no trace elsewhere; gridencoder.binned_backward routes a lone C = 2 fp16 table there."""
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward, binned_backward_pair
    B = 150001
    g = torch.Generator(device="cuda").manual_seed(21)
    # image-order-like coherent samples: a slow walk through the cube, so that consecutive lanes share cells on the coarse levels
    t = torch.linspace(0, 1, B, device="cuda")
    x = torch.stack([0.5 + 0.45 * torch.sin(37 * t), 0.5 + 0.45 * torch.cos(23 * t), 0.05 + 0.9 * t], -1).contiguous()
    x = (x + 1e-3 * torch.rand(B, 3, device="cuda", generator=g)).clamp(0, 1).contiguous()
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    e2 = GridEncoder(level_dim=2, desired_resolution=2048).cuda()
    rows = e1.embeddings.shape[0]
    d1 = torch.randn(16, B, 1, device="cuda", generator=g) * 1e-3
    d2 = (torch.randn(16, B, 2, device="cuda", generator=g) * 0.05).half()
    a1 = torch.zeros(rows, 1, device="cuda"); a2 = torch.zeros(rows, 2, device="cuda", dtype=torch.float16)
    assert binned_backward_pair(e1, e2, d1, d2, x, a1, a2, 16)
    b2 = torch.zeros(rows, 2, device="cuda", dtype=torch.float16)
    assert binned_backward(e2, d2, x, b2, 16)                      # -> the pair entry with grad1 = NULL
    # levels whose partitions one work item owns (all hashed levels) are bit-reproducible; the small dense levels that are split over tile
    # groups end in fp16 float atomics whose order differs from launch to launch (two identical calls differ there as well)
    first_hashed = int(np.asarray(e1.host_offsets)[8])
    assert torch.equal(a2[first_hashed:], b2[first_hashed:])
    d = (a2[:first_hashed].float() - b2[:first_hashed].float()).abs()
    assert float(d.max()) <= 2.0 ** -9 * float(a2[:first_hashed].float().abs().max())
    assert float(b2.float().abs().sum()) > 0
    # added onto what is there (not overwritten), like the single-table path
    assert binned_backward(e2, d2, x, b2, 16)
    ref = (a2.float() * 2)
    assert float((b2.float() - ref).abs().max()) <= 2.0 ** -10 * float(ref.abs().max())


@pytest.mark.parametrize("max_level", [16, 5])
def test_density_only_pair_backward_equals_the_pair_call(be, max_level):
    """grad2 = NULL: the density table alone through the shared-fill kernels (the SDF head's stacked finite-difference samples; progressive
    max_level < 16 included), against the full pair call: equal up to fp32 rounding of merged runs / atomic order."""
    torch = be["torch"]
    from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward, binned_backward_pair
    B = 120007
    g = torch.Generator(device="cuda").manual_seed(22)
    t = torch.linspace(0, 1, B, device="cuda")
    x = torch.stack([0.5 + 0.45 * torch.sin(31 * t), 0.5 + 0.45 * torch.cos(19 * t), 0.05 + 0.9 * t], -1)
    x = (x + 1e-3 * torch.rand(B, 3, device="cuda", generator=g)).clamp(0, 1).contiguous()
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    e2 = GridEncoder(level_dim=2, desired_resolution=2048).cuda()
    rows = e1.embeddings.shape[0]
    d1 = torch.randn(16, B, 1, device="cuda", generator=g) * 1e-3
    d2 = (torch.randn(16, B, 2, device="cuda", generator=g) * 0.05).half()
    a1 = torch.zeros(rows, 1, device="cuda"); a2 = torch.zeros(rows, 2, device="cuda", dtype=torch.float16)
    assert binned_backward_pair(e1, e2, d1, d2, x, a1, a2, max_level)
    b1 = torch.zeros(rows, 1, device="cuda")
    assert binned_backward(e1, d1, x, b1, max_level)               # -> the pair entry with grad2 = NULL
    offs = np.asarray(e1.host_offsets)
    hi = int(offs[max_level])
    assert float(b1[hi:].abs().max() if hi < rows else 0.0) == 0.0          # levels beyond max_level are not touched
    # the lone density table merges same-cell runs on ALL levels (its callers are the SDF head's adjacent finite-difference copies), the pair
    # call on levels 0..8: one more fp32 rounding per merged run on the fine levels, atomic-order noise on the small split dense levels
    d = (a1[:hi] - b1[:hi]).abs()
    assert float(d.max()) <= 4e-6 * float(a1[:hi].abs().max())
    assert float(b1.abs().sum()) > 0


def test_precomputed_tv_terms_equal_the_in_place_stencil(be):
    """n2m_grid_tv_terms + n2m_grid_encode_backward_binned_pair_tvt (the TV terms of the batch evaluated ahead of the backward, on another
    stream in the step executor) against the pair call that gathers the stencil inside its fill: one device function computes the term,
    so the density-table gradient is the same bit for bit on every level a single work item owns (fp32 atomic-order noise on the small split
    dense levels, as between two identical calls); with the inner / outer weighting of bound > 1 and a loss scale on the device."""
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd.gridencoder import GridEncoder, _host_offsets
    p = L.ptr
    B = 100003
    g = torch.Generator(device="cuda").manual_seed(23)
    t = torch.linspace(0, 1, B, device="cuda")
    x = torch.stack([0.5 + 0.49 * torch.sin(29 * t), 0.5 + 0.49 * torch.cos(17 * t), 0.01 + 0.98 * t], -1)
    x = (x + 1e-3 * torch.rand(B, 3, device="cuda", generator=g)).clamp(0, 1).contiguous()
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    with torch.no_grad():
        e1.embeddings.uniform_(-1e-2, 1e-2)
    rows = e1.embeddings.shape[0]
    ho = _host_offsets(e1)
    S, H0 = float(np.log2(e1.per_level_scale)), int(e1.base_resolution)
    d1 = torch.randn(16, B, device="cuda", generator=g) * 1e-3
    d2 = (torch.randn(16, B, 2, device="cuda", generator=g) * 0.05).half()
    scale = torch.tensor(512.0, device="cuda")
    lam, lam_out, inner = 1e-4, 1e-3, 0.25
    emb = e1.embeddings.detach().contiguous()
    need = L.lib().n2m_grid_binned_pair_workspace_bytes(B, 16, ho.ctypes.data)
    ws = L.workspace(x.device, need)
    L.grid_backward_config(1, 1.0)
    outs = []
    for split in (False, True):
        g1 = torch.empty(rows, 1, device="cuda"); g2 = torch.empty(rows, 2, device="cuda", dtype=torch.float16)
        finf = torch.zeros((), device="cuda")
        common = (p(d1), p(d2), p(x), ho.ctypes.data, p(g1), p(g2), B, 16, 16, S, H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id)
        tail = (p(finf), 1.0, 0.0, 1, p(ws), ws.numel(), L.stream())
        if split:
            terms = torch.empty(16, B, device="cuda")
            L.call("n2m_grid_tv_terms", p(x), p(emb), ho.ctypes.data, B, 16, S, H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id, lam,
                   lam_out, inner, p(scale), 1.0, 0.0, p(terms), L.stream())
            assert float(terms.abs().max()) > 0
            L.call("n2m_grid_encode_backward_binned_pair_tvt", *common, p(terms), *tail, 0)
        else:
            L.call("n2m_grid_encode_backward_binned_pair", *common, p(emb), lam, lam_out, inner, p(scale), *tail)
        torch.cuda.synchronize()
        outs.append((g1, g2))
    (a1, a2), (b1, b2) = outs
    lo = int(np.asarray(e1.host_offsets)[8])
    assert torch.equal(a1[lo:], b1[lo:]) and torch.equal(a2[lo:], b2[lo:])
    assert float((a1[:lo] - b1[:lo]).abs().max()) <= 2e-6 * float(a1[:lo].abs().max())
    # and the halves of the multi-rank step equal the full call
    g1 = torch.empty(rows, 1, device="cuda"); g2 = torch.empty(rows, 2, device="cuda", dtype=torch.float16)
    common = (p(d1), p(d2), p(x), ho.ctypes.data, p(g1), p(g2), B, 16, 16, S, H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id)
    for half in (1, 2):
        L.call("n2m_grid_encode_backward_binned_pair_tvt", *common, p(terms), p(finf), 1.0, 0.0, 1, p(ws), ws.numel(), L.stream(), half)
    torch.cuda.synchronize()
    assert torch.equal(g1[lo:], b1[lo:]) and torch.equal(g2[lo:], b2[lo:])


@pytest.mark.parametrize("eps,max_level,with_tv", [(1e-4, 16, True), (1e-4, 16, False), (2e-3, 16, True), (1e-4, 9, False)])
def test_sdf_copies_folded_into_the_batch_backward_equal_the_stacked_pass(be, eps, max_level, with_tv):
    """SDF recipe: the table backward of the six finite-difference copies (nerf/network.py:143-154).  Stacked form: the batch through the
    pair call, then the 6 M copies (n2m_sdf_offsets' points) through the density-only call.  Folded form: n2m_sdf_fold_plan marks the
    copies that share their centre's cell on every level, n2m_grid_encode_backward_binned_pair_fold adds those to the centre's entries,
    the others go through the density-only call as the compact list n2m_sdf_fold_gather fills.  Same sums up to fp32 association;
    the colour table does not see the copies at all.  Flags and list are checked against a torch statement of the cell test."""
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd.gridencoder import GridEncoder, _host_offsets
    p = L.ptr
    M = 60011
    g = torch.Generator(device="cuda").manual_seed(31)
    t = torch.linspace(0, 1, M, device="cuda")
    xyz = torch.stack([0.9 * torch.sin(23 * t), 0.9 * torch.cos(13 * t), -0.95 + 1.9 * t], -1)
    xyz = (xyz + 1e-3 * torch.rand(M, 3, device="cuda", generator=g)).clamp(-1, 1).contiguous()
    xyz[:7] = torch.tensor([1.0, -1.0, 0.3], device="cuda")                     # on the faces of the cube: copies are clamped
    bound = 1.0
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    with torch.no_grad():
        e1.embeddings.uniform_(-1e-2, 1e-2)
    rows = e1.embeddings.shape[0]
    ho = _host_offsets(e1)
    S, H0 = float(np.log2(e1.per_level_scale)), int(e1.base_resolution)
    geo = (16, max_level, S, H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id)
    d1 = torch.randn(16, M, device="cuda", generator=g) * 1e-3
    d2 = (torch.randn(16, M, 2, device="cuda", generator=g) * 0.05).half()
    d6 = torch.randn(16, 6 * M, device="cuda", generator=g) * 1e-3
    pts = torch.empty(M, 6, 3, device="cuda"); pts01 = torch.empty(M, 6, 3, device="cuda")
    L.call("n2m_sdf_offsets", p(xyz), M, eps, bound, p(pts), p(pts01), L.stream())
    scale_t = torch.tensor(128.0, device="cuda")
    emb = e1.embeddings.detach().contiguous()
    tv = (p(emb), 1e-4, 1e-4, 0.5, p(scale_t)) if with_tv else (None, 0.0, 0.0, 1.0, None)
    need = L.lib().n2m_grid_binned_pair_workspace_bytes(6 * M, 16, ho.ctypes.data)
    ws = L.workspace(xyz.device, need)
    L.grid_backward_config(1, 1.0)
    finf = torch.zeros((), device="cuda")

    def batch_args(g1, g2):
        return (p(d1), p(d2), p(xyz), ho.ctypes.data, p(g1), p(g2), M, *geo, *tv, p(finf), 0.5, 0.5, 1, p(ws), ws.numel())

    def copies(grad, points, n, g1):
        L.call("n2m_grid_encode_backward_binned_pair", p(grad), None, p(points), ho.ctypes.data, p(g1), None, n, *geo, None, 0.0, 0.0, 1.0, None,
               p(finf), 1.0, 0.0, 0, p(ws), ws.numel(), L.stream())

    a1 = torch.empty(rows, 1, device="cuda"); a2 = torch.empty(rows, 2, device="cuda", dtype=torch.float16)
    L.call("n2m_grid_encode_backward_binned_pair", *batch_args(a1, a2), L.stream())
    only_batch = a1.clone()
    copies(d6, pts01, 6 * M, a1)
    # ---- folded
    cap = 6 * M
    flags = torch.full((16, M), 0x3F, dtype=torch.uint8, device="cuda")       # garbage: the plan defines the flags of ALL 16 levels
    left_pts = torch.empty(16, cap, 3, device="cuda"); left_src = torch.empty(16, cap, dtype=torch.int32, device="cuda")
    left_g = torch.empty(16 * cap, device="cuda")
    cnt = torch.zeros(2, 32, dtype=torch.int32, device="cuda")
    cnt[1] = 12345
    L.call("n2m_sdf_fold_plan", p(xyz), M, eps, bound, 16, max_level, S, H0, int(bool(e1.align_corners)), p(flags), p(left_pts), p(left_src), cap, p(cnt), 0,
           L.stream())
    Ks = cnt[0, :max_level].cpu().numpy()
    assert int(cnt[1].abs().sum()) == 0, "the other parity's counters are cleared"
    # levels above the active ones fold nothing: a fold call over all 16 levels (progressive levels + TV on every level, engine._step_sdf) reads
    # their flags, and a garbage flag would add the copies' gradients onto rows the reference leaves to the TV term alone
    assert int(flags[max_level:].max() if max_level < 16 else 0) == 0
    # the cell test, in torch, with the library's level scales (exp2f(level * S) * H - 1 in fp32): copy c folds on level l iff
    # floor(p01 * scale + 0.5) of its moved axis equals the centre's
    c01 = (xyz + bound) / (2 * bound)
    folded_pairs = 0
    for l in range(max_level):
        scale = float(np.float32(np.exp2(np.float32(l) * np.float32(S)) * np.float32(H0)) - np.float32(1.0))
        cc = torch.floor(c01 * scale + 0.5)
        same = torch.stack([torch.floor(pts01[:, c, c >> 1] * scale + 0.5) == cc[:, c >> 1] for c in range(6)], 1)      # [M,6]
        want = (same.int() << torch.arange(6, device="cuda")).sum(1)
        diff = (flags[l].int() != want)
        assert float(diff.float().mean()) <= 1e-3, f"level {l}: flags differ from the torch statement on {int(diff.sum())} samples"      # (the scale may differ by an ulp of exp2f: cells right on a boundary)
        fl = flags[l].int()
        K = int(Ks[l])
        assert K == 6 * M - int(sum(((fl >> c) & 1).sum() for c in range(6)))
        folded_pairs += 6 * M - K
        src = left_src[l, :K].long()
        listed = torch.nonzero(((fl.unsqueeze(1) >> torch.arange(6, device="cuda")) & 1).reshape(-1) == 0).reshape(-1)
        assert torch.equal(torch.sort(src)[0], listed)
        assert torch.equal(left_pts[l, :K], pts01.view(-1, 3)[src]), "the listed copies carry n2m_sdf_offsets' own coordinates"
    frac = folded_pairs / (6 * M * max_level)
    if eps == 1e-4:
        assert frac > 0.9, "nearly all (copy, level) pairs of a 1e-4 offset share their centre's cell"
    B = int(Ks.max())
    b1 = torch.empty(rows, 1, device="cuda"); b2 = torch.empty(rows, 2, device="cuda", dtype=torch.float16)
    L.call("n2m_grid_encode_backward_binned_pair_fold", *batch_args(b1, b2), p(flags), p(d6), eps, bound, L.stream())
    if B > 0:
        L.call("n2m_sdf_fold_gather", p(d6), M, max_level, p(left_src), p(left_pts), cap, p(cnt), B, p(left_g), L.stream())
        for l in range(max_level):
            K = int(Ks[l])
            assert torch.equal(left_g[l * B:l * B + K], d6[l, left_src[l, :K].long()])
            assert float(left_g[l * B + K:(l + 1) * B].abs().max() if K < B else 0.0) == 0.0 and bool((left_pts[l, K:B] == 2.0).all())
        L.call("n2m_grid_encode_backward_binned_lists", p(left_g), p(left_pts), cap, ho.ctypes.data, p(b1), B, *geo, p(finf), 0, p(ws), ws.numel(),
               L.stream())
    torch.cuda.synchronize()
    K = 6 * M * max_level - folded_pairs
    hi = int(np.asarray(e1.host_offsets)[max_level])
    lo = int(np.asarray(e1.host_offsets)[min(8, max_level)])
    assert torch.equal(a2[lo:hi], b2[lo:hi]), "colour table: untouched by the copies (levels one work item owns: bit-equal)"
    d = (a1[:hi] - b1[:hi]).abs()
    ref = float(a1[:hi].abs().max())
    print(f"eps {eps} max_level {max_level}: folded {folded_pairs} of {6 * M * max_level} (copy, level) pairs; max |diff| {float(d.max()):.3g} of {ref:.3g}")
    offs_ = np.asarray(e1.host_offsets)
    for l in range(max_level):
        dl = d[int(offs_[l]):int(offs_[l + 1])]
        print(f"   level {l}: max |diff| {float(dl.max()):.3g}  (max |a| {float(a1[int(offs_[l]):int(offs_[l + 1])].abs().max()):.3g})")
    assert float(d.max()) <= 4e-6 * ref
    assert float((a1[:hi] - only_batch[:hi]).abs().max()) > 100 * float(d.max()), "the copies' contribution is far above the tolerance"
    assert float(finf) == 0.0


@pytest.mark.parametrize("B", [262144, 600000])
@pytest.mark.parametrize("max_level", [5, 9, 12])
def test_binned_pair_backward_with_a_level_cap_equals_the_scatter_backward(be, B, max_level):
    """Fewer than 16 active levels (the SDF recipe's progressive schedule, nerf/utils.py:651-655) run the shared fill on its plain
    (tile group, level) grid.  On that grid every level's first workgroup used to clear the level maxima again -- levels then reached the
    accumulate kernels with a maximum of zero (skipped) or too small a fixed-point unit: wrong sums on the coarse levels at training batch
    sizes, found in round 3.  Both forms of the call (density table alone, adding; both tables, overwriting) against the scatter backward
    n2m_grid_encode_backward, which shares no code with the binned path."""
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd.gridencoder import GridEncoder, _host_offsets
    p = L.ptr
    g = torch.Generator(device="cuda").manual_seed(B + max_level)
    t = torch.linspace(0, 1, B, device="cuda")
    x = torch.stack([0.5 + 0.45 * torch.sin(31 * t), 0.5 + 0.45 * torch.cos(19 * t), 0.05 + 0.9 * t], -1)
    x = (x + 1e-3 * torch.rand(B, 3, device="cuda", generator=g)).clamp(0, 1).contiguous()
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    rows = e1.embeddings.shape[0]
    ho = _host_offsets(e1)
    offs = np.asarray(e1.host_offsets)
    S, H0 = float(np.log2(e1.per_level_scale)), int(e1.base_resolution)
    emb = e1.embeddings.detach().contiguous()
    d1 = torch.randn(16, B, device="cuda", generator=g) * 1e-3
    d2 = (torch.randn(16, B, 2, device="cuda", generator=g) * 0.05).half()
    need = L.lib().n2m_grid_binned_pair_workspace_bytes(B, 16, ho.ctypes.data)
    ws = L.workspace(x.device, need)
    L.grid_backward_config(1, 1.0)
    finf = torch.zeros((), device="cuda")
    geo = (16, max_level, S, H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id)
    ref = torch.zeros(rows, 1, device="cuda")
    L.call("n2m_grid_encode_backward", p(d1), p(x), p(emb), p(e1.offsets), p(ref), B, 3, 1, 16, max_level, S, H0, None, None, e1.gridtype_id,
           int(bool(e1.align_corners)), e1.interp_id, L.F32, L.stream())
    hi = int(offs[max_level])
    tol = 1e-5 * float(ref.abs().max())
    for rep in range(3):                                                # the failure depended on which workgroup started when
        lone = torch.zeros(rows, 1, device="cuda")
        L.call("n2m_grid_encode_backward_binned_pair", p(d1), None, p(x), ho.ctypes.data, p(lone), None, B, *geo, None, 0.0, 0.0, 1.0, None, p(finf),
               1.0, 0.0, 0, p(ws), ws.numel(), L.stream())
        g1 = torch.full((rows, 1), 7.0, device="cuda"); g2 = torch.full((rows, 2), 7.0, device="cuda", dtype=torch.float16)
        L.call("n2m_grid_encode_backward_binned_pair", p(d1), p(d2), p(x), ho.ctypes.data, p(g1), p(g2), B, *geo, None, 0.0, 0.0, 1.0, None, p(finf),
               1.0, 0.0, 1, p(ws), ws.numel(), L.stream())
        torch.cuda.synchronize()
        for name, got in (("density table alone", lone), ("both tables", g1)):
            for l in range(max_level):
                d = float((got[int(offs[l]):int(offs[l + 1])] - ref[int(offs[l]):int(offs[l + 1])]).abs().max())
                assert d <= tol, f"{name}, level {l}: max |diff| {d:.3g} against the scatter backward (values up to {float(ref.abs().max()):.3g})"
        assert float(g1[hi:].abs().max()) == 0.0 and float(g2[hi:].float().abs().max()) == 0.0, "overwrite mode defines the untouched levels as zero"
        assert float(g2[:hi].float().abs().max()) > 0


@pytest.mark.parametrize("hot_fraction", [1.0, 0.5])
def test_binned_pair_backward_with_a_hot_cell_takes_the_overflow_log(be, hot_fraction):
    """The partition-major update log gives every (level, partition) a region of twice its expected share; a batch whose samples pile up
    in one cell (here: 2^17 samples, all or half of them inside a cube of 1e-5) overruns the regions of the eight rows' partitions on every
    level, and the rest goes through the shared overflow log.  The sums stay exact: fp32 table against the scatter backward
    n2m_grid_encode_backward (which shares no code with the binned path), fp16 table against the same scatter run in fp32 (each product is
    rounded to half first, like the reference's: 2^-11 relative)."""
    torch = be["torch"]
    from nerf2mesh_amd import _lib as L
    from nerf2mesh_amd.gridencoder import GridEncoder, _host_offsets
    p = L.ptr
    B = 2 ** 17
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(B, 3, device="cuda", generator=g)
    n_hot = int(B * hot_fraction)
    x[:n_hot] = torch.tensor([0.3712, 0.6161, 0.4443], device="cuda") + 1e-5 * torch.rand(n_hot, 3, device="cuda", generator=g)
    x = x[torch.randperm(B, device="cuda", generator=g)].contiguous()
    e1 = GridEncoder(level_dim=1, desired_resolution=2048).cuda()
    e2 = GridEncoder(level_dim=2, desired_resolution=2048).cuda()
    rows = e1.embeddings.shape[0]
    ho = _host_offsets(e1)
    S, H0 = float(np.log2(e1.per_level_scale)), int(e1.base_resolution)
    d1 = torch.randn(16, B, device="cuda", generator=g) * 1e-3
    d2f = torch.randn(16, B, 2, device="cuda", generator=g) * 0.01
    d2 = d2f.half()
    need = L.lib().n2m_grid_binned_pair_workspace_bytes(B, 16, ho.ctypes.data)
    ws = L.workspace(x.device, need)
    L.grid_backward_config(1, 1.0)
    finf = torch.zeros((), device="cuda")
    geo = (16, 16, S, H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id)
    ref1 = torch.zeros(rows, 1, device="cuda"); ref2 = torch.zeros(rows, 2, device="cuda")
    L.call("n2m_grid_encode_backward", p(d1), p(x), p(e1.embeddings.detach()), p(e1.offsets), p(ref1), B, 3, 1, 16, 16, S, H0, None, None, e1.gridtype_id,
           int(bool(e1.align_corners)), e1.interp_id, L.F32, L.stream())
    L.call("n2m_grid_encode_backward", p(d2.float().contiguous()), p(x), p(e2.embeddings.detach().float().contiguous()), p(e2.offsets), p(ref2), B, 3, 2, 16, 16,
           S, H0, None, None, e2.gridtype_id, int(bool(e2.align_corners)), e2.interp_id, L.F32, L.stream())
    for overwrite in (0, 1):
        g1 = torch.full((rows, 1), 7.0 * overwrite, device="cuda"); g2 = torch.full((rows, 2), 7.0 * overwrite, device="cuda", dtype=torch.float16)
        L.call("n2m_grid_encode_backward_binned_pair", p(d1), p(d2), p(x), ho.ctypes.data, p(g1), p(g2), B, *geo, None, 0.0, 0.0, 1.0, None, p(finf),
               1.0, 0.0, overwrite, p(ws), ws.numel(), L.stream())
        torch.cuda.synchronize()
        assert float(finf) == 0.0
        # a hot row sums ~1e5 terms: the scatter adds them in fp32 arrival order (relative error ~1e-4 of the row), the binned sum is exact
        np.testing.assert_allclose(g1.cpu().numpy(), ref1.cpu().numpy(), rtol=2e-3, atol=2e-5 * float(ref1.abs().max()))
        np.testing.assert_allclose(g2.float().cpu().numpy(), ref2.cpu().numpy(), rtol=4e-3, atol=2e-3 * float(ref2.abs().max()))
