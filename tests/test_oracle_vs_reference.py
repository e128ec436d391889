"""Pins oracle/n2m_oracle.c against the reference's OWN kernels compiled for the host (oracle/_ref).

The reference ships no tests or golden vectors, so the strongest pin available is its
kernel source executed serially on the CPU (oracle/build_ref.py).  Integer outputs and every fp32 path must
agree BIT FOR BIT; the fp16 grid path too (same rounding points); SH agrees to fp32 rounding (different
association order, see n2m_oracle.c).  Runs wherever oracle/_ref exists (this container; the GPU box gets the
prebuilt .so files); the same comparisons against committed fixtures live in test_golden.py.
"""
import numpy as np
import pytest

from conftest import lego_offsets


def T(ref_mods):
    import torch
    return torch


def t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))


def make_rays(scene, n, seed=0):
    torch, S = scene["torch"], scene["S"]
    g = torch.Generator().manual_seed(seed)
    o, d = S.random_rays(scene["poses"], n, g)
    return o.numpy(), d.numpy()


def test_morton_known_answers(oracle, ref):
    rm = ref[0]
    # known answers derivable from the reference text
    c = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [127, 127, 127], [1023, 1023, 1023]], np.int32)
    got = oracle.morton3D(c)
    assert got.tolist() == [1, 2, 4, 2 ** 21 - 1, 2 ** 30 - 1]
    rng = np.random.default_rng(0)
    c = rng.integers(0, 1024, (5000, 3)).astype(np.int32)
    out = t(np.zeros(5000, np.int32))
    rm.morton3D(t(c), 5000, out)
    assert np.array_equal(out.numpy(), oracle.morton3D(c))
    back = t(np.zeros((5000, 3), np.int32))
    rm.morton3D_invert(out, 5000, back)
    assert np.array_equal(back.numpy(), c)
    assert np.array_equal(oracle.morton3D_invert(out.numpy()), c)


def test_packbits_flatten(oracle, ref):
    rm = ref[0]
    rng = np.random.default_rng(1)
    grid = rng.normal(size=(2, 4096)).astype(np.float32)
    grid[0, :8] = [0.5, 0.5000001, 0.4999999, -1, np.nan, np.inf, 0.5, 1]
    bf = t(np.zeros(1024, np.uint8))
    rm.packbits(t(grid), 1024, 0.5, bf)
    assert np.array_equal(bf.numpy(), oracle.packbits(grid, 0.5))
    rays = np.array([[0, 3], [3, 0], [3, 5], [8, 1]], np.int32)
    res = t(np.zeros(9, np.int32))
    rm.flatten_rays(t(rays), 4, 9, res)
    assert np.array_equal(res.numpy(), oracle.flatten_rays(rays, 9))


def test_near_far(oracle, ref, scene):
    rm = ref[0]
    o, d = make_rays(scene, 3000)
    d[:50, 0] = 0.0           # axis-parallel rays: 1/0 = inf paths
    d[50:60] = 0.0            # degenerate direction -> NaN paths
    o[60:80] *= 0.1           # origins inside the box
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    for box, mn in ((aabb, 0.05), (aabb * np.array([0.5, 0.7, 0.2, 0.5, 0.7, 0.2], np.float32), 0.2)):
        nears, fars = t(np.zeros(3000, np.float32)), t(np.zeros(3000, np.float32))
        rm.near_far_from_aabb(t(o), t(d), t(box), 3000, mn, nears, fars)
        on, of = oracle.near_far_from_aabb(o, d, box, mn)
        assert np.array_equal(nears.numpy().view(np.uint32), on.view(np.uint32))
        assert np.array_equal(fars.numpy().view(np.uint32), of.view(np.uint32))


def test_sph_from_ray(oracle, ref, scene):
    rm = ref[0]
    o, d = make_rays(scene, 500)
    c = t(np.zeros((500, 2), np.float32))
    rm.sph_from_ray(t(o), t(d), 5.0, 500, c)
    np.testing.assert_allclose(oracle.sph_from_ray(o, d, 5.0), c.numpy(), rtol=0, atol=2e-6)


def _ref_march_train(rm, o, d, bits, bound, contract, dt_gamma, max_steps, C, H, nears, fars, noises):
    N = o.shape[0]
    rays = t(np.zeros((N, 2), np.int32))
    counter = t(np.zeros(1, np.int32))
    args = (t(o), t(d), t(bits), bound, contract, dt_gamma, max_steps, N, C, H, t(nears), t(fars))
    rm.march_rays_train(*args, None, None, None, rays, counter, t(noises))
    M = int(counter[0])
    xyzs, dirs, ts = t(np.zeros((M, 3), np.float32)), t(np.zeros((M, 3), np.float32)), t(np.zeros((M, 2), np.float32))
    rm.march_rays_train(*args, xyzs, dirs, ts, rays, counter, t(noises))
    return xyzs.numpy(), dirs.numpy(), ts.numpy(), rays.numpy()


@pytest.mark.parametrize("cfg", [
    dict(bound=1.0, contract=False, dt_gamma=0.0, C=1),              # lego recipe (scripts/runall_syn.sh:1)
    dict(bound=1.0, contract=False, dt_gamma=1 / 256, C=1),          # default dt_gamma (main.py:56)
    dict(bound=4.0, contract=False, dt_gamma=1 / 256, C=3),          # cascades
    dict(bound=4.0, contract=True, dt_gamma=0.0, C=2),               # contraction (grid bound 2)
    dict(bound=1.0, contract=False, dt_gamma=0.0, C=1, max_steps=64, dense=True),  # max_steps cap
])
def test_march_rays_train(oracle, ref, scene, cfg):
    rm = ref[0]
    torch, S = scene["torch"], scene["S"]
    C, H = cfg["C"], 128 if cfg["C"] == 1 else 64
    rng = np.random.default_rng(5)
    if C == 1 and H == 128 and not cfg.get("dense"):
        bits = scene["bits"]
    elif cfg.get("dense"):
        bits = np.full(C * H ** 3 // 8, 255, np.uint8)
    else:
        grid = S.scene_density_grid(H=H, cascade=C, bound=cfg["bound"] if not cfg["contract"] else 2.0).numpy()
        grid += (rng.random(grid.shape) < 0.02).astype(np.float32) * 50   # sprinkle far cells
        bits = oracle.packbits(grid, 10.0)
    N = 1500
    o, d = make_rays(scene, N, seed=9)
    if cfg["bound"] > 1:
        o = o * 1.1
    b = cfg["bound"]
    aabb = np.array([-b, -b, -b, b, b, b], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.05)
    noises = rng.random(N).astype(np.float32)
    ms = cfg.get("max_steps", 1024)
    rx, rd, rt, rr = _ref_march_train(rm, o, d, bits, b, cfg["contract"], cfg["dt_gamma"], ms, C, H, nears, fars, noises)
    ox, od, ot, orr = oracle.march_rays_train(o, d, b, cfg["contract"], bits, C, H, nears, fars, noises, cfg["dt_gamma"], ms)
    assert rr[:, 1].sum() > 1000, "scene produced no samples: test is vacuous"
    assert np.array_equal(rr, orr)                                   # bit-exact ray/sample indexing
    for a, bb in ((rx, ox), (rd, od), (rt, ot)):
        assert np.array_equal(a.view(np.uint32), bb.view(np.uint32))


def test_march_composite_inference(oracle, ref, scene):
    rm = ref[0]
    N = 2000
    o, d = make_rays(scene, N, seed=11)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.05)
    rng = np.random.default_rng(2)
    alive = np.arange(N, dtype=np.int32)
    rays_t_r, rays_t_o = nears.copy(), nears.copy()
    ws_r, dp_r, im_r = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    ws_o, dp_o, im_o = ws_r.copy(), dp_r.copy(), im_r.copy()
    alive_r, alive_o = alive.copy(), alive.copy()
    total = 0
    for it in range(6):
        n_alive = alive_r.shape[0]
        if n_alive == 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        noises = np.zeros(n_alive, np.float32)
        M = n_alive * n_step
        xyzs, dirs, ts = t(np.zeros((M, 3), np.float32)), t(np.zeros((M, 3), np.float32)), t(np.zeros((M, 2), np.float32))
        rm.march_rays(n_alive, n_step, t(alive_r), t(rays_t_r), t(o), t(d), 1.0, False, 0.0, 1024, 1, 128, t(scene["bits"]),
                      t(nears), t(fars), xyzs, dirs, ts, t(noises))
        ox, od, ot = oracle.march_rays(n_alive, n_step, alive_o, rays_t_o, o, d, 1.0, False, scene["bits"], 1, 128, nears, fars, noises)
        assert np.array_equal(xyzs.numpy().view(np.uint32), ox.view(np.uint32))
        assert np.array_equal(ts.numpy().view(np.uint32), ot.view(np.uint32))
        assert np.array_equal(dirs.numpy().view(np.uint32), od.view(np.uint32))
        total += int((ot[:, 0] > 0).sum())
        sig = (rng.random(M) * 40).astype(np.float32)
        rgb = rng.random((M, 3)).astype(np.float32)
        ar, tr = t(alive_r), t(rays_t_r)
        wr, dr_, ir = t(ws_r), t(dp_r), t(im_r)
        rm.composite_rays(n_alive, n_step, 1e-2, False, ar, tr, t(sig), t(rgb), ts, wr, dr_, ir)
        oracle.composite_rays(n_alive, n_step, alive_o, rays_t_o, sig, rgb, ot, ws_o, dp_o, im_o, 1e-2, False)
        assert np.array_equal(ar.numpy(), alive_o)
        # rays_t of rays that ended is left as-is by both; compare all
        assert np.array_equal(tr.numpy().view(np.uint32), rays_t_o.view(np.uint32))
        for a, bb in ((wr, ws_o), (dr_, dp_o), (ir, im_o)):
            assert np.array_equal(a.numpy().view(np.uint32), bb.view(np.uint32))
        alive_r = ar.numpy()[ar.numpy() >= 0].copy()
        alive_o = oracle.compact_alive(alive_o).copy()
        assert np.array_equal(alive_r, alive_o)
        rays_t_r, ws_r, dp_r, im_r = tr.numpy().copy(), wr.numpy().copy(), dr_.numpy().copy(), ir.numpy().copy()
    assert total > 500


@pytest.mark.parametrize("alpha_mode", [False, True])
def test_composite_train(oracle, ref, alpha_mode):
    rm = ref[0]
    rng = np.random.default_rng(3)
    N = 300
    counts = rng.integers(0, 40, N).astype(np.int32)
    counts[:5] = 0
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    rays = np.stack([offs, counts], 1).astype(np.int32)
    M = int(counts.sum())
    rays[7] = [M - 2, 10]                                            # overflowing ray: guard path (:521)
    sig = (rng.random(M) * (1.0 if alpha_mode else 60.0)).astype(np.float32)
    if alpha_mode:
        sig = np.clip(sig, 0, 0.98)
    rgb = rng.random((M, 3)).astype(np.float32)
    ts = np.stack([np.cumsum(rng.random(M)).astype(np.float32) * 0.01 + 2, np.full(M, 0.0034, np.float32)], 1).astype(np.float32)
    w, ws, dp, im = t(np.zeros(M, np.float32)), t(np.zeros(N, np.float32)), t(np.zeros(N, np.float32)), t(np.zeros((N, 3), np.float32))
    rm.composite_rays_train_forward(t(sig), t(rgb), t(ts), t(rays), M, N, 1e-4, alpha_mode, w, ws, dp, im)
    ow, ows, odp, oim = oracle.composite_rays_train_forward(sig, rgb, ts, rays, 1e-4, alpha_mode)
    for a, bb in ((w, ow), (ws, ows), (dp, odp), (im, oim)):
        assert np.array_equal(a.numpy().view(np.uint32), bb.view(np.uint32))
    gw, gws, gd, gi = (rng.normal(size=M).astype(np.float32), rng.normal(size=N).astype(np.float32),
                       rng.normal(size=N).astype(np.float32), rng.normal(size=(N, 3)).astype(np.float32))
    gs, gr = t(np.zeros(M, np.float32)), t(np.zeros((M, 3), np.float32))
    rm.composite_rays_train_backward(t(gw), t(gws), t(gd), t(gi), t(sig), t(rgb), t(ts), t(rays), ws, dp, im, M, N, 1e-4,
                                     alpha_mode, gs, gr)
    ogs, ogr = oracle.composite_rays_train_backward(gw, gws, gd, gi, sig, rgb, ts, rays, ows, odp, oim, 1e-4, alpha_mode)
    assert np.array_equal(gs.numpy().view(np.uint32), ogs.view(np.uint32))
    assert np.array_equal(gr.numpy().view(np.uint32), ogr.view(np.uint32))


def test_level_geometry_known_answers(oracle):
    # derived from gridencoder/grid.py:107-134 + gridencoder.cu:138-139
    offs, S = lego_offsets(1.0)
    assert int(offs[-1]) == 6119864
    assert np.diff(offs).tolist() == [4920, 13824, 32768, 85184, 216000] + [524288] * 11
    scales, res = oracle.level_geometry(offs, S, 16)
    assert res.tolist() == [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]
    offs16, S16 = lego_offsets(16.0)
    assert int(offs16[-1]) == 6837544
    assert np.diff(offs16).tolist() == [4920, 21952, 97336, 421880] + [524288] * 12


@pytest.mark.parametrize("D,C,half,gridtype,align,interp", [
    (3, 1, False, 0, False, 0),     # density encoder (nerf/network.py:66)
    (3, 2, True, 0, False, 0),      # colour encoder under autocast (grid.py:45)
    (3, 2, False, 0, False, 0),
    (3, 4, True, 1, False, 1),      # tiled + smoothstep
    (2, 8, False, 0, True, 0),      # 2-D, align_corners
    (3, 2, True, 0, False, 1),
    (4, 2, False, 0, False, 0),
])
def test_grid_encode(oracle, ref, D, C, half, gridtype, align, interp):
    ge = ref[1]
    rng = np.random.default_rng(4)
    L, H = (16, 16) if D == 3 else (8, 8)
    pls = 1.3819129 if D == 3 else 1.5
    offs = oracle.level_offsets(D, L, pls, H, 19 if D == 3 else 14, align)
    S = float(np.log2(pls))
    rows = int(offs[-1])
    emb = (rng.random((rows, C), dtype=np.float32) * 2 - 1) * (0.5 if half else 1e-1)
    B = 777
    x = rng.random((B, D), dtype=np.float32)
    x[0] = 0.0; x[1] = 1.0; x[2, 0] = -0.001; x[3, 1] = 1.001     # edges + out-of-range rows
    embt = emb.astype(np.float16) if half else emb
    for max_level in (L, 5):
        out = t(np.zeros((L, B, C), embt.dtype))
        dy = t(np.zeros((B, L * D * C), embt.dtype))
        ge.grid_encode_forward(t(x), t(embt), t(offs), out, B, D, C, L, max_level, S, H, dy, gridtype, align, interp)
        oo, ody = oracle.grid_encode_forward(x, embt, offs, S, H, max_level, True, gridtype, align, interp)
        assert np.array_equal(out.numpy().view(np.uint16 if half else np.uint32), oo.view(np.uint16 if half else np.uint32))
        assert np.array_equal(dy.numpy().view(np.uint16 if half else np.uint32), ody.view(np.uint16 if half else np.uint32))
        # sample-major variant == permute of the level-major result (grid.py:63)
        bm = oracle.grid_encode_forward(x, embt, offs, S, H, max_level, False, gridtype, align, interp, sample_major=True)
        assert np.array_equal(bm, oo.transpose(1, 0, 2).reshape(B, L * C))
        if half and C % 2:
            continue
        g = (rng.normal(size=(L, B, C)) * (64 if half else 1)).astype(embt.dtype)
        ge_out = t(np.zeros_like(embt))
        gin = t(np.zeros((B, D), embt.dtype))
        ge.grid_encode_backward(t(g), t(x), t(embt), t(offs), ge_out, B, D, C, L, max_level, S, H, dy, gin, gridtype, align, interp)
        og, ogi = oracle.grid_encode_backward(g, x, embt, offs, S, H, max_level, ody, gridtype, align, interp)
        assert np.array_equal(ge_out.numpy().view(np.uint16 if half else np.uint32), og.view(np.uint16 if half else np.uint32))
        assert np.array_equal(gin.numpy().view(np.uint16 if half else np.uint32), ogi.view(np.uint16 if half else np.uint32))
        gbm = oracle.grid_encode_backward(np.ascontiguousarray(g.transpose(1, 0, 2)).reshape(B, L * C), x, embt, offs, S, H,
                                          max_level, None, gridtype, align, interp, sample_major=True)
        assert np.array_equal(gbm, og)


def test_grad_total_variation(oracle, ref):
    ge = ref[1]
    rng = np.random.default_rng(6)
    offs, S = lego_offsets(1.0)
    emb = (rng.random((int(offs[-1]), 1), dtype=np.float32) * 2 - 1) * 1e-2
    x = rng.random((2000, 3), dtype=np.float32)
    x[:3] = [[0, 0, 0], [1, 1, 1], [1.2, 0.5, 0.5]]
    g_r = t(rng.normal(size=emb.shape).astype(np.float32) * 1e-6)
    g_o = g_r.numpy().copy()
    ge.grad_total_variation(t(x), t(emb), g_r, t(offs), 1e-3, 2000, 3, 1, 16, S, 16, 0, False)
    oracle.grad_total_variation(x, emb, g_o, offs, 1e-3, S, 16, 0, False)
    assert np.array_equal(g_r.numpy().view(np.uint32), g_o.view(np.uint32))


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_encode(oracle, ref, degree):
    sh = ref[2]
    rng = np.random.default_rng(7)
    B = 400
    v = rng.normal(size=(B, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[0] = [0, 0, 1]; v[1] = [1, 0, 0]; v[2] = [0, -1, 0]
    out = t(np.zeros((B, degree ** 2), np.float32))
    dy = t(np.zeros((B, 3 * degree ** 2), np.float32))
    sh.sh_encode_forward(t(v), out, B, 3, degree, dy)
    oo, ody = oracle.sh_encode_forward(v, degree, True)
    # fp32 polynomial evaluation (reference) vs double recurrence rounded once (oracle): a few ulp of the
    # largest term; entries are O(1), derivative entries up to O(100) at degree 8.
    np.testing.assert_allclose(oo, out.numpy(), rtol=0, atol=3e-6)
    np.testing.assert_allclose(ody, dy.numpy(), rtol=2e-6, atol=6e-5)
    g = rng.normal(size=(B, degree ** 2)).astype(np.float32)
    gi = t(np.zeros((B, 3), np.float32))
    sh.sh_encode_backward(t(g), t(v), B, 3, degree, dy, gi)
    ogi = oracle.sh_encode_backward(g, v, degree, dy.numpy())
    assert np.array_equal(gi.numpy().view(np.uint32), ogi.view(np.uint32))


@pytest.mark.parametrize("degree", [1, 4, 6, 10])
def test_freq_encode(oracle, degree):
    """freqencoder.cu:30-94 on the host (its __sinf maps to sinf in the shim) == the oracle, bit for bit, forward and backward."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import build_ref
    if not build_ref.available():
        pytest.skip("oracle/_ref not built")
    fq = build_ref.load_freq()
    rng = np.random.default_rng(9)
    B, D = 300, 3
    C = D + 2 * degree * D
    x = (rng.normal(size=(B, D)) * 3).astype(np.float32)
    out = t(np.zeros((B, C), np.float32))
    fq.freq_encode_forward(t(x), B, D, degree, C, out)
    oo = oracle.freq_encode_forward(x, degree)
    assert np.array_equal(out.numpy().view(np.uint32), oo.view(np.uint32))
    g = rng.normal(size=(B, C)).astype(np.float32)
    gi = t(np.zeros((B, D), np.float32))
    fq.freq_encode_backward(t(g), out, B, D, degree, C, gi)
    ogi = oracle.freq_encode_backward(g, oo, D, degree)
    assert np.array_equal(gi.numpy().view(np.uint32), ogi.view(np.uint32))
    # independent property: the columns really are sin / cos of 2^f x
    f = min(degree - 1, 3)
    np.testing.assert_allclose(oo[:, D + 2 * f * D:D + (2 * f + 1) * D], np.sin(x.astype(np.float64) * 2 ** f), atol=2e-6)
    np.testing.assert_allclose(oo[:, D + (2 * f + 1) * D:D + (2 * f + 2) * D], np.cos(x.astype(np.float64) * 2 ** f), atol=4e-6)


def test_sh_orthonormality(oracle):
    # independent property: Monte-Carlo orthonormality of the basis on the sphere
    rng = np.random.default_rng(8)
    v = rng.normal(size=(400000, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    Y = oracle.sh_encode_forward(v.astype(np.float32), 6).astype(np.float64)
    G = 4 * np.pi * (Y.T @ Y) / Y.shape[0]
    assert np.abs(G - np.eye(36)).max() < 0.03
