"""Closed-form checks of stage-1 rasterisation, run on BOTH implementations: the scalar oracle (oracle/n2m_raster_oracle.c, CPU) and
the HIP kernels behind the nvdiffrast.torch facade (GPU).  nvdiffrast itself is not under /root/reference (parity unpinned), so these
self-consistency properties -- analytic coverage and barycentrics, watertight shared edges, perspective-correct interpolation, depth
order, antialias == exact area coverage, non-silhouette edges untouched -- are what pins the semantics; the
HIP kernels are additionally compared with the oracle value for value in test_raster_parity.py."""
import numpy as np
import pytest


class _HipRaster:
    """The oracle's numpy interface (rasterize / interpolate / antialias on arrays) on the HIP kernels."""

    def __init__(self):
        import torch
        from nerf2mesh_amd import raster
        self.torch, self.dr = torch, raster
        self.ctx = raster.RasterizeGLContext(output_db=False)

    def _t(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def rasterize(self, pos, tri, H, W):
        rast, _ = self.dr.rasterize(self.ctx, self._t(pos)[None], self._t(tri), (H, W))
        return rast[0].cpu().numpy()

    def interpolate(self, attr, rast, tri):
        out, _ = self.dr.interpolate(self._t(attr)[None], self._t(rast)[None], self._t(tri))
        return out[0].cpu().numpy()

    def antialias(self, color, rast, pos, tri):
        return self.dr.antialias(self._t(color)[None], self._t(rast)[None], self._t(pos)[None], self._t(tri))[0].cpu().numpy()


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def impl(request):
    if request.param == "oracle":
        from oracle import oracle as o
        o.lib()
        return o
    return _HipRaster()


def quad_mesh(x0, x1, y0, y1, z=0.0, w=1.0):
    pos = np.array([[x0, y0, z, 1], [x1, y0, z, 1], [x1, y1, z, 1], [x0, y1, z, 1]], np.float32) * np.array([w, w, w, w], np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return pos, tri


def test_single_triangle_barycentrics_reproduce_positions(impl):
    pos = np.array([[-0.8, -0.7, 0.1, 1], [0.9, -0.5, 0.3, 1], [0.1, 0.85, -0.2, 1]], np.float32)
    tri = np.array([[0, 1, 2]], np.int32)
    H = W = 64
    rast = impl.rasterize(pos, tri, H, W)
    cov = rast[..., 3] > 0
    # area of the triangle in pixels ~ number of covered centres
    area = 0.5 * abs(np.cross(pos[1, :2] - pos[0, :2], pos[2, :2] - pos[0, :2])) * (W / 2) * (H / 2)
    assert abs(cov.sum() - area) < 0.06 * area
    # interpolating the NDC positions returns the pixel centre (affine case, w = 1)
    out = impl.interpolate(pos[:, :3].copy(), rast, tri)
    ys, xs = np.nonzero(cov)
    np.testing.assert_allclose(out[ys, xs, 0], (xs + 0.5) / W * 2 - 1, atol=2e-6)
    np.testing.assert_allclose(out[ys, xs, 1], (ys + 0.5) / H * 2 - 1, atol=2e-6)      # row 0 is y = -1
    np.testing.assert_allclose(out[ys, xs, 2], rast[ys, xs, 2], atol=2e-6)             # z/w channel
    assert np.all(rast[~cov] == 0)


def test_watertight_shared_edges(impl):
    # a fan of triangles around an interior point covering the whole screen: every pixel exactly once, no gaps
    rng = np.random.default_rng(0)
    n = 23
    ang = np.sort(rng.random(n)) * 2 * np.pi
    ring = np.stack([3 * np.cos(ang), 3 * np.sin(ang)], 1)
    centre = np.array([[0.137, -0.291]])
    xy = np.concatenate([centre, ring]).astype(np.float32)
    pos = np.concatenate([xy, np.zeros((n + 1, 1), np.float32), np.ones((n + 1, 1), np.float32)], 1)
    tri = np.array([[0, 1 + i, 1 + (i + 1) % n] for i in range(n)], np.int32)
    H, W = 96, 80
    rast = impl.rasterize(pos, tri, H, W)
    assert np.all(rast[..., 3] > 0), "gap between adjacent triangles"
    # permuting the draw order changes only the ids, not which pixels are covered; and with distinct depths per triangle
    # the coverage of each triangle is order independent: count pixels per triangle under a reversed order
    rev = impl.rasterize(pos, tri[::-1].copy(), H, W)
    ids_fwd = rast[..., 3].astype(int) - 1
    ids_rev = n - 1 - (rev[..., 3].astype(int) - 1)
    assert np.array_equal(ids_fwd, ids_rev), "a pixel on a shared edge was claimed by both triangles (tie rule not exclusive)"
    # pixel centres exactly on an edge: axis-aligned quad split along the diagonal through pixel centres
    pos2, tri2 = quad_mesh(-1, 1, -1, 1)
    r2 = impl.rasterize(pos2, tri2, 16, 16)
    assert np.all(r2[..., 3] > 0)
    assert np.array_equal(np.unique(r2[..., 3]), [1, 2])


def test_perspective_correct_and_depth_order(impl):
    # one triangle with varying w: attribute interpolation must be perspective correct
    pos = np.array([[-1.5, -1.2, 0.2, 1.0], [3.0, -2.0, 1.0, 2.5], [0.3, 4.5, 2.0, 4.0]], np.float32)
    tri = np.array([[0, 1, 2]], np.int32)
    H = W = 48
    rast = impl.rasterize(pos, tri, H, W)
    ys, xs = np.nonzero(rast[..., 3] > 0)
    assert len(ys) > 200
    attr = np.array([[1.0], [5.0], [-2.0]], np.float32)
    out = impl.interpolate(attr, rast, tri)[ys, xs, 0]
    # closed form: screen-space barycentrics s_i, then a = sum(s_i a_i / w_i) / sum(s_i / w_i)
    sp = pos[:, :2] / pos[:, 3:4]
    px, py = (xs + 0.5) / W * 2 - 1, (ys + 0.5) / H * 2 - 1
    def ef(a, b, x, y): return (b[0] - a[0]) * (y - a[1]) - (b[1] - a[1]) * (x - a[0])
    s0, s1, s2 = ef(sp[1], sp[2], px, py), ef(sp[2], sp[0], px, py), ef(sp[0], sp[1], px, py)
    S = s0 + s1 + s2
    s0, s1, s2 = s0 / S, s1 / S, s2 / S
    num = s0 * attr[0, 0] / pos[0, 3] + s1 * attr[1, 0] / pos[1, 3] + s2 * attr[2, 0] / pos[2, 3]
    den = s0 / pos[0, 3] + s1 / pos[1, 3] + s2 / pos[2, 3]
    np.testing.assert_allclose(out, num / den, rtol=2e-5, atol=2e-5)
    # depth test: a nearer quad hides a farther one regardless of order
    pa, ta = quad_mesh(-0.5, 0.5, -0.5, 0.5, z=0.2)
    pb, tb = quad_mesh(-1, 1, -1, 1, z=0.6)
    pos = np.concatenate([pb, pa]); tri = np.concatenate([tb, ta + 4])
    r = impl.rasterize(pos, tri, 32, 32)
    assert set(np.unique(r[12:20, 12:20, 3])) <= {3.0, 4.0}
    assert set(np.unique(r[0:4, 0:4, 3])) <= {1.0, 2.0}
    np.testing.assert_allclose(r[16, 16, 2], 0.2, atol=1e-6)


@pytest.mark.parametrize("xe", [0.13, 0.37, 0.5, 0.62, 0.91])
def test_antialias_equals_area_coverage_for_an_axis_aligned_edge(impl, xe):
    # white quad covering x < edge on black: the blended pixel values equal the exact covered fraction of each pixel
    H, W = 8, 16
    edge_px = 6 + xe                                   # edge position in pixels
    x1 = edge_px / W * 2 - 1
    pos, tri = quad_mesh(-3.0, x1, -3.0, 3.0)
    rast = impl.rasterize(pos, tri, H, W)
    color = (rast[..., 3:4] > 0).astype(np.float32)
    out = impl.antialias(color, rast, pos, tri)[..., 0]
    expect = np.clip(edge_px - np.arange(W), 0, 1)      # covered fraction of pixel i = clamp(edge - i, 0, 1)
    np.testing.assert_allclose(out[3], expect, atol=2e-5)
    # interior shared edge (the quad's diagonal) is not a silhouette: nothing changes away from the boundary
    np.testing.assert_allclose(out[:, :5], 1.0, atol=1e-6)


def test_antialias_ignores_non_silhouette_edges(impl):
    pos, tri = quad_mesh(-3, 3, -3, 3)
    rast = impl.rasterize(pos, tri, 16, 16)
    rng = np.random.default_rng(1)
    color = rng.random((16, 16, 3)).astype(np.float32)
    out = impl.antialias(color, rast, pos, tri)
    np.testing.assert_array_equal(out, color)


def test_bbox_form_of_the_oracle_rasteriser_is_the_same_image():
    """oracle.rasterize(bbox=True) -- each triangle confined to its pixel box, the form bench.py times as the stage-1 CPU baseline --
    against the brute-force per-pixel loop over all triangles: bit-identical, including triangles behind the camera (w <= 0: full image)
    and triangles that leave the frame."""
    import torch
    from nerf2mesh_amd import synthetic as S
    from oracle import oracle as orc
    v, f = S.scene_mesh(1500)
    poses = S.make_cameras(4, seed=0)
    for cam, res, zoom in ((1, 96, 1.0), (2, 64, 6.0)):                 # zoom 6: most of the mesh outside the frame, some vertices behind it
        mvp = S.mvp_matrix(poses[cam], res, res, S.LEGO_FOCAL * res / S.LEGO_HW * zoom)
        pos = (torch.cat([v, torch.ones_like(v[:, :1])], 1) @ mvp.T).numpy()
        if zoom > 1:
            pos[::7, 3] *= -1.0                                           # a few vertices at w < 0
        a = orc.rasterize(pos, f.numpy(), res, res)
        b = orc.rasterize(pos, f.numpy(), res, res, bbox=True)
        assert np.array_equal(a, b)
        assert (a[..., 3] > 0).mean() > 0.02
