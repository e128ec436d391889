"""GPU: HIP rasterize / interpolate / antialias (include/n2m_raster.h, called through the nvdiffrast.torch facade) against the
scalar oracle, plus finite-difference checks of the three backward kernels.  Triangle ids must match the oracle exactly;
floats within the stated tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dr():
    from nerf2mesh_amd import backends
    backends.install()
    import nvdiffrast.torch as dr
    import torch
    return dr, torch


def random_mesh(rng, n_tri=600, persp=True):
    """A soup of small-ish triangles at assorted depths (some behind others, some partly off screen)."""
    c = rng.uniform(-1.1, 1.1, (n_tri, 1, 2))
    off = rng.normal(0, 0.12, (n_tri, 3, 2))
    xy = (c + off).reshape(-1, 2)
    z = np.repeat(rng.uniform(-0.8, 0.8, (n_tri, 1)), 3, 1).reshape(-1, 1) + rng.normal(0, 0.02, (n_tri * 3, 1))
    w = rng.uniform(0.7, 2.5, (n_tri * 3, 1)) if persp else np.ones((n_tri * 3, 1))
    pos = np.concatenate([xy * w, z * w, w], 1).astype(np.float32)
    tri = np.arange(n_tri * 3, dtype=np.int32).reshape(-1, 3)
    return pos, tri


@pytest.mark.parametrize("H,W,persp", [(96, 128, True), (64, 64, False), (203, 117, True)])
def test_rasterize_matches_oracle(dr, oracle, H, W, persp):
    dr, torch = dr
    rng = np.random.default_rng(H + W)
    pos, tri = random_mesh(rng, 500, persp)
    pos[:12, 3] = -0.5                                      # a few triangles behind the eye / crossing w = 0
    big = np.array([[-2.5, -2.0, 0.95, 1], [2.5, -2.0, 0.95, 1], [0, 3.0, 0.95, 1]], np.float32)   # a huge far triangle (big-box path)
    pos = np.concatenate([pos, big]); tri = np.concatenate([tri, [[len(pos) - 3, len(pos) - 2, len(pos) - 1]]]).astype(np.int32)
    ctx = dr.RasterizeGLContext(output_db=False)
    rast, _ = dr.rasterize(ctx, torch.from_numpy(pos).cuda()[None], torch.from_numpy(tri).cuda(), (H, W))
    ref = oracle.rasterize(pos, tri, H, W)
    got = rast[0].cpu().numpy()
    assert np.array_equal(got[..., 3], ref[..., 3]), f"{(got[..., 3] != ref[..., 3]).sum()} pixels with a different triangle id"
    np.testing.assert_allclose(got[..., :3], ref[..., :3], rtol=0, atol=3e-5)     # same formulas, fp contraction aside
    assert (ref[..., 3] > 0).mean() > 0.5


def test_rasterize_size_classes_match_oracle(dr, oracle):
    """The rasteriser's three size classes (raster.hip: a lane walks boxes up to 16 pixels, boxes up to 1024 get 16 lanes, anything larger
    or touching w <= 0 a workgroup) at their boundaries: right triangles whose bounding boxes hold exactly 15, 16, 17, 18, 1024 and 1056
    pixel centres, thin 17 x 1 / 1 x 17 / 1 x 1024 slivers, overlapping in depth, one partly off screen -- ids bit-exact against the scalar
    oracle, (u, v, z/w) to 3e-5."""
    dr, torch = dr
    H, W = 160, 192
    boxes = [(5, 3), (4, 4), (17, 1), (1, 17), (6, 3), (32, 32), (33, 32), (1024 // 8, 8), (2, 8), (16, 1), (1, 16), (40, 30)]
    rng = np.random.default_rng(7)
    pos, tri = [], []
    for i, (bw, bh) in enumerate(boxes * 3):
        x0, y0 = int(rng.integers(-8, W - 4)), int(rng.integers(-8, H - 4))          # some start off screen (the box is clipped to the image)
        z = float(rng.uniform(-0.9, 0.9))
        px = np.array([[x0 + 0.25, y0 + 0.25], [x0 + bw - 0.25, y0 + 0.25], [x0 + 0.25, y0 + bh - 0.25]], np.float64)
        if i % 2:
            px = px[[0, 2, 1]]                                                       # both orientations
        ndc = np.stack([px[:, 0] / W * 2 - 1, px[:, 1] / H * 2 - 1], 1)
        w = float(rng.uniform(0.8, 2.0)) if i % 3 == 0 else 1.0
        pos += [[ndc[k, 0] * w, ndc[k, 1] * w, z * w, w] for k in range(3)]
        tri.append([3 * i, 3 * i + 1, 3 * i + 2])
    pos, tri = np.asarray(pos, np.float32), np.asarray(tri, np.int32)
    ctx = dr.RasterizeGLContext(output_db=False)
    rast, _ = dr.rasterize(ctx, torch.from_numpy(pos).cuda()[None], torch.from_numpy(tri).cuda(), (H, W))
    ref = oracle.rasterize(pos, tri, H, W)
    got = rast[0].cpu().numpy()
    assert np.array_equal(got[..., 3], ref[..., 3]), f"{(got[..., 3] != ref[..., 3]).sum()} pixels with a different triangle id"
    np.testing.assert_allclose(got[..., :3], ref[..., :3], rtol=0, atol=3e-5)
    seen = set(np.unique(ref[..., 3]).astype(int)) - {0}
    assert len(seen) >= 20                                                            # most of the 36 triangles own pixels


def test_rasterize_large_mesh_properties(dr):
    """Full-size stage-1 case: ~300k faces at 1600x1600 (ssaa 2). Properties: each box face tessellation is watertight, ids valid,
    re-running gives identical output (deterministic z-buffer), interpolating ones gives the coverage mask."""
    dr, torch = dr
    from nerf2mesh_amd import synthetic as S
    v, f = S.scene_mesh(300000, device="cuda")
    pose = S.make_cameras(4, seed=1)[2].cuda()
    mvp = S.mvp_matrix(pose)
    clip = (torch.nn.functional.pad(v, (0, 1), value=1.0) @ mvp.T)[None].contiguous()
    ctx = dr.RasterizeGLContext(output_db=False)
    r1, _ = dr.rasterize(ctx, clip, f, (1600, 1600))
    r2, _ = dr.rasterize(ctx, clip, f, (1600, 1600))
    assert torch.equal(r1, r2)
    ids = r1[0, ..., 3]
    assert float(ids.max()) <= f.shape[0] and float(ids.min()) == 0
    cov = ids > 0
    assert 0.03 < float(cov.float().mean()) < 0.6
    mask, _ = dr.interpolate(torch.ones_like(v[:, :1])[None], r1, f)
    assert torch.equal(mask[0, ..., 0] > 0, cov)
    np.testing.assert_allclose(mask[0, ..., 0][cov].cpu().numpy(), 1.0, atol=1e-5)
    xyz, _ = dr.interpolate(v[None], r1, f)
    # interpolated surface points lie on the boxes: inside the scene's bounding box, and reprojecting them lands on the pixel
    p = xyz[0][cov]
    assert float(p.abs().max()) <= 0.61
    q = torch.nn.functional.pad(p, (0, 1), value=1.0) @ mvp.T
    ndc = q[:, :2] / q[:, 3:4]
    ys, xs = torch.nonzero(cov, as_tuple=True)
    np.testing.assert_allclose(ndc[:, 0].cpu().numpy(), ((xs + 0.5) / 1600 * 2 - 1).cpu().numpy(), atol=2e-4)
    np.testing.assert_allclose(ndc[:, 1].cpu().numpy(), ((ys + 0.5) / 1600 * 2 - 1).cpu().numpy(), atol=2e-4)


def test_interpolate_and_antialias_match_oracle(dr, oracle):
    dr, torch = dr
    rng = np.random.default_rng(7)
    H, W = 72, 88
    # a closed-ish surface: a quad strip plus floating triangles, so that there are silhouette AND interior edges
    pos, tri = random_mesh(rng, 150, True)
    gx, gy = np.meshgrid(np.linspace(-0.9, 0.9, 9), np.linspace(-0.6, 0.6, 7))
    grid_pos = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, 0.5), np.ones(gx.size)], 1).astype(np.float32)
    base = len(pos)
    quads = []
    for j in range(6):
        for i in range(8):
            a = base + j * 9 + i
            quads += [[a, a + 1, a + 10], [a, a + 10, a + 9]]
    pos = np.concatenate([pos, grid_pos]); tri = np.concatenate([tri, np.array(quads, np.int32)]).astype(np.int32)
    P, T = torch.from_numpy(pos).cuda()[None], torch.from_numpy(tri).cuda()
    ctx = dr.RasterizeCudaContext()
    rast, _ = dr.rasterize(ctx, P, T, (H, W))
    ref_rast = oracle.rasterize(pos, tri, H, W)
    assert np.array_equal(rast[0, ..., 3].cpu().numpy(), ref_rast[..., 3])
    attr = rng.normal(size=(len(pos), 5)).astype(np.float32)
    out, _ = dr.interpolate(torch.from_numpy(attr).cuda(), rast, T)
    ref_out = oracle.interpolate(attr, rast[0].cpu().numpy(), tri)
    np.testing.assert_allclose(out[0].cpu().numpy(), ref_out, rtol=0, atol=2e-6)
    color = rng.random((H, W, 3)).astype(np.float32)
    aa = dr.antialias(torch.from_numpy(color).cuda()[None], rast, P, T)
    ref_aa = oracle.antialias(color, rast[0].cpu().numpy(), pos, tri)
    changed = np.abs(ref_aa - color).max(-1) > 0
    assert changed.mean() > 0.02, "test scene has no silhouettes"
    np.testing.assert_allclose(aa[0].cpu().numpy(), ref_aa, rtol=0, atol=3e-5)
    # explicit topology hash gives the same result
    th = dr.antialias_construct_topology_hash(T)
    aa2 = dr.antialias(torch.from_numpy(color).cuda()[None], rast, P, T, topology_hash=th)
    np.testing.assert_allclose(aa2[0].cpu().numpy(), aa[0].cpu().numpy(), atol=1e-6)


def test_rasterize_and_interpolate_backward_fd(dr):
    """One triangle far larger than the screen (coverage cannot change): d(u,v)/d(pos) and interpolate grads vs central differences."""
    dr, torch = dr
    torch.manual_seed(0)
    H = W = 24
    pos0 = torch.tensor([[-6.0, -5.0, 0.2, 1.3], [7.0, -4.0, 0.5, 2.1], [0.5, 8.0, 0.1, 1.7]], device="cuda")
    tri = torch.tensor([[0, 1, 2]], dtype=torch.int32, device="cuda")
    attr0 = torch.randn(3, 4, device="cuda")
    wr = torch.randn(H, W, 2, device="cuda")
    wo = torch.randn(H, W, 4, device="cuda")
    ctx = dr.RasterizeGLContext()

    def loss(pos, attr):
        rast, _ = dr.rasterize(ctx, pos[None], tri, (H, W))
        out, _ = dr.interpolate(attr[None], rast, tri)
        return (rast[0, ..., :2] * wr).sum() + (out[0] * wo).sum()

    pos = pos0.clone().requires_grad_(True)
    attr = attr0.clone().requires_grad_(True)
    loss(pos, attr).backward()
    eps = 2e-3
    for (i, j) in [(0, 0), (0, 1), (1, 3), (2, 0), (2, 3), (1, 1)]:
        d = torch.zeros_like(pos0); d[i, j] = eps
        fd = (loss(pos0 + d, attr0) - loss(pos0 - d, attr0)).item() / (2 * eps)
        assert abs(fd - pos.grad[i, j].item()) <= 2e-2 * max(1.0, abs(fd)), (i, j, fd, pos.grad[i, j].item())
    assert abs(pos.grad[:, 2]).max().item() == 0.0          # z carries no gradient
    for (i, j) in [(0, 0), (1, 2), (2, 3)]:
        d = torch.zeros_like(attr0); d[i, j] = 1e-2
        fd = (loss(pos0, attr0 + d) - loss(pos0, attr0 - d)).item() / 2e-2
        assert abs(fd - attr.grad[i, j].item()) <= 2e-3 * max(1.0, abs(fd))


def test_antialias_backward_fd(dr):
    """A bright quad over a dark background: gradients of the antialiased image w.r.t. colours and the silhouette's vertex
    positions against central differences (edge kept away from pixel centres so coverage stays fixed)."""
    dr, torch = dr
    H, W = 12, 16
    x1 = (9 + 0.3) / W * 2 - 1
    y1 = (7 + 0.7) / H * 2 - 1
    pos0 = torch.tensor([[-3.0, -3.0, 0.0, 1.0], [x1, -3.0, 0.0, 1.0], [x1, y1, 0.0, 1.0], [-3.0, y1, 0.0, 1.0]], device="cuda") * 1.5
    tri = torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32, device="cuda")
    ctx = dr.RasterizeGLContext()
    torch.manual_seed(1)
    col0 = torch.rand(1, H, W, 3, device="cuda")
    wgt = torch.randn(1, H, W, 3, device="cuda")

    def loss(pos, col):
        rast, _ = dr.rasterize(ctx, pos[None], tri, (H, W))
        inside = (rast[..., 3:] > 0).float()
        c = col * (0.2 + 0.8 * inside)
        return (dr.antialias(c, rast, pos[None], tri, pos_gradient_boost=1.0) * wgt).sum()

    pos = pos0.clone().requires_grad_(True)
    col = col0.clone().requires_grad_(True)
    loss(pos, col).backward()
    assert pos.grad.abs().sum().item() > 0, "no image-space gradient reached the silhouette vertices"
    eps = 1e-3
    for (i, j) in [(1, 0), (2, 0), (2, 1), (3, 1), (1, 3), (2, 3)]:
        d = torch.zeros_like(pos0); d[i, j] = eps
        fd = (loss(pos0 + d, col0) - loss(pos0 - d, col0)).item() / (2 * eps)
        assert abs(fd - pos.grad[i, j].item()) <= 3e-2 * max(1.0, abs(fd)), (i, j, fd, pos.grad[i, j].item())
    for idx in [(0, 7, 9, 1), (0, 7, 10, 0), (0, 3, 3, 2), (0, 8, 5, 1)]:
        d = torch.zeros_like(col0); d[idx] = 1e-2
        fd = (loss(pos0, col0 + d) - loss(pos0, col0 - d)).item() / 2e-2
        assert abs(fd - col.grad[idx].item()) <= 2e-3 * max(1.0, abs(fd)), idx


def test_torch_scatter_shim(dr):
    dr, torch = dr
    import torch_scatter
    out = torch.zeros(5, device="cuda")
    torch_scatter.scatter_add(torch.tensor([1.0, 2.0, 3.0, 4.0], device="cuda"), torch.tensor([0, 3, 3, 4], device="cuda"), out=out)
    assert out.tolist() == [1.0, 0.0, 0.0, 5.0, 4.0]
