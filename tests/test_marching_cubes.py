"""Marching cubes: the device kernels against the CPU restatement, and both against order-free properties.

PARITY UNPINNED with respect to PyMCubes (absent here; see oracle/marching_cubes.py): what is checked is (i) the case table -- two
independent derivations of the documented rule and the committed .inc agree, (ii) HIP == oracle bit for bit (vertex floats, indices,
order), (iii) properties any correct extraction has: closed and consistently oriented surfaces, Euler characteristic, distance to the
analytic surface, positive enclosed volume (normals towards lower values)."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fields():
    """name -> (volume, iso): small cases the pure-Python emission loop finishes in a second."""
    rng = np.random.default_rng(5)
    out = {}
    x = np.linspace(-1, 1, 24)
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    out["sphere"] = ((0.6 - np.sqrt(X * X + Y * Y + Z * Z)).astype(np.float32), 0.0)
    out["torus"] = ((0.2 - np.sqrt((np.sqrt(X * X + Y * Y) - 0.55) ** 2 + Z * Z)).astype(np.float32), 0.0)
    noise = np.pad(rng.normal(size=(14, 14, 14)).astype(np.float32), 1, constant_values=-5.0)
    out["noise"] = (noise, 0.0)                                       # all 256 cases, closed because of the empty border
    out["ragged"] = (rng.normal(size=(5, 9, 33)).astype(np.float32), 0.25)      # open surface, three different extents
    ties = rng.integers(-1, 2, size=(9, 8, 7)).astype(np.float32)     # values exactly equal to iso count as solid
    out["ties"] = (ties, 0.0)
    out["ragged4"] = (rng.normal(size=(5, 9, 32)).astype(np.float32), -0.3)     # last extent a multiple of 4: four nodes per thread
    out["ties4"] = (rng.integers(-1, 2, size=(7, 9, 8)).astype(np.float32), 0.0)
    out["thin"] = (rng.normal(size=(1, 6, 6)).astype(np.float32), 0.0)          # a single layer of nodes: vertices, no cell
    out["point"] = (np.ones((1, 1, 1), np.float32), 0.0)
    out["empty"] = (np.full((6, 6, 6), -1.0, np.float32), 0.0)
    out["full"] = (np.full((6, 6, 6), 2.0, np.float32), 0.0)
    return out


# ------------------------------------------------------------------------------------------------------------ CPU

def test_case_table_two_derivations_and_the_committed_file_agree():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_mc_table
    from oracle import marching_cubes as omc
    a, b = gen_mc_table.table(), omc.case_table()
    assert [[tuple(t) for t in c] for c in a] == [[tuple(t) for t in c] for c in b]
    assert a[0] == [] and a[255] == [] and max(len(c) for c in a) == 5
    # case k and its complement cut the same edges; every triangle uses three distinct crossed edges
    for k in range(256):
        edges = lambda c: sorted({e for t in c for e in t})
        assert edges(a[k]) == edges(a[255 - k])
        for t in a[k]:
            assert len(set(t)) == 3
    # the committed include file is what the generator prints
    src = open(os.path.join(ROOT, "nerf2mesh_amd", "csrc", "mc_table.inc")).read()
    rows = re.findall(r"\{([^{}]*)\},", src[src.index("kMcTris"):])
    assert len(rows) == 256
    for k, row in enumerate(rows):
        flat = [int(v) for v in row.split(",")]
        want = [e for t in a[k] for e in t]
        assert flat[:len(want)] == want and all(v == 255 for v in flat[len(want):]), k
    assert int(re.search(r"N2M_MC_MAX_TRIS (\d+)", src).group(1)) == 5


def test_oracle_surfaces_are_closed_oriented_and_where_they_should_be():
    from oracle import marching_cubes as omc
    f = _fields()
    v, t = omc.marching_cubes(*f["sphere"], div=23.0, mul=2.0, add=-1.0)
    assert omc.directed_edge_imbalance(t) == 0
    assert len(v) - 3 * len(t) // 2 + len(t) == 2                                  # V - E + F of a sphere
    assert np.abs(np.linalg.norm(v, axis=1) - 0.6).max() < 2.5e-3                   # linear interpolation of a distance field: O(h^2)
    assert 0.97 * 4 / 3 * np.pi * 0.6 ** 3 < omc.signed_volume(v, t) < 4 / 3 * np.pi * 0.6 ** 3     # inscribed, normals outward
    v, t = omc.marching_cubes(*f["torus"], div=23.0, mul=2.0, add=-1.0)
    assert omc.directed_edge_imbalance(t) == 0 and len(v) - 3 * len(t) // 2 + len(t) == 0
    vol, iso = f["noise"]
    v, t = omc.marching_cubes(vol, iso)
    assert omc.directed_edge_imbalance(t) == 0 and omc.signed_volume(v, t) > 0
    s = ~(vol < iso)
    case = sum(s[x:13 + x + 2, y:13 + y + 2, z:13 + z + 2].astype(int) << (x | y << 1 | z << 2) for x in (0, 1) for y in (0, 1) for z in (0, 1))
    assert len(np.unique(case)) == 256                                              # the noise volume exercises every case
    for name in ("point", "empty", "full"):
        v, t = omc.marching_cubes(*f[name])
        assert v.shape == (0, 3) and t.shape == (0, 3)
    v, t = omc.marching_cubes(*f["thin"])
    assert len(v) > 0 and len(t) == 0


def test_c_restatement_equals_the_python_restatement():
    """oracle/n2m_oracle.c (n2m_oracle_marching_cubes, used for the full-size checks) against oracle/marching_cubes.py, bit for bit."""
    from oracle import marching_cubes as omc
    for name, (vol, iso) in _fields().items():
        for kw in (dict(), dict(div=7.0, mul=2.0, add=-1.0)):
            a, b = omc.marching_cubes(vol, iso, **kw), omc.marching_cubes_c(vol, iso, **kw)
            assert a[0].shape == b[0].shape and a[1].shape == b[1].shape, name
            assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1]), name


def test_every_extracted_surface_is_closed_and_outward():
    """The claim behind the rule-generated table: whatever the volume, the surface is closed (every directed edge has its reverse) and
    oriented towards lower values (positive enclosed volume, equal to the solid volume up to the half-cell the interpolation moves the
    surface by).  300 random volumes of three kinds -- white noise (every ambiguous face configuration), blobs, binary -- padded with
    an empty border so that nothing runs into the volume's faces."""
    from oracle import marching_cubes as omc
    rng = np.random.default_rng(11)
    for trial in range(300):
        kind = trial % 3
        shp = tuple(int(v) for v in rng.integers(3, 9, size=3))
        if kind == 0:
            vol = rng.normal(size=shp)
        elif kind == 1:
            g = [np.linspace(-1, 1, s) for s in shp]
            X, Y, Z = np.meshgrid(*g, indexing="ij")
            c = rng.uniform(-0.5, 0.5, size=(3, 3))
            vol = sum(np.exp(-((X - a) ** 2 + (Y - b) ** 2 + (Z - d) ** 2) * 6.0) for a, b, d in c) - rng.uniform(0.3, 0.9)
        else:
            vol = (rng.random(shp) < rng.uniform(0.2, 0.8)).astype(np.float64) - 0.5
        vol = np.pad(vol.astype(np.float32), 1, constant_values=-0.5 if kind == 2 else -3.0)
        v, t = omc.marching_cubes_c(vol, 0.0)
        if len(t) == 0:
            continue
        assert omc.directed_edge_imbalance(t) == 0, (trial, shp)
        assert int(t.min()) >= 0 and int(t.max()) == len(v) - 1 and len(np.unique(t)) == len(v)
        vol_mesh, n_solid = omc.signed_volume(v, t), int((vol >= 0).sum())
        assert vol_mesh > 0, (trial, shp)
        if kind == 2:        # binary volume: every vertex at an edge midpoint; a lone solid node is an octahedron of volume 1/6
            assert n_solid / 6.0 - 1e-6 <= vol_mesh <= n_solid + 1e-6, (trial, vol_mesh, n_solid)


def test_mesh_filters_restate_the_pymeshlab_selections():
    import torch
    from nerf2mesh_amd import export
    v = torch.arange(18, dtype=torch.float32).view(6, 3)
    t = torch.tensor([[0, 1, 2], [2, 3, 4], [3, 4, 5]], dtype=torch.int32)
    v2, t2 = export.remove_vertices(v, t, torch.tensor([False, False, False, False, False, True]))
    assert v2.shape[0] == 5 and t2.tolist() == [[0, 1, 2], [2, 3, 4]]
    v2, t2 = export.remove_vertices(v, t, torch.tensor([True, False, False, False, False, False]))
    assert torch.equal(v2, v[1:]) and t2.tolist() == [[1, 2, 3], [2, 3, 4]]
    # remove the last two faces; no dilation: vertex 2 survives through face 0, 3..5 go
    v2, t2 = export.remove_faces(v, t, torch.tensor([0, 1, 1]), dilation=0)
    assert v2.shape[0] == 3 and t2.tolist() == [[0, 1, 2]]
    # one dilation step grows the kept selection over the face sharing vertex 2, a second one over the last face
    v2, t2 = export.remove_faces(v, t, torch.tensor([0, 1, 1]), dilation=1)
    assert t2.tolist() == [[0, 1, 2], [2, 3, 4]] and v2.shape[0] == 5
    v2, t2 = export.remove_faces(v, t, torch.tensor([0, 1, 1]), dilation=2)
    assert t2.shape[0] == 3 and v2.shape[0] == 6


# ------------------------------------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sphere", "torus", "noise", "ragged", "ties", "ragged4", "ties4", "thin", "point", "empty", "full"])
def test_hip_marching_cubes_equals_the_oracle_bit_for_bit(name):
    import torch
    from nerf2mesh_amd.marching_cubes import marching_cubes
    from oracle import marching_cubes as omc
    vol, iso = _fields()[name]
    R = vol.shape[0]
    for kw in (dict(), dict(div=max(R - 1.0, 1.0), mul=2.0, add=-1.0)):
        ov, ot = omc.marching_cubes(vol, iso, **kw)
        v, t = marching_cubes(torch.from_numpy(vol).cuda(), iso, **kw)
        assert v.dtype == torch.float32 and t.dtype == torch.int32
        assert tuple(v.shape) == ov.shape and tuple(t.shape) == ot.shape, (v.shape, ov.shape, t.shape, ot.shape)
        assert np.array_equal(t.cpu().numpy(), ot)
        assert np.array_equal(v.cpu().numpy().view(np.uint32), ov.view(np.uint32))
    # double output: the same positions before the final rounding
    v64, t64 = marching_cubes(torch.from_numpy(vol).cuda(), iso, dtype=torch.float64)
    ov, ot = omc.marching_cubes(vol, iso)
    assert np.array_equal(v64.cpu().numpy().astype(np.float32).view(np.uint32), ov.view(np.uint32)) and np.array_equal(t64.cpu().numpy(), ot)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(192, 192, 192), (200, 168, 132), (97, 101, 103)])
def test_hip_marching_cubes_equals_the_c_oracle_at_size(shape):
    """A wavy field with hundreds of components that run into the volume's faces (open surfaces), at export-like sizes: vertices (float
    bits), triangles and their order equal the plain-C restatement.  (200, 168, 132): four nodes per thread; (97, 101, 103): one."""
    import torch
    from nerf2mesh_amd.marching_cubes import marching_cubes
    from oracle import marching_cubes as omc
    ax = [np.linspace(-1, 1, r, dtype=np.float32) for r in shape]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    vol = (np.sin(7 * X) * np.cos(5 * Y) + np.sin(6 * Z + 3 * X * Y)).astype(np.float32)
    vol[::17, ::13, ::11] = 0.2                                          # exact ties with the iso value
    kw = dict(div=shape[0] - 1.0, mul=2.0, add=-1.0)
    ov, ot = omc.marching_cubes_c(vol, 0.2, **kw)
    v, t = marching_cubes(torch.from_numpy(vol).cuda(), 0.2, **kw)
    assert tuple(v.shape) == ov.shape and tuple(t.shape) == ot.shape and ov.shape[0] > 50000
    assert np.array_equal(t.cpu().numpy(), ot)
    assert np.array_equal(v.cpu().numpy().view(np.uint32), ov.view(np.uint32))


@pytest.mark.gpu
def test_hip_marching_cubes_at_full_size_properties():
    """256^3 (the reference's --mcubes-style resolutions are 128..512): two tori and a sphere, disjoint.  Closed, consistently oriented,
    Euler characteristic 2 + 0 + 0, every vertex within O(h^2) of the analytic surface; run twice: identical output (no atomics)."""
    import torch
    from nerf2mesh_amd.marching_cubes import marching_cubes
    R = 256
    x = torch.linspace(-1, 1, R, device="cuda", dtype=torch.float64)
    X, Y, Z = torch.meshgrid(x, x, x, indexing="ij")
    d_sph = torch.sqrt((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z + 0.45) ** 2) - 0.3
    d_t1 = torch.sqrt((torch.sqrt(X * X + Y * Y) - 0.6) ** 2 + (Z - 0.5) ** 2) - 0.15
    d_t2 = torch.sqrt((torch.sqrt((X + 0.3) ** 2 + (Z + 0.4) ** 2) - 0.35) ** 2 + (Y + 0.45) ** 2) - 0.1
    dist = torch.minimum(torch.minimum(d_sph, d_t1), d_t2)
    vol = (-dist).float()
    v, t = marching_cubes(vol, 0.0, div=R - 1.0, mul=2.0, add=-1.0)
    v2, t2 = marching_cubes(vol, 0.0, div=R - 1.0, mul=2.0, add=-1.0)
    assert torch.equal(v, v2) and torch.equal(t, t2)
    V, F_ = v.shape[0], t.shape[0]
    assert V > 100000 and F_ % 2 == 0
    tl = t.long()
    assert int(tl.min()) == 0 and int(tl.max()) == V - 1 and torch.unique(tl).numel() == V          # every vertex used, none out of range
    e = torch.cat([tl[:, [0, 1]], tl[:, [1, 2]], tl[:, [2, 0]]])
    key, rev = e[:, 0] * V + e[:, 1], e[:, 1] * V + e[:, 0]
    ku, kc = torch.unique(key, return_counts=True)
    assert int(kc.max()) == 1                                                                          # 2-manifold: a directed edge once ...
    assert torch.equal(ku, torch.sort(rev).values)                                                     # ... and its reverse once
    assert V - (3 * F_) // 2 + F_ == 2                                                                 # sphere 2 + torus 0 + torus 0
    vd = v.double()
    def sd(p):
        a = torch.sqrt((p[:, 0] - 0.5) ** 2 + (p[:, 1] - 0.5) ** 2 + (p[:, 2] + 0.45) ** 2) - 0.3
        b = torch.sqrt((torch.sqrt(p[:, 0] ** 2 + p[:, 1] ** 2) - 0.6) ** 2 + (p[:, 2] - 0.5) ** 2) - 0.15
        c = torch.sqrt((torch.sqrt((p[:, 0] + 0.3) ** 2 + (p[:, 2] + 0.4) ** 2) - 0.35) ** 2 + (p[:, 1] + 0.45) ** 2) - 0.1
        return torch.minimum(torch.minimum(a, b), c)
    h = 2.0 / (R - 1)
    assert float(sd(vd).abs().max()) < 0.6 * h * h / 0.1 + 1e-6            # interpolation error of a distance field <= h^2 / (8 r_min), with slack
    tri = vd[tl]
    vol6 = (tri[:, 0] * torch.cross(tri[:, 1], tri[:, 2], dim=1)).sum()
    want = 4 / 3 * np.pi * 0.3 ** 3 + 2 * np.pi ** 2 * 0.6 * 0.15 ** 2 + 2 * np.pi ** 2 * 0.35 * 0.1 ** 2
    assert 0.99 * want < float(vol6) / 6 < want                               # inscribed polyhedra, normals outward
    # a capacity below the totals truncates instead of writing out of bounds
    from nerf2mesh_amd import _lib as L
    ws = torch.empty(int(L.lib().n2m_marching_cubes_workspace_bytes(R, R, R)), dtype=torch.uint8, device="cuda")
    totals = torch.zeros(2, dtype=torch.int64, device="cuda")
    L.call("n2m_marching_cubes_count", vol.data_ptr(), R, R, R, 0.0, ws.data_ptr(), ws.numel(), totals.data_ptr(), L.stream())
    assert totals.tolist() == [V, F_]
    vb = torch.full((1000 + 8, 3), 7.0, device="cuda")
    tb = torch.full((500 + 8, 3), -7, dtype=torch.int32, device="cuda")
    L.call("n2m_marching_cubes_emit", vol.data_ptr(), R, R, R, 0.0, ws.data_ptr(), ws.numel(), R - 1.0, 2.0, -1.0, vb.data_ptr(), 0, 1000, tb.data_ptr(), 500,
           L.stream())
    assert torch.equal(vb[:1000], v[:1000]) and torch.equal(tb[:500], t[:500]) and bool((vb[1000:] == 7).all()) and bool((tb[500:] == -7).all())
    with pytest.raises(RuntimeError, match="too small"):
        L.call("n2m_marching_cubes_count", vol.data_ptr(), R, R, R, 0.0, ws.data_ptr(), 1024, totals.data_ptr(), L.stream())
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        marching_cubes(vol.cpu(), 0.0)


@pytest.mark.gpu
def test_mcubes_shim_has_pymcubes_surface():
    """`import mcubes` of an unchanged reference checkout resolves to the shim: numpy in, (float64 index-space vertices, uint64 triangles)
    out, bool volumes accepted (the outer cascades pass one, nerf/renderer.py:614-616)."""
    import torch
    from nerf2mesh_amd import backends
    from nerf2mesh_amd.marching_cubes import marching_cubes
    backends.install()
    import mcubes
    assert os.path.dirname(mcubes.__file__) == backends.path()
    vol, iso = _fields()["sphere"]
    v, t = mcubes.marching_cubes(vol, iso)
    assert isinstance(v, np.ndarray) and v.dtype == np.float64 and t.dtype == np.uint64 and v.shape[1] == 3 and t.shape[1] == 3
    dv, dt = marching_cubes(torch.from_numpy(vol).cuda(), iso, dtype=torch.float64)
    assert np.array_equal(v, dv.cpu().numpy()) and np.array_equal(t.astype(np.int32), dt.cpu().numpy())
    assert v.min() >= 0 and v.max() <= 23                                      # index space
    vb, tb = mcubes.marching_cubes(vol > iso, 0.5)
    assert len(vb) == len(v) and np.allclose(vb % 1.0 % 0.5, 0.0)             # a binary volume puts every vertex at an edge midpoint


@pytest.mark.gpu
def test_export_stage0_feeds_stage1(tmp_path):
    """The stage-0 -> stage-1 hand-over of the reference (export_stage0 writes mesh_0.ply, the stage-1 renderer loads it,
    nerf/renderer.py:472-546, :123-165) inside this package: train the lego recipe briefly, extract at the grid resolution and at 192^3,
    read the PLY back, attach it with init_stage1 and rasterise one view."""
    import torch
    from nerf2mesh_amd import export, synthetic as S
    from nerf2mesh_amd.engine import Stage0Engine
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    torch.manual_seed(0)
    opt = make_options(O=True, bound=1, dt_gamma=0, iters=30000, fused_mlp=True)
    poses = S.make_cameras(100, seed=0)
    eng = Stage0Engine(NeRFNetwork(opt), opt, poses, torch.device("cuda:0"), seed=0)
    eng.mark_untrained()
    for _ in range(400):
        eng.train_step()
    model = eng.model
    assert model.mean_density > 0
    out = model.export_stage0(str(tmp_path / "grid"))
    v, t = out[0]
    assert v.shape[0] > 1000 and t.shape[0] > 1000 and float(v.abs().max()) <= 1.0
    rv, rt = export.read_ply(str(tmp_path / "grid" / "mesh_0.ply"))
    assert np.array_equal(rv, v.cpu().numpy()) and np.array_equal(rt, t.cpu().numpy())
    out = model.export_stage0(str(tmp_path / "fine"), resolution=192)
    v, t = out[0]
    assert t.shape[0] > 1000 and int(t.max()) < v.shape[0]
    # not a degenerate speck (the synthetic truck spans ~1.1 x 0.6 x 0.7), inside the unit cube
    ext = v.max(0).values - v.min(0).values
    assert float(ext.min()) > 0.3 and float(v.abs().max()) <= 1.0
    # visibility filter path + stage-1 attach
    class DS: pass
    ds = DS()
    ds.H = ds.W = 200
    ds.mvps = torch.stack([S.mvp_matrix(poses[i], H=200, W=200, focal=S.LEGO_FOCAL / 4) for i in range(0, 100, 5)]).cuda()
    out = model.export_stage0(str(tmp_path / "vis"), resolution=192, dataset=ds)
    v2, t2 = out[0]
    assert 0 < t2.shape[0] <= t.shape[0] and int(t2.max()) < v2.shape[0]
    rv, rt = export.read_ply(str(tmp_path / "vis" / "mesh_0.ply"))
    model.init_stage1(torch.from_numpy(rv), torch.from_numpy(rt))
    assert model.vertices.shape[0] == v2.shape[0] and model.triangles.shape[0] == t2.shape[0]
