"""Round 6: the forward lookup leaves the FINISHED total-variation terms of its samples (n2m_grid_encode_forward_packed_tvterms).

The TV term of gridencoder.cu:505-609 for a sample on a level needs the density table at its cell c = floor(x * scale + 0.5) and at c's six axis
neighbours.  c is vertex 000 of the sample's interpolation cell: the centre and the +x / +y / +z neighbours are corners the lookup of the same sample
holds in registers; it gathers -y, -z and (hashed level, even cell; dense level) -x, forms the term like the fill does, and the table backward reads it
back (n2m_grid_encode_backward_binned_pair_tvt) instead of gathering the stencil in its tile's dependent chain.  Held here: the lookup's outputs do not
change; the terms equal n2m_grid_tv_terms' bit for bit (plain table and packed density column, with and without a scale, inner / outer weights,
samples on the cube's faces and outside it); they are what a torch statement of the stencil says they are; the table gradients equal the in-place
stencil's bit for bit; and a training run of the executor ends in identical bits with the switch on and off."""
import numpy as np
import pytest
import torch

from test_tv_corners import P1, P2, _fwd_args, _setup

pytestmark = pytest.mark.gpu


def _terms_from_lookup(e1, xyz, pk, M, lam, lam_outer, inner01, scale_t):
    from nerf2mesh_amd import _lib as L
    h1, h2 = torch.empty(16, M, device="cuda"), torch.empty(16, M, 2, device="cuda", dtype=torch.float16)
    tv = torch.full((16, M), float("nan"), device="cuda")
    L.call("n2m_grid_encode_forward_packed_tvterms", *_fwd_args(e1, xyz, pk, h1, h2, M), lam, lam_outer, inner01,
           L.ptr(scale_t) if scale_t is not None else None, L.ptr(tv), L.stream())
    torch.cuda.synchronize()
    return h1, h2, tv


@pytest.mark.parametrize("with_scale, inner01", [(True, 0.5), (False, 0.5), (True, 0.25)])
def test_lookup_terms_equal_the_stand_alone_terms_and_the_lookup_is_unchanged(with_scale, inner01):
    from nerf2mesh_amd import _lib as L
    p = L.ptr
    e1, e2, xyz, pk, ho, g = _setup()
    M = xyz.shape[0]
    h1a, h2a = torch.empty(16, M, device="cuda"), torch.empty(16, M, 2, device="cuda", dtype=torch.float16)
    L.call("n2m_grid_encode_forward_packed", *_fwd_args(e1, xyz, pk, h1a, h2a, M), L.stream())
    scale_t = torch.tensor(8192.0, device="cuda") if with_scale else None
    lam, lam_outer = 1e-4, 1e-3
    h1b, h2b, tv = _terms_from_lookup(e1, xyz, pk, M, lam, lam_outer, inner01, scale_t)
    assert torch.equal(h1a, h1b) and torch.equal(h2a, h2b), "the lookup's outputs must not change"
    assert bool(torch.isfinite(tv).all()), "every (level, sample) gets a term"
    S, H0 = float(np.log2(e1.per_level_scale)), int(e1.base_resolution)
    emb = e1.embeddings.detach().contiguous()
    for stride, table in ((1, emb), (2, pk)):       # the stand-alone kernel on the plain table and on the packed rows' density column
        want = torch.full((16, M), float("nan"), device="cuda")
        L.grid_backward_config(stride, 1.0)
        try:
            L.call("n2m_grid_tv_terms", p(xyz), p(table), ho.ctypes.data, M, 16, S, H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id, lam,
                   lam_outer, inner01, p(scale_t) if with_scale else None, 0.5, 0.5, p(want), L.stream())
        finally:
            L.grid_backward_config(1, 1.0)
        torch.cuda.synchronize()
        assert torch.equal(tv, want), f"stride {stride}: {int((tv != want).sum())} terms differ"
    x01 = xyz * 0.5 + 0.5
    outside = ~((x01 >= 0) & (x01 <= 1)).all(-1)
    assert int(outside.sum()) >= 4 and bool((tv[:, outside] == 0).all()), "samples outside the unit cube: zero term (gridencoder.cu:537)"
    assert float(tv.abs().max()) > 0


def test_lookup_terms_are_the_stencil_of_the_reference_in_torch():
    """gridencoder.cu:505-609 restated in torch (fp32, the reference's order of neighbours) on four levels: one dense, three hashed."""
    e1, e2, xyz, pk, ho, g = _setup(M=30011, seed=3)
    M = xyz.shape[0]
    lam = 1e-4
    _, _, tv = _terms_from_lookup(e1, xyz, pk, M, lam, lam, 0.5, None)
    x01 = xyz * 0.5 + 0.5
    inside = ((x01 >= 0) & (x01 <= 1)).all(-1)
    S, H0 = float(np.log2(e1.per_level_scale)), int(e1.base_resolution)
    emb = e1.embeddings.detach()[:, 0]
    for l in (2, 6, 11, 15):
        size = int(ho[l + 1] - ho[l])
        scale = float(np.float32(np.exp2(np.float32(l) * np.float32(S)) * np.float32(H0)) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        dense = (res + 1) ** 3 <= size
        cell = torch.floor(x01 * scale + 0.5).to(torch.int64)

        def row(c):
            if dense:
                return c[:, 0] + c[:, 1] * (res + 1) + c[:, 2] * (res + 1) ** 2
            return (c[:, 0] ^ ((c[:, 1] * P1) & 0xFFFFFFFF) ^ ((c[:, 2] * P2) & 0xFFFFFFFF)) & (size - 1)

        centre = emb[int(ho[l]) + row(cell)]
        tot, sq = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
        for d in range(3):
            for step in (1, -1):
                c = cell.clone()
                c[:, d] += step
                ok = (cell[:, d] < res) if step == 1 else (cell[:, d] > 0)
                c[:, d] = torch.where(ok, c[:, d], cell[:, d])
                dv = centre - emb[int(ho[l]) + row(c)]
                dv = torch.where(ok, dv, torch.zeros_like(dv))
                tot = tot + dv
                sq = sq + dv * dv
        want = (lam / 6.0) * tot * torch.rsqrt(sq + 1e-9)
        got = tv[l]
        # (cell boundaries may differ by an ulp of the level scale; rsqrt: v_rsq_f32 vs torch's -- 1 ulp)
        close = (got - want).abs() <= 2e-6 * want.abs() + 1e-12
        assert float((close | ~inside).float().mean()) >= 0.9995, f"level {l}: {int((~close & inside).sum())} terms differ from the torch statement"


@pytest.mark.parametrize("half", [0, 1])
def test_table_gradients_equal_the_in_place_stencils(half):
    from nerf2mesh_amd import _lib as L
    p = L.ptr
    e1, e2, xyz, pk, ho, g = _setup(M=90011, seed=9)
    M = xyz.shape[0]
    rows = e1.embeddings.shape[0]
    scale_t = torch.tensor(4096.0, device="cuda")
    lam, lam_outer, inner01 = 1e-4, 1e-3, 0.4
    _, _, tv = _terms_from_lookup(e1, xyz, pk, M, lam, lam_outer, inner01, scale_t)
    d1 = torch.randn(16, M, device="cuda", generator=g) * 1e-3
    d2 = (torch.randn(16, M, 2, device="cuda", generator=g) * 0.05).half()
    d1[:, M // 2:] = 0                      # the tails of the rays: no gradient, a TV term only
    d2[:, M // 2:] = 0
    emb = e1.embeddings.detach().contiguous()
    need = L.lib().n2m_grid_binned_pair_workspace_bytes(M, 16, ho.ctypes.data)
    ws = L.workspace(xyz.device, need)
    finf = torch.zeros((), device="cuda")
    S, H0 = float(np.log2(e1.per_level_scale)), int(e1.base_resolution)
    out = {}
    for use in (False, True):
        g1 = torch.full((rows, 1), float("nan"), device="cuda")
        g2 = torch.full((rows, 2), float("nan"), device="cuda", dtype=torch.float16)
        L.grid_backward_config(1, 1.0)
        common = (p(d1), p(d2), p(xyz), ho.ctypes.data, p(g1), p(g2), M, 16, 16, S, H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id)
        tail = (p(finf), 0.5, 0.5, 1, p(ws), ws.numel(), L.stream())
        halves = (0,) if half == 0 else (1, 2)
        for h in halves:
            if use:
                L.call("n2m_grid_encode_backward_binned_pair_tvt", *common, p(tv), *tail, h)
            elif h == 0:
                L.call("n2m_grid_encode_backward_binned_pair", *common, p(emb), lam, lam_outer, inner01, p(scale_t), *tail)
            else:
                L.call("n2m_grid_encode_backward_binned_pair_half", *common, p(emb), lam, lam_outer, inner01, p(scale_t), *tail, h)
        torch.cuda.synchronize()
        out[use] = (g1, g2)
    assert float(finf) == 0.0
    assert bool(torch.isfinite(out[True][0]).all()) and float(out[True][0].abs().max()) > 0
    assert torch.equal(out[False][0], out[True][0]), "density-table gradient (carries the TV terms)"
    assert torch.equal(out[False][1], out[True][1]), "colour-table gradient"


@pytest.mark.parametrize("recipe", ["lego", "garden"])
def test_executor_trains_to_identical_bits_with_the_terms_from_the_lookup(monkeypatch, recipe):
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.engine import Stage0Engine
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    dev = torch.device("cuda", 0)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("N2M_TV_FWD", flag)
        torch.manual_seed(0)
        if recipe == "lego":
            opt = make_options(O=True, bound=1, dt_gamma=0, iters=30000, fused_mlp=True)
        else:       # config 4's recipe: inner / outer TV weights, five cascades
            opt = make_options(O=True, bound=16, dt_gamma=1 / 256, lambda_entropy=1e-3, enable_cam_near_far=True, scene="garden", iters=30000, fused_mlp=True)
        opt.num_rays, opt.num_points = 4096, 1 << 16
        model = NeRFNetwork(opt)
        if recipe == "garden":
            model.update_aabb(synthetic.pts_aabb("garden"))          # main.py:234-235
        eng = Stage0Engine(model, opt, synthetic.make_cameras(20, seed=0), dev, seed=0)
        assert eng.tv_fwd == (flag == "1")
        eng.mark_untrained()
        for _ in range(40):          # (across two occupancy refreshes and GradScaler's first steps)
            eng.train_step()
        torch.cuda.synchronize()
        res[flag] = [p.detach().clone() for p in eng.model.parameters()]
    for a, b in zip(res["1"], res["0"]):
        assert torch.equal(a, b)
