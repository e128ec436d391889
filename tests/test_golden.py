"""Committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py from the reference's own
kernels compiled for the host).  CPU: the oracle must reproduce them bit for bit (fp32 / fp16 / integer alike; SH to
fp32 rounding).  GPU: the HIP library must reproduce them to the bars of test_hip_parity.py.  No access to
/root/reference is needed at test time."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    v = {1: np.uint8, 2: np.uint16, 4: np.uint32}[a.dtype.itemsize]
    return a.shape == b.shape and np.array_equal(a.view(v), b.view(v))


# ------------------------------------------------------------------------------------------------------ CPU: oracle
def test_oracle_utils(oracle):
    g = load("utils")
    assert np.array_equal(oracle.packbits(g["grid"], float(g["thresh"])), g["bits"])
    n, f = oracle.near_far_from_aabb(g["rays_o"], g["rays_d"], g["aabb"], float(g["min_near"]))
    assert same_bits(n, g["nears"]) and same_bits(f, g["fars"])
    assert np.array_equal(oracle.morton3D(g["coords"]), g["morton"])
    assert np.array_equal(oracle.morton3D_invert(g["morton"]), g["coords"])


@pytest.mark.parametrize("name", ["march_lego", "march_gamma"])
def test_oracle_march(oracle, name):
    u, g = load("utils"), load(name)
    x, d, ts, rays = oracle.march_rays_train(u["rays_o"], u["rays_d"], 1.0, False, u["bits"], 1, int(g["H"]), u["nears"], u["fars"],
                                             g["noises"], float(g["dt_gamma"]), 1024)
    assert np.array_equal(rays, g["rays"]) and same_bits(x, g["xyzs"]) and same_bits(ts, g["ts"])


def test_oracle_composite(oracle):
    m, c = load("march_lego"), load("composite")
    w, ws, dp, im = oracle.composite_rays_train_forward(c["sigmas"], c["rgbs"], m["ts"], m["rays"], 1e-4, False)
    assert same_bits(w, c["weights"]) and same_bits(ws, c["weights_sum"]) and same_bits(dp, c["depth"]) and same_bits(im, c["image"])
    gs, gr = oracle.composite_rays_train_backward(c["grad_weights"], c["grad_weights_sum"], c["grad_depth"], c["grad_image"], c["sigmas"],
                                                  c["rgbs"], m["ts"], m["rays"], c["weights_sum"], c["depth"], c["image"], 1e-4, False)
    assert same_bits(gs, c["grad_sigmas"]) and same_bits(gr, c["grad_rgbs"])


@pytest.mark.parametrize("name", ["grid_c1_f32", "grid_c2_f16"])
def test_oracle_grid(oracle, name):
    g = load(name)
    out, dy = oracle.grid_encode_forward(g["inputs"], g["embeddings"], g["offsets"], float(g["S"]), int(g["H"]), None, True)
    assert same_bits(out, g["outputs"]) and same_bits(dy, g["dy_dx"])
    ge, gi = oracle.grid_encode_backward(g["grad"], g["inputs"], g["embeddings"], g["offsets"], float(g["S"]), int(g["H"]), None, g["dy_dx"])
    assert same_bits(ge, g["grad_embeddings"]) and same_bits(gi, g["grad_inputs"])
    if "tv_grad_in" in g:
        tv = g["tv_grad_in"].copy()
        oracle.grad_total_variation(g["inputs"], g["embeddings"], tv, g["offsets"], 1e-2, float(g["S"]), int(g["H"]))
        assert same_bits(tv, g["tv_grad_out"])


def test_oracle_sh(oracle):
    g = load("sh")
    for deg in (1, 4, 8):
        out, dy = oracle.sh_encode_forward(g["inputs"], deg, True)
        np.testing.assert_allclose(out, g[f"out{deg}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(dy, g[f"dy{deg}"], rtol=2e-6, atol=6e-5)


def test_oracle_freq(oracle):
    g = load("freq")
    for deg in (1, 4, 6):
        out = oracle.freq_encode_forward(g["inputs"], deg)
        assert np.array_equal(out.view(np.uint32), g[f"out{deg}"].view(np.uint32))
        gi = oracle.freq_encode_backward(g[f"grad{deg}"], g[f"out{deg}"], 3, deg)
        assert np.array_equal(gi.view(np.uint32), g[f"grad_inputs{deg}"].view(np.uint32))


# ------------------------------------------------------------------------------------------------------- GPU: HIP
@pytest.fixture(scope="module")
def hip():
    import torch
    from nerf2mesh_amd import _lib, raymarching, gridencoder, shencoder, backends
    _lib.lib()
    backends.install()
    import _gridencoder, _shencoder
    return {"torch": torch, "rm": raymarching, "ge": _gridencoder, "sh": _shencoder}


def dev(hip, a):
    return hip["torch"].from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_hip_utils_and_march(hip):
    rm, u = hip["rm"], load("utils")
    assert np.array_equal(rm.packbits(dev(hip, u["grid"]), float(u["thresh"])).cpu().numpy(), u["bits"])
    n, f = rm.near_far_from_aabb(dev(hip, u["rays_o"]), dev(hip, u["rays_d"]), dev(hip, u["aabb"]), float(u["min_near"]))
    assert same_bits(n.cpu().numpy(), u["nears"]) and same_bits(f.cpu().numpy(), u["fars"])
    assert np.array_equal(rm.morton3D(dev(hip, u["coords"])).cpu().numpy(), u["morton"])
    for name in ("march_lego", "march_gamma"):
        g = load(name)
        x, d, ts, rays = rm.march_rays_train(dev(hip, u["rays_o"]), dev(hip, u["rays_d"]), 1.0, False, dev(hip, u["bits"]), 1, int(g["H"]),
                                             dev(hip, u["nears"]), dev(hip, u["fars"]), True, float(g["dt_gamma"]), 1024, dev(hip, g["noises"]))
        assert np.array_equal(rays.cpu().numpy(), g["rays"])                 # bit-exact ray/sample indexing
        assert same_bits(x.cpu().numpy(), g["xyzs"]) and same_bits(ts.cpu().numpy(), g["ts"])


@pytest.mark.gpu
def test_hip_composite(hip):
    rm, m, c = hip["rm"], load("march_lego"), load("composite")
    torch = hip["torch"]
    sig = dev(hip, c["sigmas"]).requires_grad_(True)
    rgb = dev(hip, c["rgbs"]).requires_grad_(True)
    w, ws, dp, im = rm.composite_rays_train(sig, rgb, dev(hip, m["ts"]), dev(hip, m["rays"]), 1e-4, False)
    np.testing.assert_allclose(w.detach().cpu().numpy(), c["weights"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(im.detach().cpu().numpy(), c["image"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ws.detach().cpu().numpy(), c["weights_sum"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dp.detach().cpu().numpy(), c["depth"], rtol=2e-4, atol=2e-5)
    torch.autograd.backward([w, ws, dp, im], [dev(hip, c["grad_weights"]), dev(hip, c["grad_weights_sum"]), dev(hip, c["grad_depth"]),
                                              dev(hip, c["grad_image"])])
    np.testing.assert_allclose(rgb.grad.cpu().numpy(), c["grad_rgbs"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(sig.grad.cpu().numpy(), c["grad_sigmas"], rtol=1e-3, atol=2e-5 * max(1.0, float(np.abs(c["grad_sigmas"]).max())))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["grid_c1_f32", "grid_c2_f16"])
def test_hip_grid(hip, name):
    torch, ge, g = hip["torch"], hip["ge"], load(name)
    x, emb, offs = dev(hip, g["inputs"]), dev(hip, g["embeddings"]), dev(hip, g["offsets"])
    L, B, C = g["outputs"].shape
    out = torch.zeros(L, B, C, dtype=emb.dtype, device="cuda")
    dy = torch.zeros(B, L * 3 * C, dtype=emb.dtype, device="cuda")
    ge.grid_encode_forward(x, emb, offs, out, B, 3, C, L, L, float(g["S"]), int(g["H"]), dy, 0, False, 0)
    assert same_bits(out.cpu().numpy(), g["outputs"]) and same_bits(dy.cpu().numpy(), g["dy_dx"])     # bit-identical to the reference
    gemb = torch.zeros_like(emb)
    gin = torch.zeros(B, 3, dtype=emb.dtype, device="cuda")
    ge.grid_encode_backward(dev(hip, g["grad"]), x, emb, offs, gemb, B, 3, C, L, L, float(g["S"]), int(g["H"]), dy, gin, 0, False, 0)
    ref = g["grad_embeddings"].astype(np.float32)
    tol = 3e-2 if emb.dtype == torch.float16 else 1e-5
    np.testing.assert_allclose(gemb.float().cpu().numpy(), ref, rtol=tol * 10, atol=tol * np.abs(ref).max())
    assert same_bits(gin.cpu().numpy(), g["grad_inputs"])
    if "tv_grad_in" in g:
        tv = dev(hip, g["tv_grad_in"])
        ge.grad_total_variation(x, emb, tv, offs, 1e-2, B, 3, C, L, float(g["S"]), int(g["H"]), 0, False)
        np.testing.assert_allclose(tv.cpu().numpy(), g["tv_grad_out"], rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
def test_hip_sh(hip):
    torch, sh, g = hip["torch"], hip["sh"], load("sh")
    for deg in (1, 4, 8):
        out = torch.empty(64, deg * deg, device="cuda")
        dy = torch.empty(64, 3 * deg * deg, device="cuda")
        sh.sh_encode_forward(dev(hip, g["inputs"]), out, 64, 3, deg, dy)
        np.testing.assert_allclose(out.cpu().numpy(), g[f"out{deg}"], rtol=0, atol=4e-6)
        np.testing.assert_allclose(dy.cpu().numpy(), g[f"dy{deg}"], rtol=4e-6, atol=1e-4)


@pytest.mark.gpu
def test_hip_freq(hip):
    """fp32 tolerance: device sinf vs the host libm behind the fixtures (<= 2 ulp of values in [-1, 1]); the backward is a few
    multiply-adds of those values in the reference's order."""
    torch, g = hip["torch"], load("freq")
    import _freqencoder as fq
    for deg in (1, 4, 6):
        C = 3 + 2 * deg * 3
        out = torch.empty(64, C, device="cuda")
        fq.freq_encode_forward(dev(hip, g["inputs"]), 64, 3, deg, C, out)
        np.testing.assert_allclose(out.cpu().numpy(), g[f"out{deg}"], rtol=0, atol=3e-7)
        gi = torch.empty(64, 3, device="cuda")
        fq.freq_encode_backward(dev(hip, g[f"grad{deg}"]), dev(hip, g[f"out{deg}"]), 64, 3, deg, C, gi)
        np.testing.assert_allclose(gi.cpu().numpy(), g[f"grad_inputs{deg}"], rtol=1e-5, atol=1e-5)
