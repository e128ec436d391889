"""CPU tests of the host-side logic: options, level geometry, network parameter inventory, synthetic rays."""
import math

import numpy as np
import pytest
import torch

from nerf2mesh_amd import synthetic as S
from nerf2mesh_amd.gridencoder import level_offsets
from nerf2mesh_amd.options import make_options


def test_options_recipes():
    o = make_options(O=True, bound=1, dt_gamma=0)            # scripts/runall_syn.sh:1
    assert o.fp16 and o.mark_untrained and o.adaptive_num_rays and o.cuda_ray and not o.contract
    o = make_options(O=True, bound=16, sdf=True)             # main.py:138-157
    assert o.contract and o.density_thresh == 0.001 and o.progressive_level and not o.mark_untrained
    with pytest.raises(TypeError):
        make_options(nonsense=1)


def test_level_offsets_known_answers():
    pls = np.exp2(np.log2(2048 / 16) / 15)
    offs = level_offsets(3, 16, pls, 16, 19)
    assert offs[-1] == 6119864
    assert np.diff(offs).tolist() == [4920, 13824, 32768, 85184, 216000] + [524288] * 11
    pls16 = np.exp2(np.log2(2048 * 16 / 16) / 15)
    assert level_offsets(3, 16, pls16, 16, 19)[-1] == 6837544
    pls2 = np.exp2(np.log2(2048 * 2 / 16) / 15)
    assert level_offsets(3, 16, pls2, 16, 19)[-1] == 6328848


def test_network_inventory_matches_reference_checkpoint_keys():
    from nerf2mesh_amd.network import NeRFNetwork
    m = NeRFNetwork(make_options(O=True, bound=1, dt_gamma=0))
    sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert sum(p.numel() for p in m.parameters()) == 18367240
    assert sd["encoder.embeddings"] == (6119864, 1) and sd["encoder_color.embeddings"] == (6119864, 2)
    assert sd["sigma_net.net.0.weight"] == (32, 19) and sd["sigma_net.net.1.weight"] == (1, 32)
    assert sd["color_net.net.0.weight"] == (64, 35) and sd["color_net.net.2.weight"] == (6, 64)
    assert sd["specular_net.net.0.weight"] == (32, 6) and sd["specular_net.net.1.weight"] == (3, 32)
    assert sd["density_grid"] == (1, 128 ** 3) and sd["density_bitfield"] == (128 ** 3 // 8,)
    assert sd["encoder.offsets"] == (17,) and m.encoder.offsets.dtype == torch.int32
    assert float(m.encoder.embeddings.abs().max()) <= 1e-4
    m5 = NeRFNetwork(make_options(bound=16))
    assert m5.cascade == 5 and m5.density_bitfield.numel() == 5 * 128 ** 3 // 8


def test_rays_follow_get_rays():
    poses = S.make_cameras(4, seed=1)
    # pixel centre of the image looks straight down -z of the camera, i.e. at the origin
    pix = torch.tensor([400 * 800 + 400])
    o, d = S.rays_from_pixels(poses, torch.tensor([2]), pix)
    assert abs(float(o.norm()) - S.LEGO_RADIUS) < 1e-4
    fwd = -poses[2, :3, 2]
    dn = d[0] / d[0].norm()
    assert float((dn - fwd).abs().max()) < 2e-3                     # half-pixel offset only
    # the direction's camera-space z is exactly -1 (un-normalised, so t is z-depth: nerf/utils.py:282-288)
    assert abs(float((d[0] @ poses[2, :3, 2])) + 1.0) < 1e-6
    assert abs(S.LEGO_FOCAL - 1111.111) < 0.01


def test_scene_and_gt():
    g = S.scene_density_grid(H=64)
    occ = float((g > 0).float().mean())
    assert 0.01 < occ < 0.10
    poses = S.make_cameras(8, seed=2)
    o, d = S.random_rays(poses, 2000, torch.Generator().manual_seed(0))
    rgba = S.render_gt(o, d)
    assert 0.03 < float(rgba[:, 3].mean()) < 0.6
    assert float(rgba[:, :3].max()) <= 1.0
    # Morton helper agrees with its definition
    c = torch.tensor([[1, 0, 0], [0, 1, 0], [0, 0, 1], [127, 127, 127]])
    assert S.morton3D_torch(c).tolist() == [1, 2, 4, 2 ** 21 - 1]


def test_uniform_laplacian_matches_dense_operator_and_its_gradient():
    """trainer.UniformLaplacian == the reference's laplacian_smooth_loss (nerf/utils.py:176-221): mean_i || ((D - A) v)_i ||_2 with A the
    0/1 adjacency of the unique edges -- built densely here, and the unchanged reference function itself when the checkout is present;
    the custom backward (the neighbour sum is self-adjoint on a symmetric edge list) must equal autograd's."""
    import torch
    from nerf2mesh_amd.trainer import UniformLaplacian
    torch.manual_seed(0)
    faces = torch.tensor([[0, 1, 2], [0, 2, 3], [0, 3, 4], [1, 2, 5], [2, 3, 5], [3, 4, 5]])
    V = 6
    v = torch.randn(V, 3, dtype=torch.float64, requires_grad=True)
    A = torch.zeros(V, V, dtype=torch.float64)
    for f in faces.tolist():
        for a, b in ((0, 1), (1, 2), (2, 0)):
            A[f[a], f[b]] = A[f[b], f[a]] = 1
    ref = ((torch.diag(A.sum(1)) - A) @ v).norm(dim=1).mean()
    g_ref, = torch.autograd.grad(ref, v)
    lap = UniformLaplacian(faces, V)
    lap.deg = lap.deg.double()
    got = lap(v)
    g_got, = torch.autograd.grad(got, v)
    assert torch.allclose(got, ref, rtol=1e-9) and torch.allclose(g_got, g_ref, rtol=1e-8, atol=1e-12)
    from oracle import ref_python as RP
    if RP.available():
        ns = RP.load("ref")
        vf = v.detach().float().requires_grad_(True)
        theirs = ns.utils.laplacian_smooth_loss(vf, faces)
        g_theirs, = torch.autograd.grad(theirs, vf)
        lap32 = UniformLaplacian(faces, V)
        v32 = v.detach().float().requires_grad_(True)
        mine = lap32(v32)
        g_mine, = torch.autograd.grad(mine, v32)
        assert torch.allclose(mine, theirs, rtol=1e-5) and torch.allclose(g_mine, g_theirs, rtol=1e-4, atol=1e-6)


def test_plain_lambda_lr_equals_torch_lambda_lr():
    """trainer.LambdaLR (main.py:239's schedule without torch's per-step Python overhead) against torch.optim.lr_scheduler.LambdaLR:
    same learning rates for every parameter group over the warm-up and the decay, same state after load_state_dict."""
    import torch
    from nerf2mesh_amd.trainer import LambdaLR
    iters = 3000
    f = lambda it: 0.01 + 0.99 * (it / 500) if it <= 500 else 0.1 ** ((it - 500) / (iters - 500))
    def make():
        p = [torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(3))]
        return torch.optim.Adam([{"params": [p[0]], "lr": 1e-2}, {"params": [p[1]], "lr": 3e-4}])
    a, b = make(), make()
    sa, sb = torch.optim.lr_scheduler.LambdaLR(a, f), LambdaLR(b, f)
    for it in range(1200):
        for ga, gb in zip(a.param_groups, b.param_groups):
            assert abs(ga["lr"] - gb["lr"]) <= 1e-15 + 1e-12 * ga["lr"], (it, ga["lr"], gb["lr"])
        a.step(); sa.step(); sb.step()
    assert sb.get_last_lr() == [g["lr"] for g in b.param_groups]
    c = make()
    sc = LambdaLR(c, f)
    sc.load_state_dict(sb.state_dict())
    assert [g["lr"] for g in c.param_groups] == [g["lr"] for g in b.param_groups]


def test_bench_watchdog_ends_a_hung_phase_with_an_error_line():
    """bench.py --gpus N must end in a JSON line, not in a hang (RCCL / IPC set-up has never run on an 8-GPU node): a phase that exceeds its limit
    makes the watchdog thread print ONE line with an `error` field naming the phase on rank 0 and end the process (exit code 4)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "import bench\n"
            "bench._start_watchdog()\n"
            "bench._phase('process group set-up + first collective', 0.5)\n"
            "time.sleep(30)\n"                       # the 'hung collective'
            "print('not reached')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, RANK="0", WORLD_SIZE="8"))
    assert r.returncode == 4, (r.returncode, r.stderr[-500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "not reached" not in r.stdout
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 8 and "process group set-up" in d["error"] and d["metric"] == "train_samples_per_sec"
    # a rank other than 0 ends too, silently
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, RANK="3", WORLD_SIZE="8"))
    assert r.returncode == 4 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_peer_store_chunks_are_aligned_at_every_world_size():
    """Peer-store mode pads the coarse half's chunks to a multiple of four rows (engine.peer_chunk_rows): at the standard table split / W is
    481 390 (W = 4) and 240 695 (W = 8) -- not multiples of four, the second one odd -- which kept n2m_adam_step_peer's fused form (16-byte row
    pairs) off for those world sizes until round 6.  Every chunk must start on a multiple of four rows, hold a multiple of four, and the chunks
    must tile [0, split) exactly."""
    from nerf2mesh_amd.engine import peer_chunk_rows
    from nerf2mesh_amd.gridencoder import level_offsets
    off = level_offsets(3, 16, float(np.exp2(np.log2(2048 / 16) / 15)), 16, 19, False)
    split = off[8]
    assert split == 1925560
    for W in (2, 3, 4, 8):
        cs = peer_chunk_rows(split, W)
        assert cs is not None and cs % 4 == 0
        starts = [min(split, r * cs) for r in range(W)]
        lens = [min(cs, split - s0) for s0 in starts]
        assert sum(lens) == split and all(l > 0 and l % 4 == 0 for l in lens) and all(s0 % 4 == 0 for s0 in starts), (W, cs, lens)
        assert (W - 1) * cs < split <= W * cs                      # what n2m_grid_backward_peer_route checks
    assert peer_chunk_rows(split, 8) == 240696 and peer_chunk_rows(split, 4) == 481392 and peer_chunk_rows(split, 2) == 962780
    assert peer_chunk_rows(split + 2, 4) is None                   # a half that is not a multiple of four rows has no such layout
    assert peer_chunk_rows(split, 2, pad=52) == 962832 and peer_chunk_rows(1000, 2, pad=600) is None
