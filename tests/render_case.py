"""BASELINE config 1 as a scripted case: one occupancy upkeep + one 64x64-crop render iteration (forward, backward, eval render),
written against the model API the reference's NeRFNetwork and nerf2mesh_amd.network.NeRFNetwork share.

`run_case(model, ...)` is executed
  * by tests/golden/make_golden_render.py on the UNCHANGED reference Python (nerf/renderer.py:676-813 `render`, :1074-1149
    `update_extra_state`, :985-1071 `mark_untrained_grid`, nerf/network.py:81-189, SDF branch :135-156 / renderer :724-739) over the
    reference's own kernels compiled for the host (oracle/_ref), CPU tensors, fp32  ->  tests/golden/render_*.npz;
  * by tests/test_reference_render.py on (a) nerf2mesh_amd's restated renderer/network and (b) the unchanged reference Python over the
    HIP `_backend` modules, both on the GPU, and compared with those fixtures.

Everything random is drawn from seeded CPU generators (weights, the occupancy refresh's `rand_like` jitter) so that CPU and GPU
runs consume identical numbers.
"""
import contextlib
import math
import types

import numpy as np
import torch

from nerf2mesh_amd import synthetic as S

GRID = 128
N_CAMS = 8
CROP = 64
LEVEL0_ROWS = 4920          # (16+1)^3 = 4913 rounded up to a multiple of 8 (gridencoder/grid.py:127-133)


# BASELINE config 4's shape (scripts/runall_360_outdoor.sh:2: --bound 16 --enable_cam_near_far --lambda_entropy 1e-3, dt_gamma 1/256 default):
# 5 cascades, 6 837 544 table rows, a colmap-style AABB tighter than the bound, per-camera near/far, entropy loss, inner/outer TV
GARDEN = dict(bound=16.0, rows=6837544, aabb=[-6.0, -5.0, -4.0, 7.0, 6.0, 5.0], dt_gamma=1.0 / 256, lambda_entropy=1e-3, lambda_tv=1e-3)


def make_state(sdf, rows=6119864, seed=1234, garden=False):
    """Deterministic parameters (reference key names).  The density head is shaped by hand so that the occupancy grid is
    sparse like a trained scene: level 0 of the density table is the constant 1 (a bias the bias-free MLP lacks), six hidden units
    carry relu(+-x_i), so the head sees |x|+|y|+|z|; all remaining weights and table rows are random."""
    g = torch.Generator().manual_seed(seed)

    def u(*shape, scale=1.0):
        return (torch.rand(*shape, generator=g) * 2 - 1) * scale

    sd = {}
    emb = u(rows, 1, scale=0.5)
    emb[:LEVEL0_ROWS] = 1.0
    sd["encoder.embeddings"] = emb
    sd["encoder_color.embeddings"] = u(rows, 2, scale=0.5)
    w0 = u(32, 19, scale=0.2)
    w0[:7] = 0
    for i in range(3):
        w0[2 * i, i] = 1.0
        w0[2 * i + 1, i] = -1.0
    w0[6, 3] = 1.0                                   # input 3 = level-0 feature = 1
    w1 = u(1, 32, scale=0.05 if sdf else 0.3)
    if sdf:
        w1[0, :6] = 1.0
        w1[0, 6] = -0.4                              # sdf ~ |x|_1 - 0.4 + noise
    elif garden:
        w1[0, :6] = -1.2
        w1[0, 6] = 4.0                               # log sigma ~ 4 - 1.2 |x|_1 + noise: occupied out to |x|_1 ~ 4, i.e. cascades 0..2
    else:
        w1[0, :6] = -12.0
        w1[0, 6] = 6.0                               # log sigma ~ 6 - 12 |x|_1 + noise
    sd["sigma_net.net.0.weight"], sd["sigma_net.net.1.weight"] = w0, w1
    sd["color_net.net.0.weight"] = u(64, 35, scale=1 / math.sqrt(35))
    sd["color_net.net.1.weight"] = u(64, 64, scale=1 / 8)
    sd["color_net.net.2.weight"] = u(6, 64, scale=1 / 2)       # wide logits: colours spread over (0, 1)
    sd["specular_net.net.0.weight"] = u(32, 6, scale=1 / math.sqrt(6))
    sd["specular_net.net.1.weight"] = -u(3, 32, scale=2 / math.sqrt(32)).abs()      # specular mostly < 0.5: the clamp at 1 bites on some samples only
    if sdf:
        sd["variance"] = torch.tensor(0.5)
    return sd


def cameras():
    poses = S.make_cameras(N_CAMS, seed=0)
    intr = np.array([S.LEGO_FOCAL, S.LEGO_FOCAL, S.LEGO_HW / 2, S.LEGO_HW / 2], dtype=np.float32)
    return poses, intr


def cam_near_far():
    """[N_CAMS, 2] per-camera (near, far) as colmap_provider.py:270 derives them from the sparse points (seeded stand-in)."""
    g = torch.Generator().manual_seed(31)
    return torch.stack([1.2 + 0.6 * torch.rand(N_CAMS, generator=g), 4.5 + 1.5 * torch.rand(N_CAMS, generator=g)], dim=1)


def _morton_of_meshgrid(H):
    """morton index of the i-th cell of custom_meshgrid(arange(H) x3) flattened (nerf/renderer.py:1097-1099)."""
    from oracle import oracle as orc
    ax = np.arange(H, dtype=np.int32)
    xx, yy, zz = np.meshgrid(ax, ax, ax, indexing="ij")
    coords = np.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1).astype(np.int32)
    return torch.from_numpy(orc.morton3D(coords).astype(np.int64))


@contextlib.contextmanager
def seeded_jitter(seed, order, H=GRID):
    """Replaces torch.rand_like (the occupancy refresh's jitter, nerf/renderer.py:1110) with draws from a seeded CPU generator.
    order="meshgrid": the i-th row belongs to the i-th meshgrid cell (reference loop order);
    order="morton":   the same numbers re-ordered for a caller that walks the cells in Morton order (nerf2mesh_amd/renderer.py)."""
    g = torch.Generator().manual_seed(seed)
    perm = _morton_of_meshgrid(H) if order == "morton" else None
    orig = torch.rand_like

    def rand_like(x, **kw):
        n = torch.rand(x.shape, generator=g, dtype=torch.float32)
        if perm is not None:
            out = torch.empty_like(n)
            out[perm] = n
            n = out
        return n.to(x.device)
    torch.rand_like = rand_like
    try:
        yield
    finally:
        torch.rand_like = orig


def loss_weights(n):
    g = torch.Generator().manual_seed(77)
    return torch.rand(n, 3, generator=g), torch.rand(n, generator=g), torch.rand(n, generator=g)


def run_case(model, mark_untrained, jitter_order, device, sdf=False, bitfield_override=None, ctx=contextlib.nullcontext, garden=False):
    """The scripted iteration.  `mark_untrained(model, poses, intrinsics[, cam_near_far])` adapts the two signatures of
    mark_untrained_grid.  garden=True adds what BASELINE config 4 adds to the iteration: update_aabb (main.py:234-235), the per-ray
    cam_near_far clamp (nerf/renderer.py:689-691), dt_gamma 1/256, the entropy loss on weights / weights_sum (nerf/utils.py:728-733: the
    only source of `grad_weights` in composite_rays_train's backward) and the inner / outer TV split of post_train_step
    (nerf/utils.py:812-821).  Returns a dict of numpy arrays (what the fixtures hold)."""
    out = {}
    poses, intr = cameras()
    model.train()
    dt_gamma = GARDEN["dt_gamma"] if garden else 0
    with ctx():
        if garden:
            model.update_aabb(np.asarray(GARDEN["aabb"], dtype=np.float32))
            out["aabb_train"] = model.aabb_train.detach().cpu().numpy().copy()
            mark_untrained(model, poses, intr, cam_near_far().to(device))
        else:
            mark_untrained(model, poses, intr)
        out["untrained"] = np.packbits((model.density_grid.detach() < 0).cpu().numpy().reshape(-1))
        for k, seed in enumerate((100, 101)):
            with seeded_jitter(seed, jitter_order):
                model.update_extra_state()
        out["density_grid"] = model.density_grid.detach().cpu().numpy().copy()
        out["mean_density"] = np.float32(model.mean_density)
        out["density_bitfield"] = model.density_bitfield.detach().cpu().numpy().copy()
        if bitfield_override is not None:            # march on the fixture's bit field: integer outputs must then match exactly
            model.density_bitfield.copy_(torch.from_numpy(bitfield_override).to(model.density_bitfield.device))

        o, d = S.crop_rays(poses, cam=0, size=CROP)
        o, d = o.to(device), d.to(device)
        extra = {}
        if garden:      # per-ray [N, 2] as random_image_batch hands them out (colmap_provider.py:563-565); a seeded spread around camera 0's pair
            g = torch.Generator().manual_seed(32)
            cnf = cam_near_far()[0].unsqueeze(0) + (torch.rand(o.shape[0], 2, generator=g) - 0.5) * torch.tensor([0.4, 1.0])
            extra["cam_near_far"] = cnf.to(device).contiguous()
        res = model.render(o, d, dt_gamma=dt_gamma, bg_color=1, perturb=False, max_steps=1024, shading="full", **extra)
        out["num_points"] = np.int64(res["num_points"])
        out["image"] = res["image"].detach().cpu().numpy()
        out["depth"] = res["depth"].detach().cpu().numpy()
        out["weights_sum"] = res["weights_sum"].detach().cpu().numpy()
        out["xyzs_head"] = res["xyzs"][:4096].detach().cpu().numpy()
        wi, ww, wd = (t.to(device) for t in loss_weights(o.shape[0]))
        loss = (res["image"] * wi).sum() + (res["weights_sum"] * ww).sum() + (res["depth"] * wd).sum()
        if garden:      # nerf/utils.py:728-733, weighted up so that grad_weights is a visible share of the sample gradients
            w = res["weights"].clamp(1e-5, 1 - 1e-5)
            w2 = res["weights_sum"].clamp(1e-5, 1 - 1e-5)
            ent = lambda p: (-p * torch.log2(p) - (1 - p) * torch.log2(1 - p)).mean()
            out["entropy"] = np.float32((ent(w) + ent(w2)).item())
            out["weights_head"] = res["weights"][:4096].detach().cpu().numpy()
            loss = loss + (GARDEN["lambda_entropy"] * 1e3 * o.shape[0]) * (ent(w) + ent(w2))
        if sdf:
            out["normal_head"] = res["normal"][:4096].detach().cpu().numpy()
            loss = loss + 0.1 * ((res["normal"].norm(dim=-1) - 1) ** 2).mean() * o.shape[0]     # eikonal term, nerf/utils.py:740-743
        for p in model.parameters():
            p.grad = None
        loss.backward()
        out["loss"] = np.float32(loss.item())
        if garden:      # post_train_step, nerf/utils.py:812-821: in-place TV on the density table's gradient, outer samples ten-fold
            xyzs = res["xyzs"].detach()
            inner = xyzs.abs().amax(dim=-1) <= 1
            out["tv_inner"] = np.int64(int(inner.sum()))
            model.encoder.grad_total_variation(GARDEN["lambda_tv"], xyzs[inner].contiguous(), model.bound)
            model.encoder.grad_total_variation(GARDEN["lambda_tv"] * 10, xyzs[~inner].contiguous(), model.bound)
        for name, p in model.named_parameters():
            if p.grad is None:
                continue
            gnp = p.grad.detach().float().cpu().numpy()
            if "embeddings" in name:
                out["grad_sum." + name] = np.float64(np.abs(gnp.astype(np.float64)).sum())
                out["grad_head." + name] = gnp[:65536].copy()           # levels 0-2 and the start of level 3: dense, all touched
                nz = np.flatnonzero(np.abs(gnp).sum(-1))
                out["grad_nnz." + name] = np.int64(nz.size)
            else:
                out["grad." + name] = gnp

        model.eval()
        with torch.no_grad():
            res = model.render(o, d, dt_gamma=dt_gamma, bg_color=1, perturb=False, max_steps=1024, shading="full", **extra)
        out["eval_image"] = res["image"].detach().cpu().numpy()
        out["eval_depth"] = res["depth"].detach().cpu().numpy()
        model.train()
    return out


def compress_for_fixture(out, stride=32):
    """What is committed: everything except the full density grid (8 MB; 40 MB with 5 cascades) -- a strided subsample of it instead."""
    fx = dict(out)
    grid = fx.pop("density_grid")
    fx["density_grid_stride"] = np.int64(stride)
    fx["density_grid_sub"] = grid.reshape(-1)[::stride].copy()
    fx["density_grid_sum"] = np.float64(np.clip(grid, 0, None).astype(np.float64).sum())
    fx["density_grid_neg"] = np.int64((grid < 0).sum())
    return fx


def dataset_stub(poses, intr, cam_near_far=None):
    """The attributes mark_untrained_grid reads from the reference's dataset object (nerf/renderer.py:989-991)."""
    ns = types.SimpleNamespace(poses=poses, intrinsics=intr)
    if cam_near_far is not None:
        ns.cam_near_far = cam_near_far
    return ns
