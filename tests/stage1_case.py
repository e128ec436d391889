"""BASELINE config 3 as a scripted case: one stage-1 iteration on a small mesh, written against the model API the reference's
NeRFNetwork (stage 1) and nerf2mesh_amd.network.NeRFNetwork share.

`run_case(model, ...)` is executed
  * by tests/golden/make_golden_stage1.py on the UNCHANGED reference Python (nerf/renderer.py:816-921 `render_stage1`, :924-943
    `update_triangles_errors`, :947-981 `mark_unseen_triangles`; mesh loading :123-165) over the scalar C rasteriser
    (oracle/nvdiffrast_oracle.py, forward only) and the reference's own grid kernels compiled for the host (oracle/_ref), CPU, fp32
    -> tests/golden/render_stage1.npz;
  * by tests/test_stage1_reference.py on (a) the unchanged reference Python over the HIP facade (backends/nvdiffrast/torch.py,
    backends/torch_scatter.py, backends/_gridencoder.py) and (b) nerf2mesh_amd's restated renderer, both on the GPU, with gradients.

nvdiffrast itself is un-vendored: rasterize / interpolate / antialias stay PARITY-UNPINNED; what this case pins is everything the caller
does around them.
"""
import contextlib
import os

import numpy as np
import torch

from nerf2mesh_amd import synthetic as S

import render_case as RC

H0 = W0 = 64                   # the view (ssaa 2 -> rasterised at 128 x 128)
FACES = 1500                   # target face count of the box-scene mesh
N_VIS = 4                      # cameras of the visibility vote
VIS_RES = 96


def mesh():
    v, f = S.scene_mesh(FACES)
    return v.numpy().astype(np.float32), f.numpy().astype(np.int32)


def write_workspace(root):
    """<root>/mesh_stage0/mesh_0.ply -- what NeRFRenderer.__init__ loads for stage 1 (nerf/renderer.py:137-141)."""
    from nerf2mesh_amd import export
    os.makedirs(os.path.join(root, "mesh_stage0"), exist_ok=True)
    v, f = mesh()
    export.write_ply(os.path.join(root, "mesh_stage0", "mesh_0.ply"), v, f)
    return v, f


def view(cam, H=H0, W=W0):
    """rays + mvp of one camera at H x W: nerf/utils.py:242-290 (get_rays), nerf/provider.py:266-276 (projection @ inverse(pose))."""
    poses, _ = RC.cameras()
    focal = S.LEGO_FOCAL * H / S.LEGO_HW
    pix = torch.arange(H * W)
    o, d = S.rays_from_pixels(poses, torch.full_like(pix, cam), pix, H, W, focal)
    return o, d, S.mvp_matrix(poses[cam], H, W, focal)


def targets(n):
    g = torch.Generator().manual_seed(4242)
    return torch.rand(n, 3, generator=g), torch.rand(n, 4, generator=g), (torch.rand(n, 3, generator=g) * 2 - 1) * 2e-3


def run_case(model, device, grad, laplacian=None, ctx=contextlib.nullcontext):
    """One stage-1 iteration as nerf/utils.py:708-721 scripts it (render_stage1 -> per-pixel loss -> update_triangles_errors -> mean
    [-> + Laplacian -> backward]) followed by the visibility vote of export_stage0 (mark_unseen_triangles).  Returns numpy arrays."""
    out = {}
    n = H0 * W0
    model.train()
    with ctx():
        o, d, mvp = view(1)
        o, d, mvp = o.to(device), d.to(device), mvp.to(device)
        bg, rgba, off = (t.to(device) for t in targets(n))
        V = model.vertices.shape[0]
        with torch.no_grad():                       # vertices + offsets is what gets rendered: start from a seeded non-zero offset
            model.vertices_offsets.copy_(targets(V)[2].to(device))
        gt_mask = rgba[:, 3:]
        gt_rgb = rgba[:, :3] * gt_mask + bg * (1 - gt_mask)
        with contextlib.nullcontext() if grad else torch.no_grad():
            res = model.render_stage1(o, d, mvp, H0, W0, index=None, bg_color=bg, shading="full")
            loss = ((res["image"] - gt_rgb) ** 2).mean(-1)                                            # lambda_rgb = 1
            loss = loss + 0.1 * (res["weights_sum"].view(-1) - gt_mask.view(-1)) ** 2                 # lambda_mask = 0.1
        out["trig_id"] = model.triangles_errors_id.detach().cpu().numpy().astype(np.int32)
        model.update_triangles_errors(loss.detach())
        out["image"] = res["image"].detach().cpu().numpy()
        out["depth"] = res["depth"].detach().cpu().numpy()
        out["weights_sum"] = res["weights_sum"].detach().cpu().numpy()
        out["triangles_errors"] = model.triangles_errors.detach().cpu().numpy().copy()
        out["triangles_errors_cnt"] = model.triangles_errors_cnt.detach().cpu().numpy().copy()
        out["loss"] = np.float32(loss.mean().item())
        if grad:
            total = loss.mean()
            if laplacian is not None:
                lap = laplacian(model.vertices + model.vertices_offsets, model.triangles)
                out["laplacian"] = np.float32(lap.item())
                total = total + 0.001 * lap                                                           # lambda_lap (main.py:85)
            for p in model.parameters():
                p.grad = None
            total.backward()
            for name, p in model.named_parameters():
                if p.grad is None:
                    continue
                g = p.grad.detach().float().cpu().numpy()
                if "embeddings" in name:
                    out["grad_sum." + name] = np.float64(np.abs(g.astype(np.float64)).sum())
                    out["grad_head." + name] = g[:65536].copy()
                else:
                    out["grad." + name] = g
        # visibility vote over N_VIS cameras at VIS_RES^2 (nerf/renderer.py:947-981; called from export_stage0 :548)
        v_np, f_np = mesh()
        mvps = torch.stack([view(c, VIS_RES, VIS_RES)[2] for c in range(N_VIS)])
        unseen = model.mark_unseen_triangles(v_np, f_np, mvps, VIS_RES, VIS_RES)
        out["unseen"] = np.packbits(unseen.detach().cpu().numpy().astype(bool))
        out["unseen_count"] = np.int64(int(unseen.sum()))
    return out
