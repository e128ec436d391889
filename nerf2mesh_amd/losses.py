"""Fused loss heads of the training step (C ABI: include/n2m_hip.h, "training-step helpers")."""
import os

import torch
from torch.autograd import Function

from . import _lib as L

_p = L.ptr
_TICKETS = {}


def _ticket(device):
    t = _TICKETS.get(device)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=device)      # the kernel leaves it zero again
        _TICKETS[device] = t
    return t


class _photo_loss(Function):
    @staticmethod
    def forward(ctx, image, weights_sum, gt_rgba, bg, lambda_rgb, lambda_mask):
        image, weights_sum, gt_rgba = image.float().contiguous(), weights_sum.float().contiguous(), gt_rgba.float().contiguous()
        N = image.shape[0]
        bg_t, bg_s = (bg.float().contiguous(), 0.0) if torch.is_tensor(bg) else (None, float(bg))
        if bg_t is not None and bg_t.shape != (N, 3):
            bg_t = bg_t.expand(N, 3).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=image.device)
        partial = torch.empty((N + 255) // 256, dtype=torch.float32, device=image.device)
        L.call("n2m_photo_loss_forward", _p(image), _p(weights_sum), _p(gt_rgba), _p(bg_t), bg_s, float(lambda_rgb), float(lambda_mask), N,
               _p(partial), _p(_ticket(image.device)), _p(loss), L.stream())
        ctx.save_for_backward(image, weights_sum, gt_rgba, bg_t)
        ctx.cfg = (bg_s, float(lambda_rgb), float(lambda_mask), N)
        return loss.view(())

    @staticmethod
    def backward(ctx, grad_loss):
        image, weights_sum, gt_rgba, bg_t = ctx.saved_tensors
        bg_s, lr, lm, N = ctx.cfg
        g = grad_loss.float().contiguous()
        d_image = torch.empty_like(image)
        d_ws = torch.empty_like(weights_sum)
        L.call("n2m_photo_loss_backward", _p(image), _p(weights_sum), _p(gt_rgba), _p(bg_t), bg_s, lr, lm, N, _p(g), _p(d_image), _p(d_ws),
               L.stream())
        return d_image, d_ws, None, None, None, None


def photo_loss(image, weights_sum, gt_rgba, bg, lambda_rgb=1.0, lambda_mask=0.0):
    """mean over rays of lambda_rgb * mse(image + (1-weights_sum)*bg, gt.rgb*gt.a + bg*(1-gt.a)).mean(-1) + lambda_mask * (weights_sum - gt.a)^2
    -- the stage-0 loss of nerf/utils.py:658-683 with the background blend of nerf/renderer.py:747 folded in.
    `image` is the composited colour BEFORE that blend; bg is a float (uniform) or an [N,3] tensor."""
    return _photo_loss.apply(image, weights_sum, gt_rgba, bg, lambda_rgb, lambda_mask)


class _stage1_head(Function):
    """n2m_stage1_head: clamp / alpha * rgb / depth / T / ssaa reduction / background blend / per-pixel loss and its mean in one launch;
    the gradient of the mean w.r.t. the two antialias outputs is formed in the same pass (it does not depend on the loss value) and
    scaled by the incoming gradient in backward."""

    @staticmethod
    def forward(ctx, aa_alpha, aa_rgb, rast, gt_rgba, bg, h0, w0, ssaa, lambda_rgb, lambda_mask, tri_err=None, tri_cnt=None, seed=None):
        # aa_alpha is None: aa_rgb is the [1, h, w, 4] output of ONE antialias call on RGB + alpha
        dev = aa_rgb.device
        packed = aa_alpha is None
        aa_rgb, rast = aa_rgb.float().contiguous(), rast.float().contiguous()
        aa_alpha = None if packed else aa_alpha.float().contiguous()
        gt_rgba = gt_rgba.float().contiguous()
        N = h0 * w0
        bg_t, bg_s = (bg.float().reshape(N, 3).contiguous(), 0.0) if torch.is_tensor(bg) else (None, float(bg))
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        image, depth, ws, trig, loss_px = f(N, 3), f(N), f(N), f(N), f(N)
        partial = f((N + 255) // 256)
        need = any(ctx.needs_input_grad[:2])
        if packed:
            d_rgb = torch.empty_like(aa_rgb) if need else None
            pa, pda = aa_rgb.data_ptr() + 12, (d_rgb.data_ptr() + 12 if need else None)
            d_alpha = None
        else:
            d_alpha, d_rgb = (torch.empty_like(aa_alpha), torch.empty_like(aa_rgb)) if need else (None, None)
            pa, pda = _p(aa_alpha), _p(d_alpha)
        L.call("n2m_stage1_head", pa, _p(aa_rgb), _p(rast), int(h0), int(w0), int(ssaa), _p(gt_rgba), _p(bg_t), bg_s, float(lambda_rgb),
               float(lambda_mask), _p(image), _p(depth), _p(ws), _p(trig), _p(loss_px), pda, _p(d_rgb), _p(partial), _p(tri_err), _p(tri_cnt),
               int(packed), _p(seed), None, L.stream())
        ctx.grads = (d_alpha, d_rgb)
        ctx.seeded = seed is not None
        ctx.seed_ref = seed
        loss = partial.sum() / N
        ctx.mark_non_differentiable(image, depth, ws, trig, loss_px)
        return loss, image, depth, ws, trig, loss_px

    @staticmethod
    def backward(ctx, g, *unused):
        d_alpha, d_rgb = ctx.grads
        ctx.grads = None
        if ctx.seeded:
            # the kernel has applied the incoming gradient already -- which is only right when `g` IS the seed: the head loss enters the total
            # with weight 1 and the total is differentiated with gradient = seed.  N2M_DEBUG_SEED=1 checks it (a host sync per step).
            if os.environ.get("N2M_DEBUG_SEED") and ctx.seed_ref is not None:
                assert torch.allclose(g.float().reshape(()), ctx.seed_ref.float().reshape(())), "stage1_head: the incoming gradient is not the seed it was given"
            return d_alpha, d_rgb, None, None, None, None, None, None, None, None, None, None, None
        return (d_alpha * g if d_alpha is not None else None), d_rgb * g, None, None, None, None, None, None, None, None, None, None, None


def stage1_head(aa_alpha, aa_rgb, rast, gt_rgba, bg, h0, w0, ssaa, lambda_rgb=1.0, lambda_mask=0.0, tri_err=None, tri_cnt=None, seed=None):
    """(loss, image [N,3], depth [N], weights_sum [N], trig_id [N] float, loss_px [N]) of one stage-1 view from the two antialias outputs
    (before their clamp) and the rasteriser's image: nerf/renderer.py:886-913 + the loss of nerf/utils.py:708-721, N = h0 * w0.
    tri_err / tri_cnt [faces] f32: `update_triangles_errors` (nerf/renderer.py:924-943) done by the same launch.
    aa_alpha = None: aa_rgb is the [1, h, w, 4] output of ONE antialias call on the RGB + alpha image (same values per channel).
    seed (device scalar): the gradient that WILL flow into the returned loss -- the loss scale, when the loss enters the total with weight 1 and
    the total is differentiated with `gradient = seed` (FusedAdamAMP.backward); the kernel then writes the scaled gradients itself."""
    return _stage1_head.apply(aa_alpha, aa_rgb, rast, gt_rgba, bg, h0, w0, ssaa, lambda_rgb, lambda_mask, tri_err, tri_cnt, seed)


class _gather_rows(Function):
    """x[idx] for a [N, C] fp32 array and an int64 index list (n2m_gather_rows); backward scatters into zeros (idx unique)."""

    @staticmethod
    def forward(ctx, x, idx):
        x = x.float().contiguous()
        out = torch.empty(idx.shape[0], x.shape[1], dtype=torch.float32, device=x.device)
        L.call("n2m_gather_rows", _p(x), _p(idx), idx.shape[0], x.shape[1], _p(out), L.stream())
        ctx.save_for_backward(idx)
        ctx.n = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        g = g.float().contiguous()
        d = torch.zeros(ctx.n, g.shape[1], dtype=torch.float32, device=g.device)
        L.call("n2m_scatter_rows", _p(g), _p(idx), idx.shape[0], g.shape[1], _p(d), L.stream())
        return d, None


class _scatter_rows(Function):
    """zeros(N, C) with rows idx set to src (n2m_scatter_rows; idx unique); backward gathers."""

    @staticmethod
    def forward(ctx, src, idx, n):
        src = src.float().contiguous()
        out = torch.zeros(n, src.shape[1], dtype=torch.float32, device=src.device)
        L.call("n2m_scatter_rows", _p(src), _p(idx), idx.shape[0], src.shape[1], _p(out), L.stream())
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        g = g.float().contiguous()
        d = torch.empty(idx.shape[0], g.shape[1], dtype=torch.float32, device=g.device)
        L.call("n2m_gather_rows", _p(g), _p(idx), idx.shape[0], g.shape[1], _p(d), L.stream())
        return d, None, None


def gather_rows(x, idx):
    """x[idx] ([N, C] fp32, idx int64 [K]) -- the covered pixels of a stage-1 frame (nerf/renderer.py:875-881)."""
    return _gather_rows.apply(x, idx)


def scatter_rows(src, idx, n):
    """[n, C] zeros with rows idx (unique) set to src -- `rgbs[mask] = mask_rgbs` (nerf/renderer.py:881)."""
    return _scatter_rows.apply(src, idx, n)
