"""Fused loss heads of the training step (C ABI: include/n2m_hip.h, "training-step helpers")."""
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
