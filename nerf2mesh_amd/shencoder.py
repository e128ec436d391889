"""Spherical-harmonics direction encoder on libn2m_hip.so -- mirror of the reference's shencoder/sphere_harmonics.py."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L

_p = L.ptr


class _sh_encoder(Function):
    @staticmethod
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        inputs = inputs.float().contiguous()               # fp32 for precision (sphere_harmonics.py:16)
        B, D = inputs.shape
        C2 = degree ** 2
        outputs = torch.empty(B, C2, dtype=torch.float32, device=inputs.device)
        dy_dx = torch.empty(B, D * C2, dtype=torch.float32, device=inputs.device) if calc_grad_inputs else None
        L.call("n2m_sh_encode_forward", _p(inputs), _p(outputs), B, D, int(degree), _p(dy_dx), L.stream())
        ctx.save_for_backward(inputs, dy_dx)
        ctx.cfg = (B, D, int(degree))
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        B, D, degree = ctx.cfg
        grad = grad.float().contiguous()
        grad_inputs = torch.zeros_like(inputs)
        L.call("n2m_sh_encode_backward", _p(grad), _p(inputs), B, D, degree, _p(dy_dx), _p(grad_inputs), L.stream())
        return grad_inputs, None, None


def sh_encode(inputs, degree, calc_grad_inputs=False):
    """unit vectors [B,3] -> real SH basis [B, degree^2], degree in 1..8 (sphere_harmonics.py:14-58)."""
    return _sh_encoder.apply(inputs, degree, calc_grad_inputs)


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        inputs = inputs / torch.norm(inputs, dim=-1, keepdim=True)     # the kernel expects unit vectors (:79-82)
        prefix = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        out = sh_encode(inputs, self.degree, inputs.requires_grad)
        return out.reshape(prefix + [self.output_dim])
