"""Frequency encoding operator on libn2m_hip.so -- the host-side mirror of the reference's freqencoder/freq.py
(`freq_encode` autograd Function :15-54, `FreqEncoder` module :59-83)."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L

_p = L.ptr


class _freq_encoder(Function):
    @staticmethod
    def forward(ctx, inputs, degree, output_dim):
        inputs = (inputs if inputs.is_cuda else inputs.cuda()).float().contiguous()      # custom_fwd(cast_inputs=float32), freq.py:17
        B, D = inputs.shape
        outputs = torch.empty(B, output_dim, dtype=torch.float32, device=inputs.device)
        L.call("n2m_freq_encode_forward", _p(inputs), B, D, int(degree), int(output_dim), _p(outputs), L.stream())
        ctx.save_for_backward(outputs)
        ctx.dims = (B, D, int(degree), int(output_dim))
        return outputs

    @staticmethod
    def backward(ctx, grad):
        outputs, = ctx.saved_tensors
        B, D, degree, C = ctx.dims
        grad = grad.float().contiguous()
        grad_inputs = torch.empty(B, D, dtype=torch.float32, device=grad.device)
        L.call("n2m_freq_encode_backward", _p(grad), _p(outputs), B, D, degree, C, _p(grad_inputs), L.stream())
        return grad_inputs, None, None


freq_encode = _freq_encoder.apply


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        prefix = list(inputs.shape[:-1])
        out = freq_encode(inputs.reshape(-1, self.input_dim), self.degree, self.output_dim)
        return out.reshape(prefix + [self.output_dim])
