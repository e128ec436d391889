"""trunc_exp: exp in fp32 whose backward clamps the argument to [-15, 15] (reference activation.py:5-17)."""
import torch
from torch.autograd import Function


class _trunc_exp(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.float()
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


def trunc_exp(x):
    # run outside autocast so the exp is evaluated in fp32, like custom_fwd(cast_inputs=torch.float32)
    with torch.autocast(device_type=x.device.type, enabled=False):
        return _trunc_exp.apply(x)
