"""`_shencoder` for ROCm: the function table of shencoder/src/bindings.cpp:5-8 over libn2m_hip.so."""
import torch

from nerf2mesh_amd import _lib as L

_p = L.ptr


def _f32(**ts):
    for k, t in ts.items():
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"{k} must be a float32 tensor (sphere_harmonics.py:16 casts inputs to float32)")


def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
    L.check_cuda(inputs=inputs, outputs=outputs, dy_dx=dy_dx)
    _f32(inputs=inputs, outputs=outputs, dy_dx=dy_dx)
    L.call("n2m_sh_encode_forward", _p(inputs), _p(outputs), B, D, C, _p(dy_dx), L.stream())


def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
    L.check_cuda(grad=grad, inputs=inputs, dy_dx=dy_dx, grad_inputs=grad_inputs)
    _f32(grad=grad, inputs=inputs, dy_dx=dy_dx, grad_inputs=grad_inputs)
    L.call("n2m_sh_encode_backward", _p(grad), _p(inputs), B, D, C, _p(dy_dx), _p(grad_inputs), L.stream())
