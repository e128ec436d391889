"""`_freqencoder` for ROCm: the function table of freqencoder/src/bindings.cpp:5-8 over libn2m_hip.so."""
import torch

from nerf2mesh_amd import _lib as L

_p = L.ptr


def _f32(**ts):
    for k, t in ts.items():
        if t.dtype != torch.float32:
            raise RuntimeError(f"{k} must be a float32 tensor (freq.py:17 casts inputs to float32)")


def freq_encode_forward(inputs, B, D, deg, C, outputs):
    L.check_cuda(inputs=inputs, outputs=outputs)
    _f32(inputs=inputs, outputs=outputs)
    L.call("n2m_freq_encode_forward", _p(inputs), B, D, deg, C, _p(outputs), L.stream())


def freq_encode_backward(grad, outputs, B, D, deg, C, grad_inputs):
    L.check_cuda(grad=grad, outputs=outputs, grad_inputs=grad_inputs)
    _f32(grad=grad, outputs=outputs, grad_inputs=grad_inputs)
    L.call("n2m_freq_encode_backward", _p(grad), _p(outputs), B, D, deg, C, _p(grad_inputs), L.stream())
