"""`_gridencoder` for ROCm: the function table of gridencoder/src/bindings.cpp:5-9 over libn2m_hip.so.

Checks mirror grid_encode_forward/backward's TORCH_CHECKs (gridencoder.cu:448-464,473-495): RuntimeError for
non-device / non-contiguous / wrongly-typed tensors and for unsupported C or D (":380,397").
"""
import torch

from nerf2mesh_amd import _lib as L

_p = L.ptr


def _dtype_id(t, name):
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.float16:
        return L.F16
    raise RuntimeError(f"{name} must be a float32 or float16 tensor (got {t.dtype})")


def _check(inputs, embeddings, offsets, **more):
    L.check_cuda(inputs=inputs, embeddings=embeddings, offsets=offsets, **more)
    if inputs.dtype != torch.float32:
        raise RuntimeError("inputs must be a float32 tensor")
    if offsets.dtype != torch.int32:
        raise RuntimeError("offsets must be an int tensor")


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L_, max_level, S, H, dy_dx, gridtype, align_corners,
                        interp):
    _check(inputs, embeddings, offsets, outputs=outputs, dy_dx=dy_dx)
    dt = _dtype_id(embeddings, "embeddings")
    if outputs.dtype != embeddings.dtype or (dy_dx is not None and dy_dx.dtype != embeddings.dtype):
        raise RuntimeError("outputs / dy_dx must have the dtype of embeddings")
    L.call("n2m_grid_encode_forward", _p(inputs), _p(embeddings), _p(offsets), _p(outputs), B, D, C, L_, max_level, float(S), H,
           _p(dy_dx), gridtype, int(bool(align_corners)), interp, dt, L.stream())


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L_, max_level, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners, interp):
    _check(inputs, embeddings, offsets, grad=grad, grad_embeddings=grad_embeddings, dy_dx=dy_dx, grad_inputs=grad_inputs)
    dt = _dtype_id(grad, "grad")   # the reference dispatches on grad's dtype (gridencoder.cu:497-498)
    if grad_embeddings.dtype != grad.dtype:
        raise RuntimeError("grad_embeddings must have the dtype of grad")
    L.call("n2m_grid_encode_backward", _p(grad), _p(inputs), _p(embeddings), _p(offsets), _p(grad_embeddings), B, D, C, L_,
           max_level, float(S), H, _p(dy_dx), _p(grad_inputs), gridtype, int(bool(align_corners)), interp, dt, L.stream())


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L_, S, H, gridtype, align_corners):
    L.check_cuda(inputs=inputs, embeddings=embeddings, grad=grad, offsets=offsets)
    dt = _dtype_id(embeddings, "embeddings")
    if inputs.dtype != embeddings.dtype or grad.dtype != embeddings.dtype:
        raise RuntimeError("inputs and grad must have the dtype of embeddings (the kernel reads all three as scalar_t)")
    L.call("n2m_grad_total_variation", _p(inputs), _p(embeddings), _p(grad), _p(offsets), weight, B, D, C, L_, float(S), H,
           gridtype, int(bool(align_corners)), dt, L.stream())
