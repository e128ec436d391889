"""TEST INFRASTRUCTURE ONLY -- numpy front-end of the CPU oracle (oracle/n2m_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as
the checker / the timed CPU baseline.  The product package never imports it.

Every function mirrors one entry of the reference `_backend` tables on numpy arrays:
outputs are allocated here and returned (the C side writes in place like the reference kernels do).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libn2m_oracle.so")
_SRCS = [os.path.join(_HERE, "n2m_oracle.c"), os.path.join(_HERE, "n2m_raster_oracle.c")]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in _SRCS):
        subprocess.run(["make", "-C", _HERE, "libn2m_oracle.so"] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


_u32, _i32, _f32, _vp = ctypes.c_uint32, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


def _call(name, *args):
    fn = getattr(lib(), "n2m_oracle_" + name)
    fn.restype = None
    fn(*args)


# ------------------------------------------------------------------------------------------- raymarching

def near_far_from_aabb(rays_o, rays_d, aabb, min_near):
    rays_o, rays_d, aabb = _c(rays_o, np.float32).reshape(-1, 3), _c(rays_d, np.float32).reshape(-1, 3), _c(aabb, np.float32)
    N = rays_o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    _call("near_far_from_aabb", _p(rays_o), _p(rays_d), _p(aabb), _u32(N), _f32(min_near), _p(nears), _p(fars))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    rays_o, rays_d = _c(rays_o, np.float32).reshape(-1, 3), _c(rays_d, np.float32).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), np.float32)
    _call("sph_from_ray", _p(rays_o), _p(rays_d), _f32(radius), _u32(N), _p(coords))
    return coords


def morton3D(coords):
    coords = _c(coords, np.int32).reshape(-1, 3)
    out = np.empty(coords.shape[0], np.int32)
    _call("morton3D", _p(coords), _u32(coords.shape[0]), _p(out))
    return out


def morton3D_invert(indices):
    indices = _c(indices, np.int32).reshape(-1)
    out = np.empty((indices.shape[0], 3), np.int32)
    _call("morton3D_invert", _p(indices), _u32(indices.shape[0]), _p(out))
    return out


def packbits(grid, thresh, bitfield=None):
    grid = _c(grid, np.float32)
    N = grid.size // 8
    if bitfield is None:
        bitfield = np.empty(N, np.uint8)
    _call("packbits", _p(grid), _u32(N), _f32(thresh), _p(bitfield))
    return bitfield


def flatten_rays(rays, M):
    rays = _c(rays, np.int32).reshape(-1, 2)
    res = np.zeros(M, np.int32)
    _call("flatten_rays", _p(rays), _u32(rays.shape[0]), _u32(M), _p(res))
    return res


def march_rays_train(rays_o, rays_d, bound, contract, bitfield, C, H, nears, fars, noises, dt_gamma=0.0,
                     max_steps=1024, counter0=0):
    """Both passes of raymarching/raymarching.py:229-241. Returns xyzs, dirs, ts, rays."""
    rays_o, rays_d = _c(rays_o, np.float32).reshape(-1, 3), _c(rays_d, np.float32).reshape(-1, 3)
    bitfield = _c(bitfield, np.uint8)
    nears, fars, noises = _c(nears, np.float32), _c(fars, np.float32), _c(noises, np.float32)
    N = rays_o.shape[0]
    rays = np.empty((N, 2), np.int32)
    counter = np.array([counter0], np.int32)
    args = (_p(rays_o), _p(rays_d), _p(bitfield), _f32(bound), ctypes.c_int(int(contract)), _f32(dt_gamma),
            _u32(max_steps), _u32(N), _u32(C), _u32(H), _p(nears), _p(fars))
    _call("march_rays_train", *args, None, None, None, _p(rays), _p(counter), _p(noises))
    M = int(counter[0])
    xyzs, dirs, ts = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    _call("march_rays_train", *args, _p(xyzs), _p(dirs), _p(ts), _p(rays), _p(counter), _p(noises))
    return xyzs, dirs, ts, rays


def composite_rays_train_forward(sigmas, rgbs, ts, rays, T_thresh=1e-4, alpha_mode=False):
    sigmas, rgbs, ts, rays = _c(sigmas, np.float32), _c(rgbs, np.float32), _c(ts, np.float32), _c(rays, np.int32)
    M, N = sigmas.shape[0], rays.shape[0]
    weights = np.zeros(M, np.float32)
    weights_sum, depth, image = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    _call("composite_rays_train_forward", _p(sigmas), _p(rgbs), _p(ts), _p(rays), _u32(M), _u32(N), _f32(T_thresh),
          ctypes.c_int(int(alpha_mode)), _p(weights), _p(weights_sum), _p(depth), _p(image))
    return weights, weights_sum, depth, image


def composite_rays_train_backward(grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas, rgbs, ts, rays,
                                  weights_sum, depth, image, T_thresh=1e-4, alpha_mode=False):
    a = [_c(x, np.float32) for x in (grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas, rgbs, ts)]
    rays = _c(rays, np.int32)
    b = [_c(x, np.float32) for x in (weights_sum, depth, image)]
    M, N = a[4].shape[0], rays.shape[0]
    grad_sigmas, grad_rgbs = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    _call("composite_rays_train_backward", *[_p(x) for x in a], _p(rays), *[_p(x) for x in b], _u32(M), _u32(N),
          _f32(T_thresh), ctypes.c_int(int(alpha_mode)), _p(grad_sigmas), _p(grad_rgbs))
    return grad_sigmas, grad_rgbs


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, contract, bitfield, C, H, nears, fars,
               noises, dt_gamma=0.0, max_steps=1024):
    rays_alive, rays_t = _c(rays_alive, np.int32), _c(rays_t, np.float32)
    rays_o, rays_d = _c(rays_o, np.float32).reshape(-1, 3), _c(rays_d, np.float32).reshape(-1, 3)
    bitfield, nears, fars, noises = _c(bitfield, np.uint8), _c(nears, np.float32), _c(fars, np.float32), _c(noises, np.float32)
    M = n_alive * n_step
    xyzs, dirs, ts = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    _call("march_rays", _u32(n_alive), _u32(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), _f32(bound),
          ctypes.c_int(int(contract)), _f32(dt_gamma), _u32(max_steps), _u32(C), _u32(H), _p(bitfield), _p(nears),
          _p(fars), _p(xyzs), _p(dirs), _p(ts), _p(noises))
    return xyzs, dirs, ts


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image, T_thresh=1e-2,
                   alpha_mode=False):
    """In place on rays_alive, rays_t, weights_sum, depth, image (must be C-contiguous arrays of the right dtype)."""
    for a, dt in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32), (depth, np.float32), (image, np.float32)):
        assert a.dtype == dt and a.flags.c_contiguous
    sigmas, rgbs, ts = _c(sigmas, np.float32), _c(rgbs, np.float32), _c(ts, np.float32)
    _call("composite_rays", _u32(n_alive), _u32(n_step), _f32(T_thresh), ctypes.c_int(int(alpha_mode)), _p(rays_alive),
          _p(rays_t), _p(sigmas), _p(rgbs), _p(ts), _p(weights_sum), _p(depth), _p(image))


def compact_alive(rays_alive):
    rays_alive = _c(rays_alive, np.int32)
    out = np.empty_like(rays_alive)
    n = np.zeros(1, np.int32)
    _call("compact_alive", _p(rays_alive), _u32(rays_alive.shape[0]), _p(out), _p(n))
    return out[: int(n[0])]


# ------------------------------------------------------------------------------------------- gridencoder

def level_offsets(D=3, L=16, per_level_scale=2.0, base_resolution=16, log2_hashmap_size=19, align_corners=False):
    """Row offsets of the multiresolution table (host logic of gridencoder/grid.py:121-135)."""
    offs, off = [], 0
    cap = 2 ** log2_hashmap_size
    for i in range(L):
        res = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(cap, (res if align_corners else res + 1) ** D)
        n = int(np.ceil(n / 8) * 8)
        offs.append(off)
        off += n
    offs.append(off)
    return np.asarray(offs, np.int32)


def _emb_dtype(emb):
    if emb.dtype == np.float32:
        return 0
    if emb.dtype == np.float16:
        return 1
    raise TypeError(emb.dtype)


def grid_encode_forward(inputs, emb, offsets, S, H, max_level=None, calc_dy_dx=False, gridtype=0, align_corners=False,
                        interp=0, sample_major=False):
    inputs, offsets = _c(inputs, np.float32), _c(offsets, np.int32)
    emb = np.ascontiguousarray(emb)
    dt = _emb_dtype(emb)
    B, D = inputs.shape
    C, L = emb.shape[1], offsets.shape[0] - 1
    max_level = L if max_level is None else min(max_level, L)
    if sample_major:
        out = np.zeros((B, L * C), emb.dtype)
        _call("grid_encode_forward_bm", _p(inputs), _p(emb), _p(offsets), _p(out), _u32(B), _u32(D), _u32(C), _u32(L),
              _u32(max_level), _f32(S), _u32(H), _u32(gridtype), ctypes.c_int(int(align_corners)), _u32(interp), ctypes.c_int(dt))
        return out
    out = np.zeros((L, B, C), emb.dtype)
    dy_dx = np.zeros((B, L * D * C), emb.dtype) if calc_dy_dx else None
    _call("grid_encode_forward", _p(inputs), _p(emb), _p(offsets), _p(out), _u32(B), _u32(D), _u32(C), _u32(L), _u32(max_level),
          _f32(S), _u32(H), _p(dy_dx), _u32(gridtype), ctypes.c_int(int(align_corners)), _u32(interp), ctypes.c_int(dt))
    return (out, dy_dx) if calc_dy_dx else out


def grid_encode_backward(grad, inputs, emb, offsets, S, H, max_level=None, dy_dx=None, gridtype=0, align_corners=False,
                         interp=0, sample_major=False):
    inputs, offsets = _c(inputs, np.float32), _c(offsets, np.int32)
    emb = np.ascontiguousarray(emb)
    grad = np.ascontiguousarray(grad, dtype=emb.dtype)
    dt = _emb_dtype(emb)
    B, D = inputs.shape
    C, L = emb.shape[1], offsets.shape[0] - 1
    max_level = L if max_level is None else min(max_level, L)
    grad_emb = np.zeros_like(emb)
    if sample_major:
        _call("grid_encode_backward_bm", _p(grad), _p(inputs), _p(emb), _p(offsets), _p(grad_emb), _u32(B), _u32(D), _u32(C),
              _u32(L), _u32(max_level), _f32(S), _u32(H), _u32(gridtype), ctypes.c_int(int(align_corners)), _u32(interp), ctypes.c_int(dt))
        return grad_emb
    grad_inputs = np.zeros((B, D), emb.dtype) if dy_dx is not None else None
    if dy_dx is not None:
        dy_dx = np.ascontiguousarray(dy_dx, dtype=emb.dtype)
    _call("grid_encode_backward", _p(grad), _p(inputs), _p(emb), _p(offsets), _p(grad_emb), _u32(B), _u32(D), _u32(C), _u32(L),
          _u32(max_level), _f32(S), _u32(H), _p(dy_dx), _p(grad_inputs), _u32(gridtype), ctypes.c_int(int(align_corners)),
          _u32(interp), ctypes.c_int(dt))
    return (grad_emb, grad_inputs) if dy_dx is not None else grad_emb


def grid_encode_backward_exact(grad, inputs, offsets, S, H, C, is_half, max_level=None, gridtype=0, align_corners=False, interp=0):
    """(sum, abs_sum, count) per table entry [rows, C]: the double-precision sum of exactly the terms the reference's backward adds
    (half tables: each term rounded to half like gridencoder.cu:326), the sum of their magnitudes and their number.  grad is level-major
    [L, B, C] in the table's dtype.  For parity bars on kernels whose summation order differs from the reference's."""
    inputs, offsets = _c(inputs, np.float32), _c(offsets, np.int32)
    grad = np.ascontiguousarray(grad, dtype=np.float16 if is_half else np.float32)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    max_level = L if max_level is None else min(max_level, L)
    rows = int(offsets[-1])
    total, mag = np.zeros((rows, C), np.float64), np.zeros((rows, C), np.float64)
    cnt = np.zeros((rows, C), np.uint32)
    _call("grid_encode_backward_exact", _p(grad), _p(inputs), _p(offsets), _p(total), _p(mag), _p(cnt), _u32(B), _u32(D), _u32(C), _u32(L),
          _u32(max_level), _f32(S), _u32(H), _u32(gridtype), ctypes.c_int(int(align_corners)), _u32(interp), ctypes.c_int(1 if is_half else 0))
    return total, mag, cnt


def grad_total_variation(inputs, emb, grad, offsets, weight, S, H, gridtype=0, align_corners=False):
    """In place on `grad` (fp32, C-contiguous)."""
    inputs, emb, offsets = _c(inputs, np.float32), _c(emb, np.float32), _c(offsets, np.int32)
    assert grad.dtype == np.float32 and grad.flags.c_contiguous
    B, D = inputs.shape
    C, L = emb.shape[1], offsets.shape[0] - 1
    _call("grad_total_variation", _p(inputs), _p(emb), _p(grad), _p(offsets), _f32(weight), _u32(B), _u32(D), _u32(C), _u32(L),
          _f32(S), _u32(H), _u32(gridtype), ctypes.c_int(int(align_corners)))


def level_geometry(offsets, S, H):
    offsets = _c(offsets, np.int32)
    L = offsets.shape[0] - 1
    scales, res = np.empty(L, np.float32), np.empty(L, np.uint32)
    _call("level_geometry", _p(offsets), _u32(L), _f32(S), _u32(H), _p(scales), _p(res))
    return scales, res


# --------------------------------------------------------------------------------------------- shencoder

def sh_encode_forward(inputs, degree, calc_dy_dx=False):
    inputs = _c(inputs, np.float32)
    B, D = inputs.shape
    out = np.empty((B, degree * degree), np.float32)
    dy_dx = np.empty((B, D * degree * degree), np.float32) if calc_dy_dx else None
    _call("sh_encode_forward", _p(inputs), _p(out), _u32(B), _u32(D), _u32(degree), _p(dy_dx))
    return (out, dy_dx) if calc_dy_dx else out


def sh_encode_backward(grad, inputs, degree, dy_dx, grad_inputs=None):
    grad, inputs, dy_dx = _c(grad, np.float32), _c(inputs, np.float32), _c(dy_dx, np.float32)
    B, D = inputs.shape
    if grad_inputs is None:
        grad_inputs = np.zeros((B, D), np.float32)
    _call("sh_encode_backward", _p(grad), _p(inputs), _u32(B), _u32(D), _u32(degree), _p(dy_dx), _p(grad_inputs))
    return grad_inputs


def freq_encode_forward(inputs, degree):
    """[B,D] -> [B, D + 2*degree*D] (freqencoder/freq.py:15-37)."""
    inputs = _c(inputs, np.float32)
    B, D = inputs.shape
    C = D + 2 * degree * D
    out = np.empty((B, C), np.float32)
    _call("freq_encode_forward", _p(inputs), _u32(B), _u32(D), _u32(degree), _u32(C), _p(out))
    return out


def freq_encode_backward(grad, outputs, D, degree):
    grad, outputs = _c(grad, np.float32), _c(outputs, np.float32)
    B, C = outputs.shape
    gi = np.zeros((B, D), np.float32)
    _call("freq_encode_backward", _p(grad), _p(outputs), _u32(B), _u32(D), _u32(degree), _u32(C), _p(gi))
    return gi


# ------------------------------------------------------------------------------------------- stage-1 raster

def rasterize(pos, tri, H, W, bbox=False):
    """bbox=True: every triangle confined to its pixel box -- the same image bit for bit at O(covered pixels) (the CPU-baseline form)."""
    pos, tri = _c(pos, np.float32).reshape(-1, 4), _c(tri, np.int32).reshape(-1, 3)
    rast = np.zeros((H, W, 4), np.float32)
    _call("rasterize_bbox" if bbox else "rasterize", _p(pos), _p(tri), _u32(pos.shape[0]), _u32(tri.shape[0]), _u32(H), _u32(W), _p(rast))
    return rast


def interpolate(attr, rast, tri):
    attr, rast, tri = _c(attr, np.float32), _c(rast, np.float32), _c(tri, np.int32).reshape(-1, 3)
    H, W = rast.shape[0], rast.shape[1]
    out = np.zeros((H, W, attr.shape[1]), np.float32)
    _call("interpolate", _p(attr), _p(rast), _p(tri), _u32(attr.shape[0]), _u32(tri.shape[0]), _u32(attr.shape[1]), _u32(H), _u32(W), _p(out))
    return out


def antialias(color, rast, pos, tri):
    color, rast = _c(color, np.float32), _c(rast, np.float32)
    pos, tri = _c(pos, np.float32).reshape(-1, 4), _c(tri, np.int32).reshape(-1, 3)
    H, W, C = color.shape
    out = np.zeros_like(color)
    _call("antialias", _p(color), _p(rast), _p(pos), _p(tri), _u32(pos.shape[0]), _u32(tri.shape[0]), _u32(C), _u32(H), _u32(W), _p(out))
    return out
