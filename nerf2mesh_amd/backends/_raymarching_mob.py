"""`_raymarching_mob` for ROCm: the function table of raymarching/src/bindings.cpp:5-20 over libn2m_hip.so.

All functions return None and write into the pre-allocated output tensors, like the reference extension.
The reference defines CHECK_* macros for this module but applies none (raymarching.cu:13-16); here device
and contiguity are verified because a stray CPU tensor would otherwise be dereferenced as a device pointer.
"""
import torch

from nerf2mesh_amd import _lib as L

_p, _chk = L.ptr, L.check_cuda


def _f32(**ts):
    for k, t in ts.items():
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"{k} must be a float32 tensor (got {t.dtype}); the wrappers cast with custom_fwd(cast_inputs=torch.float32)")


def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    _chk(rays_o=rays_o, rays_d=rays_d, aabb=aabb, nears=nears, fars=fars)
    _f32(rays_o=rays_o, rays_d=rays_d, aabb=aabb, nears=nears, fars=fars)
    L.call("n2m_near_far_from_aabb", _p(rays_o), _p(rays_d), _p(aabb), N, min_near, _p(nears), _p(fars), L.stream())


def sph_from_ray(rays_o, rays_d, radius, N, coords):
    _chk(rays_o=rays_o, rays_d=rays_d, coords=coords)
    _f32(rays_o=rays_o, rays_d=rays_d, coords=coords)
    L.call("n2m_sph_from_ray", _p(rays_o), _p(rays_d), radius, N, _p(coords), L.stream())


def morton3D(coords, N, indices):
    _chk(coords=coords, indices=indices)
    if coords.dtype != torch.int32 or indices.dtype != torch.int32:
        raise RuntimeError("morton3D: coords and indices must be int32 tensors")
    L.call("n2m_morton3D", _p(coords), N, _p(indices), L.stream())


def morton3D_invert(indices, N, coords):
    _chk(coords=coords, indices=indices)
    if coords.dtype != torch.int32 or indices.dtype != torch.int32:
        raise RuntimeError("morton3D_invert: coords and indices must be int32 tensors")
    L.call("n2m_morton3D_invert", _p(indices), N, _p(coords), L.stream())


def packbits(grid, N, density_thresh, bitfield):
    _chk(grid=grid, bitfield=bitfield)
    _f32(grid=grid)
    if bitfield.dtype != torch.uint8:
        raise RuntimeError("packbits: bitfield must be a uint8 tensor")
    L.call("n2m_packbits", _p(grid), N, density_thresh, _p(bitfield), L.stream())


def flatten_rays(rays, N, M, res):
    _chk(rays=rays, res=res)
    L.call("n2m_flatten_rays", _p(rays), N, M, _p(res), L.stream())


def march_rays_train(rays_o, rays_d, grid, bound, contract, dt_gamma, max_steps, N, C, H, nears, fars, xyzs, dirs, ts,
                     rays, counter, noises):
    _chk(rays_o=rays_o, rays_d=rays_d, grid=grid, nears=nears, fars=fars, xyzs=xyzs, dirs=dirs, ts=ts, rays=rays,
         counter=counter, noises=noises)
    _f32(rays_o=rays_o, rays_d=rays_d, nears=nears, fars=fars, xyzs=xyzs, dirs=dirs, ts=ts, noises=noises)
    L.call("n2m_march_rays_train", _p(rays_o), _p(rays_d), _p(grid), bound, int(bool(contract)), dt_gamma, max_steps, N, C, H,
           _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(ts), _p(rays), _p(counter), _p(noises), L.stream())


def composite_rays_train_forward(sigmas, rgbs, ts, rays, M, N, T_thresh, alpha_mode, weights, weights_sum, depth, image):
    _chk(sigmas=sigmas, rgbs=rgbs, ts=ts, rays=rays, weights=weights, weights_sum=weights_sum, depth=depth, image=image)
    _f32(sigmas=sigmas, rgbs=rgbs, ts=ts, weights=weights, weights_sum=weights_sum, depth=depth, image=image)
    L.call("n2m_composite_rays_train_forward", _p(sigmas), _p(rgbs), _p(ts), _p(rays), M, N, T_thresh, int(bool(alpha_mode)),
           _p(weights), _p(weights_sum), _p(depth), _p(image), L.stream())


def composite_rays_train_backward(grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas, rgbs, ts, rays,
                                  weights_sum, depth, image, M, N, T_thresh, alpha_mode, grad_sigmas, grad_rgbs):
    _chk(grad_weights=grad_weights, grad_weights_sum=grad_weights_sum, grad_depth=grad_depth, grad_image=grad_image,
         sigmas=sigmas, rgbs=rgbs, ts=ts, rays=rays, weights_sum=weights_sum, depth=depth, image=image,
         grad_sigmas=grad_sigmas, grad_rgbs=grad_rgbs)
    _f32(grad_weights=grad_weights, grad_weights_sum=grad_weights_sum, grad_depth=grad_depth, grad_image=grad_image,
         sigmas=sigmas, rgbs=rgbs, ts=ts, grad_sigmas=grad_sigmas, grad_rgbs=grad_rgbs)
    L.call("n2m_composite_rays_train_backward", _p(grad_weights), _p(grad_weights_sum), _p(grad_depth), _p(grad_image),
           _p(sigmas), _p(rgbs), _p(ts), _p(rays), _p(weights_sum), _p(depth), _p(image), M, N, T_thresh,
           int(bool(alpha_mode)), _p(grad_sigmas), _p(grad_rgbs), L.stream())


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, contract, dt_gamma, max_steps, C, H, grid,
               nears, fars, xyzs, dirs, ts, noises):
    _chk(rays_alive=rays_alive, rays_t=rays_t, rays_o=rays_o, rays_d=rays_d, grid=grid, nears=nears, fars=fars, xyzs=xyzs,
         dirs=dirs, ts=ts, noises=noises)
    _f32(rays_t=rays_t, rays_o=rays_o, rays_d=rays_d, nears=nears, fars=fars, xyzs=xyzs, dirs=dirs, ts=ts, noises=noises)
    L.call("n2m_march_rays", n_alive, n_step, _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), bound, int(bool(contract)),
           dt_gamma, max_steps, C, H, _p(grid), _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(ts), _p(noises), L.stream())


def composite_rays(n_alive, n_step, T_thresh, alpha_mode, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image):
    _chk(rays_alive=rays_alive, rays_t=rays_t, sigmas=sigmas, rgbs=rgbs, ts=ts, weights_sum=weights_sum, depth=depth, image=image)
    _f32(rays_t=rays_t, sigmas=sigmas, rgbs=rgbs, ts=ts, weights_sum=weights_sum, depth=depth, image=image)
    L.call("n2m_composite_rays", n_alive, n_step, T_thresh, int(bool(alpha_mode)), _p(rays_alive), _p(rays_t), _p(sigmas),
           _p(rgbs), _p(ts), _p(weights_sum), _p(depth), _p(image), L.stream())
