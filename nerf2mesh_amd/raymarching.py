"""Ray-marching operators on libn2m_hip.so -- the host-side mirror of the reference's raymarching/raymarching.py.

Same operator names, argument meaning and return values (so a caller written against the reference's
`raymarching` module works against this one), restructured for ROCm:

* every op enqueues on torch's current HIP stream through the C ABI (include/n2m_hip.h) -- no pybind layer;
* `march_rays_train` gets deterministic, ray-ordered sample packing (exclusive scan instead of the reference's
  arrival-order atomicAdd, raymarching.cu:471) and reads the sample count back through one pinned-host copy;
* `composite_rays_train` runs one wavefront per ray (prefix product / prefix sums across 64 samples);
* fp32 everywhere: inputs are cast like the reference's custom_fwd(cast_inputs=torch.float32).

Reference call sites: nerf/renderer.py:688,711,717,741,776,796,1020,1100,1142.
"""
import torch
from torch.autograd import Function

from . import _lib as L

_p = L.ptr


def _f32c(t):
    return t.float().contiguous()


def _dev(t):
    # the reference moves stray CPU tensors to the GPU (`if not x.is_cuda: x = x.cuda()`, raymarching.py:34-35)
    return t if t.is_cuda else t.cuda()


class _near_far_from_aabb(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        rays_o = _f32c(_dev(rays_o)).view(-1, 3)
        rays_d = _f32c(_dev(rays_d)).view(-1, 3)
        aabb = _f32c(_dev(aabb))
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=torch.float32, device=rays_o.device)
        fars = torch.empty(N, dtype=torch.float32, device=rays_o.device)
        L.call("n2m_near_far_from_aabb", _p(rays_o), _p(rays_d), _p(aabb), N, float(min_near), _p(nears), _p(fars), L.stream())
        return nears, fars


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """rays_o, rays_d [N,3], aabb [6] -> nears, fars [N]; a miss gives FLT_MAX for both (raymarching.py:19-49)."""
    return _near_far_from_aabb.apply(rays_o, rays_d, aabb, min_near)


def sph_from_ray(rays_o, rays_d, radius):
    """(theta, phi) in [-1,1]^2 of each ray's far intersection with the sphere of `radius` (raymarching.py:52-80)."""
    rays_o = _f32c(_dev(rays_o)).view(-1, 3)
    rays_d = _f32c(_dev(rays_d)).view(-1, 3)
    N = rays_o.shape[0]
    coords = torch.empty(N, 2, dtype=torch.float32, device=rays_o.device)
    L.call("n2m_sph_from_ray", _p(rays_o), _p(rays_d), float(radius), N, _p(coords), L.stream())
    return coords


def morton3D(coords):
    """int [N,3] in [0,1024) -> int32 [N] Morton codes, x in bit 0 (raymarching.py:82-103)."""
    coords = _dev(coords).int().contiguous()
    N = coords.shape[0]
    indices = torch.empty(N, dtype=torch.int32, device=coords.device)
    L.call("n2m_morton3D", _p(coords), N, _p(indices), L.stream())
    return indices


def morton3D_invert(indices):
    """int32 [N] -> int32 [N,3] (raymarching.py:105-125)."""
    indices = _dev(indices).int().contiguous()
    N = indices.shape[0]
    coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
    L.call("n2m_morton3D_invert", _p(indices), N, _p(coords), L.stream())
    return coords


def packbits(grid, thresh, bitfield=None):
    """density grid [C, H^3] fp32 -> bitfield [C*H^3/8] u8, bit i of byte n <=> grid[8n+i] > thresh; writes into
    `bitfield` when given (raymarching.py:128-154)."""
    grid = _f32c(_dev(grid))
    N = grid.numel() // 8
    if bitfield is None:
        bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
    if torch.is_tensor(thresh):        # threshold computed on the device: no host read-back (n2m_packbits_dev)
        L.call("n2m_packbits_dev", _p(grid), N, _p(thresh.float().contiguous()), _p(bitfield), L.stream())
    else:
        L.call("n2m_packbits", _p(grid), N, float(thresh), _p(bitfield), L.stream())
    return bitfield


def flatten_rays(rays, M):
    """rays [N,2] (offset,count) -> int32 [M] ray id of every sample (raymarching.py:157-178)."""
    rays = _dev(rays).contiguous()
    res = torch.zeros(M, dtype=torch.int32, device=rays.device)
    L.call("n2m_flatten_rays", _p(rays), rays.shape[0], M, _p(res), L.stream())
    return res


class _march_rays_train(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, bound, contract, density_bitfield, C, H, nears, fars, perturb=False, dt_gamma=0,
                max_steps=1024, noises=None):
        rays_o = _f32c(_dev(rays_o)).view(-1, 3)
        rays_d = _f32c(_dev(rays_d)).view(-1, 3)
        bits = _dev(density_bitfield).contiguous()
        nears, fars = _f32c(nears), _f32c(fars)
        dev = rays_o.device
        N = rays_o.shape[0]
        if noises is None:
            noises = torch.rand(N, dtype=torch.float32, device=dev) if perturb else torch.zeros(N, dtype=torch.float32, device=dev)
        else:
            noises = _f32c(noises)
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        rays = torch.empty(N, 2, dtype=torch.int32, device=dev)
        s = L.stream()
        args = (_p(rays_o), _p(rays_d), _p(bits), float(bound), int(bool(contract)), float(dt_gamma), int(max_steps), N, int(C),
                int(H), _p(nears), _p(fars))
        # pass 1: per-ray counts -> ray-ordered offsets, total in `counter`
        L.call("n2m_march_rays_train", *args, None, None, None, _p(rays), _p(counter), _p(noises), s)
        M = int(counter.item())   # the one host sync of the step (the reference has the same one, raymarching.py:232)
        xyzs = torch.empty(M, 3, dtype=torch.float32, device=dev)
        dirs = torch.empty(M, 3, dtype=torch.float32, device=dev)
        ts = torch.empty(M, 2, dtype=torch.float32, device=dev)
        if M > 0:   # pass 2 fills every row (offsets are dense), so no zero-fill is needed
            L.call("n2m_march_rays_train", *args, _p(xyzs), _p(dirs), _p(ts), _p(rays), _p(counter), _p(noises), s)
        return xyzs, dirs, ts, rays


def march_rays_train(rays_o, rays_d, bound, contract, density_bitfield, C, H, nears, fars, perturb=False, dt_gamma=0,
                     max_steps=1024, noises=None):
    """Occupancy-grid ray marching for training (raymarching.py:184-245).

    Returns xyzs [M,3] (contracted coordinates), dirs [M,3] (the un-normalised ray direction), ts [M,2] =
    (t after the step, dt) and rays [N,2] int32 = (offset, count); samples of ray i are rows
    rays[i,0] .. rays[i,0]+rays[i,1].  `noises` (optional, [N] in [0,1)) replaces the internal torch.rand draw.
    """
    return _march_rays_train.apply(rays_o, rays_d, bound, contract, density_bitfield, C, H, nears, fars, perturb, dt_gamma,
                                   max_steps, noises)


_ZERO_CACHE = {}


def _zeros_like_cached(ref, n):
    """A persistent all-zero fp32 buffer of at least n elements on ref's device (never written by anyone)."""
    z = _ZERO_CACHE.get(ref.device)
    if z is None or z.numel() < n:
        z = torch.zeros(int(n * 1.5) + 1024, dtype=torch.float32, device=ref.device)
        _ZERO_CACHE[ref.device] = z
    return z


class _composite_rays_train(Function):
    @staticmethod
    def forward(ctx, sigmas, rgbs, ts, rays, T_thresh=1e-4, alpha_mode=False, rays_tile_samples=False):
        sigmas, rgbs, ts = _f32c(sigmas), _f32c(rgbs), _f32c(ts)
        rays = rays.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        dev = sigmas.device
        # the kernels write every sample inside a ray's range (zeros after the early stop); the fill is only for samples no ray owns
        mk = torch.empty if rays_tile_samples else torch.zeros
        weights = mk(M, dtype=torch.float32, device=dev)
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        L.call("n2m_composite_rays_train_forward", _p(sigmas), _p(rgbs), _p(ts), _p(rays), M, N, float(T_thresh),
               int(bool(alpha_mode)), _p(weights), _p(weights_sum), _p(depth), _p(image), L.stream())
        ctx.save_for_backward(sigmas, rgbs, ts, rays, weights_sum, depth, image)
        ctx.cfg = (M, N, float(T_thresh), int(bool(alpha_mode)), bool(rays_tile_samples))
        ctx.set_materialize_grads(False)      # unused outputs (weights, depth in the plain rgb loss) arrive as None, not as fresh zero tensors
        return weights, weights_sum, depth, image

    @staticmethod
    def backward(ctx, grad_weights, grad_weights_sum, grad_depth, grad_image):
        sigmas, rgbs, ts, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh, alpha_mode, tiled = ctx.cfg
        zeros = _zeros_like_cached(sigmas, max(M, 3 * N))          # stands in for every gradient that is None (read-only)
        gw = _f32c(grad_weights) if grad_weights is not None else zeros[:M]
        gws = _f32c(grad_weights_sum) if grad_weights_sum is not None else zeros[:N]
        gd = _f32c(grad_depth) if grad_depth is not None else zeros[:N]
        gi = _f32c(grad_image) if grad_image is not None else zeros[:3 * N].view(N, 3)
        both = (torch.empty if tiled else torch.zeros)(M, 4, dtype=torch.float32, device=sigmas.device)      # one buffer for the two outputs
        grad_sigmas = both.view(-1)[:M]
        grad_rgbs = both.view(-1)[M:].view(M, 3)
        L.call("n2m_composite_rays_train_backward", _p(gw), _p(gws), _p(gd), _p(gi), _p(sigmas), _p(rgbs), _p(ts), _p(rays),
               _p(weights_sum), _p(depth), _p(image), M, N, T_thresh, alpha_mode, _p(grad_sigmas), _p(grad_rgbs), L.stream())
        return grad_sigmas, grad_rgbs, None, None, None, None, None


def composite_rays_train(sigmas, rgbs, ts, rays, T_thresh=1e-4, alpha_mode=False, rays_tile_samples=False):
    """Front-to-back compositing of packed samples; differentiable in sigmas and rgbs (raymarching.py:248-305).
    Returns weights [M], weights_sum [N], depth [N], image [N,3].
    rays_tile_samples (not in the reference signature): the (offset, count) ranges of `rays` cover [0, M) without gaps -- true for the
    output of march_rays_train -- so the outputs need no zero-fill."""
    return _composite_rays_train.apply(sigmas, rgbs, ts, rays, T_thresh, alpha_mode, rays_tile_samples)


@torch.no_grad()
def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, contract, density_bitfield, C, H, near, far,
               perturb=False, dt_gamma=0, max_steps=1024):
    """Inference marcher: up to n_step samples for each of the first n_alive entries of rays_alive, fixed stride
    (empty slots are all-zero rows) (raymarching.py:311-359)."""
    rays_o = _f32c(_dev(rays_o)).view(-1, 3)
    rays_d = _f32c(_dev(rays_d)).view(-1, 3)
    dev = rays_o.device
    M = n_alive * n_step
    xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    ts = torch.zeros(M, 2, dtype=torch.float32, device=dev)
    noises = torch.rand(n_alive, dtype=torch.float32, device=dev) if perturb else torch.zeros(n_alive, dtype=torch.float32, device=dev)
    L.call("n2m_march_rays", n_alive, n_step, _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), float(bound),
           int(bool(contract)), float(dt_gamma), int(max_steps), int(C), int(H), _p(density_bitfield.contiguous()), _p(near),
           _p(far), _p(xyzs), _p(dirs), _p(ts), _p(noises), L.stream())
    return xyzs, dirs, ts


@torch.no_grad()
def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image, T_thresh=1e-2,
                   alpha_mode=False):
    """Inference compositing, in place on weights_sum / depth / image / rays_alive / rays_t (raymarching.py:362-386)."""
    sigmas, rgbs = _f32c(sigmas), _f32c(rgbs)
    L.call("n2m_composite_rays", n_alive, n_step, float(T_thresh), int(bool(alpha_mode)), _p(rays_alive), _p(rays_t), _p(sigmas),
           _p(rgbs), _p(ts), _p(weights_sum), _p(depth), _p(image), L.stream())
    return tuple()


@torch.no_grad()
def compact_alive(rays_alive):
    """Order-preserving removal of the finished rays (entries < 0): the on-device form of
    `rays_alive = rays_alive[rays_alive >= 0]` (nerf/renderer.py:798). Returns the compacted int32 tensor."""
    n = rays_alive.shape[0]
    out = torch.empty_like(rays_alive)
    count = torch.zeros(1, dtype=torch.int32, device=rays_alive.device)
    L.call("n2m_compact_alive", _p(rays_alive), n, _p(out), _p(count), L.stream())
    return out[: int(count.item())]


# ------------------------------------------------------------------------------------------------------------------
# Split form of march_rays_train for software pipelining.  Pass 1 (counting + offsets) depends on the rays and the
# occupancy bit field only -- not on the network parameters -- so a training loop can issue it for batch i+1 while the
# GPU is still busy with step i, and pick up the sample count without ever draining the queue: `march_rays_train_begin`
# enqueues pass 1 and an asynchronous copy of the counter into pinned host memory; `march_rays_train_finish` waits for
# that copy only (an event, normally long complete) and enqueues pass 2.  Results are identical to march_rays_train.
class MarchTicket:
    __slots__ = ("args", "rays", "counter", "noises", "host_count", "event", "done", "keep", "spec", "cap")


@torch.no_grad()
def march_rays_train_fused(rays_o, rays_d, bound, contract, density_bitfield, C, H, nears, fars, noises, dt_gamma=0, max_steps=1024,
                           max_points=0, out=None, rays=None, counter=None, workspace=None):
    """march_rays_train with one march per ray (n2m_march_rays_train_fused: count + recorded chunks, then a replay kernel): offsets (ray
    order, from 0), the sample count and the samples of every ray that fits `max_points` rows.  Returns (xyzs, dirs, ts, rays,
    counter) with the sample arrays at capacity max_points: rows [0, counter) are defined when counter <= max_points (else the rays
    that did not fit are missing, like raymarching.cu:417, and the caller re-marches with n2m_march_rays_train_write)."""
    rays_o = _f32c(_dev(rays_o)).view(-1, 3)
    rays_d = _f32c(_dev(rays_d)).view(-1, 3)
    bits = _dev(density_bitfield).contiguous()
    nears, fars, noises = _f32c(nears), _f32c(fars), _f32c(noises)
    dev = rays_o.device
    N = rays_o.shape[0]
    cap = int(max_points)
    if out is None:
        buf = torch.empty(max(cap, 1) * 8, dtype=torch.float32, device=dev)
        out = (buf[:3 * cap].view(-1, 3), buf[3 * cap:6 * cap].view(-1, 3), buf[6 * cap:8 * cap].view(-1, 2))
    if rays is None:
        rays = torch.empty(N, 2, dtype=torch.int32, device=dev)
    if counter is None:
        counter = torch.empty(1, dtype=torch.int32, device=dev)
    need = int(L.lib().n2m_march_fused_workspace_bytes(N))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    L.call("n2m_march_rays_train_fused", _p(rays_o), _p(rays_d), _p(bits), float(bound), int(bool(contract)), float(dt_gamma), int(max_steps),
           N, int(C), int(H), _p(nears), _p(fars), _p(out[0]), _p(out[1]), _p(out[2]), _p(rays), _p(counter), _p(noises), cap,
           _p(workspace), need, L.stream())
    return out[0], out[1], out[2], rays, counter


@torch.no_grad()
def march_rays_train_begin(rays_o, rays_d, bound, contract, density_bitfield, C, H, nears, fars, perturb=False, dt_gamma=0,
                           max_steps=1024, noises=None, expect_points=0):
    """Pass 1 (count + offset scan) of march_rays_train, with the sample count on its way to the host; march_rays_train_finish
    completes the call.  expect_points > 0: pass 2 is queued right behind it into buffers of that many rows (n2m_march_rays_train_write
    skips rays that do not fit), so that finish() only has to slice them -- unless the batch turned out larger, then it re-runs."""
    rays_o = _f32c(_dev(rays_o)).view(-1, 3)
    rays_d = _f32c(_dev(rays_d)).view(-1, 3)
    bits = _dev(density_bitfield).contiguous()
    nears, fars = _f32c(nears), _f32c(fars)
    dev = rays_o.device
    N = rays_o.shape[0]
    if noises is None:
        noises = torch.rand(N, dtype=torch.float32, device=dev) if perturb else torch.zeros(N, dtype=torch.float32, device=dev)
    t = MarchTicket()
    t.counter = torch.zeros(1, dtype=torch.int32, device=dev)
    t.rays = torch.empty(N, 2, dtype=torch.int32, device=dev)
    t.noises = _f32c(noises)
    t.keep = (rays_o, rays_d, bits, nears, fars)
    t.args = (_p(rays_o), _p(rays_d), _p(bits), float(bound), int(bool(contract)), float(dt_gamma), int(max_steps), N, int(C), int(H),
              _p(nears), _p(fars))
    L.call("n2m_march_rays_train", *t.args, None, None, None, _p(t.rays), _p(t.counter), _p(t.noises), L.stream())
    t.host_count = torch.empty(1, dtype=torch.int32, pin_memory=True)
    t.host_count.copy_(t.counter, non_blocking=True)
    t.event = torch.cuda.Event()                       # the HOST waits for this one: the count is all it needs to go on
    t.event.record()
    t.spec, t.cap, t.done = None, int(expect_points), t.event
    if t.cap > 0 and N > 0:
        buf = torch.empty(t.cap, 8, dtype=torch.float32, device=dev)       # one allocation: xyzs | dirs | ts
        xyzs, dirs, ts = buf.view(-1)[:3 * t.cap].view(-1, 3), buf.view(-1)[3 * t.cap:6 * t.cap].view(-1, 3), buf.view(-1)[6 * t.cap:].view(-1, 2)
        L.call("n2m_march_rays_train_write", *t.args, _p(xyzs), _p(dirs), _p(ts), _p(t.rays), _p(t.noises), t.cap, L.stream())
        t.spec = (xyzs, dirs, ts)
        t.done = torch.cuda.Event()                    # the consuming STREAM waits for this one (speculative pass 2 finished)
        t.done.record()
    return t


@torch.no_grad()
def march_rays_train_finish(t):
    t.event.synchronize()
    M = int(t.host_count[0])
    rays_o = t.keep[0]
    dev = rays_o.device
    torch.cuda.current_stream(dev).wait_event(t.done)       # pass 1 (and a speculative pass 2) may have been issued on another stream
    if t.spec is not None and M <= t.cap:
        xyzs, dirs, ts = t.spec
        return xyzs[:M], dirs[:M], ts[:M], t.rays
    xyzs = torch.empty(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.empty(M, 3, dtype=torch.float32, device=dev)
    ts = torch.empty(M, 2, dtype=torch.float32, device=dev)
    if M > 0:
        L.call("n2m_march_rays_train", *t.args, _p(xyzs), _p(dirs), _p(ts), _p(t.rays), _p(t.counter), _p(t.noises), L.stream())
    return xyzs, dirs, ts, t.rays
