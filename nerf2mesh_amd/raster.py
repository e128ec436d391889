"""Stage-1 differentiable rasterisation operators on libn2m_hip.so, with the call surface nerf2mesh uses from
`nvdiffrast.torch` (nerf/renderer.py:126-128,338-340,860-863,886-887,961-968):

    RasterizeGLContext(output_db=False) / RasterizeCudaContext()      opaque context objects
    rasterize(glctx, pos[B,V,4], tri[F,3] int32, (H, W))   -> (rast[B,H,W,4], rast_db)
    interpolate(attr[B|1,V,A] | [V,A], rast, tri)          -> (out[B,H,W,A], out_da)
    antialias(color[B,H,W,C], rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0) -> [B,H,W,C]

`nerf2mesh_amd/backends/nvdiffrast/torch.py` re-exports this module under the name the reference imports.  nvdiffrast
itself is not vendored by the reference (readme.md:28-29): semantics follow SURVEY.md Appendix B; parity with an
nvdiffrast build is unpinned.  No resolution limit (the CUDA context's 2048 cap does not apply).
C ABI: include/n2m_raster.h.
"""
import torch
from torch.autograd import Function

from . import _lib as L

_p = L.ptr
_vp, _u32, _f32 = L._vp, L._u32, L._f32


class RasterizeGLContext:
    def __init__(self, output_db=True, mode="automatic", device=None):
        self.output_db = output_db
        self.device = device


class RasterizeCudaContext:
    def __init__(self, device=None):
        self.output_db = True
        self.device = device


def _check(pos, tri):
    if pos.dim() != 3 or pos.shape[-1] != 4:
        raise ValueError("pos must have shape [minibatch, num_vertices, 4] (range mode is not supported)")
    if tri.dim() != 2 or tri.shape[1] != 3 or tri.dtype != torch.int32:
        raise ValueError("tri must be an int32 tensor of shape [num_triangles, 3]")
    L.check_cuda(pos=pos, tri=tri)


class _rasterize(Function):
    @staticmethod
    def forward(ctx, pos, tri, H, W):
        pos = pos.float().contiguous()
        B, V = pos.shape[0], pos.shape[1]
        F = tri.shape[0]
        rast = torch.empty(B, H, W, 4, dtype=torch.float32, device=pos.device)
        zbuf = torch.empty(H * W, dtype=torch.int64, device=pos.device)
        for b in range(B):
            L.call("n2m_rasterize_forward", _p(pos[b]), _p(tri), V, F, H, W, _p(zbuf), _p(rast[b]), L.stream())
        ctx.save_for_backward(pos, tri, rast)
        return rast

    @staticmethod
    def backward(ctx, d_rast):
        pos, tri, rast = ctx.saved_tensors
        B, V = pos.shape[0], pos.shape[1]
        H, W = rast.shape[1], rast.shape[2]
        d_rast = d_rast.float().contiguous()
        grad_pos = torch.zeros_like(pos)
        for b in range(B):
            L.call("n2m_rasterize_backward", _p(pos[b]), _p(tri), _p(rast[b]), _p(d_rast[b]), V, tri.shape[0], H, W, _p(grad_pos[b]),
                   L.stream())
        return grad_pos, None, None, None


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    """(rast, rast_db): rast[...,0:2] = barycentrics of the triangle's first two vertices, [...,2] = z/w,
    [...,3] = triangle id + 1 (0 = empty).  rast_db (screen-space derivatives) is returned as zeros: the reference
    creates its contexts with output_db=False and discards it (nerf/renderer.py:128,860)."""
    if ranges is not None:
        raise NotImplementedError("range mode is not used by nerf2mesh")
    _check(pos, tri)
    H, W = int(resolution[0]), int(resolution[1])
    rast = _rasterize.apply(pos, tri.contiguous(), H, W)
    return rast, torch.zeros(rast.shape[0], H, W, 4 if getattr(glctx, "output_db", False) else 0, device=rast.device)


class _interpolate(Function):
    @staticmethod
    def forward(ctx, attr, rast, tri):
        attr = attr.float().contiguous()
        rast = rast.float().contiguous()
        B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
        Ba, V, A = attr.shape
        out = torch.empty(B, H, W, A, dtype=torch.float32, device=rast.device)
        for b in range(B):
            L.call("n2m_interpolate_forward", _p(attr[b if Ba > 1 else 0]), _p(rast[b]), _p(tri), V, tri.shape[0], A, H, W, _p(out[b]),
                   L.stream())
        ctx.save_for_backward(attr, rast, tri)
        return out

    @staticmethod
    def backward(ctx, d_out):
        attr, rast, tri = ctx.saved_tensors
        B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
        Ba, V, A = attr.shape
        d_out = d_out.float().contiguous()
        need_attr, need_rast = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_attr = torch.zeros_like(attr) if need_attr else None
        grad_rast = torch.empty_like(rast) if need_rast else None
        for b in range(B):
            L.call("n2m_interpolate_backward", _p(attr[b if Ba > 1 else 0]), _p(rast[b]), _p(tri), _p(d_out[b]), V, tri.shape[0], A, H, W,
                   _p(grad_attr[b if Ba > 1 else 0]) if need_attr else None, _p(grad_rast[b]) if need_rast else None, L.stream())
        return grad_attr, grad_rast, None


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """(out, out_da): barycentric interpolation of per-vertex attributes; out_da is empty (no diff_attrs in nerf2mesh)."""
    if attr.dim() == 2:
        attr = attr.unsqueeze(0)
    L.check_cuda(rast=rast, tri=tri)
    out = _interpolate.apply(attr, rast, tri.contiguous())
    return out, torch.zeros(out.shape[0], out.shape[1], out.shape[2], 0, device=out.device)


def _next_pow2(n):
    p = 4
    while p < n:
        p *= 2
    return p


def antialias_construct_topology_hash(tri):
    """Edge -> opposite-vertex table for `tri`; pass as antialias(..., topology_hash=) to reuse across calls."""
    tri = tri.contiguous()
    F = tri.shape[0]
    cap = _next_pow2(4 * max(F, 1))
    table = torch.empty(cap, 4, dtype=torch.int32, device=tri.device)
    L.call("n2m_antialias_build_topology", _p(tri), F, _p(table), cap, L.stream())
    return table


_topology_cache = {}      # id(tri) -> (weakref to tri, version, table): an entry dies with its tensor, so a new tensor that lands on a
                          # freed one's address (same face count after remeshing) can never pick up the stale edge table


def _topology(tri):
    import weakref
    key = id(tri)
    hit = _topology_cache.get(key)
    if hit is not None and hit[0]() is tri and hit[1] == tri._version:
        return hit[2]
    table = antialias_construct_topology_hash(tri)
    _topology_cache[key] = (weakref.ref(tri, lambda _r, k=key: _topology_cache.pop(k, None)), tri._version, table)
    return table


class _antialias(Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri, table, boost):
        color = color.float().contiguous()
        rast = rast.float().contiguous()
        pos = pos.float().contiguous()
        B, H, W, C = color.shape
        V = pos.shape[1]
        out = torch.empty_like(color)
        for b in range(B):
            L.call("n2m_antialias_forward", _p(color[b]), _p(rast[b]), _p(pos[b]), _p(tri), _p(table), table.shape[0], V, tri.shape[0], C, H,
                   W, _p(out[b]), L.stream())
        ctx.save_for_backward(color, rast, pos, tri, table)
        ctx.boost = float(boost)
        return out

    @staticmethod
    def backward(ctx, d_out):
        color, rast, pos, tri, table = ctx.saved_tensors
        B, H, W, C = color.shape
        V = pos.shape[1]
        d_out = d_out.float().contiguous()
        grad_color = torch.empty_like(color)
        need_pos = ctx.needs_input_grad[2]
        grad_pos = torch.zeros_like(pos) if need_pos else None
        for b in range(B):
            L.call("n2m_antialias_backward", _p(color[b]), _p(rast[b]), _p(pos[b]), _p(tri), _p(table), table.shape[0], _p(d_out[b]), V,
                   tri.shape[0], C, H, W, ctx.boost, _p(grad_color[b]), _p(grad_pos[b]) if need_pos else None, L.stream())
        return grad_color, None, grad_pos, None, None, None


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    """Silhouette antialiasing: pixel pairs with different triangle ids are blended by the coverage of the silhouette
    edge between them; gradients reach `color` and, through the edge position, `pos`."""
    _check(pos, tri)
    tri = tri.contiguous()
    table = topology_hash if topology_hash is not None else _topology(tri)
    return _antialias.apply(color, rast, pos, tri, table, pos_gradient_boost)
