"""TEST INFRASTRUCTURE ONLY -- an `nvdiffrast.torch` stand-in over the scalar C rasteriser of oracle/n2m_raster_oracle.c, so that the
reference's UNCHANGED stage-1 callers (nerf/renderer.py:816-921 `render_stage1`, :924-943 `update_triangles_errors`, :947-981
`mark_unseen_triangles`) can run on the CPU in this container (oracle/ref_python.py, backend "ref") and produce
tests/golden/render_stage1.npz.

PARITY UNPINNED for the three operators themselves: nvdiffrast is an un-vendored, unpinned dependency of the reference (readme.md:28-29,
not under /root/reference, not installable here); the C functions restate its published semantics.  What the
fixture pins is the CALLER: clip transform, supersampling, masking, shading of covered pixels, alpha / depth / T composition, the ssaa
reduction, the per-face error scatter and the visibility vote -- everything nerf/renderer.py does around the three calls.

Forward only (no autograd): the fixture holds forward quantities; gradients of the stage-1 step are compared on the GPU between the
unchanged reference Python over the HIP facade and the restated renderer (tests/test_stage1_reference.py).
Call signatures follow the call sites nerf/renderer.py:126-128,338-340,860-863,886-887,961-968.
"""
import numpy as np
import torch

from oracle import oracle as orc


class RasterizeGLContext:
    def __init__(self, output_db=True, mode="automatic", device=None):
        self.output_db = output_db


class RasterizeCudaContext:
    def __init__(self, device=None):
        self.output_db = True


def _np(t, dt):
    return np.ascontiguousarray(t.detach().cpu().numpy(), dtype=dt)


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    """pos [1,V,4] clip space, tri [F,3] int32 -> (rast [1,H,W,4] = (u, v, z/w, triangle id + 1), rast_db zeros)."""
    assert pos.dim() == 3 and pos.shape[0] == 1 and ranges is None
    H, W = int(resolution[0]), int(resolution[1])
    rast = torch.from_numpy(orc.rasterize(_np(pos[0], np.float32), _np(tri, np.int32), H, W)).unsqueeze(0)
    return rast, torch.zeros_like(rast)


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """attr [1,V,A] (or [V,A]) -> (out [1,H,W,A], None): u a0 + v a1 + (1-u-v) a2 on covered pixels, 0 elsewhere."""
    a = attr[0] if attr.dim() == 3 else attr
    out = orc.interpolate(_np(a, np.float32), _np(rast[0], np.float32), _np(tri, np.int32))
    return torch.from_numpy(out).unsqueeze(0), None


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    out = orc.antialias(_np(color[0], np.float32), _np(rast[0], np.float32), _np(pos[0], np.float32), _np(tri, np.int32))
    return torch.from_numpy(out).unsqueeze(0)


def antialias_construct_topology_hash(tri):
    return None
