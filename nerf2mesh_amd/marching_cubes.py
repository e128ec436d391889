"""Marching cubes on the device (include/n2m_hip.h, csrc/marchingcubes.hip): what the reference does with
`mcubes.marching_cubes(sigmas, density_thresh)` on the host (nerf/renderer.py:524-527, :563, :616) without moving the volume over PCIe.
Conventions and the case table: tools/gen_mc_table.py."""
import torch

from . import _lib as L

_p = L.ptr


def marching_cubes(volume, isovalue, div=1.0, mul=1.0, add=0.0, dtype=torch.float32):
    """volume [R0, R1, R2] float32 CUDA tensor, isovalue a Python float.  Returns (vertices [V, 3] dtype, triangles [T, 3] int32) on the
    device; vertices = ((index-space position / div) * mul) + add evaluated in double.  With the defaults the coordinates are PyMCubes'
    (index space); `div=resolution - 1, mul=2, add=-1` is the reference's mapping to [-1, 1] (nerf/renderer.py:529)."""
    if not (torch.is_tensor(volume) and volume.is_cuda):
        raise RuntimeError("marching_cubes: volume must be a CUDA tensor (the extraction runs on the device; there is no host path)")
    if volume.dim() != 3:
        raise ValueError(f"marching_cubes: volume must be 3-dimensional, got {tuple(volume.shape)}")
    if dtype not in (torch.float32, torch.float64):
        raise ValueError("marching_cubes: vertices are float32 or float64")
    vol = volume.detach().float().contiguous()
    R0, R1, R2 = (int(s) for s in vol.shape)
    dev = vol.device
    need = int(L.lib().n2m_marching_cubes_workspace_bytes(R0, R1, R2))
    if need == 0:
        raise ValueError(f"marching_cubes: volume {R0} x {R1} x {R2} is outside the supported range")
    with torch.cuda.device(dev):
        ws = L.workspace(dev, need, slot=3)
        totals = torch.zeros(2, dtype=torch.int64, device=dev)
        s = L.stream()
        L.call("n2m_marching_cubes_count", _p(vol), R0, R1, R2, float(isovalue), _p(ws), ws.numel(), _p(totals), s)
        nv, nt = (int(v) for v in totals.tolist())                    # the one host read of the protocol (count -> allocate -> emit)
        if nv >= 1 << 29 or nt >= 1 << 31:
            raise RuntimeError(f"marching_cubes: {nv} vertices / {nt} triangles exceed the 29-bit vertex / 31-bit triangle ids")
        vertices = torch.empty(nv, 3, dtype=dtype, device=dev)
        triangles = torch.empty(nt, 3, dtype=torch.int32, device=dev)
        if nv or nt:
            L.call("n2m_marching_cubes_emit", _p(vol), R0, R1, R2, float(isovalue), _p(ws), ws.numel(), float(div), float(mul), float(add),
                   _p(vertices), int(dtype == torch.float64), nv, _p(triangles), nt, s)
    return vertices, triangles
