"""`import mcubes` for an unchanged reference checkout (nerf/renderer.py:16 imports PyMCubes; call sites :525, :527, :563, :616):
`marching_cubes(volume, isovalue) -> (vertices float64 [V, 3] in index space, triangles [T, 3])` as numpy arrays, computed on the device
by libn2m_hip.so.  The volume arrives as a host array (the reference copies it down at :518) and is sent back up -- the restated caller
(nerf2mesh_amd.renderer.export_stage0) keeps it on the device instead.  Vertex / triangle ORDER and the triangulation of ambiguous cells
are this library's (tools/gen_mc_table.py), the surface is the same."""
import numpy as np
import torch

from nerf2mesh_amd.marching_cubes import marching_cubes as _mc


def marching_cubes(volume, isovalue):
    vol = torch.as_tensor(np.ascontiguousarray(volume, dtype=np.float32)).cuda()
    v, t = _mc(vol, float(isovalue), dtype=torch.float64)
    return v.cpu().numpy(), t.cpu().numpy().astype(np.uint64)      # PyMCubes returns unsigned 64-bit indices
