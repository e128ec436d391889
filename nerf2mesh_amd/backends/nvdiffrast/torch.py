"""The subset of nvdiffrast.torch that nerf2mesh calls (nerf/renderer.py:15,126-128,338-340,860-887,961-968)."""
from nerf2mesh_amd.raster import (RasterizeCudaContext, RasterizeGLContext, antialias,  # noqa: F401
                                  antialias_construct_topology_hash, interpolate, rasterize)
