"""`import nvdiffrast.torch as dr` for ROCm: resolves to nerf2mesh_amd.raster (HIP kernels) when
nerf2mesh_amd/backends is on sys.path (nerf2mesh_amd.backends.install())."""
__version__ = "0.0-n2m-hip"
