"""Drop-in `_backend` modules for an UNCHANGED reference checkout.

The reference's operator wrappers do `import _raymarching_mob as _backend` / `import _gridencoder as _backend`
/ `import _shencoder as _backend` (raymarching/raymarching.py:9-12, gridencoder/grid.py:9-12,
shencoder/sphere_harmonics.py:9-12) and fall back to a JIT CUDA build only when that import fails.
Putting this directory on sys.path (`nerf2mesh_amd.backends.install()`) makes those imports resolve to the
modules here, which expose the same function tables (same names, positional arguments and in-place
semantics as the pybind11 modules of raymarching/src/bindings.cpp:5-20, gridencoder/src/bindings.cpp:5-9,
shencoder/src/bindings.cpp:5-8) on top of libn2m_hip.so.
"""
import os
import sys


def path():
    return os.path.dirname(os.path.abspath(__file__))


def install():
    """Prepend this directory to sys.path so `import _raymarching_mob` etc. find the HIP-backed modules."""
    p = path()
    if p not in sys.path:
        sys.path.insert(0, p)
    # a test harness may have parked empty stand-ins under these names (oracle/ref_python.py does, to import the reference Python without
    # its third-party packages): they must not shadow the modules of this directory
    for name in ("mcubes", "nvdiffrast.torch", "nvdiffrast", "torch_scatter", "_raymarching_mob", "_gridencoder", "_shencoder", "_freqencoder"):
        m = sys.modules.get(name)
        if m is not None and getattr(m, "__n2m_stub__", False):
            del sys.modules[name]
    return p
