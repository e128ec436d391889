import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ------------------------------------------------------------------------------------------------ fixtures

@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def ref():
    """The reference's own kernels compiled for the host (oracle/_ref). Skips when they are not built."""
    from oracle import build_ref
    if not build_ref.available():
        if not build_ref.build(verbose=False):
            pytest.skip("oracle/_ref not built and /root/reference absent")
    return build_ref.load()


@pytest.fixture(scope="session")
def scene():
    """Small deterministic marching scene: cameras, occupancy bitfield (bound 1, cascade 1, H 128)."""
    import torch
    from nerf2mesh_amd import synthetic as S
    from oracle import oracle as o
    poses = S.make_cameras(16, seed=3)
    grid = S.scene_density_grid(H=128, cascade=1, bound=1.0)
    bits = o.packbits(grid.numpy(), 10.0)
    return {"poses": poses, "grid": grid.numpy(), "bits": bits, "torch": torch, "S": S}


def lego_offsets(bound=1.0, L=16, C=None):
    from oracle import oracle as o
    pls = float(np.exp2(np.log2(2048 * bound / 16) / (L - 1)))
    return o.level_offsets(3, L, pls, 16, 19), float(np.log2(pls))
