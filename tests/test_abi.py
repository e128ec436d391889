"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares,
the ctypes table matches the header, and the `_backend` shim modules expose the reference's function tables.
No compute is launched (there is no GPU in the build container)."""
import ctypes
import glob
import inspect
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(n2m_[a-zA-Z0-9_]+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def libpath():
    from nerf2mesh_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    syms = declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    lib.n2m_abi_version.restype = ctypes.c_int
    assert lib.n2m_abi_version() == 1


def test_ctypes_table_covers_the_header(libpath):
    from nerf2mesh_amd import _lib
    declared = set(declared_symbols()) - {"n2m_abi_version", "n2m_last_error", "n2m_prof_name"}
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    # argument counts agree with the header prototypes
    src = "\n".join(re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S) for h in glob.glob(os.path.join(ROOT, "include", "*.h")))
    for name, argtypes in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(args) == len(argtypes), f"{name}: header has {len(args)} parameters, ctypes table {len(argtypes)}"


def test_argument_validation_without_gpu(libpath):
    """Entry points reject bad arguments before touching the device, with the reference's messages."""
    from nerf2mesh_amd import _lib as L
    with pytest.raises(RuntimeError, match="C must be 1, 2, 4, or 8"):
        L.call("n2m_grid_encode_forward", 1, 1, 1, 1, 8, 3, 3, 16, 16, 0.5, 16, None, 0, 0, 0, L.F32, None)
    with pytest.raises(RuntimeError, match="D must be 2, 3, 4 or 5"):
        L.call("n2m_grid_encode_forward", 1, 1, 1, 1, 8, 7, 2, 16, 16, 0.5, 16, None, 0, 0, 0, L.F32, None)
    with pytest.raises(RuntimeError, match="NULL"):
        L.call("n2m_near_far_from_aabb", None, None, None, 4, 0.05, None, None, None)
    with pytest.raises(RuntimeError, match="degree in \\[1, 8\\]"):
        L.call("n2m_sh_encode_forward", 1, 1, 8, 3, 9, None, None)
    with pytest.raises(RuntimeError, match="fp32 tables only"):
        L.call("n2m_grad_total_variation", 1, 1, 1, 1, 1e-3, 8, 3, 2, 16, 0.5, 16, 0, 0, L.F16, None)
    with pytest.raises(RuntimeError, match="odd C"):
        L.call("n2m_grid_encode_backward", 1, 1, 1, 1, 1, 8, 3, 1, 16, 16, 0.5, 16, None, None, 0, 0, 0, L.F16, None)
    # round 4: the stage-1 executor's strided forms and the peer-store exchange (include/n2m_peer.h)
    import ctypes
    with pytest.raises(RuntimeError, match="row stride below the row length"):
        L.call("n2m_gather_rows_strided", 16, 16, 0, 3, 2, 16, 3, None)
    with pytest.raises(RuntimeError, match="pixel stride below the attribute count"):
        L.call("n2m_interpolate_backward_strided", 16, 16, 16, 16, 2, 4, 4, 3, 8, 8, None, 16, None)
    with pytest.raises(RuntimeError, match="second buffer"):
        L.call("n2m_antialias_backward_seeded", 16, 16, 16, 16, 16, 8, 32, 4, 4, 4, 8, 8, 1.0, 32, None, None)
    route = L.PeerRoute()
    route.world, route.split_row, route.rows_c, route.rows_f = 2, 10, 4, 4            # the chunks do not cover the coarse half: 2 * 4 < 10
    with pytest.raises(RuntimeError, match="split_row <= world \\* rows_c"):
        L.call("n2m_grid_backward_peer_route", ctypes.byref(route))
    route.rows_c = 12                                                                   # ... or leave the last rank nothing: (2 - 1) * 12 >= 10
    with pytest.raises(RuntimeError, match="split_row <= world \\* rows_c"):
        L.call("n2m_grid_backward_peer_route", ctypes.byref(route))
    route.rows_c = 8                                                                    # a padded layout (round 6): chunks of 8 and 2 rows -- accepted
    with pytest.raises(RuntimeError, match="NULL staging slot"):
        L.call("n2m_grid_backward_peer_route", ctypes.byref(route))
    route.rows_c = 4
    route.split_row = 8
    with pytest.raises(RuntimeError, match="NULL staging slot"):
        L.call("n2m_grid_backward_peer_route", ctypes.byref(route))
    L.call("n2m_grid_backward_peer_route", None)                                       # clearing is always fine
    with pytest.raises(RuntimeError, match="16-byte aligned"):                          # round 6: the forward lookup's corner records
        L.call("n2m_grid_backward_tv_corners", 8)
    L.call("n2m_grid_backward_tv_corners", None)
    with pytest.raises(RuntimeError, match="> 32|> 16"):
        L.call("n2m_grid_backward_merge_levels", 99)
    L.call("n2m_grid_backward_merge_levels", 0)
    ptrs = L.PeerPtrs()
    with pytest.raises(RuntimeError, match="1..8 flags"):
        L.call("n2m_peer_signal", ctypes.byref(ptrs), 1, None)
    ptrs.count, ptrs.ptr[0] = 1, 16
    with pytest.raises(RuntimeError, match="multiple of 4"):
        L.call("n2m_peer_copy", None, ctypes.byref(ptrs), 6, None)
    with pytest.raises(RuntimeError, match="1..8 ranks"):
        L.call("n2m_peer_reduce_slices", 16, None, 0, 8, 16, None, None, None)


REFERENCE_TABLES = {   # raymarching/src/bindings.cpp:5-20, gridencoder/src/bindings.cpp:5-9, shencoder/src/bindings.cpp:5-8
    "_raymarching_mob": {"flatten_rays": 4, "packbits": 4, "near_far_from_aabb": 7, "sph_from_ray": 5, "morton3D": 3,
                         "morton3D_invert": 3, "march_rays_train": 18, "composite_rays_train_forward": 12,
                         "composite_rays_train_backward": 17, "march_rays": 19, "composite_rays": 12},
    "_gridencoder": {"grid_encode_forward": 15, "grid_encode_backward": 17, "grad_total_variation": 13},
    "_shencoder": {"sh_encode_forward": 6, "sh_encode_backward": 7},
    "_freqencoder": {"freq_encode_forward": 6, "freq_encode_backward": 7},      # freqencoder/src/bindings.cpp:5-8
}


def test_backend_shims_have_the_reference_function_tables():
    from nerf2mesh_amd import backends
    backends.install()
    for mod, table in REFERENCE_TABLES.items():
        m = __import__(mod)
        assert os.path.dirname(m.__file__) == backends.path()
        for fn, nargs in table.items():
            f = getattr(m, fn)
            assert len(inspect.signature(f).parameters) == nargs, f"{mod}.{fn}"


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not present")
def test_shim_signatures_match_reference_headers():
    """Parameter counts above are the ones of the reference's C++ prototypes."""
    for pkg, hdr, mod in (("raymarching", "raymarching.h", "_raymarching_mob"), ("gridencoder", "gridencoder.h", "_gridencoder"),
                          ("shencoder", "shencoder.h", "_shencoder"), ("freqencoder", "freqencoder.h", "_freqencoder")):
        src = open(f"/root/reference/{pkg}/src/{hdr}").read()
        for fn, nargs in REFERENCE_TABLES[mod].items():
            m = re.search(r"void\s+" + fn + r"\s*\((.*?)\)\s*;", src, flags=re.S)
            assert m, fn
            assert len(m.group(1).split(",")) == nargs, fn


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under nerf2mesh_amd/ may reference it."""
    for path in glob.glob(os.path.join(ROOT, "nerf2mesh_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
        assert "n2m_oracle" not in src, path
    # native side: no source includes, links or opens anything under oracle/ (comments may NAME oracle/n2m_oracle.c as the arithmetic they follow)
    for path in glob.glob(os.path.join(ROOT, "nerf2mesh_amd", "csrc", "*")):
        src = open(path).read()
        assert "oracle/" not in src.replace("oracle/n2m_oracle.c", ""), path
        for line in src.splitlines():
            if "oracle" in line:
                assert line.lstrip().startswith(("//", "*", "/*")), f"{path}: {line.strip()}"
    from nerf2mesh_amd import build
    assert not any("oracle" in f for f in build.FLAGS) and "oracle" not in build.CSRC


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from nerf2mesh_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU / PyTorch fallback"):
        _lib.lib()
