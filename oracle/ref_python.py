"""TEST INFRASTRUCTURE ONLY -- runs the reference's UNCHANGED Python callers (nerf/renderer.py, nerf/network.py,
raymarching/raymarching.py, gridencoder/grid.py, shencoder/sphere_harmonics.py, encoding.py, activation.py) so that the
caller-level restatement in nerf2mesh_amd/{renderer,network}.py and the drop-in `_backend` modules can be checked against them.

Where the Python comes from:
  * this container: the sources where they lie under /root/reference (never copied);
  * the GPU box (no /root/reference): `oracle/_ref/pyref/` = the same files byte-compiled by `compile_pyref()` below
    (`py_compile`, sourceless `.pyc`, git-ignored like the `.so` files next to them; the compiled form travels, the sources do not).

Which kernels sit under the wrappers (`backend=`):
  * "ref": oracle/_ref/_ref_*.so -- the reference's own .cu kernels compiled for the host (oracle/build_ref.py); CPU tensors, fp32
    (torch.cuda.amp.autocast disables itself without a device); `Tensor.cuda` is patched to the identity for the
    duration of a call because the wrappers hard-code `.cuda()` (raymarching/raymarching.py:34-35 ...);
  * "hip": nerf2mesh_amd/backends/_{raymarching_mob,gridencoder,shencoder,freqencoder}.py over libn2m_hip.so -- the product's drop-in
    boundary; device tensors.
Both function tables can live in one process: `use_backend()` swaps the `_backend` global of the three wrapper modules.

Modules the reference imports at module scope but that this path never calls (cv2, xatlas, pymeshlab, ... SURVEY 8b) are
registered as empty stubs.  Stage 1 (nerf/renderer.py:123-165,816-981): `nvdiffrast.torch` resolves to the HIP facade
(nerf2mesh_amd/backends/nvdiffrast/torch.py) when backend == "hip" and to the scalar C rasteriser (oracle/nvdiffrast_oracle.py, forward
only) when backend == "ref"; `torch_scatter.scatter_add` to backends/torch_scatter.py / `index_add_`; `trimesh.load` to a PLY reader.
"""
import contextlib
import importlib
import os
import py_compile
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("N2M_REFERENCE", "/root/reference")
PYREF = os.path.join(HERE, "_ref", "pyref")

# reference files on this path (relative to the checkout) -> compiled for the GPU box
PY_FILES = [
    "activation.py", "encoding.py", "meshutils.py",
    "raymarching/__init__.py", "raymarching/raymarching.py",
    "gridencoder/__init__.py", "gridencoder/grid.py",
    "shencoder/__init__.py", "shencoder/sphere_harmonics.py",
    "freqencoder/__init__.py", "freqencoder/freq.py",
    "nerf/renderer.py", "nerf/network.py", "nerf/utils.py",
]

_STUBS = ["cv2", "mcubes", "trimesh", "xatlas", "pymeshlab", "imageio", "tensorboardX", "torch_ema", "lpips", "pytorch3d",
          "pytorch3d.structures", "pytorch3d.loss", "dearpygui", "dearpygui.dearpygui", "torch_scatter"]


def compile_pyref(verbose=False) -> bool:
    """Byte-compile the reference's Python files of this path into oracle/_ref/pyref (sourceless layout). False without a checkout."""
    if not os.path.isdir(REFERENCE):
        return False
    for rel in PY_FILES:
        src = os.path.join(REFERENCE, rel)
        dst = os.path.join(PYREF, rel + "c")
        if os.path.exists(dst) and os.path.getmtime(dst) > os.path.getmtime(src):
            continue
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        py_compile.compile(src, cfile=dst, dfile=rel, doraise=True)
        if verbose:
            print(f"[oracle/_ref] pyref: {rel}")
    # `nerf` has no __init__.py in the reference (namespace package): nothing to compile for it
    return True


def root():
    """Directory to put on sys.path: the checkout when present, else the byte-compiled copy. None when neither exists."""
    if os.path.isdir(REFERENCE):
        return REFERENCE
    if os.path.exists(os.path.join(PYREF, "nerf", "renderer.pyc")):
        return PYREF
    return None


def available() -> bool:
    return root() is not None


def _stub(name):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__path__ = []          # lets `import a.b` resolve through sys.modules
    m.__n2m_stub__ = True
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        setattr(_stub(parent), child, m)
    return m


# installed in this image but not necessarily on every box: imported for real when present, stubbed otherwise
_OPTIONAL = {"matplotlib": [], "matplotlib.pyplot": [], "sklearn": [], "sklearn.neighbors": ["NearestNeighbors"], "scipy": [],
             "scipy.ndimage": ["binary_dilation", "binary_erosion"], "tqdm": ["tqdm", "trange"], "rich": [], "rich.console": ["Console"],
             "packaging": [], "packaging.version": []}


def _install_stubs():
    for n in _STUBS:
        _stub(n)
    for n, names in _OPTIONAL.items():
        try:
            importlib.import_module(n)
        except Exception:
            m = _stub(n)
            for a in names:
                setattr(m, a, None)
    sys.modules["torch_ema"].ExponentialMovingAverage = type("ExponentialMovingAverage", (), {})
    sys.modules["pytorch3d.structures"].Meshes = type("Meshes", (), {})
    for fn in ("mesh_laplacian_smoothing", "mesh_normal_consistency", "mesh_edge_loss"):
        setattr(sys.modules["pytorch3d.loss"], fn, None)
    sys.modules["lpips"].LPIPS = type("LPIPS", (), {})


_BACKENDS = {}
_state = {"loaded": False}


def _backend_tables(kind):
    """(raymarching, gridencoder, shencoder, freqencoder) function tables of one kind."""
    if kind in _BACKENDS:
        return _BACKENDS[kind]
    if kind == "ref":
        from oracle import build_ref
        if not build_ref.available():
            assert build_ref.build(verbose=False), "oracle/_ref is not built and /root/reference is absent"
        rm, ge, sh = build_ref.load()
        fq = build_ref.load_freq()
    elif kind == "hip":
        from nerf2mesh_amd import backends
        bdir = backends.path()

        def imp(n):
            spec = importlib.util.spec_from_file_location("n2m_hip" + n, os.path.join(bdir, n + ".py"))
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            return m
        rm, ge, sh, fq = (imp(n) for n in ("_raymarching_mob", "_gridencoder", "_shencoder", "_freqencoder"))
    else:
        raise ValueError(kind)
    _BACKENDS[kind] = (rm, ge, sh, fq)
    return _BACKENDS[kind]


def load(backend="ref"):
    """Import the unchanged reference modules; returns a namespace with `.network` (nerf.network), `.renderer`, `.utils`,
    `.raymarching`, `.grid`.  `backend` selects the kernels under the wrappers (see module docstring)."""
    r = root()
    assert r is not None, "neither /root/reference nor oracle/_ref/pyref is present"
    import torch  # noqa: F401
    if not _state["loaded"]:
        _install_stubs()
        _stub("nvdiffrast")
        _stub("nvdiffrast.torch")
        rm, ge, sh, fq = _backend_tables(backend)
        # the wrappers' `import _raymarching_mob as _backend` seam (raymarching/raymarching.py:9-12 etc.)
        sys.modules["_raymarching_mob"], sys.modules["_gridencoder"] = rm, ge
        sys.modules["_shencoder"], sys.modules["_freqencoder"] = sh, fq
        if r not in sys.path:
            sys.path.insert(0, r)
        _state["loaded"] = True
    ns = types.SimpleNamespace()
    ns.raymarching = importlib.import_module("raymarching.raymarching")
    ns.grid = importlib.import_module("gridencoder.grid")
    ns.sh = importlib.import_module("shencoder.sphere_harmonics")
    ns.freq = importlib.import_module("freqencoder.freq")
    # nerf/utils.py:45-52 decorates two colour-space helpers (not on this path) with torch.jit.script, which wants source text:
    # with the byte-compiled copy the decorator is the identity for the duration of the import
    import torch.jit
    script = torch.jit.script
    if r == PYREF:
        torch.jit.script = lambda fn, *a, **k: fn
    try:
        ns.utils = importlib.import_module("nerf.utils")
        ns.renderer = importlib.import_module("nerf.renderer")
        ns.network = importlib.import_module("nerf.network")
    finally:
        torch.jit.script = script
    use_backend(backend)
    return ns


def use_backend(kind):
    """Point the reference wrappers' `_backend` globals at the "ref" (CPU, reference kernels) or "hip" (libn2m_hip.so) tables;
    with "hip", nerf/renderer.py's `dr` (nvdiffrast.torch, :15) becomes the HIP facade."""
    rm, ge, sh, fq = _backend_tables(kind)
    sys.modules["raymarching.raymarching"]._backend = rm
    sys.modules["gridencoder.grid"]._backend = ge
    sys.modules["shencoder.sphere_harmonics"]._backend = sh
    sys.modules["freqencoder.freq"]._backend = fq
    if kind == "hip":
        # the facade FILES a maintainer puts on sys.path (nerf2mesh_amd/backends/nvdiffrast/torch.py, backends/torch_scatter.py), loaded by path
        from nerf2mesh_amd import backends
        bdir = backends.path()

        def imp(name, rel):
            spec = importlib.util.spec_from_file_location(name, os.path.join(bdir, rel))
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            return m
        sys.modules["nerf.renderer"].dr = imp("n2m_hip_nvdiffrast_torch", os.path.join("nvdiffrast", "torch.py"))
        sys.modules["torch_scatter"].scatter_add = imp("n2m_hip_torch_scatter", "torch_scatter.py").scatter_add
    else:
        # stage 1 on the CPU: nvdiffrast.torch = the scalar C rasteriser (forward only, parity unpinned -- oracle/nvdiffrast_oracle.py);
        # torch_scatter.scatter_add(src, index, out=out) = out.index_add_ (its documented semantics for a 1-D index)
        sys.modules["nerf.renderer"].dr = importlib.import_module("oracle.nvdiffrast_oracle")
        sys.modules["torch_scatter"].scatter_add = lambda src, index, dim=-1, out=None, dim_size=None: out.index_add_(0, index, src)
    sys.modules["nerf.renderer"].TORCH_SCATTER = None      # nerf/renderer.py:43,934-938 caches the module on first use
    sys.modules["trimesh"].load = _trimesh_load


def _trimesh_load(path, **kwargs):
    """Stand-in for trimesh.load(path, force='mesh', skip_material=True, process=False) (nerf/renderer.py:137-141): the reference reads
    `.vertices` [V,3] and `.faces` [F,3] of a binary little-endian PLY (what trimesh, and nerf2mesh_amd.export.write_ply, write)."""
    import numpy as np
    with open(path, "rb") as fp:
        data = fp.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    head = data[:end].decode("ascii").split()
    assert "binary_little_endian" in head, "the stand-in reads binary little-endian PLY only"
    nv, nf = int(head[head.index("vertex") + 1]), int(head[head.index("face") + 1])
    v = np.frombuffer(data, dtype="<f4", count=nv * 3, offset=end).reshape(nv, 3)
    rec = np.frombuffer(data, dtype=np.dtype([("n", "u1"), ("idx", "<i4", (3,))]), count=nf, offset=end + nv * 12)
    assert (rec["n"] == 3).all()
    return types.SimpleNamespace(vertices=v.astype(np.float64), faces=rec["idx"].astype(np.int64))


@contextlib.contextmanager
def cpu_mode():
    """`Tensor.cuda()` / `Module.cuda()` -> identity while the reference Python runs on CPU tensors over the "ref" tables
    (test-only patch)."""
    import torch
    t_cuda, m_cuda = torch.Tensor.cuda, torch.nn.Module.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = t_cuda, m_cuda


def reference_opt(**kw):
    """The `opt` fields NeRFNetwork / NeRFRenderer read (nerf/renderer.py:68-168, nerf/network.py:57-79), main.py defaults."""
    d = dict(bound=1.0, contract=False, grid_size=128, min_near=0.05, density_thresh=10, ind_num=500, ind_dim=0, cuda_ray=True,
             trainable_density_grid=False, stage=0, gui=False, tcnn=False, sdf=False, fp16=False, lr=1e-2,
             normal_anneal_epsilon=1e-4, cos_anneal_ratio=1.0, lambda_density=0, workspace="", mesh="", ckpt="scratch",
             ssaa=2, pos_gradient_boost=1, enable_offset_nerf_grad=False, lr_vert=1e-4)              # stage 1: main.py:49-50,108
    d.update(kw)
    return types.SimpleNamespace(**d)
