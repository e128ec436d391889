#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- builds `oracle/_ref/`: the reference's own kernels, compiled for the CPU.

The reference's arithmetic for this path lives in three CUDA files
(`raymarching/src/raymarching.cu`, `gridencoder/src/gridencoder.cu`, `shencoder/src/shencoder.cu`) plus
their pybind11 `bindings.cpp`.  They cannot run as CUDA here, but the kernels use nothing beyond the basic
SIMT model (one thread per item, no shared memory / barriers / warp intrinsics), so this recipe compiles the
sources *where they lie* under /root/reference with g++ against `oracle/ref_shim/` (a host stand-in for
`cuda.h` & friends that sweeps blockIdx/threadIdx serially).  Two textual rewrites are applied to the
stream fed to the compiler (never written to disk, never committed):

  1. `kernel<<<grid, block>>>(args);`  ->  `ref_launch(grid, block, [&]{ kernel(args); });`
  2. `CHECK_CUDA(x);` lines dropped (the tensors are CPU tensors here).

Everything else -- every arithmetic statement of every kernel and every host wrapper -- is the reference's.
Outputs (python extension modules taking CPU torch tensors, same function table as the CUDA builds):

  oracle/_ref/_ref_raymarching.so   <- raymarching/src/{raymarching.cu,bindings.cpp}
  oracle/_ref/_ref_gridencoder.so   <- gridencoder/src/{gridencoder.cu,bindings.cpp}
  oracle/_ref/_ref_shencoder.so     <- shencoder/src/{shencoder.cu,bindings.cpp}
  oracle/_ref/_ref_freqencoder.so   <- freqencoder/src/{freqencoder.cu,bindings.cpp}

`oracle/_ref/` is git-ignored (binaries only) but travels to the GPU box with the snapshot.
Used by: tests/ (to pin oracle/n2m_oracle.c), tests/golden/make_golden.py (fixture generation).
Never imported by the product package.
"""
import os
import re
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
SHIM = os.path.join(HERE, "ref_shim")
REFERENCE = os.environ.get("N2M_REFERENCE", "/root/reference")

EXTS = {
    "_ref_raymarching": ("raymarching", "raymarching.cu"),
    "_ref_gridencoder": ("gridencoder", "gridencoder.cu"),
    "_ref_shencoder": ("shencoder", "shencoder.cu"),
    "_ref_freqencoder": ("freqencoder", "freqencoder.cu"),
}

_LAUNCH = re.compile(r"(\w+(?:<[^<>;]*>)?)\s*<<<(.*?)>>>\s*\((.*?)\);", re.S)


def rewrite(src: str) -> str:
    src = "\n".join(l for l in src.splitlines() if not re.match(r"\s*CHECK_CUDA\(", l)) + "\n"
    # a launch may span several lines; its argument list ends at the first `);`
    return _LAUNCH.sub(lambda m: f"ref_launch({m.group(2)}, [&]{{ {m.group(1)}({m.group(3)}); }});", src)


def available() -> bool:
    return all(os.path.exists(os.path.join(OUT, n + ".so")) for n in EXTS)


def build(force: bool = False, verbose: bool = True) -> bool:
    """Build all three modules. Returns False (and builds nothing) when /root/reference is absent."""
    if not os.path.isdir(REFERENCE):
        return False
    import torch
    from torch.utils import cpp_extension as ce

    os.makedirs(OUT, exist_ok=True)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = [SHIM] + ce.include_paths() + [sysconfig.get_paths()["include"]]
    common = ["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-w",
              f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"] + [f"-I{p}" for p in inc]
    for name, (pkg, cu) in EXTS.items():
        so = os.path.join(OUT, name + ".so")
        srcdir = os.path.join(REFERENCE, pkg, "src")
        cu_path = os.path.join(srcdir, cu)
        bind_path = os.path.join(srcdir, "bindings.cpp")
        if (not force and os.path.exists(so)
                and os.path.getmtime(so) > max(os.path.getmtime(cu_path), os.path.getmtime(__file__),
                                               os.path.getmtime(os.path.join(SHIM, "cuda.h")))):
            continue
        if verbose:
            print(f"[oracle/_ref] {name}: compiling {cu_path} for the host", flush=True)
        obj_cu = os.path.join(OUT, name + ".cu.o")
        obj_b = os.path.join(OUT, name + ".bind.o")
        with open(cu_path) as f:
            stream = rewrite(f.read())
        subprocess.run(common + ["-x", "c++", "-c", "-", "-o", obj_cu], input=stream.encode(), check=True)
        subprocess.run(common + [f"-DTORCH_EXTENSION_NAME={name}", f"-I{srcdir}", "-c", bind_path, "-o", obj_b],
                       check=True)
        subprocess.run(["g++", "-shared", obj_cu, obj_b, "-o", so, f"-L{tlib}", f"-Wl,-rpath,{tlib}",
                        "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python"], check=True)
        os.remove(obj_cu)
        os.remove(obj_b)
    # the reference's Python callers of this path, byte-compiled (oracle/_ref/pyref): what the GPU box imports instead of the checkout
    if os.path.dirname(HERE) not in sys.path:
        sys.path.insert(0, os.path.dirname(HERE))
    from oracle import ref_python
    ref_python.compile_pyref(verbose=verbose)
    return True


def load():
    """Import the modules (after `import torch`). Returns (raymarching, gridencoder, shencoder)."""
    import importlib
    import torch  # noqa: F401  (must be loaded before the extension modules)
    if OUT not in sys.path:
        sys.path.insert(0, OUT)
    return tuple(importlib.import_module(n) for n in ("_ref_raymarching", "_ref_gridencoder", "_ref_shencoder"))


def load_freq():
    """The reference's freqencoder module (freqencoder/src/freqencoder.cu) compiled for the host."""
    import importlib
    import torch  # noqa: F401
    if OUT not in sys.path:
        sys.path.insert(0, OUT)
    return importlib.import_module("_ref_freqencoder")


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("built" if ok else f"{REFERENCE} not present: nothing built")
