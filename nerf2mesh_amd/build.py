"""Builds nerf2mesh_amd/lib/libn2m_hip.so: every HIP source under csrc/, cross-compiled for gfx950.

    python -m nerf2mesh_amd.build [--force] [--asm]

hipcc works without a GPU, so this runs in the CPU-only build container; the .so is git-ignored but
travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libn2m_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: index-deciding arithmetic must match the oracle bit for bit (no FMA contraction);
# -munsafe-fp-atomics: float/half2 atomic adds become global_atomic_add_f32 / global_atomic_pk_add_f16
#  instead of CAS loops (all accumulation targets are coarse-grained device allocations).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    inc = os.path.join(os.path.dirname(HERE), "include")
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(inc, f) for f in os.listdir(inc)] + [__file__]
    return max(os.path.getmtime(f) for f in files)


def build(force=False, verbose=True, save_asm=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) > _deps_mtime():
        return LIB
    hdr_m = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc")))
    hdr_m = max(hdr_m, os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "n2m_hip.h")), os.path.getmtime(__file__))

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_m):
            return obj
        cmd = [HIPCC] + FLAGS + ["-c", path, "-o", obj]
        if save_asm:
            cmd += ["-save-temps=obj"]
        if verbose:
            print("[n2m build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=OBJDIR)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[n2m build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_asm="--asm" in sys.argv))
