#!/usr/bin/env python3
"""A/B builds of one source of libn2m_hip.so: compiles csrc/<src>.hip with extra flags and links it with the other objects of the regular
build into nerf2mesh_amd/lib/libn2m_hip_<name>.so (select it with N2M_HIP_LIB=<path>).     tools/build_variant.py <name> <src> [flags...]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf2mesh_amd import build as B
name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build(verbose=False)
obj = os.path.join(B.OBJDIR, f"{src}_{name}.o")
subprocess.run([B.HIPCC] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, src + ".hip"), "-o", obj], check=True, cwd=B.OBJDIR)
objs = [os.path.join(B.OBJDIR, f.replace(".hip", ".o")) for f in B.sources() if f != src + ".hip"] + [obj]
out = os.path.join(B.LIBDIR, f"libn2m_hip_{name}.so")
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
