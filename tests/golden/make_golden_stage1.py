#!/usr/bin/env python3
"""Generates tests/golden/render_stage1.npz: the scripted stage-1 iteration of tests/stage1_case.py run by the reference's UNCHANGED
Python (nerf/renderer.py:123-165 mesh loading, :816-921 `render_stage1`, :924-943 `update_triangles_errors`, :947-981
`mark_unseen_triangles`; nerf/network.py `rgb`) on the CPU: grid encoder = the reference's own kernels compiled for the host
(oracle/_ref), `nvdiffrast.torch` = the scalar C rasteriser (oracle/nvdiffrast_oracle.py -- nvdiffrast is not under /root/reference:
the three raster operators stay PARITY-UNPINNED, the fixture pins the caller around them), fp32, forward quantities only.

Needs /root/reference (build container only).      python tests/golden/make_golden_stage1.py
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_python as RP      # noqa: E402
import render_case as RC                 # noqa: E402
import stage1_case as SC                 # noqa: E402


def reference_model(ns, workspace, device="cpu", fp16=False):
    """NeRFNetwork(opt) with opt.stage = 1: the constructor itself loads <workspace>/mesh_stage0/mesh_0.ply (nerf/renderer.py:123-165)."""
    opt = RP.reference_opt(stage=1, workspace=workspace, fp16=fp16)
    ctx = RP.cpu_mode() if device == "cpu" else __import__("contextlib").nullcontext()
    with ctx:
        model = ns.network.NeRFNetwork(opt)
    missing = model.load_state_dict(RC.make_state(False), strict=False)
    assert not missing.unexpected_keys, missing
    return model.to(device)


def main():
    assert os.path.isdir(RP.REFERENCE), "needs the reference checkout"
    ns = RP.load("ref")
    RP.use_backend("ref")
    t0 = time.time()
    with tempfile.TemporaryDirectory() as ws:
        SC.write_workspace(ws)
        model = reference_model(ns, ws)
        out = SC.run_case(model, "cpu", grad=False, ctx=RP.cpu_mode)
    path = os.path.join(HERE, "render_stage1.npz")
    np.savez_compressed(path, **out)
    cov = (out["weights_sum"] > 0).mean()
    print(f"stage1: {time.time() - t0:.0f} s, coverage {cov:.3f}, faces with errors {(out['triangles_errors_cnt'] > 0).sum()}, "
          f"unseen {out['unseen_count']}, loss {out['loss']:.5f}, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
