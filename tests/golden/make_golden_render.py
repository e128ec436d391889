#!/usr/bin/env python3
"""Generates tests/golden/render_{nerf,sdf,garden}.npz: BASELINE config 1 (and, "garden", config 4's shape: bound 16, 5 cascades,
update_aabb, cam_near_far, dt_gamma 1/256, entropy loss, inner/outer TV) run by the reference's UNCHANGED Python callers
(nerf/renderer.py `render` :676-813, `update_extra_state` :1074-1149, `mark_untrained_grid` :985-1071; nerf/network.py :81-189 and
the SDF branch :135-156 / renderer :724-739) over the reference's own kernels compiled for the host (oracle/_ref), CPU tensors, fp32
(torch.cuda.amp.autocast switches itself off without a device).  The scripted iteration is tests/render_case.py::run_case.

Needs /root/reference (build container only).      python tests/golden/make_golden_render.py [nerf|sdf]
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_python as RP      # noqa: E402
import render_case as RC                 # noqa: E402


def reference_model(ns, sdf, device="cpu", garden=False):
    opt = RP.reference_opt(sdf=sdf, density_thresh=0.001 if sdf else 10)        # main.py:141
    if garden:
        opt.bound = RC.GARDEN["bound"]
    with RP.cpu_mode():
        model = ns.network.NeRFNetwork(opt)
    missing = model.load_state_dict(RC.make_state(sdf, rows=RC.GARDEN["rows"] if garden else 6119864, garden=garden), strict=False)
    assert not missing.unexpected_keys and all("embeddings" not in k and "net" not in k for k in missing.missing_keys), missing
    return model.to(device)


def main(which):
    assert os.path.isdir(RP.REFERENCE), "needs the reference checkout"
    ns = RP.load("ref")
    for name in which:
        sdf, garden = name == "sdf", name == "garden"
        t0 = time.time()
        model = reference_model(ns, sdf, garden=garden)
        out = RC.run_case(model, lambda m, poses, intr, cnf=None: m.mark_untrained_grid(RC.dataset_stub(poses, intr, cnf)), "meshgrid", "cpu",
                          sdf=sdf, ctx=RP.cpu_mode, garden=garden)
        fx = RC.compress_for_fixture(out, stride=97 if garden else 32)
        path = os.path.join(HERE, f"render_{name}.npz")
        np.savez_compressed(path, **fx)
        occ = np.unpackbits(out["density_bitfield"]).mean()
        print(f"{name}: {time.time() - t0:.0f} s, num_points {out['num_points']}, occupancy {occ:.4f}, mean density {out['mean_density']:.5f}, "
              f"untrained {fx['density_grid_neg']}, image mean {out['image'].mean():.4f}, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    main([a for a in sys.argv[1:] if a in ("nerf", "sdf", "garden")] or ["nerf", "sdf", "garden"])
