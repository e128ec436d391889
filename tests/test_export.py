"""On-disk contract writers (nerf2mesh_amd/export.py) against the layout nerf/renderer.py:137-141,404-468 and renderer.html:424-472 expect."""
import json
import os

import numpy as np


def test_ply_round_trip_and_header(tmp_path):
    from nerf2mesh_amd import export
    rng = np.random.default_rng(0)
    v = rng.normal(size=(57, 3)).astype(np.float32)
    f = rng.integers(0, 57, (101, 3)).astype(np.int32)
    p = str(tmp_path / "mesh_0.ply")
    export.write_ply(p, v, f)
    head = open(p, "rb").read(200).decode("ascii", "ignore")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 57\n") and "property list uchar int vertex_indices" in head
    v2, f2 = export.read_ply(p)
    assert np.array_equal(v, v2) and np.array_equal(f, f2)
    # an ascii file with extra vertex properties (what other tools write) reads too
    with open(tmp_path / "a.ply", "w") as fp:
        fp.write("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n"
                 "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0 255\n1 0 0 255\n0 1 0 255\n3 0 1 2\n")
    v3, f3 = export.read_ply(str(tmp_path / "a.ply"))
    assert v3.shape == (3, 3) and f3.tolist() == [[0, 1, 2]] and v3[1, 0] == 1


def test_obj_mtl_and_mlp_json_layout(tmp_path):
    import torch
    from nerf2mesh_amd import export
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    f = np.array([[0, 1, 2]], np.int32)
    vt = np.array([[0.1, 0.2], [0.9, 0.2], [0.1, 0.8]], np.float32)
    export.write_obj(str(tmp_path), 0, v, f, vt, f)
    obj = open(tmp_path / "mesh_0.obj").read().splitlines()
    assert obj[0].strip() == "mtllib mesh_0.mtl" and obj[1].startswith("v 0.0 0.0 0.0")
    u0, v0 = (float(t) for t in [l for l in obj if l.startswith("vt")][0].split()[1:])
    assert abs(u0 - 0.1) < 1e-6 and abs(v0 - 0.8) < 1e-6                                                     # v flipped: 1 - v
    assert [l for l in obj if l.startswith("f ")][0].strip() == "f 1/1 2/2 3/3" and "usemtl defaultMat " in obj
    mtl = open(tmp_path / "mesh_0.mtl").read()
    assert mtl.startswith("newmtl defaultMat") and "map_Kd feat0_0.jpg" in mtl
    model = NeRFNetwork(make_options(bound=1))
    mlp = export.write_mlp_json(str(tmp_path / "mlp.json"), model)
    back = json.load(open(tmp_path / "mlp.json"))
    assert set(back) == {"net.0.weight", "net.1.weight", "bound", "cascade"} and back["cascade"] == 1
    w0 = np.asarray(back["net.0.weight"])
    assert w0.shape == (6, 32) and np.allclose(w0, model.specular_net.net[0].weight.detach().numpy().T)      # transposed: [in, out]
    assert np.asarray(back["net.1.weight"]).shape == (32, 3)
