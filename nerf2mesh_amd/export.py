"""On-disk contract of nerf2mesh's two stages (what its web renderer and its own stage 1 read), without the third-party mesh tooling:

* `mesh_stage0/mesh_{cas}.ply`  -- what stage 1 loads with `trimesh.load` (nerf/renderer.py:137-141): binary little-endian PLY,
  float32 vertices + int32 triangle lists (`write_ply` / `read_ply`);
* `mesh_stage1/mesh_{cas}.obj` + `.mtl` -- `v`, `vt` (v flipped: `1 - v`), `f a/ta b/tb c/tc`, material `defaultMat` with
  `map_Kd feat0_{cas}.jpg` (nerf/renderer.py:404-440);
* `mesh_stage1/mlp.json` -- the specular head's weights TRANSPOSED (`p.T.tolist()` under the names `net.0.weight`, `net.1.weight`) plus
  `bound` and `cascade` (nerf/renderer.py:452-468), consumed by renderer.html:424-472.

Marching cubes, mesh cleaning/decimation (PyMCubes, pymeshlab) and UV unwrapping (xatlas) are host-side third-party steps of the
reference's export path (nerf/renderer.py:298-672) and are out of this package's scope (OUT rows); the writers
here take their results as arrays.
"""
import json
import os
import struct

import numpy as np


def write_ply(path, vertices, triangles):
    """Binary little-endian PLY: `vertices` [V,3] float32, `triangles` [F,3] int32 (list property uchar/int, what trimesh writes)."""
    v = np.ascontiguousarray(vertices, dtype="<f4").reshape(-1, 3)
    f = np.ascontiguousarray(triangles, dtype="<i4").reshape(-1, 3)
    header = ("ply\nformat binary_little_endian 1.0\n"
              f"element vertex {v.shape[0]}\nproperty float x\nproperty float y\nproperty float z\n"
              f"element face {f.shape[0]}\nproperty list uchar int vertex_indices\nend_header\n")
    rec = np.empty(f.shape[0], dtype=[("n", "u1"), ("idx", "<i4", (3,))])
    rec["n"] = 3
    rec["idx"] = f
    with open(path, "wb") as fp:
        fp.write(header.encode("ascii"))
        fp.write(v.tobytes())
        fp.write(rec.tobytes())


def read_ply(path):
    """(vertices [V,3] float32, triangles [F,3] int32) of a PLY written by `write_ply` or by trimesh (binary little-endian or ascii,
    x/y/z float vertices possibly followed by other per-vertex properties, triangle faces)."""
    with open(path, "rb") as fp:
        data = fp.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    lines = data[:end].decode("ascii").splitlines()
    fmt = [l.split()[1] for l in lines if l.startswith("format")][0]
    elems, cur = [], None
    for l in lines:
        t = l.split()
        if t[:1] == ["element"]:
            cur = {"name": t[1], "count": int(t[2]), "props": []}
            elems.append(cur)
        elif t[:1] == ["property"] and cur is not None:
            cur["props"].append(t[1:])
    sizes = {"char": 1, "uchar": 1, "int8": 1, "uint8": 1, "short": 2, "ushort": 2, "int16": 2, "uint16": 2, "int": 4, "uint": 4, "int32": 4,
             "uint32": 4, "float": 4, "float32": 4, "double": 8, "float64": 8}
    codes = {"char": "i1", "uchar": "u1", "int8": "i1", "uint8": "u1", "short": "<i2", "ushort": "<u2", "int16": "<i2", "uint16": "<u2",
             "int": "<i4", "uint": "<u4", "int32": "<i4", "uint32": "<u4", "float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8"}
    verts = faces = None
    if fmt == "ascii":
        tok = data[end:].split()
        pos = 0
        for e in elems:
            if e["name"] == "vertex":
                n = len(e["props"])
                arr = np.array(tok[pos:pos + n * e["count"]], dtype=np.float64).reshape(e["count"], n)
                names = [p[-1] for p in e["props"]]
                verts = arr[:, [names.index(c) for c in "xyz"]].astype(np.float32)
                pos += n * e["count"]
            elif e["name"] == "face":
                out = []
                for _ in range(e["count"]):
                    k = int(tok[pos])
                    out.append([int(t) for t in tok[pos + 1:pos + 1 + k]])
                    pos += 1 + k
                faces = np.asarray(out, np.int32)
        return verts, faces
    assert fmt == "binary_little_endian", f"unsupported PLY format {fmt}"
    off = end
    for e in elems:
        if e["name"] == "vertex":
            dt = np.dtype([(p[-1], codes[p[0]]) for p in e["props"]])
            arr = np.frombuffer(data, dtype=dt, count=e["count"], offset=off)
            verts = np.stack([arr[c].astype(np.float32) for c in "xyz"], -1)
            off += dt.itemsize * e["count"]
        elif e["name"] == "face":
            p = e["props"][0]
            assert p[0] == "list", "face element without a list property"
            dt = np.dtype([("n", codes[p[1]]), ("idx", codes[p[2]], (3,))])
            arr = np.frombuffer(data, dtype=dt, count=e["count"], offset=off)
            assert (arr["n"] == 3).all(), "only triangle meshes"
            faces = arr["idx"].astype(np.int32)
            off += dt.itemsize * e["count"]
        else:
            off += sum(sizes[p[0]] for p in e["props"]) * e["count"]
    return verts, faces


def write_obj(path_prefix, cas, vertices, triangles, uvs, uv_triangles, texture_ext="jpg"):
    """`mesh_{cas}.obj` + `mesh_{cas}.mtl` under `path_prefix` exactly as nerf/renderer.py:404-440 lays them out."""
    v, f = np.asarray(vertices), np.asarray(triangles)
    vt, ft = np.asarray(uvs), np.asarray(uv_triangles)
    obj_file = os.path.join(path_prefix, f"mesh_{cas}.obj")
    with open(obj_file, "w") as fp:
        fp.write(f"mtllib mesh_{cas}.mtl \n")
        for p in v:
            fp.write(f"v {p[0]} {p[1]} {p[2]} \n")
        for t in vt:
            fp.write(f"vt {t[0]} {1 - t[1]} \n")
        fp.write("usemtl defaultMat \n")
        for i in range(len(f)):
            fp.write(f"f {f[i, 0] + 1}/{ft[i, 0] + 1} {f[i, 1] + 1}/{ft[i, 1] + 1} {f[i, 2] + 1}/{ft[i, 2] + 1} \n")
    with open(os.path.join(path_prefix, f"mesh_{cas}.mtl"), "w") as fp:
        fp.write("newmtl defaultMat \nKa 1 1 1 \nKd 1 1 1 \nKs 0 0 0 \nTr 1 \nillum 1 \nNs 0 \n")
        fp.write(f"map_Kd feat0_{cas}.{texture_ext} \n")
    return obj_file


def write_mlp_json(path, model):
    """`mlp.json`: specular_net parameters transposed + bound + cascade (nerf/renderer.py:452-468)."""
    mlp = {k: p.detach().cpu().numpy().T.tolist() for k, p in model.specular_net.named_parameters()}
    mlp["bound"] = model.bound
    mlp["cascade"] = model.cascade
    with open(path, "w") as fp:
        json.dump(mlp, fp, indent=2)
    return mlp


# ------------------------------------------------------------------------------------------------ mesh filters
# The two pymeshlab selections export_stage0 uses between marching cubes and the PLY (meshutils.py:63-143), on device tensors.
# (clean_mesh / decimate_mesh, meshutils.py:27-60,146-190, are CPU mesh post-processing through pymeshlab: out of scope)

def remove_vertices(vertices, triangles, selected):
    """Deletes the selected vertices and every face that touches one (`meshing_remove_selected_vertices`, meshutils.py:122-143);
    the remaining vertices keep their order.  vertices [V,3], triangles [F,3] int, selected [V] bool -> (vertices', triangles')."""
    import torch
    keep_v = ~selected
    keep_f = keep_v[triangles.long()].all(dim=1)
    remap = torch.cumsum(keep_v.to(torch.int64), 0) - 1
    return vertices[keep_v], remap[triangles[keep_f].long()].to(triangles.dtype)


def remove_faces(vertices, triangles, remove, dilation=5):
    """meshutils.py:63-92: the KEPT faces (remove == 0) are grown `dilation` times over faces sharing a vertex with them
    (`apply_selection_dilatation`), the rest is deleted, then unreferenced vertices are dropped."""
    import torch
    tri = triangles.long()
    keep = ~remove.bool()
    for _ in range(int(dilation)):
        touched = torch.zeros(vertices.shape[0], dtype=torch.bool, device=vertices.device)
        touched[tri[keep].reshape(-1)] = True
        keep = touched[tri].any(dim=1)
    tri = tri[keep]
    used = torch.zeros(vertices.shape[0], dtype=torch.bool, device=vertices.device)
    used[tri.reshape(-1)] = True
    remap = torch.cumsum(used.to(torch.int64), 0) - 1
    return vertices[used], remap[tri].to(triangles.dtype)


# ------------------------------------------------------------------------------------------------ texture atlas / images
def grid_atlas(n_faces, margin=0.12, device="cpu"):
    """A trivial UV atlas: face i gets its own right triangle in cell i // 2 of a G x G grid (two faces per cell, `margin` of the cell
    kept free around each).  Returns (vt [3 F, 2] in [0, 1], ft [F, 3] int32).  Stand-in for the xatlas unwrap of nerf/renderer.py:312-322
    (xatlas is an un-vendored dependency and UV unwrapping is outside the hot path): valid and seam-free per face,
    but it spends the texture uniformly per face instead of per area."""
    import math
    import torch
    F = int(n_faces)
    G = max(1, math.ceil(math.sqrt((F + 1) // 2)))
    s, m = 1.0 / G, margin / G
    i = torch.arange(F, device=device)
    cell = i // 2
    ox, oy = (cell % G).float() * s, (cell // G).float() * s
    lower = torch.tensor([[m, m], [s - 2 * m, m], [m, s - 2 * m]], device=device)
    upper = torch.tensor([[s - m, s - m], [2 * m, s - m], [s - m, 2 * m]], device=device)
    tri = torch.where((i % 2 == 0)[:, None, None], lower[None], upper[None])                       # [F, 3, 2]
    vt = (tri + torch.stack([ox, oy], dim=-1)[:, None, :]).reshape(-1, 2).contiguous()
    ft = torch.arange(3 * F, dtype=torch.int32, device=device).reshape(F, 3)
    return vt, ft


def write_jpg(path, rgb):
    """[H, W, 3] uint8 RGB -> JPEG.  The reference writes through cv2.imwrite (default quality 95, nerf/renderer.py:399-400) after an
    RGB -> BGR swap for cv2's channel order: the file holds the RGB image either way."""
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(rgb, dtype=np.uint8), "RGB").save(path, quality=95)
