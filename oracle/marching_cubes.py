"""CPU restatement of the marching-cubes path (TEST INFRASTRUCTURE: only tests/ may import this; the product never does).

PARITY UNPINNED.  The reference extracts its stage-0 mesh with `mcubes.marching_cubes(sigmas, density_thresh)` (nerf/renderer.py:524-527,
outer cascades :563, :616) -- PyMCubes, an un-vendored pip dependency that is absent from /root/reference and from this image, so neither
its case table nor its vertex order can be checked here.  What is restated is the published algorithm (Lorensen & Cline 1987): classify
the 8 corners of every cell against the iso value (PyMCubes: bit set when value < iso; here the complementary "solid" bit, same surface),
put one vertex on every grid edge whose ends differ, at the linear interpolation PyMCubes uses
(x1 + (iso - f1) / (f2 - f1), doubles, lower corner first), and triangulate every cell from a 256-case table.  The table is generated from
the rule documented in tools/gen_mc_table.py (ambiguous faces never join their solid corners); this file derives it a SECOND time with
different code (coordinates instead of index tables, the orientation from the trilinear field's gradient instead of Newell normals) and
tests/test_marching_cubes.py asserts that the two agree case by case.  Vertex order (grid node, then axis) and triangle order (cell, then
table order) are this repo's conventions; the tests also check order-free properties (closedness, orientation, distance to analytic
surfaces, Euler characteristic).
"""
import itertools

import numpy as np

_CORNERS = np.array([[c & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], dtype=np.int64)


def _edge_id(p, q):
    """Edge id of the cube edge between corner coordinates p and q (4 * axis + u + 2 v; u, v = the other coordinates in axis order)."""
    p, q = np.asarray(p), np.asarray(q)
    axis = int(np.nonzero(p != q)[0][0])
    rest = [int(p[a]) for a in range(3) if a != axis]
    return 4 * axis + rest[0] + 2 * rest[1]


def _edge_ends(e):
    axis, u, v = e // 4, e & 1, (e >> 1) & 1
    lo = np.zeros(3, np.int64)
    rest = [a for a in range(3) if a != axis]
    lo[rest[0]], lo[rest[1]] = u, v
    hi = lo.copy()
    hi[axis] = 1
    return lo, hi


def _trilinear_grad(vals, p):
    """Gradient at p of the trilinear interpolant of the 8 corner values (vals indexed x | y << 1 | z << 2)."""
    g = np.zeros(3)
    for c in range(8):
        xyz = _CORNERS[c]
        w = [(p[a] if xyz[a] else 1 - p[a]) for a in range(3)]
        for a in range(3):
            dw = (1.0 if xyz[a] else -1.0)
            others = [w[b] for b in range(3) if b != a]
            g[a] += vals[c] * dw * others[0] * others[1]
    return g


def case_table():
    """256 lists of triangles (edge-id triples), derived from coordinates."""
    table = []
    for case in range(256):
        solid = np.array([(case >> c) & 1 for c in range(8)], bool)
        sign = {tuple(_CORNERS[c]): bool(solid[c]) for c in range(8)}
        links = {}
        for axis, side in itertools.product(range(3), range(2)):
            a, b = [x for x in range(3) if x != axis]
            def pt(s, t):
                p = [0, 0, 0]
                p[axis], p[a], p[b] = side, s, t
                return tuple(p)
            ring = [pt(0, 0), pt(1, 0), pt(1, 1), pt(0, 1)]
            cuts = []                                            # (edge id, ring position i) for ring edges i -> i + 1 whose ends differ
            for i in range(4):
                p, q = ring[i], ring[(i + 1) % 4]
                if sign[p] != sign[q]:
                    cuts.append((_edge_id(p, q), i))
            if len(cuts) == 2:
                pairs = [(cuts[0][0], cuts[1][0])]
            elif len(cuts) == 4:
                by_pos = dict((i, e) for e, i in cuts)
                pairs = [(by_pos[(i - 1) % 4], by_pos[i]) for i in range(4) if sign[ring[i]]]      # around each solid corner
            else:
                pairs = []
            for e0, e1 in pairs:
                links.setdefault(e0, []).append(e1)
                links.setdefault(e1, []).append(e0)
        tris, todo = [], set(links)
        loops = []
        while todo:
            start = min(todo)
            loop, prev, cur = [start], start, links[start][0]
            while cur != start:
                loop.append(cur)
                n0, n1 = links[cur]
                prev, cur = cur, (n1 if n0 == prev else n0)
            todo -= set(loop)
            # orientation: fan normals against the trilinear field of +1 (solid) / -1 (empty): outward = down the gradient
            vals = np.where(solid, 1.0, -1.0)
            mids = [(np.add(*_edge_ends(e)) / 2.0) for e in loop]
            score = 0.0
            for i in range(1, len(loop) - 1):
                p0, p1, p2 = mids[0], mids[i], mids[i + 1]
                n = np.cross(p1 - p0, p2 - p0)
                score += float(np.dot(n, -_trilinear_grad(vals, (p0 + p1 + p2) / 3.0)))
            assert abs(score) > 1e-9, (case, loop)
            if score < 0:
                loop = loop[::-1]
            k = loop.index(min(loop))
            loops.append(loop[k:] + loop[:k])
        for loop in sorted(loops, key=lambda l: l[0]):
            tris += [(loop[0], loop[i], loop[i + 1]) for i in range(1, len(loop) - 1)]
        table.append(tris)
    return table


_TABLE = None


def marching_cubes(field, iso, div=1.0, mul=1.0, add=0.0):
    """vertices [V, 3] float32 = float32(((index-space position) / div) * mul + add), triangles [T, 3] int32.  field [R0, R1, R2] float32,
    iso a Python float (double).  Solid = not (value < iso).  Vertices: for every grid node in C order, its +x, +y, +z edges that are
    crossed; triangles: for every cell in C order, the case table's triangles."""
    global _TABLE
    if _TABLE is None:
        _TABLE = case_table()
    f = np.asarray(field, dtype=np.float32).astype(np.float64)
    R0, R1, R2 = f.shape
    solid = ~(f < float(iso))
    vid = -np.ones((R0, R1, R2, 3), np.int64)
    verts = []
    shp = (R0, R1, R2)
    for n in _active_nodes(solid):
        i, j, k = (int(v) for v in n)
        for axis in range(3):
            m = [i, j, k]
            m[axis] += 1
            if m[axis] >= shp[axis] or solid[i, j, k] == solid[tuple(m)]:
                continue
            f1, f2 = f[i, j, k], f[tuple(m)]
            pos = np.array([i, j, k], np.float64)
            pos[axis] = pos[axis] + (float(iso) - f1) / (f2 - f1)
            vid[i, j, k, axis] = len(verts)
            verts.append(((pos / div) * mul + add).astype(np.float32))
    tris = []
    if min(shp) >= 2:
        s = solid
        case = np.zeros((R0 - 1, R1 - 1, R2 - 1), np.int64)
        for c in range(8):
            x, y, z = _CORNERS[c]
            case |= s[x:R0 - 1 + x, y:R1 - 1 + y, z:R2 - 1 + z].astype(np.int64) << c
        for cell in np.argwhere((case != 0) & (case != 255)):
            i, j, k = (int(v) for v in cell)
            for tri in _TABLE[int(case[i, j, k])]:
                ids = []
                for e in tri:
                    lo, _ = _edge_ends(e)
                    ids.append(int(vid[i + lo[0], j + lo[1], k + lo[2], e // 4]))
                assert min(ids) >= 0
                tris.append(ids)
    v = np.stack(verts).astype(np.float32) if verts else np.zeros((0, 3), np.float32)
    t = np.asarray(tris, np.int32).reshape(-1, 3)
    return v, t


def marching_cubes_c(field, iso, div=1.0, mul=1.0, add=0.0):
    """The same extraction by the plain-C restatement (oracle/n2m_oracle.c: n2m_oracle_marching_cubes) -- for volumes the emission loop
    above is too slow for.  The case table is this module's (passed to the C side, which holds none)."""
    import ctypes
    from . import oracle as o
    global _TABLE
    if _TABLE is None:
        _TABLE = case_table()
    stride = 15
    num = np.array([len(c) for c in _TABLE], np.uint8)
    tris = np.full((256, stride), 255, np.uint8)
    for k, c in enumerate(_TABLE):
        flat = [e for t in c for e in t]
        tris[k, :len(flat)] = flat
    f = np.ascontiguousarray(field, dtype=np.float32)
    R0, R1, R2 = f.shape
    fn = o.lib().n2m_oracle_marching_cubes
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_uint32, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                   ctypes.c_uint64, ctypes.c_void_p]
    counts = np.zeros(2, np.uint64)
    args = (f.ctypes.data, R0, R1, R2, float(iso), num.ctypes.data, tris.ctypes.data, stride, float(div), float(mul), float(add))
    fn(*args, None, 0, None, 0, counts.ctypes.data)
    v = np.empty((int(counts[0]), 3), np.float32)
    t = np.empty((int(counts[1]), 3), np.int32)
    fn(*args, v.ctypes.data, v.shape[0], t.ctypes.data, t.shape[0], counts.ctypes.data)
    return v, t


def _active_nodes(solid):
    """Nodes that own at least one crossed edge, in C order."""
    R0, R1, R2 = solid.shape
    own = np.zeros(solid.shape, bool)
    own[:-1] |= solid[:-1] != solid[1:]
    own[:, :-1] |= solid[:, :-1] != solid[:, 1:]
    own[:, :, :-1] |= solid[:, :, :-1] != solid[:, :, 1:]
    return np.argwhere(own)


def directed_edge_imbalance(triangles):
    """Number of directed edges (a, b) whose reverse (b, a) does not occur equally often: 0 for a closed, consistently oriented surface."""
    t = np.asarray(triangles, np.int64)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    key = e[:, 0] * (int(t.max()) + 1 if len(t) else 1) + e[:, 1]
    rev = e[:, 1] * (int(t.max()) + 1 if len(t) else 1) + e[:, 0]
    ku, kc = np.unique(key, return_counts=True)
    ru, rc = np.unique(rev, return_counts=True)
    if len(ku) != len(ru) or not np.array_equal(ku, ru):
        return int(len(np.setxor1d(ku, ru))) + 1
    return int((kc != rc).sum())


def signed_volume(vertices, triangles):
    v = np.asarray(vertices, np.float64)[np.asarray(triangles, np.int64)]
    return float(np.einsum("ij,ij->i", v[:, 0], np.cross(v[:, 1], v[:, 2])).sum() / 6.0)
