#!/usr/bin/env python3
"""Generates nerf2mesh_amd/csrc/mc_table.inc: the 256-case triangle table of the marching-cubes kernels.

The reference calls PyMCubes (`mcubes.marching_cubes`, nerf/renderer.py:524-527, :563, :616), an un-vendored dependency that is not in
this image, so its literal case table is not available.  This table is derived from a RULE instead -- which also makes it watertight by
construction, something the classic complement-symmetric tables are not on ambiguous faces:

  corners   c = x | y << 1 | z << 2 (x, y, z in {0, 1});   case bit c is set when corner c is SOLID, i.e. not (value < iso)
  edges     e = 4 * axis + (u | v << 1): the edge runs along `axis`; (u, v) are the other two coordinates of its corners in increasing
            axis order (axis 0: (y, z), axis 1: (x, z), axis 2: (x, y)).  Its vertex belongs to the grid node at its lower corner.
  faces     every face of the cube with 2 crossed edges gets one segment between them; a face with 4 crossed edges (solid corners on one
            diagonal) gets two segments, each cutting off ONE SOLID corner (the solid corners of an ambiguous face are never joined
            through it).  The choice depends on the face's own four corners only, so the two cells sharing a face always agree.
  loops     every crossed edge then has exactly two segments; following them gives closed loops.  A loop is oriented so that its
            normal (Newell, edge midpoints) points from its solid edge ends to its empty ones -- outward, towards lower values -- rotated
            to start at its smallest edge id, and cut into a triangle fan (l0, l[i], l[i+1]).  Loops are emitted by smallest edge id.

    python tools/gen_mc_table.py > nerf2mesh_amd/csrc/mc_table.inc
"""
import sys


def corner_xyz(c):
    return (c & 1, (c >> 1) & 1, (c >> 2) & 1)


def edge_corners(e):
    axis, u, v = e // 4, e & 1, (e >> 1) & 1
    others = [a for a in range(3) if a != axis]
    lo = [0, 0, 0]
    lo[others[0]], lo[others[1]] = u, v
    hi = list(lo)
    hi[axis] = 1
    cid = lambda p: p[0] | p[1] << 1 | p[2] << 2
    return cid(lo), cid(hi)


def edge_between(c0, c1):
    for e in range(12):
        if set(edge_corners(e)) == {c0, c1}:
            return e
    raise ValueError((c0, c1))


def faces():
    """Each face as its four corners in cyclic order."""
    out = []
    for axis in range(3):
        o = [a for a in range(3) if a != axis]
        for side in range(2):
            ring = []
            for (a, b) in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = [0, 0, 0]
                p[axis], p[o[0]], p[o[1]] = side, a, b
                ring.append(p[0] | p[1] << 1 | p[2] << 2)
            out.append(ring)
    return out


def case_triangles(case):
    solid = [(case >> c) & 1 for c in range(8)]
    crossed = [e for e in range(12) if solid[edge_corners(e)[0]] != solid[edge_corners(e)[1]]]
    if not crossed:
        return []
    adj = {e: [] for e in crossed}
    for ring in faces():
        es = [edge_between(ring[i], ring[(i + 1) % 4]) for i in range(4)]       # es[i] joins ring[i] and ring[i+1]
        cr = [i for i in range(4) if es[i] in adj]
        if len(cr) == 2:
            a, b = es[cr[0]], es[cr[1]]
            adj[a].append(b); adj[b].append(a)
        elif len(cr) == 4:
            for i in range(4):                                                     # corner ring[i] sits between es[i-1] and es[i]
                if solid[ring[i]]:
                    a, b = es[(i - 1) % 4], es[i]
                    adj[a].append(b); adj[b].append(a)
    assert all(len(v) == 2 for v in adj.values()), (case, adj)
    mid = lambda e: [(p + q) / 2 for p, q in zip(corner_xyz(edge_corners(e)[0]), corner_xyz(edge_corners(e)[1]))]
    seen, loops = set(), []
    for e0 in crossed:
        if e0 in seen:
            continue
        loop, prev, cur = [e0], e0, adj[e0][0]
        seen.add(e0)
        while cur != e0:
            loop.append(cur); seen.add(cur)
            a, b = adj[cur]
            assert a != b, case                                                    # a two-edge loop cannot occur on a cube
            prev, cur = cur, (b if a == prev else a)
        assert len(loop) >= 3, (case, loop)
        # orientation
        pts = [mid(e) for e in loop]
        nrm = [0.0, 0.0, 0.0]
        for i in range(len(pts)):
            p, q = pts[i], pts[(i + 1) % len(pts)]
            nrm[0] += (p[1] - q[1]) * (p[2] + q[2])
            nrm[1] += (p[2] - q[2]) * (p[0] + q[0])
            nrm[2] += (p[0] - q[0]) * (p[1] + q[1])
        s_sum, e_sum = [0.0] * 3, [0.0] * 3
        for e in loop:
            for c in edge_corners(e):
                tgt = s_sum if solid[c] else e_sum
                for k in range(3):
                    tgt[k] += corner_xyz(c)[k]
        d = [e_sum[k] / len(loop) - s_sum[k] / len(loop) for k in range(3)]
        dot = sum(nrm[k] * d[k] for k in range(3))
        assert abs(dot) > 1e-9, (case, loop)
        if dot < 0:
            loop = loop[::-1]
        i0 = loop.index(min(loop))
        loops.append(loop[i0:] + loop[:i0])
    loops.sort(key=lambda l: l[0])
    tris = []
    for l in loops:
        for i in range(1, len(l) - 1):
            tris.append((l[0], l[i], l[i + 1]))
    return tris


def table():
    return [case_triangles(c) for c in range(256)]


def main():
    t = table()
    max_t = max(len(x) for x in t)
    w = sys.stdout.write
    w("// GENERATED by tools/gen_mc_table.py (rule-based marching-cubes case table; see that file for the conventions). Do not edit.\n")
    w(f"#define N2M_MC_MAX_TRIS {max_t}\n")
    w("#ifndef N2M_MC_QUAL\n#define N2M_MC_QUAL static\n#endif\n")
    w("N2M_MC_QUAL const unsigned char kMcNumTris[256] = {\n")
    for r in range(0, 256, 32):
        w("    " + ", ".join(str(len(x)) for x in t[r:r + 32]) + ",\n")
    w("};\n")
    w(f"N2M_MC_QUAL const unsigned char kMcTris[256][{3 * max_t}] = {{\n")
    for x in t:
        flat = [e for tri in x for e in tri]
        flat += [255] * (3 * max_t - len(flat))
        w("    {" + ", ".join(f"{e:3d}" for e in flat) + "},\n")
    w("};\n")


if __name__ == "__main__":
    main()
