#!/usr/bin/env python3
"""Generates include/p3d_mc_table.h: the 256-case triangle table of the iso-surface extractor.

The reference calls skimage.measure.marching_cubes(method='lewiner') (_util/eg3d_metrics3d.py:186-210); skimage is
not installed here and its case tables are not under /root/reference, so no table is restated from anywhere: this
script DERIVES one from first principles, and the derivation is the specification:

  corners   v = 0..7 at offsets (da,db,dc) = (v>>2&1, v>>1&1, v&1) along the volume axes (a slowest, c fastest)
  edges     e = 4*axis + rank; axis 0/1/2 = along c/b/a; rank = position of the edge's lower corner among the four corners
            whose axis bit is clear, in increasing corner order
  case      bit v set  <=>  value(corner v) > level          ("inside")
  per face  the crossed edges of a face are joined pairwise; a face with four crossings (inside corners on a diagonal)
            always ISOLATES its inside corners.  The rule reads only the face's own four flags, so the two cubes sharing
            the face draw the same segments -> the mesh is watertight by construction.
  per cube  segments chain into closed loops (every crossed edge lies on exactly two faces); each loop is rotated to start
            at its smallest edge id, oriented so that its normal points from the inside corners to the outside ones
            (gradient_direction='descent': outward = towards lower values) and cut into triangles by the first
            triangulation (fixed enumeration order) none of whose chords lies in a cube face.

Run:  python tools/gen_mc_table.py  (rewrites the header; tests/test_mcubes_cpu.py regenerates and compares)."""
import os
import numpy as np

AXBIT = (1, 2, 4)  # axis 0 = c (bit 0), 1 = b (bit 1), 2 = a (bit 2)


def corner_pos(v):  # (a, b, c)
    return np.array([(v >> 2) & 1, (v >> 1) & 1, v & 1], dtype=np.float64)


def edges():
    out = []
    for axis in range(3):
        lows = [v for v in range(8) if not v & AXBIT[axis]]
        for v0 in lows:
            out.append((v0, v0 | AXBIT[axis]))
    return out  # e -> (v0, v1), v0 < v1


EDGES = edges()
EDGE_ID = {frozenset(e): i for i, e in enumerate(EDGES)}


def faces():
    out = []
    for axis in range(3):
        u, w = [AXBIT[x] for x in range(3) if x != axis]
        for side in (0, 1):
            base = AXBIT[axis] * side
            out.append([base, base | u, base | u | w, base | w])  # cyclic
    return out


FACES = faces()


def loops_of_case(case):
    inside = [(case >> v) & 1 for v in range(8)]
    link = {}
    for q in FACES:
        fe = [(q[i], q[(i + 1) % 4]) for i in range(4)]
        crossed = [i for i in range(4) if inside[fe[i][0]] != inside[fe[i][1]]]
        pairs = []
        if len(crossed) == 2:
            pairs.append((crossed[0], crossed[1]))
        elif len(crossed) == 4:
            for i in range(4):
                if inside[q[i]]:  # isolate the inside corner q[i]: its two face edges are (i-1) and i
                    pairs.append(((i - 1) % 4, i))
        for x, y in pairs:
            ex, ey = EDGE_ID[frozenset(fe[x])], EDGE_ID[frozenset(fe[y])]
            link.setdefault(ex, []).append(ey)
            link.setdefault(ey, []).append(ex)
    assert all(len(v) == 2 for v in link.values())
    seen, loops = set(), []
    for start in sorted(link):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            a, b = link[cur]
            nxt = a if a != prev or (a == b and len(loop) == 1) else b
            if nxt == start and len(loop) > 1:
                break
            if nxt in seen:  # 2-cycle guard (cannot happen for a cube, asserted)
                raise AssertionError("degenerate loop")
            loop.append(nxt)
            seen.add(nxt)
            prev, cur = cur, nxt
        loops.append(loop)
    # orientation
    out = []
    for loop in loops:
        pts = [0.5 * (corner_pos(EDGES[e][0]) + corner_pos(EDGES[e][1])) for e in loop]
        nrm = np.zeros(3)
        for i in range(len(pts)):  # Newell
            p, q = pts[i], pts[(i + 1) % len(pts)]
            nrm += np.cross(p, q)
        d = np.zeros(3)
        for e in loop:
            v0, v1 = EDGES[e]
            vin, vout = (v0, v1) if inside[v0] else (v1, v0)
            d += corner_pos(vout) - corner_pos(vin)
        s = float(np.dot(nrm, d))
        assert abs(s) > 1e-9, (case, loop)
        if s < 0:
            loop = [loop[0]] + loop[1:][::-1]
        out.append(loop)
    return out


def share_face(e0, e1):
    c = set(EDGES[e0]) | set(EDGES[e1])
    return any(c <= set(q) for q in FACES)


def triangulations(poly):
    """All triangulations of the polygon (list of vertex labels, cyclic), each a list of triangles that keeps the polygon's
    orientation; deterministic order (the triangle on the edge poly[0]-poly[-1] picks its apex in increasing position)."""
    if len(poly) < 3:
        return [[]]
    if len(poly) == 3:
        return [[tuple(poly)]]
    out = []
    for k in range(1, len(poly) - 1):
        for left in triangulations(poly[:k + 1]):
            for right in triangulations(poly[k:]):
                out.append(left + [(poly[0], poly[k], poly[-1])] + right)
    return out


def triangulate(loop):
    """First triangulation none of whose chords lies in a cube face.  A chord in a face would put a whole triangle, or an
    edge the neighbouring cube may also draw, into the shared face: the mesh would stop being a 2-manifold there."""
    n = len(loop)
    pos = {e: i for i, e in enumerate(loop)}
    for tris in triangulations(loop):
        ok = True
        for t in tris:
            for x, y in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
                adjacent = (pos[x] - pos[y]) % n in (1, n - 1)
                if not adjacent and share_face(x, y):
                    ok = False
        if ok:
            return tris
    raise AssertionError(("no face-free triangulation", loop))


def table():
    tri = []
    for case in range(256):
        row = []
        for loop in loops_of_case(case):
            for t in triangulate(loop):
                row += list(t)
        tri.append(row)
    return tri


def header_text():
    tri = table()
    mx = max(len(r) for r in tri) // 3
    width = 3 * mx + 1
    L = []
    L.append("/* GENERATED by tools/gen_mc_table.py — do not edit.  Triangle table of the iso-surface extractor: derived, not copied")
    L.append(" * (see the generator's docstring for the rule).  Part of the contract shared by the HIP kernels and oracle/p3d_oracle.c. */")
    L.append("#ifndef P3D_MC_TABLE_H")
    L.append("#define P3D_MC_TABLE_H")
    L.append("#ifndef P3D_MC_QUAL /* storage qualifier: the HIP translation unit defines it as `static __device__ const` */")
    L.append("#define P3D_MC_QUAL static const")
    L.append("#endif")
    L.append(f"#define P3D_MC_MAXTRI {mx}")
    L.append(f"#define P3D_MC_ROW {width}")
    L.append("/* edge e joins corners P3D_MC_EDGE[e][0] < P3D_MC_EDGE[e][1]; corner v sits at (a,b,c) offsets (v>>2&1, v>>1&1, v&1);")
    L.append(" * the edge is OWNED by the grid point of its lower corner, slot e>>2 (0: along c, 1: along b, 2: along a) */")
    L.append("P3D_MC_QUAL unsigned char P3D_MC_EDGE[12][2] = {" + ", ".join("{%d,%d}" % e for e in EDGES) + "};")
    L.append("P3D_MC_QUAL unsigned char P3D_MC_NTRI[256] = {")
    for i in range(0, 256, 32):
        L.append("    " + ",".join(str(len(r) // 3) for r in tri[i:i + 32]) + ",")
    L.append("};")
    L.append("/* P3D_MC_TRI[case][3*t + k]: edge id of corner k of triangle t; -1 terminated */")
    L.append("P3D_MC_QUAL signed char P3D_MC_TRI[256][P3D_MC_ROW] = {")
    for r in tri:
        rr = r + [-1] * (width - len(r))
        L.append("    {" + ",".join("%2d" % x for x in rr) + "},")
    L.append("};")
    L.append("#endif")
    return "\n".join(L) + "\n"


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "p3d_mc_table.h")
    txt = header_text()
    open(path, "w").write(txt)
    tri = table()
    print("wrote", path, "max triangles per case", max(len(r) for r in tri) // 3, "total", sum(len(r) for r in tri) // 3)
