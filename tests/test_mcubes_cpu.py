"""Iso-surface extractor, CPU side: the case table is what its generator derives, and the oracle's meshes
have the properties the specification promises (oracle/p3d_oracle_mc.c header: parity with skimage is UNPINNED, so the
properties are the pin)."""
import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import oracle as O


def test_table_header_is_what_the_generator_derives():
    import gen_mc_table as g
    assert open(os.path.join(ROOT, "include", "p3d_mc_table.h")).read() == g.header_text()
    tri = g.table()
    assert max(len(r) for r in tri) == 15 and len(tri[0]) == 0 and len(tri[255]) == 0
    for case, row in enumerate(tri):  # every crossed edge of the case is used, no uncrossed edge is
        crossed = {e for e, (v0, v1) in enumerate(g.EDGES) if ((case >> v0) ^ (case >> v1)) & 1}
        assert set(row) == crossed


def edge_stats(faces):
    """directed edge -> count; a closed, consistently oriented 2-manifold has every directed edge exactly once and its
    reverse exactly once."""
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]).astype(np.int64)
    key = e[:, 0] * (1 << 32) + e[:, 1]
    rkey = e[:, 1] * (1 << 32) + e[:, 0]
    uniq, cnt = np.unique(key, return_counts=True)
    return uniq, cnt, np.isin(rkey, uniq)


def padded_noise(n, seed):
    v = np.full((n, n, n), -1.0, np.float32)
    v[1:-1, 1:-1, 1:-1] = np.random.default_rng(seed).standard_normal((n - 2,) * 3).astype(np.float32)
    return v


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_noise_volume_closed_oriented_manifold(seed):
    """White noise exercises all 256 cases incl. every ambiguous face; the border is 'outside', so the surface is closed."""
    vol = padded_noise(20, seed)
    verts, faces, normals, values = O.marching_cubes(vol, level=0.1)
    assert len(faces) > 5000 and faces.min() == 0 and faces.max() == len(verts) - 1
    uniq, cnt, has_rev = edge_stats(faces)
    assert (cnt == 1).all() and has_rev.all()
    assert len(np.unique(faces)) == len(verts)  # no orphan vertices
    assert not (faces[:, 0] == faces[:, 1]).any() and not (faces[:, 1] == faces[:, 2]).any()
    # vertices sit on grid edges, strictly between the end points, value = the larger end
    frac = verts - np.floor(verts)
    assert ((frac > 0).sum(1) <= 1).all()
    lo = np.floor(verts).astype(int)
    f0 = vol[lo[:, 0], lo[:, 1], lo[:, 2]]
    ax = np.argmax(frac, 1)
    hi = lo.copy(); hi[np.arange(len(hi)), ax] += 1
    f1 = vol[hi[:, 0], hi[:, 1], hi[:, 2]]
    assert ((f0 > 0.1) != (f1 > 0.1)).all()
    assert np.array_equal(values, np.maximum(f0, f1))
    interp = f0 + frac[np.arange(len(ax)), ax] * (f1 - f0)
    assert np.abs(interp - 0.1).max() < 1e-5


def sphere(n, r):
    g = np.arange(n, dtype=np.float32) - (n - 1) / 2
    a, b, c = np.meshgrid(g, g, g, indexing="ij")
    return (r - np.sqrt(a * a + b * b + c * c)).astype(np.float32)


def test_sphere_topology_area_volume_orientation():
    n, r = 48, 17.3
    verts, faces, normals, values = O.marching_cubes(sphere(n, r), level=0.0)
    V, F = len(verts), len(faces)
    uniq, cnt, has_rev = edge_stats(faces)
    assert (cnt == 1).all() and has_rev.all()
    E = len(uniq) // 2
    assert V - E + F == 2  # Euler characteristic of a sphere
    p = verts - (n - 1) / 2
    assert np.abs(np.linalg.norm(p, axis=1) - r).max() < 0.05  # on the level set (linear interpolation of a distance field)
    tri = p[faces]
    cr = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    area = 0.5 * np.linalg.norm(cr, axis=1).sum()
    vol6 = np.einsum("ij,ij->i", tri[:, 0], cr).sum()  # 6 x signed volume, > 0 for outward-facing triangles
    assert abs(area / (4 * np.pi * r * r) - 1) < 0.01
    assert abs(vol6 / 6 / (4 / 3 * np.pi * r ** 3) - 1) < 0.01
    # vertex normals: unit, outward ('descent': towards lower values), aligned with the face normals
    assert np.abs(np.linalg.norm(normals, axis=1) - 1).max() < 1e-5
    assert (np.einsum("ij,ij->i", normals, p / np.linalg.norm(p, axis=1, keepdims=True)) > 0.99).all()


def test_flip0_reads_the_unflipped_grid():
    vol = padded_noise(12, 5)
    a = O.marching_cubes(vol, 0.0)
    b = O.marching_cubes(vol[::-1].copy(), 0.0, flip0=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_empty_and_full_volumes():
    for fill in (-1.0, 1.0):
        v, f, nrm, val = O.marching_cubes(np.full((6, 6, 6), fill, np.float32), 0.0)
        assert len(v) == 0 and len(f) == 0
