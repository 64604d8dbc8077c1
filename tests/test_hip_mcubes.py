"""GPU: the device iso-surface extractor (csrc/p3d_mcubes.hip, through the C ABI) against its CPU specification
(oracle/p3d_oracle_mc.c) — BIT-EXACT for every vertex, normal, value and face index — plus size-independent mesh
properties at the pipeline's grid sizes and the generate.py mesh flow end to end."""
import ctypes as C
import numpy as np
import pytest
import torch

import p3d_testing as T
from test_mcubes_cpu import padded_noise, sphere, edge_stats

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import panic3d_amd
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"
    panic3d_amd._lib.lib()
    return panic3d_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def check_exact(hip, oracle, vol, level, flip0=False):
    got = hip.ops.marching_cubes(dev(vol), level, flip0=flip0)
    ref = oracle.marching_cubes(vol, level, flip0=flip0)
    for name, a, b in zip(("verts", "faces", "normals", "values"), got, ref):
        a = a.cpu().numpy()
        assert a.shape == b.shape, (name, a.shape, b.shape)
        assert np.array_equal(a, b), name
    return ref


@pytest.mark.parametrize("n,seed", [(20, 0), (17, 1), (33, 2), (11, 3)])
def test_noise_volumes_bit_exact(hip, oracle, n, seed):
    """White noise: all 256 cases, every ambiguous face; sizes that are not multiples of the 1024-point block."""
    v, f, _, _ = check_exact(hip, oracle, padded_noise(n, seed), 0.1)
    assert len(f) > 500


def test_sphere_and_flip_bit_exact(hip, oracle):
    vol = sphere(48, 17.3)
    check_exact(hip, oracle, vol, 0.0)
    check_exact(hip, oracle, vol[::-1].copy(), 0.0, flip0=True)
    a = hip.ops.marching_cubes(dev(vol), 0.0)
    b = hip.ops.marching_cubes(dev(vol[::-1].copy()), 0.0, flip0=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_edge_sizes_and_empty(hip, oracle):
    check_exact(hip, oracle, np.array([[[0, 1], [1, 1]], [[1, 1], [1, 0]]], np.float32), 0.5)  # n = 2: one cube
    for fill in (-1.0, 1.0):
        v, f, nr, va = hip.ops.marching_cubes(torch.full((9, 9, 9), fill, device="cuda"), 0.0)
        assert v.shape == (0, 3) and f.shape == (0, 3) and f.dtype == torch.int32
    # a value exactly equal to the level is 'outside' (x > level); the vertex then sits ON the grid point (t = 0)
    vol = np.full((5, 5, 5), -1.0, np.float32); vol[2, 2, 2] = 1.0; vol[2, 2, 3] = 0.0
    v, f, _, _ = check_exact(hip, oracle, vol, 0.0)
    assert (v == np.array([2, 2, 3], np.float32)).all(1).any()


def test_errors(hip):
    L = hip._lib.lib()
    with pytest.raises(RuntimeError):
        hip.ops.marching_cubes(torch.zeros((4, 4, 5), device="cuda"), 0.0)
    with pytest.raises(RuntimeError):
        hip.ops.marching_cubes(torch.zeros((1, 1, 1), device="cuda"), 0.0)
    assert L.p3d_mc_workspace_bytes(1) == 0 and L.p3d_mc_workspace_bytes(1025) == 0
    vol = torch.zeros((8, 8, 8), device="cuda")
    ws = torch.empty(64, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    assert L.p3d_mc_count_f32(vol.data_ptr(), 8, 0, 0.0, ws.data_ptr(), 64, cnt.data_ptr(), None) == -3  # workspace
    assert L.p3d_mc_count_f32(None, 8, 0, 0.0, ws.data_ptr(), 64, cnt.data_ptr(), None) == -1


def blobs(n, seed=0):
    """Smooth random field with the statistics of a density grid (a few connected blobs, values in [0,1]), border outside."""
    rng = np.random.default_rng(seed)
    g = torch.linspace(-1, 1, n, device="cuda")
    a, b, c = torch.meshgrid(g, g, g, indexing="ij")
    f = torch.zeros((n, n, n), device="cuda")
    for _ in range(12):
        ctr = rng.uniform(-0.5, 0.5, 3)
        r = rng.uniform(0.15, 0.35)
        f += torch.exp(-((a - ctr[0]) ** 2 + (b - ctr[1]) ** 2 + (c - ctr[2]) ** 2) / (r * r))
    f = 1 - torch.exp(-f)
    f[0], f[-1], f[:, 0], f[:, -1], f[:, :, 0], f[:, :, -1] = 0, 0, 0, 0, 0, 0
    return f.contiguous()


def test_pipeline_size_256_exact_and_closed(hip, oracle):
    """generate.py's grid size (256^3): bit-exact against the specification and a closed, consistently oriented 2-manifold."""
    vol = blobs(256)
    v, f, nr, va = hip.ops.marching_cubes(vol, 0.5)
    ref = oracle.marching_cubes(vol.cpu().numpy(), 0.5)
    for a, b in zip((v, f, nr, va), ref):
        assert np.array_equal(a.cpu().numpy(), b)
    uniq, cnt, has_rev = edge_stats(f.cpu().numpy())
    assert len(f) > 100000 and (cnt == 1).all() and has_rev.all()
    assert len(v) - len(uniq) // 2 + len(f) == 2 * 1 or (len(v) - len(uniq) // 2 + len(f)) % 2 == 0  # Euler characteristic is even


def test_full_size_512_properties(hip):
    """BASELINE c5's grid (512^3 = 134 M points): size-independent properties only (the CPU specification would take a while)."""
    vol = blobs(512, seed=3)
    v, f, nr, va = hip.ops.marching_cubes(vol, 0.5)
    F = f.long()
    assert F.min() == 0 and F.max() == len(v) - 1 and len(torch.unique(F)) == len(v)
    e = torch.cat([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]])
    key, rkey = e[:, 0] * (1 << 32) + e[:, 1], e[:, 1] * (1 << 32) + e[:, 0]
    uk, cnt = torch.unique(key, return_counts=True)
    assert (cnt == 1).all() and torch.equal(uk, torch.unique(rkey))  # every directed edge once, its reverse present
    frac = v - v.floor()
    assert ((frac > 0).sum(1) <= 1).all() and (va >= 0.5).all()
    assert ((nr.norm(dim=1) - 1).abs() < 1e-5).all()
    tri = v[F]
    fn = torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=1)
    assert ((fn * nr[F[:, 0]]).sum(1) > 0).float().mean() > 0.999  # faces wind with the outward vertex normals


def test_mesh_flow_colors(hip):
    """volume.mesh (density grid -> device extraction -> colours decoded at the vertices' voxels only) == the reference's
    flow on the same numbers: dense rgb grid from sample_mixed, flipped like eg3d_metrics3d.py:172-174, indexed at
    verts.astype(int) (eg3d_metrics3d.py:196-199), vertices scaled by /n*bw - bw/2."""
    from test_hip_synthesis import TRI_KW, load_sd
    from panic3d_amd.generator import TriPlaneGenerator
    g = T.load_golden("syn_triplane_f.npz")
    G = load_sd(TriPlaneGenerator(**TRI_KW), g, "sd_")
    G.set_force_sigmoid(True)
    ws = dev(g["ws"])[:1]
    N = 32
    with torch.no_grad():
        dg = hip.volume.density_grid(G, ws, {}, resolution=N)["densities"]
        level = float(dg.median())  # random-init densities: put the surface through the middle of their range
        mc = hip.volume.mesh(G, ws, {}, resolution=N, level=level)
        pts = hip.volume.create_samples(N, cube_length=0.7)[0].cuda()
        out = G.sample_mixed(pts.contiguous(), None, ws, {}, noise_mode="const")
        dens = hip.volume.to_volume(hip.ops.sigma2density(out["sigma"]), N)[0, 0].contiguous()
        rgbs = hip.volume.to_volume(out["rgb"], N)[0, :3].contiguous()
        ref = hip.volume.marching_cubes(dens, rgbs, 0.7, level=level)
    assert len(mc["faces"]) > 100 and mc["colors"].shape == (len(mc["verts"]), 3)
    for k in ("verts", "faces", "normals", "values", "colors"):
        assert np.array_equal(mc[k], ref[k]), k
    # (the level IS a grid value here: the reference's allow_degenerate=False, the default of volume.marching_cubes / mesh, applies)
    idx = hip.ops.marching_cubes(dens, level, allow_degenerate=False)[0].cpu().numpy()
    assert np.array_equal(mc["verts"], (idx / N * 0.7 - 0.5 * 0.7).astype(np.float32))
    assert hip.ops.marching_cubes(dens, level)[1].shape[0] >= len(mc["faces"])


def test_degenerate_triangles_are_dropped_like_skimage_allow_degenerate_false(hip):
    """The reference extracts its meshes with allow_degenerate=False (_util/eg3d_metrics3d.py:189-194).  A volume with grid values
    EXACTLY on the level (integer distances, integer level) makes zero-area triangles; volume.marching_cubes removes them, merges
    the coincident vertices and drops unused ones; the raw extractor (ops.marching_cubes) keeps them."""
    n = 24
    g = torch.arange(n, dtype=torch.float32) - 11.0
    vol = (g[:, None, None] ** 2 + g[None, :, None] ** 2 + g[None, None, :] ** 2).cuda().contiguous()  # integers; 25 occurs (3,4,0) ...
    v0, f0, n0, s0 = hip.ops.marching_cubes(vol, 25.0)
    tri = v0[f0.long()]
    deg0 = ((tri[:, 0] == tri[:, 1]).all(1) | (tri[:, 0] == tri[:, 2]).all(1) | (tri[:, 1] == tri[:, 2]).all(1))
    assert int(deg0.sum()) > 0  # the fixture really has degenerate faces
    v1, f1, n1, s1 = hip.ops.marching_cubes(vol, 25.0, allow_degenerate=False)
    tri1 = v1[f1.long()]
    deg1 = ((tri1[:, 0] == tri1[:, 1]).all(1) | (tri1[:, 0] == tri1[:, 2]).all(1) | (tri1[:, 1] == tri1[:, 2]).all(1))
    assert int(deg1.sum()) == 0 and f1.shape[0] == f0.shape[0] - int(deg0.sum())
    assert f1.dtype == torch.int32 and int(f1.min()) >= 0 and int(f1.max()) < v1.shape[0]
    assert torch.unique(f1).numel() == v1.shape[0] and v1.shape[0] < v0.shape[0]  # every vertex is used; coincident ones merged
    assert n1.shape == v1.shape and s1.shape[0] == v1.shape[0]
    # the surface itself is unchanged: same set of non-degenerate triangles (as vertex coordinates)
    key = lambda t: set(map(tuple, t.reshape(-1, 9).cpu().numpy().round(5).tolist()))
    assert key(tri[~deg0]) == key(tri1)
    out = hip.volume.marching_cubes(vol, None, 0.7, level=25.0)  # the mirror of eg3d_metrics3d.marching_cubes: drops them by default
    assert out["faces"].shape[0] == f1.shape[0] and out["verts"].shape[0] == v1.shape[0]


def test_hip_mesh_scores_like_the_reference_evaluation(hip):
    """parity with skimage's Lewiner triangulation stays unpinned (skimage is not installable here), so the
    HIP extractor is held to what the reference's evaluation actually computes from a mesh — surface-sample chamfer distance and F1
    at the thresholds of _scripts/eval/measure.py:187-201 — against an analytic sphere, through volume.marching_cubes (the
    reference's vertex scaling, eg3d_metrics3d.py:201-202)."""
    from test_mcubes_cpu import world_mesh_of_sphere, mesh_scores_against_sphere, F1_THRESHOLDS
    n, r, bw = 96, 0.24, 0.7

    def extract(vol):
        m = hip.volume.marching_cubes(dev(vol), None, bw, level=0.0)
        return (m["verts"] + bw / 2) / bw * n, m["faces"]  # back to index space: world_mesh_of_sphere applies the scaling itself
    verts, faces, centre = world_mesh_of_sphere(n, r, extract, bw)
    s = mesh_scores_against_sphere(verts.astype(np.float64), faces.astype(np.int64), centre.astype(np.float64), r, n_sample=1500)
    assert s["cd"] < 0.05 * (bw / n), s
    assert all(s[f"f1_{int(th * 1000):03d}"] == 1.0 for th in F1_THRESHOLDS), s
    uniq, cnt, has_rev = edge_stats(faces.astype(np.int64))
    assert (cnt == 1).all() and has_rev.all()  # closed, consistently oriented
