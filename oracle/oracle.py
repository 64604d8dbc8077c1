"""ctypes/numpy front-end of oracle/libp3d_oracle.so (see p3d_oracle.c).  TEST INFRASTRUCTURE ONLY.

All arrays are numpy float32/int32, C-contiguous, in the REFERENCE's layouts
(planes [N,3,32,H,W] as produced by training/triplane.py:200-206).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLAG_CROP, FLAG_CULL, FLAG_BINARIZE, FLAG_FORCE_SIGMOID, FLAG_WHITE_BACK = 1, 2, 4, 8, 16
FLAG_DISPARITY = 4096


def build(force=False):
    so = os.path.join(_HERE, "libp3d_oracle.so")
    srcs = [os.path.join(_HERE, "p3d_oracle.c"), os.path.join(_HERE, "p3d_oracle_mc.c"),
            os.path.join(_HERE, "..", "include", "p3d_numerics.h"), os.path.join(_HERE, "..", "include", "p3d_mc_table.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libp3d_oracle.so")
        build()
        _LIB = C.CDLL(so)
    return _LIB


class Opts(C.Structure):
    _fields_ = [("coord_scale", C.c_float), ("ray_start", C.c_float), ("ray_end", C.c_float),
                ("depth_delta", C.c_float), ("crop_limit", C.c_float), ("cull_thresh", C.c_float),
                ("Sc", C.c_int32), ("Sf", C.c_int32), ("plane_mode", C.c_int32), ("flags", C.c_int32)]


class Dumps(C.Structure):
    _fields_ = [("depths_coarse", C.c_void_p), ("sigma_coarse", C.c_void_p), ("rgb_coarse", C.c_void_p),
                ("weights_coarse", C.c_void_p), ("depths_fine", C.c_void_p), ("inds", C.c_void_p),
                ("sigma_fine", C.c_void_p), ("perm", C.c_void_p), ("depth_unclamped", C.c_void_p),
                ("tminmax", C.c_void_p)]


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.c_void_p)


def make_opts(rendering_options, triplane_crop=None, cull_clouds=None, binarize_clouds=None, force_sigmoid=True):
    """Translate the reference's rendering_kwargs + forward() arguments (renderer.py:162) into Opts.
    The double->float conversions happen here exactly as the contract states them."""
    ro = rendering_options
    bw = float(ro["box_warp"])
    Sc = int(ro["depth_resolution"])
    Sf = int(ro.get("depth_resolution_importance", 0))
    flags = 0
    crop_limit = 0.0
    thr = 0.0
    if triplane_crop:
        flags |= FLAG_CROP
        crop_limit = bw / 2 - float(triplane_crop)
    if binarize_clouds:
        flags |= FLAG_BINARIZE
        thr = float(binarize_clouds)
    elif cull_clouds:
        flags |= FLAG_CULL
        thr = float(cull_clouds)
    if force_sigmoid:
        flags |= FLAG_FORCE_SIGMOID
    if ro.get("white_back", False):
        flags |= FLAG_WHITE_BACK
    if ro.get("ray_start") == "auto" and ro.get("ray_end") == "auto":  # per-ray limits go to render(ray_limits=...)
        rs = re = 0.0
    else:
        rs, re = float(ro["ray_start"]), float(ro["ray_end"])
    if ro.get("disparity_space_sampling", False):  # renderer.py:309-316: the reciprocals, and depth_delta = 1 / (Sc - 1)
        return Opts(np.float32(2.0 / bw), np.float32(1.0 / rs), np.float32(1.0 / re), np.float32(1 / (Sc - 1)),
                    np.float32(crop_limit), np.float32(thr), Sc, Sf, int(bool(ro.get("use_triplane", False))), flags | FLAG_DISPARITY)
    return Opts(np.float32(2.0 / bw), np.float32(rs), np.float32(re), np.float32((re - rs) / (Sc - 1)),
                np.float32(crop_limit), np.float32(thr), Sc, Sf, int(bool(ro.get("use_triplane", False))), flags)


def prescale_mlp(w0, b0, w1, b1, lr_mul=1.0):
    """FullyConnectedLayer.forward's w*weight_gain, b*bias_gain (networks_stylegan2.py:121-127) in float32."""
    w0 = np.asarray(w0, np.float32)
    w1 = np.asarray(w1, np.float32)
    g0 = np.float32(lr_mul / np.sqrt(w0.shape[1]))
    g1 = np.float32(lr_mul / np.sqrt(w1.shape[1]))
    b0 = np.asarray(b0, np.float32)
    b1 = np.asarray(b1, np.float32)
    if lr_mul != 1:
        b0 = b0 * np.float32(lr_mul)
        b1 = b1 * np.float32(lr_mul)
    return (w0 * g0).astype(np.float32), b0, (w1 * g1).astype(np.float32), b1


def decode(planes, coords, mlp, box_warp, plane_mode=1, flags=FLAG_FORCE_SIGMOID, crop_limit=0.0, cull_thresh=0.0,
           density_only=False):
    planes, pp = _f(planes)
    coords, pc = _f(coords)
    N, _, Cc, H, W = planes.shape
    assert Cc == 32 and coords.shape[0] == N and coords.shape[2] == 3
    M = coords.shape[1]
    (w0, p0), (b0, q0), (w1, p1), (b1, q1) = map(_f, mlp)
    sigma = np.empty((N, M, 1), np.float32)
    rgb = None if density_only else np.empty((N, M, 32), np.float32)
    lib().p3d_oracle_decode(pp, N, H, W, pc, C.c_long(M), p0, q0, p1, q1, C.c_float(np.float32(2.0 / box_warp)),
                            int(plane_mode), int(flags), C.c_float(np.float32(crop_limit)),
                            C.c_float(np.float32(cull_thresh)), sigma.ctypes.data_as(C.c_void_p),
                            None if rgb is None else rgb.ctypes.data_as(C.c_void_p))
    return sigma, rgb


def decode_features(feats, mlp, force_sigmoid=True):
    """OSGDecoder.forward (triplane.py:528-544) on sampled features [N,3,M,32] -> sigma [N,M,1], rgb [N,M,32]."""
    feats, pf = _f(feats)
    N, three, M, Cc = feats.shape
    assert three == 3 and Cc == 32
    (w0, p0), (b0, q0), (w1, p1), (b1, q1) = map(_f, mlp)
    sigma = np.empty((N, M, 1), np.float32)
    rgb = np.empty((N, M, 32), np.float32)
    lib().p3d_oracle_decode_features(pf, N, C.c_long(M), p0, q0, p1, q1, FLAG_FORCE_SIGMOID if force_sigmoid else 0,
                                     sigma.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p))
    return sigma, rgb


def sample_stratified(start, end, S, jitter):
    jitter, pj = _f(jitter)
    NR = jitter.size // S
    out = np.empty_like(jitter)
    lib().p3d_oracle_sample_stratified(C.c_float(np.float32(start)), C.c_float(np.float32(end)),
                                       C.c_float(np.float32((float(end) - float(start)) / (S - 1))), int(S), pj,
                                       C.c_long(NR), out.ctypes.data_as(C.c_void_p))
    return out


def composite(colors, sigma, depths, white_back=True):
    colors, pc = _f(colors)
    sigma, ps = _f(sigma)
    depths, pd = _f(depths)
    S, K = colors.shape[-2], colors.shape[-1]
    NR = colors.size // (S * K)
    rgb = np.empty((NR, K), np.float32)
    depth = np.empty((NR, 1), np.float32)
    w = np.empty((NR, S - 1, 1), np.float32)
    lib().p3d_oracle_composite(pc, ps, pd, C.c_long(NR), int(S), int(K), int(bool(white_back)),
                               rgb.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p),
                               w.ctypes.data_as(C.c_void_p))
    return rgb, depth, w


def importance(depths, weights, u):
    depths, pd = _f(depths)
    weights, pw = _f(weights)
    u, pu = _f(u)
    Sf = u.shape[-1]
    NR = u.size // Sf
    Sc = depths.size // NR
    assert weights.size == NR * (Sc - 1)
    out = np.empty((NR, Sf), np.float32)
    inds = np.empty((NR, Sf), np.int32)
    lib().p3d_oracle_importance(pd, pw, C.c_long(NR), int(Sc), int(Sf), pu, out.ctypes.data_as(C.c_void_p),
                                inds.ctypes.data_as(C.c_void_p))
    return out, inds


def unify_perm(depths_coarse, depths_fine):
    dc, pc = _f(depths_coarse)
    df, pf = _f(depths_fine)
    NR = dc.shape[0]
    Sc, Sf = dc.size // NR, df.size // NR
    perm = np.empty((NR, Sc + Sf), np.int32)
    lib().p3d_oracle_unify_perm(pc, pf, C.c_long(NR), Sc, Sf, perm.ctypes.data_as(C.c_void_p))
    return perm


def render(planes, rays_o, rays_d, jitter, u, mlp, opts, dumps=False, ray_limits=None):
    """ImportanceRenderer.forward (renderer.py:162-264) with injected randomness.
    jitter [N,R,Sc] = the torch.rand_like draw of renderer.py:324; u [N*R,Sf] = the torch.rand draw of :371.
    ray_limits: None, or (ray_start, ray_end) per ray [N,R(,1)] = the 'auto' limits of renderer.py:165-171 (after the patching
    of the rays that miss the box)."""
    planes, pp = _f(planes)
    rays_o, po = _f(rays_o)
    rays_d, pd = _f(rays_d)
    jitter, pj = _f(jitter)
    N, _, Cc, H, W = planes.shape
    R = rays_o.shape[1]
    Sc, Sf = opts.Sc, opts.Sf
    assert jitter.size == N * R * Sc
    if Sf > 0:
        u, pu = _f(u)
        assert u.size == N * R * Sf
    else:
        pu = None
    (w0, p0), (b0, q0), (w1, p1), (b1, q1) = map(_f, mlp)
    feat = np.empty((N, R, 32), np.float32)
    depth = np.empty((N, R, 1), np.float32)
    wsum = np.empty((N, R, 1), np.float32)
    xyz = np.empty((N, R, 3), np.float32)
    d = None
    dm = None
    if dumps:
        NR = N * R
        d = dict(depths_coarse=np.empty((NR, Sc), np.float32), sigma_coarse=np.empty((NR, Sc), np.float32),
                 rgb_coarse=np.empty((NR, Sc, 32), np.float32), weights_coarse=np.empty((NR, Sc - 1), np.float32),
                 depths_fine=np.empty((NR, Sf), np.float32), inds=np.empty((NR, Sf), np.int32),
                 sigma_fine=np.empty((NR, Sf), np.float32), perm=np.empty((NR, Sc + Sf), np.int32),
                 depth_unclamped=np.empty((NR,), np.float32), tminmax=np.empty((2,), np.float32))
        dm = Dumps(*[d[k].ctypes.data_as(C.c_void_p) for k, _ in Dumps._fields_])
    prs = pre = None
    if ray_limits is not None:
        (rs_, prs), (re_, pre) = _f(ray_limits[0]), _f(ray_limits[1])
        assert rs_.size == N * R and re_.size == N * R
    rc = lib().p3d_oracle_render_limits(pp, N, H, W, po, pd, C.c_long(R), pj, pu, p0, q0, p1, q1, prs, pre, C.byref(opts),
                                        feat.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p),
                                        wsum.ctypes.data_as(C.c_void_p), xyz.ctypes.data_as(C.c_void_p),
                                        C.byref(dm) if dm is not None else None)
    if rc != 0:
        raise RuntimeError(f"p3d_oracle_render failed: {rc}")
    return (feat, depth, wsum, xyz, d) if dumps else (feat, depth, wsum, xyz)


def math_fn(which, x):
    x, px = _f(x)
    y = np.empty_like(x)
    lib().p3d_oracle_math(px, C.c_long(x.size), {"exp": 0, "log1p01": 1, "softplus": 2, "sigmoid": 3}[which],
                          y.ctypes.data_as(C.c_void_p))
    return y


def marching_cubes(vol, level=0.5, flip0=False):
    """Iso-surface of vol [n,n,n] (p3d_oracle_mc.c): verts [V,3] in index space, faces [F,3] int32, normals [V,3], values [V]."""
    vol, pv = _f(vol)
    n = vol.shape[0]
    assert vol.shape == (n, n, n)
    cnt = np.zeros(2, dtype=np.int64)
    lib().or_mc_count(pv, n, int(bool(flip0)), C.c_float(level), cnt.ctypes.data_as(C.c_void_p))
    V, F = int(cnt[0]), int(cnt[1])
    verts, normals, values = np.empty((V, 3), np.float32), np.empty((V, 3), np.float32), np.empty((V,), np.float32)
    faces = np.empty((F, 3), np.int32)
    rc = lib().or_mc_emit(pv, n, int(bool(flip0)), C.c_float(level), verts.ctypes.data_as(C.c_void_p),
                          normals.ctypes.data_as(C.c_void_p), values.ctypes.data_as(C.c_void_p), faces.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(f"or_mc_emit failed: {rc}")
    return verts, faces, normals, values


def sigma2density(sigma, cropmask=None, cull=None):
    """get_eg3d_volume's activation + masks on a flat sigma array (p3d_oracle_sigma2density)."""
    sigma, ps = _f(sigma)
    out = np.empty_like(sigma)
    cm = None if cropmask is None else np.ascontiguousarray(cropmask, dtype=np.uint8)
    lib().p3d_oracle_sigma2density(ps, None if cm is None else cm.ctypes.data_as(C.c_void_p), C.c_long(sigma.size),
                                   C.c_float(-1.0 if cull is None else cull), out.ctypes.data_as(C.c_void_p))
    return out


def device_draws(seed, N, R, Sc, Sf):
    """The two draw tensors p3d_render_rng_f32 makes inside the kernel (include/p3d_numerics.h "device draws"), restated with
    numpy integer arithmetic: (jitter [N,R,Sc,1], u [N*R,Sf]) for oracle.render — the checker of the opt-in in-kernel generator."""
    M32 = np.uint64(0xFFFFFFFF)

    def fmix32(h):
        h = h & M32
        h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & M32
        h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & M32
        h ^= h >> np.uint64(16)
        return h

    seed = int(seed) & (2 ** 64 - 1)
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    ray = np.arange(N * R, dtype=np.uint64)

    def draws(stream, S):
        i = np.arange(S, dtype=np.uint64)
        h = fmix32(lo ^ ((ray & M32) * np.uint64(0x9E3779B1) & M32))[:, None]
        rhi = (((ray >> np.uint64(32)) * np.uint64(0x7FEB352D)) & M32)[:, None]
        k = (((i * np.uint64(2) + np.uint64(stream)) & M32) * np.uint64(0x846CA68B)) & M32
        h = fmix32(h ^ hi ^ rhi ^ k[None, :])
        return ((h >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)

    jit = draws(0, Sc).reshape(N, R, Sc, 1)
    u = draws(1, Sf) if Sf > 0 else np.zeros((N * R, 0), np.float32)
    return np.ascontiguousarray(jit), np.ascontiguousarray(u)
