"""Tensor-level operators over the C ABI (include/panic3d_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every computation is a call into
libpanic3d_hip.so.  All tensors must be float32 CUDA(ROCm) tensors; misuse raises like the reference's plugins do
(TORCH_CHECK -> RuntimeError, torch_utils/ops/bias_act.cpp:39-55).
"""
import ctypes as C
import os
import weakref

import numpy as np
import torch

from . import _lib, memo
from ._lib import Opts, Dumps

__all__ = ["make_opts", "prescale_mlp", "planes_to_nhwc", "triplane_decode", "render", "sample_stratified", "composite",
           "importance", "unify_perm"]


# torch.cuda.current_stream() builds a Stream object through three layers of Python (~7 us) and torch.cuda.current_device() goes
# through _lazy_init (~1 us); a G.f call asks ~45 times for each while the GPU waits for the launches of its 4^2 .. 32^2 layers
# (tools/host_profile.py: 0.3 ms of a 2.5 ms call).  The two C entry points below are what those wrappers end in.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda index: torch.cuda.current_stream(index).cuda_stream)
_cur_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _stream():
    return C.c_void_p(_raw_stream(_cur_device()))


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOGUARD = _NoGuard()


def _on(device):
    """The C ABI launches on the CURRENT device's stream, so a call has to run with its tensors' device current.
    `torch.cuda.device(...)` costs ~10 us of host time per entry (get + set + restore); when the device already is the current one
    — always, with one process per GPU — nothing has to be switched: a G.f call makes ~35 such calls and starts with an empty
    queue, so this is latency the GPU waits for (profiles/r04_notes.txt)."""
    idx = device.index
    if idx is None or idx == _cur_device():
        return _NOGUARD
    return torch.cuda.device(device)


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/ROCm tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def make_opts(rendering_options, triplane_crop=None, cull_clouds=None, binarize_clouds=None, force_sigmoid=False,
              early_out=True, small_launch_kernel=True, fast_color=False):
    """rendering_kwargs + ImportanceRenderer.forward arguments (renderer.py:162) -> p3d_opts.
    The double -> binary32 conversions are the ones include/p3d_numerics.h states."""
    ro = rendering_options
    auto = ro.get("ray_start") == "auto" and ro.get("ray_end") == "auto"  # per-ray limits: render(..., ray_limits=...)
    if not auto and (ro.get("ray_start") == "auto" or ro.get("ray_end") == "auto"):
        raise ValueError("ray_start and ray_end must both be 'auto' or both be numbers (renderer.py:165)")
    disparity = bool(ro.get("disparity_space_sampling", False))  # renderer.py:309-316
    if disparity and auto:
        raise NotImplementedError("disparity_space_sampling together with ray_start = ray_end = 'auto'")
    if ro.get("clamp_mode", "softplus") != "softplus":
        raise AssertionError("MipRayMarcher only supports `clamp_mode`=`softplus`!")  # ray_marcher.py:35
    if (ro.get("density_noise", 0) or 0) > 0:  # (ImportanceRenderer.forward routes such calls to forward_staged)
        raise NotImplementedError("density_noise > 0 (renderer.py:276-277) is not an option of the fused kernel: ImportanceRenderer.forward_staged")
    if ro.get("triplane_depth", 1) != 1:
        raise NotImplementedError("triplane_depth != 1")
    bw = float(ro["box_warp"])
    Sc = int(ro["depth_resolution"])
    Sf = int(ro.get("depth_resolution_importance", 0) or 0)
    flags, crop_limit, thr = 0, 0.0, 0.0
    if triplane_crop:
        flags |= _lib.P3D_FLAG_CROP
        crop_limit = bw / 2 - float(triplane_crop)
    if binarize_clouds:
        flags |= _lib.P3D_FLAG_BINARIZE
        thr = float(binarize_clouds)
    elif cull_clouds:
        flags |= _lib.P3D_FLAG_CULL
        thr = float(cull_clouds)
    if force_sigmoid:
        flags |= _lib.P3D_FLAG_FORCE_SIGMOID
    if ro.get("white_back", False):
        flags |= _lib.P3D_FLAG_WHITE_BACK
    if not early_out:  # decode every sample even where the result provably cannot matter (measurement / tests)
        flags |= _lib.P3D_FLAG_NO_EARLY_OUT
    if not small_launch_kernel:  # keep small launches on the 32-rays-per-wave kernel (tests)
        flags |= _lib.P3D_FLAG_NO_PAIR
    elif small_launch_kernel == "pair":  # ... or force the 16 rays x 2 samples kernel / the 8 rays x 4 samples one (tests, A/B);
        flags |= _lib.P3D_FLAG_PAIR16    # True: the library's size heuristic picks (p3d_render_f32)
    elif small_launch_kernel == "quad":
        flags |= _lib.P3D_FLAG_QUAD8
    if fast_color:  # opt-in tolerance mode of the final pass (include/panic3d_hip.h P3D_FLAG_FAST_COLOR)
        flags |= _lib.P3D_FLAG_FAST_COLOR
    rs, re = (0.0, 0.0) if auto else (float(ro["ray_start"]), float(ro["ray_end"]))
    if disparity:  # the kernel receives the reciprocals the reference forms in binary64 and multiplies as binary32 scalars
        flags |= _lib.P3D_FLAG_DISPARITY
        return Opts(np.float32(2.0 / bw), np.float32(1.0 / rs), np.float32(1.0 / re), np.float32(1 / max(Sc - 1, 1)),
                    np.float32(crop_limit), np.float32(thr), Sc, Sf, int(bool(ro.get("use_triplane", False))), flags)
    return Opts(np.float32(2.0 / bw), np.float32(rs), np.float32(re), np.float32((re - rs) / max(Sc - 1, 1)),
                np.float32(crop_limit), np.float32(thr), Sc, Sf, int(bool(ro.get("use_triplane", False))), flags)


def _with_flag(opts, flag):
    o = Opts.from_buffer_copy(opts)
    o.flags |= flag
    return o


def prescale_mlp(w0, b0, w1, b1, weight_gain0, bias_gain0, weight_gain1, bias_gain1):
    """FullyConnectedLayer.forward's `w * weight_gain`, `b * bias_gain` (networks_stylegan2.py:121-127), in float32."""
    w0s = (w0.detach().float() * float(weight_gain0)).contiguous()
    w1s = (w1.detach().float() * float(weight_gain1)).contiguous()
    b0s = b0.detach().float()
    b1s = b1.detach().float()
    if bias_gain0 != 1:
        b0s = b0s * float(bias_gain0)
    if bias_gain1 != 1:
        b1s = b1s * float(bias_gain1)
    return w0s, b0s.contiguous(), w1s, b1s.contiguous()


def planes_to_nhwc(planes):
    """[N,3,32,H,W] (training/triplane.py:200-206) -> channels-last [N,3,H,W,32]."""
    planes = _chk(planes, "planes")
    if planes.dim() != 5 or planes.shape[1] != 3 or planes.shape[2] != 32:
        raise RuntimeError(f"planes must be [N,3,32,H,W], got {tuple(planes.shape)}")
    N, _, Cc, H, W = planes.shape
    out = torch.empty((N, 3, H, W, Cc), dtype=torch.float32, device=planes.device)
    with _on(planes.device):
        _lib.check(_lib.lib().p3d_planes_to_nhwc_f32(_p(planes), N * 3, Cc, H, W, _p(out), _stream()), "p3d_planes_to_nhwc_f32")
    return out


def _chk_mlp(mlp):
    w0, b0, w1, b1 = (_chk(t, n) for t, n in zip(mlp, ("w0", "b0", "w1", "b1")))
    if tuple(w0.shape) != (64, 32) or tuple(b0.shape) != (64,) or tuple(w1.shape) != (33, 64) or tuple(b1.shape) != (33,):
        raise RuntimeError("decoder must be OSGDecoder-shaped: 32 -> 64 -> 33 (training/triplane.py:521-526)")
    return w0, b0, w1, b1


def triplane_decode(planes_nhwc, coords, mlp, opts, density_only=False):
    """run_model (renderer.py:266-280) on coords [N,M,3] -> sigma [N,M,1], rgb [N,M,32] (None if density_only).
    Masks are applied iff opts.flags carries CROP/CULL/BINARIZE."""
    planes_nhwc = _chk(planes_nhwc, "planes_nhwc")
    coords = _chk(coords, "coords")
    N, three, H, W, Cc = planes_nhwc.shape
    if N == 1 and coords.dim() == 3 and coords.shape[0] > 1:  # one subject, several point batches: planes are shared
        N, opts = coords.shape[0], _with_flag(opts, _lib.P3D_FLAG_SHARED_PLANES)
    if three != 3 or Cc != 32 or coords.dim() != 3 or coords.shape[0] != N or coords.shape[2] != 3:
        raise RuntimeError("planes_nhwc must be [N,3,H,W,32] (or [1,...] shared) and coords [N,M,3]")
    w0, b0, w1, b1 = _chk_mlp(mlp)
    M = coords.shape[1]
    sigma = torch.empty((N, M, 1), dtype=torch.float32, device=coords.device)
    rgb = None if density_only else torch.empty((N, M, 32), dtype=torch.float32, device=coords.device)
    with _on(coords.device):
        rc = _lib.lib().p3d_triplane_decode_f32(_p(planes_nhwc), N, H, W, _p(coords), M, _p(w0), _p(b0), _p(w1), _p(b1),
                                                C.byref(opts), _p(sigma), _p(rgb), _stream())
    _lib.check(rc, "p3d_triplane_decode_f32")
    return sigma, rgb


def decode_features(feats, mlp, force_sigmoid=True):
    """OSGDecoder.forward (training/triplane.py:528-544) on sampled features [N,3,M,32] -> sigma [N,M,1], rgb [N,M,32]."""
    feats = _chk(feats, "sampled_features")
    if feats.dim() != 4 or feats.shape[1] != 3 or feats.shape[3] != 32:
        raise RuntimeError("sampled_features must be [N,3,M,32]")
    N, _, M, _ = feats.shape
    w0, b0, w1, b1 = _chk_mlp(mlp)
    sigma = torch.empty((N, M, 1), dtype=torch.float32, device=feats.device)
    rgb = torch.empty((N, M, 32), dtype=torch.float32, device=feats.device)
    with _on(feats.device):
        rc = _lib.lib().p3d_decode_features_f32(_p(feats), N, M, _p(w0), _p(b0), _p(w1), _p(b1), int(bool(force_sigmoid)), _p(sigma),
                                                _p(rgb), _stream())
    _lib.check(rc, "p3d_decode_features_f32")
    return sigma, rgb


def grid_density(planes_nhwc, grid_n, lo, hi, voxel_size, offsets, mlp, opts, crop_limit=None, skip_cropped=False, staged=None, fast=False):
    """Density-only decode of flat indices [lo, hi) of the reference's grid_n^3 sample grid (create_samples), the points
    generated inside the kernel.  planes_nhwc [1,3,H,W,32] -> sigma [1, hi-lo, 1]; with crop_limit also the bool mask
    [1, hi-lo, 1] of triplane_crop_mask (|x| or |z| beyond the limit) evaluated on the same generated points.
    skip_cropped: do not decode masked points (sigma = -1000 there): exact for the densities, which are overwritten anyway."""
    if skip_cropped and crop_limit is not None:
        opts = _with_flag(opts, _lib.P3D_FLAG_SKIP_CROPPED)
    if fast:  # tolerance-mode decoder (f16 two-term MFMA + hardware transcendentals): sigma to ~1e-6 of the exact contract
        opts = _with_flag(opts, _lib.P3D_FLAG_FAST_COLOR)
    if staged is not None:  # default: direct (quad-cooperative / per-lane) gathers; staged=True: texel boxes through LDS (same bits)
        opts = _with_flag(opts, (_lib.P3D_FLAG_FORCE_STAGING if staged else _lib.P3D_FLAG_NO_STAGING))
    planes_nhwc = _chk(planes_nhwc, "planes_nhwc")
    if planes_nhwc.shape[0] != 1:
        raise RuntimeError("grid_density renders one subject at a time")
    _, _, H, W, _ = planes_nhwc.shape
    w0, b0, w1, b1 = _chk_mlp(mlp)
    sigma = torch.empty((1, hi - lo, 1), dtype=torch.float32, device=planes_nhwc.device)
    msk = torch.empty((1, hi - lo, 1), dtype=torch.uint8, device=planes_nhwc.device) if crop_limit is not None else None
    with _on(planes_nhwc.device):
        rc = _lib.lib().p3d_grid_density_f32(_p(planes_nhwc), H, W, int(grid_n), int(lo), int(hi), np.float32(voxel_size),
                                              np.float32(offsets[0]), np.float32(offsets[1]), np.float32(offsets[2]), _p(w0),
                                              _p(b0), _p(w1), _p(b1), C.byref(opts), _p(sigma), _p(msk),
                                              np.float32(crop_limit if crop_limit is not None else 0.0), _stream())
    _lib.check(rc, "p3d_grid_density_f32")
    return (sigma, msk.view(torch.bool)) if crop_limit is not None else sigma  # the kernel writes 0 / 1 bytes: no conversion pass


def sigma2density(sigma, cropmask=None, cull=None):
    """get_eg3d_volume's activation and masks in one pass (eg3d_metrics3d.py:65-69,153-163): 1 - exp(-softplus(sigma - 1)),
    -1000 where cropmask, then -1000 where the cull mask — evaluated on the densities, as the reference does — fires."""
    sigma = _chk(sigma, "sigma")
    out = torch.empty_like(sigma)
    cm = None
    if cropmask is not None:
        cm = cropmask.contiguous()
        cm = cm.view(torch.uint8) if cm.dtype == torch.bool else cm.to(torch.uint8)  # bool is one 0 / 1 byte: reinterpret
        if cm.numel() != sigma.numel() or not cm.is_cuda:
            raise RuntimeError("cropmask must be a CUDA tensor with one entry per sigma")
    with _on(sigma.device):
        rc = _lib.lib().p3d_sigma2density_f32(_p(sigma), _p(cm), sigma.numel(), np.float32(-1.0 if cull is None else cull), _p(out),
                                              _stream())
    _lib.check(rc, "p3d_sigma2density_f32")
    return out


def drop_degenerate_faces(verts, faces, normals, values):
    """skimage's `allow_degenerate=False` (the reference: _util/eg3d_metrics3d.py:189-194): a grid value exactly on the level puts
    several edge crossings on the same grid point, i.e. zero-area triangles.  As skimage does: a face with two coincident
    vertices is dropped, the coincident vertices are merged (the higher index follows the lower), vertices no face uses any
    more are removed and the faces re-indexed.  Index compaction on the device; nothing is downloaded."""
    if faces.shape[0] == 0:
        return verts, faces, normals, values
    f = faces.long()
    tri = verts[f]  # [F,3,3]
    vmap = torch.arange(verts.shape[0], device=verts.device)
    deg = torch.zeros(f.shape[0], dtype=torch.bool, device=verts.device)
    for a, b in ((0, 1), (0, 2), (1, 2)):
        same = (tri[:, a] == tri[:, b]).all(dim=1)
        deg |= same
        hi, lo = torch.maximum(f[same, a], f[same, b]), torch.minimum(f[same, a], f[same, b])
        vmap.scatter_reduce_(0, hi, lo, reduce="amin")  # deterministic for repeated indices (a plain indexed store is not)
    for _ in range(4):  # follow chains a -> b -> c (coincident vertices of one grid point: at most a handful)
        vmap = vmap[vmap]
    f = vmap[f[~deg]]
    used = torch.zeros(verts.shape[0], dtype=torch.bool, device=verts.device)
    used[f.reshape(-1)] = True
    new_index = torch.cumsum(used, 0) - 1
    return verts[used], new_index[f].to(torch.int32), normals[used], values[used]


def marching_cubes(vol, level, flip0=False, allow_degenerate=True):
    """Iso-surface of vol [n,n,n] (device, f32) at `level`, on the device (csrc/p3d_mcubes.hip; specification: DESIGN.md §4.5,
    case table include/p3d_mc_table.h).  flip0: vol is the un-flipped flat grid of grid_density (axis 0 is read reversed).
    Returns verts [V,3] (index space, axis 0/1/2 order), faces [F,3] int32, normals [V,3], values [V] — device tensors.
    One 16-byte D2H read (the counts) sits between the two launches groups because the caller owns the output buffers.
    allow_degenerate=False: zero-area triangles are removed like skimage does (drop_degenerate_faces)."""
    vol = _chk(vol, "vol")
    n = vol.shape[0]
    if vol.dim() != 3 or vol.shape != (n, n, n):
        raise RuntimeError("vol must be [n,n,n]")
    L = _lib.lib()
    wsb = L.p3d_mc_workspace_bytes(n)
    if wsb == 0:
        raise RuntimeError("marching_cubes: n must be in [2, 1024]")
    dev = vol.device
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    with _on(dev):
        _lib.check(L.p3d_mc_count_f32(_p(vol), n, int(bool(flip0)), np.float32(level), _p(ws), wsb, _p(counts), _stream()),
                   "p3d_mc_count_f32")
        V, F = (int(x) for x in counts.tolist())
        verts = torch.empty((V, 3), dtype=torch.float32, device=dev)
        normals = torch.empty((V, 3), dtype=torch.float32, device=dev)
        values = torch.empty((V,), dtype=torch.float32, device=dev)
        faces = torch.empty((F, 3), dtype=torch.int32, device=dev)
        _lib.check(L.p3d_mc_emit_f32(_p(vol), n, int(bool(flip0)), np.float32(level), _p(ws), wsb, V, F, _p(verts), _p(normals),
                                     _p(values), _p(faces), _stream()), "p3d_mc_emit_f32")
    if not allow_degenerate:
        return drop_degenerate_faces(verts, faces, normals, values)
    return verts, faces, normals, values


DUMP_KEYS = ("depths_coarse", "sigma_coarse", "weights_coarse", "depths_fine", "inds", "depths_sorted", "sigma_sorted",
             "depth_unclamped", "tminmax")


def render(planes_nhwc, rays_o, rays_d, jitter, u, mlp, opts, ray_tile_w=0, dumps=False, stats=None, per_view_clamp=False,
           ray_limits=None, rng_seed=None, weights_only=False):
    """ImportanceRenderer.forward (renderer.py:162-264) with the two random draws passed in:
    jitter [N,R,Sc(,1)] (torch.rand_like, :324) and u [N*R,Sf] (torch.rand, :371).
    Returns (feat [N,R,32], depth [N,R,1], wsum [N,R,1], xyz [N,R,3]) (+ dict of per-stage dumps).
    `stats`: pass a dict to receive the number of decode steps the launch executed (exact early-outs, see k_render).
    per_view_clamp: clamp each image's depth to its own sample range (N batched views = N calls of the reference).
    ray_limits: (ray_start, ray_end) per ray [N,R(,1)] for rendering_options['ray_start'] == ['ray_end'] == 'auto'
    (renderer.py:165-171; cameras.ray_limits_box + cameras.patch_ray_limits compute them).
    rng_seed (int, with jitter = u = None): the two draws are made inside the kernel by the counter-based generator of
    include/p3d_numerics.h (p3d_render_rng_f32) — no draw tensors exist.
    weights_only: the caller wants wsum and depth only (P3D_FLAG_WEIGHTS_ONLY: paste_front's occlusion pass) — returns
    (None, depth, wsum, None); where the library has a weights-only kernel the colours are never decoded, elsewhere the full
    kernel runs: wsum / depth are the full launch's bits either way."""
    if weights_only:
        if dumps:
            raise RuntimeError("render: weights_only and dumps exclude each other")
        opts = _with_flag(opts, _lib.P3D_FLAG_WEIGHTS_ONLY)
    if per_view_clamp:
        opts = _with_flag(opts, _lib.P3D_FLAG_PER_VIEW_CLAMP)
    planes_nhwc = _chk(planes_nhwc, "planes_nhwc")
    if rng_seed is not None and (jitter is not None or u is not None):
        raise RuntimeError("render: pass either the two draws (jitter, u) or rng_seed, not both")
    rays_o, rays_d = _chk(rays_o, "ray_origins"), _chk(rays_d, "ray_directions")
    if rng_seed is None:
        jitter = _chk(jitter, "jitter")
    N, three, H, W, Cc = planes_nhwc.shape
    if N == 1 and rays_o.dim() == 3 and rays_o.shape[0] > 1:  # many views of ONE subject in one launch: planes are shared
        N, opts = rays_o.shape[0], _with_flag(opts, _lib.P3D_FLAG_SHARED_PLANES)
    if three != 3 or Cc != 32 or rays_o.dim() != 3 or rays_o.shape[0] != N or rays_o.shape[2] != 3 or rays_d.shape != rays_o.shape:
        raise RuntimeError("planes_nhwc must be [N,3,H,W,32] (or [1,...] shared by all views); ray_origins / ray_directions [N,R,3]")
    R = rays_o.shape[1]
    Sc, Sf = opts.Sc, opts.Sf
    if rng_seed is None and jitter.numel() != N * R * Sc:
        raise RuntimeError(f"jitter must hold N*R*Sc = {N * R * Sc} values")
    if Sf > 0 and rng_seed is None:
        u = _chk(u, "u")
        if u.numel() != N * R * Sf:
            raise RuntimeError(f"u must hold N*R*Sf = {N * R * Sf} values")
    else:
        u = None
    w0, b0, w1, b1 = _chk_mlp(mlp)
    dev = rays_o.device
    feat = torch.empty((N, R, 32), dtype=torch.float32, device=dev)
    depth = torch.empty((N, R, 1), dtype=torch.float32, device=dev)
    wsum = torch.empty((N, R, 1), dtype=torch.float32, device=dev)
    xyz = torch.empty((N, R, 3), dtype=torch.float32, device=dev)
    L = _lib.lib()
    wsb = L.p3d_render_workspace_bytes(N, R, Sc, Sf)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    d, dm = None, None
    if dumps:
        NR, S = N * R, Sc + Sf
        f32 = dict(dtype=torch.float32, device=dev)
        d = dict(depths_coarse=torch.empty((NR, Sc), **f32), sigma_coarse=torch.empty((NR, Sc), **f32),
                 weights_coarse=torch.empty((NR, Sc - 1), **f32), depths_fine=torch.empty((NR, Sf), **f32),
                 inds=torch.empty((NR, Sf), dtype=torch.int32, device=dev), depths_sorted=torch.empty((NR, S), **f32),
                 sigma_sorted=torch.empty((NR, S), **f32), depth_unclamped=torch.empty((NR,), **f32),
                 tminmax=torch.empty((2,), **f32))
        dm = Dumps(*[_p(d[k]) for k in DUMP_KEYS])
    rs_t = re_t = None
    if ray_limits is not None and (opts.flags & _lib.P3D_FLAG_DISPARITY):
        raise NotImplementedError("disparity_space_sampling with per-ray limits")
    if ray_limits is None and not (opts.flags & _lib.P3D_FLAG_DISPARITY) and not (opts.ray_end > opts.ray_start):
        raise RuntimeError("empty depth range: rendering_options with ray_start = ray_end = 'auto' need ray_limits (per-ray limits)")
    if ray_limits is not None:  # ray_start = ray_end = 'auto' (renderer.py:165-171): per-ray limits [N,R(,1)], already patched
        rs_t, re_t = _chk(ray_limits[0], "ray_limits[0]"), _chk(ray_limits[1], "ray_limits[1]")
        if rs_t.numel() != N * R or re_t.numel() != N * R:
            raise RuntimeError(f"ray_limits must hold N*R = {N * R} values each")
    with _on(dev):
        if rng_seed is not None:
            rc = L.p3d_render_rng_f32(_p(planes_nhwc), N, H, W, _p(rays_o), _p(rays_d), R, int(ray_tile_w), int(rng_seed) & (2 ** 64 - 1),
                                      _p(w0), _p(b0), _p(w1), _p(b1), _p(rs_t), _p(re_t), C.byref(opts), _p(feat), _p(depth),
                                      _p(wsum), _p(xyz), _p(ws), wsb, C.byref(dm) if dm is not None else None, _stream())
        else:
            rc = L.p3d_render_limits_f32(_p(planes_nhwc), N, H, W, _p(rays_o), _p(rays_d), R, int(ray_tile_w), _p(jitter), _p(u),
                                         _p(w0), _p(b0), _p(w1), _p(b1), _p(rs_t), _p(re_t), C.byref(opts), _p(feat), _p(depth),
                                         _p(wsum), _p(xyz), _p(ws), wsb, C.byref(dm) if dm is not None else None, _stream())
    _lib.check(rc, "p3d_render_rng_f32" if rng_seed is not None else "p3d_render_limits_f32")
    if stats is not None:  # synchronises: wave-level decode steps executed (32 samples each) vs the full count
        steps = int(ws[8:16].view(torch.int64).item())
        tiled = bool(ray_tile_w and R % ray_tile_w == 0 and ray_tile_w % 8 == 0 and (R // ray_tile_w) % 4 == 0)
        tiles = (R // 32) * N if tiled else -(-R // 32) * N
        pair = not dumps and not (opts.flags & _lib.P3D_FLAG_NO_PAIR) and tiles <= 512  # the host's choice (p3d_render_f32)
        quad = pair and not (opts.flags & _lib.P3D_FLAG_PAIR16) and bool((opts.flags & _lib.P3D_FLAG_QUAD8) or N * R <= 8192 or
                                                                         ((opts.flags & _lib.P3D_FLAG_FAST_COLOR) and ((Sf == 96 and Sc <= 96) or Sf == 48)))
        kind = ("quad" if quad else "pair") if pair else None
        if kind == "pair":  # 16 rays x 2 samples per wave-step
            tiles = (R // 16) * N if tiled else -(-R // 16) * N
            full = tiles * ((-(-Sc // 2) + -(-(Sc + Sf) // 2)) if Sf > 0 else -(-Sc // 2))
        elif kind == "quad":  # 8 rays x 4 samples per wave-step
            tiles = (R // 8) * N if tiled else -(-R // 8) * N
            full = tiles * ((-(-Sc // 4) + -(-(Sc + Sf) // 4)) if Sf > 0 else -(-Sc // 4))
        else:
            full = tiles * (Sc + Sc + Sf if Sf > 0 else Sc)
        stats.update(decode_steps=steps, decode_steps_full=full, small_launch_kernel=pair, small_launch_kind=kind)
    if weights_only:  # (feat / xyz may have been left unwritten)
        return None, depth, wsum, None
    return (feat, depth, wsum, xyz, d) if dumps else (feat, depth, wsum, xyz)


def sample_stratified(ray_start, ray_end, S, jitter):
    """sample_stratified (renderer.py:320-324): jitter [...,S(,1)] -> depths, same shape."""
    jitter = _chk(jitter, "jitter")
    out = torch.empty_like(jitter)
    NR = jitter.numel() // S
    with _on(jitter.device):
        rc = _lib.lib().p3d_sample_stratified_f32(np.float32(ray_start), np.float32(ray_end),
                                                  np.float32((float(ray_end) - float(ray_start)) / (S - 1)), int(S),
                                                  _p(jitter), NR, _p(out), _stream())
    _lib.check(rc, "p3d_sample_stratified_f32")
    return out


def composite(colors, densities, depths, white_back=True):
    """MipRayMarcher2.run_forward (ray_marcher.py:25-57): colors [N,R,S,K], densities [N,R,S,1], depths [N,R,S,1] ->
    (rgb [N,R,K], depth [N,R,1], weights [N,R,S-1,1])."""
    colors, densities, depths = _chk(colors, "colors"), _chk(densities, "densities"), _chk(depths, "depths")
    lead = colors.shape[:-2]
    S, K = colors.shape[-2], colors.shape[-1]
    NR = colors.numel() // (S * K)
    if densities.numel() != NR * S or depths.numel() != NR * S:
        raise RuntimeError("densities / depths must be [...,S,1] matching colors [...,S,K]")
    dev = colors.device
    rgb = torch.empty(lead + (K,), dtype=torch.float32, device=dev)
    depth = torch.empty(lead + (1,), dtype=torch.float32, device=dev)
    w = torch.empty(lead + (S - 1, 1), dtype=torch.float32, device=dev)
    ws = torch.empty((_lib.lib().p3d_composite_workspace_bytes(NR, S, K),), dtype=torch.uint8, device=dev)
    with _on(dev):
        rc = _lib.lib().p3d_composite_f32(_p(colors), _p(densities), _p(depths), NR, S, K, int(bool(white_back)), _p(rgb),
                                          _p(depth), _p(w), _p(ws), _stream())
    _lib.check(rc, "p3d_composite_f32")
    return rgb, depth, w


def paste_front(weights, xyz, occ, rays_o, rays_d, front, image, thresh_weight, thresh_edges, thresh_occ, thresh_dxyz, box_warp,
                normalize_images):
    """paste_front's masks + illustration sampling + lerp in one launch (p3d_paste_front_f32; training/triplane.py:607-691).
    weights / occ [N,1,r,r], xyz / rays_o / rays_d [N,3,r,r], front [N or 1,3,S,S], image [N,3,S,S] ->
    dict(image, paste, mask, mask_weights, mask_edges, mask_occ, mask_dxyz) at S x S."""
    weights, xyz, occ = _chk(weights, "weights"), _chk(xyz, "xyz"), _chk(occ, "occ")
    rays_o, rays_d, front, image = _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d"), _chk(front, "front"), _chk(image, "image")
    N, _, r, r2 = xyz.shape
    S = image.shape[-1]
    if (r != r2 or tuple(weights.shape) != (N, 1, r, r) or tuple(occ.shape) != (N, 1, r, r) or tuple(rays_o.shape) != (N, 3, r, r)
            or tuple(rays_d.shape) != (N, 3, r, r) or tuple(image.shape) != (N, 3, S, S) or front.shape[0] not in (1, N)
            or tuple(front.shape[1:]) != (3, S, S)):
        raise RuntimeError("paste_front: weights/occ [N,1,r,r], xyz/rays [N,3,r,r], front [N|1,3,S,S], image [N,3,S,S]")
    dev = image.device
    new = lambda c: torch.empty((N, c, S, S), dtype=torch.float32, device=dev)
    out = dict(image=new(3), paste=new(3), mask=new(1), mask_weights=new(1), mask_edges=new(1), mask_occ=new(1), mask_dxyz=new(1))
    a = _lib.PasteArgs(_p(weights).value, _p(xyz).value, _p(occ).value, _p(rays_o).value, _p(rays_d).value, _p(front).value,
                       _p(image).value, *[_p(out[k]).value for k in ("image", "paste", "mask", "mask_weights", "mask_edges", "mask_occ", "mask_dxyz")],
                       N, r, S, int(front.shape[0] == 1 and N > 1), int(bool(normalize_images)),
                       np.float32(thresh_weight), np.float32(thresh_edges), np.float32(thresh_occ), np.float32(thresh_dxyz), np.float32(box_warp))
    with _on(dev):
        rc = _lib.lib().p3d_paste_front_f32(C.byref(a), _stream())
    _lib.check(rc, "p3d_paste_front_f32")
    return out


def depth_minmax(depths):
    """torch.min(depths), torch.max(depths) of ray_marcher.py:50 as one op -> float32 tensor [2] on the device."""
    depths = _chk(depths, "depths")
    out = torch.empty((2,), dtype=torch.float32, device=depths.device)
    ws = torch.empty((16,), dtype=torch.uint8, device=depths.device)
    with _on(depths.device):
        rc = _lib.lib().p3d_depth_minmax_f32(_p(depths), depths.numel(), _p(out), _p(ws), 16, _stream())
    _lib.check(rc, "p3d_depth_minmax_f32")
    return out


def importance(depths, weights, u, return_inds=False):
    """sample_importance (renderer.py:328-346): depths [N,R,Sc,1], weights [N,R,Sc-1,1], u [N*R,Sf] -> [N,R,Sf,1]."""
    depths, weights, u = _chk(depths, "depths"), _chk(weights, "weights"), _chk(u, "u")
    Sf = u.shape[-1]
    NR = u.numel() // Sf
    Sc = depths.numel() // NR
    if weights.numel() != NR * (Sc - 1):
        raise RuntimeError("weights must hold Sc-1 values per ray")
    out = torch.empty(tuple(depths.shape[:2]) + (Sf, 1) if depths.dim() == 4 else (NR, Sf), dtype=torch.float32, device=u.device)
    inds = torch.empty((NR, Sf), dtype=torch.int32, device=u.device) if return_inds else None
    with _on(u.device):
        rc = _lib.lib().p3d_importance_f32(_p(depths), _p(weights), NR, Sc, Sf, _p(u), _p(out), _p(inds), _stream())
    _lib.check(rc, "p3d_importance_f32")
    return (out, inds) if return_inds else out


def unify_perm(depths_coarse, depths_fine):
    """The sort of unify_samples (renderer.py:295): stable ascending permutation of cat([coarse, fine]) per ray -> int32 [NR,S]."""
    dc, df = _chk(depths_coarse, "depths_coarse"), _chk(depths_fine, "depths_fine")
    NR = dc.shape[0] if dc.dim() == 2 else dc.numel() // dc.shape[-2]
    Sc, Sf = dc.numel() // NR, df.numel() // NR
    perm = torch.empty((NR, Sc + Sf), dtype=torch.int32, device=dc.device)
    with _on(dc.device):
        rc = _lib.lib().p3d_unify_perm_f32(_p(dc), _p(df), NR, Sc, Sf, _p(perm), _stream())
    _lib.check(rc, "p3d_unify_perm_f32")
    return perm


# ======================================================================================================================
# StyleGAN2 synthesis operators (torch_utils/ops/{bias_act,upfirdn2d,conv2d_resample}.py, networks_stylegan2.py:40-97)
# ======================================================================================================================
_ACTS = {"linear": (0, 0.0, 1.0), "lrelu": (1, 0.2, float(np.sqrt(2)))}  # bias_act.py:23-33: index, def_alpha, def_gain


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None):
    """bias_act.bias_act (bias_act.py:54-88): clamp(act(x + b) * gain).  Only 'linear' and 'lrelu' are on the hot path."""
    x = _chk(x, "x")
    if act not in _ACTS:
        raise NotImplementedError(f"activation {act!r} is not used by the PAniC-3D generator")
    assert clamp is None or clamp >= 0  # bias_act.py:98
    idx, da, dg = _ACTS[act]
    alpha = float(alpha if alpha is not None else da)
    gain = float(gain if gain is not None else dg)
    clamp = float(clamp if clamp is not None else -1)
    if b is not None:
        b = _chk(b, "b")
        assert b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]  # bias_act.py:105-107
    C = x.shape[dim]
    outer = int(np.prod(x.shape[:dim])) if dim > 0 else 1
    inner = int(np.prod(x.shape[dim + 1:])) if dim + 1 < x.ndim else 1
    y = torch.empty_like(x)
    with _on(x.device):
        rc = _lib.lib().p3d_bias_act_f32(_p(x), _p(b), outer, C, inner, idx, alpha, gain, clamp, _p(y), _stream())
    _lib.check(rc, "p3d_bias_act_f32")
    return y


def setup_filter(f=(1, 3, 3, 1), device=None):
    """upfirdn2d.setup_filter (upfirdn2d.py:72-117) for the non-separable case: outer product, normalised to DC gain 1."""
    f = torch.as_tensor(f, dtype=torch.float32)
    if f.ndim == 1:
        f = f.ger(f)
    f = f / f.sum()
    return f.to(device) if device is not None else f


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    padding = list(padding)
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    return [int(v) for v in padding]


_FIR_CACHE = {}


def prepared_filter(f, device, gain=1.0, flip_filter=False):
    """`f * gain`, flipped for convolution unless flip_filter (upfirdn2d.py:193-196), float32 contiguous on `device` — made
    once per (filter tensor object, version, gain, flip): the reference rebuilds it on every call (three tiny launches)."""
    key = (id(f), device, float(gain), bool(flip_filter))
    hit = _FIR_CACHE.get(key)
    if memo.enabled() and hit is not None and hit[0]() is f and hit[1] == f._version:
        return hit[2]
    ff = f.detach().to(device, torch.float32) * float(gain)
    if not flip_filter:
        ff = ff.flip([0, 1])
    ff = ff.contiguous()
    if len(_FIR_CACHE) > 256:
        _FIR_CACHE.clear()
    _FIR_CACHE[key] = (weakref.ref(f), f._version, ff)
    return ff


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.upfirdn2d (upfirdn2d.py:120-167), 2-D filter, same up/down factor in x and y."""
    x = _chk(x, "x")
    if x.ndim != 4 or f is None or f.ndim != 2:
        raise NotImplementedError("upfirdn2d: [N,C,H,W] input and a 2-D filter are what the generator uses")
    px0, px1, py0, py1 = _parse_padding(padding)
    ff = prepared_filter(f, x.device, gain, flip_filter)
    N, Cc, H, W = x.shape
    fh, fw = ff.shape
    OH = (H * up + py0 + py1 - fh) // down + 1
    OW = (W * up + px0 + px1 - fw) // down + 1
    y = torch.empty((N, Cc, OH, OW), dtype=torch.float32, device=x.device)
    with _on(x.device):
        rc = _lib.lib().p3d_upfirdn2d_f32(_p(x), N * Cc, H, W, _p(ff), fh, fw, int(up), int(down), px0, px1, py0, py1, _p(y), _stream())
    _lib.check(rc, "p3d_upfirdn2d_f32")
    return y


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.upsample2d (upfirdn2d.py:315-350)."""
    px0, px1, py0, py1 = _parse_padding(padding)
    fh, fw = f.shape
    p = [px0 + (fw + up - 1) // 2, px1 + (fw - up) // 2, py0 + (fh + up - 1) // 2, py1 + (fh - up) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * up * up)


def upsample2d_add(x, f, add=None):
    """`upsample2d(x, f)` (up = 2, the 4x4 resample filter) fused with the `img.add_(y)` that follows it in SynthesisBlock.forward
    (networks_stylegan2.py:476-478): returns upsample2d(x, f) + add in one launch (polyphase FIR, same bits as upfirdn2d)."""
    x = _chk(x, "x")
    N, Cc, H, W = x.shape
    if tuple(f.shape) != (4, 4) or (2 * W) % 4 != 0:
        y = upsample2d(x, f)
        return y if add is None else y.add_(add)
    ff = prepared_filter(f, x.device, 4.0, False)
    if add is not None:
        add = _chk(add, "add")
        if tuple(add.shape) != (N, Cc, 2 * H, 2 * W):
            raise RuntimeError("add must be [N,C,2H,2W]")
    y = torch.empty((N, Cc, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    with _on(x.device):
        rc = _lib.lib().p3d_upsample2d_add_f32(_p(x), N * Cc, H, W, _p(ff), _p(add), _p(y), _stream())
    _lib.check(rc, "p3d_upsample2d_add_f32")
    return y


def torgb_weights(weight):
    """ToRGB weights [O,I,1,1] -> the transposed, channel-padded [I, 32 or 96] copy p3d_torgb_f32 streams (made once per layer)."""
    weight = _chk(weight, "weight")
    O, I = weight.shape[0], weight.shape[1]
    if weight.shape[2:] != (1, 1) or O > 96:
        raise RuntimeError("torgb_weights: [O <= 96, I, 1, 1]")
    wt = torch.empty((I, 32 if O <= 32 else 96), dtype=torch.float32, device=weight.device)
    with _on(weight.device):
        _lib.check(_lib.lib().p3d_torgb_weights_f32(_p(weight), O, I, _p(wt), _stream()), "p3d_torgb_weights_f32")
    return wt


def torgb(x, weight_t, out_channels, styles, bias=None, clamp=None, skip=None, skip_filter=None):
    """ToRGBLayer.forward (networks_stylegan2.py:366-380) + the skip connection of SynthesisBlock.forward (:476-478) in ONE launch:
    `upsample2d(skip, skip_filter) + (conv1x1(x * styles) + bias)`.  weight_t from torgb_weights; styles [N,I] already multiplied
    by the layer's weight_gain; skip [N,O,H/2,W/2] or None."""
    x, weight_t, styles = _chk(x, "x"), _chk(weight_t, "weight_t"), _chk(styles, "styles")
    N, I, H, W = x.shape
    O = int(out_channels)
    if tuple(weight_t.shape) != (I, 32 if O <= 32 else 96) or tuple(styles.shape) != (N, I):
        raise RuntimeError("torgb: x [N,I,H,W], weight_t [I,32|96] (torgb_weights), styles [N,I]")
    if bias is not None:
        bias = _chk(bias, "bias")
    skipf = None
    if skip is not None:
        skip = _chk(skip, "skip")
        if tuple(skip.shape) != (N, O, H // 2, W // 2) or H % 2 or W % 2:
            raise RuntimeError("skip must be [N,O,H/2,W/2] (the previous block's image)")
        skipf = prepared_filter(skip_filter, x.device, 4.0, False)  # upsample2d: up 2, gain up^2 (upfirdn2d.py:341-350)
        if tuple(skipf.shape) != (4, 4):
            raise NotImplementedError("skip_filter must be the 4x4 [1,3,3,1] filter")
    y = torch.empty((N, O, H, W), dtype=torch.float32, device=x.device)
    with _on(x.device):
        rc = _lib.lib().p3d_torgb_f32(_p(x), N, I, H, W, _p(weight_t), O, _p(styles), _p(bias), float(clamp if clamp is not None else -1),
                                      _p(skip), _p(skipf), _p(y), _stream())
    _lib.check(rc, "p3d_torgb_f32")
    return y


_FUSES_TORGB = {}


def conv_fuses_torgb(N, I, O, H, W, rgb_channels):
    """The LIBRARY's rule (p3d_conv_fuses_torgb) for when a plain 3x3 layer of this shape takes its block's ToRGB layer along
    (modulated_conv2d(..., rgb_weight=, rgb_styles=)); asked, not restated; memoised like takes_image."""
    key = (int(N), int(I), int(O), int(H), int(W), int(rgb_channels))
    r = _FUSES_TORGB.get(key)
    if r is None:
        r = _FUSES_TORGB[key] = bool(_lib.lib().p3d_conv_fuses_torgb(*key))
    return r


def torgb_combine(partial, bias=None, clamp=None, skip=None, skip_filter=None):
    """The second half of a ToRGB layer whose channel sums came out of conv1's launch (modulated_conv2d(..., rgb_weight=...)):
    partial [tiles,N,O,H,W] -> `upsample2d(skip, skip_filter) + (sum over tiles + bias)` [N,O,H,W] (p3d_torgb_combine_f32)."""
    partial = _chk(partial, "partial")
    T, N, O, H, W = partial.shape
    if bias is not None:
        bias = _chk(bias, "bias")
    skipf = None
    if skip is not None:
        skip = _chk(skip, "skip")
        if tuple(skip.shape) != (N, O, H // 2, W // 2) or H % 2 or W % 2:
            raise RuntimeError("skip must be [N,O,H/2,W/2] (the previous block's image)")
        skipf = prepared_filter(skip_filter, partial.device, 4.0, False)
        if tuple(skipf.shape) != (4, 4):
            raise NotImplementedError("skip_filter must be the 4x4 [1,3,3,1] filter")
    y = torch.empty((N, O, H, W), dtype=torch.float32, device=partial.device)
    with _on(partial.device):
        rc = _lib.lib().p3d_torgb_combine_f32(_p(partial), T, N, O, H, W, _p(bias), float(clamp if clamp is not None else -1), _p(skip),
                                              _p(skipf), _p(y), _stream())
    _lib.check(rc, "p3d_torgb_combine_f32")
    return y


class WeightOperand(torch.Tensor):
    """The two-term copy of a layer's weights in one of the image layouts (P3D_WLAYOUT_PLAIN / _UP: the consuming kernel's LDS image per
    chunk and channel tile, include/panic3d_hip.h): the same [2,O,9,I] float16 storage size, ANOTHER element order — the subclass
    carries which one (`p3d_layout`) so that modulated_conv2d can tell the library."""
    p3d_layout = 0


_WLAYOUT = {}


def conv_weight_layout(I, O, W, up):
    """The layout of the two-term weight copy the LIBRARY wants for a 3x3 layer of this shape (p3d_conv_weight_layout; W: the input
    map's width): asked, not restated; fixed for the life of the process."""
    key = (int(I), int(O), int(W), int(up))
    r = _WLAYOUT.get(key)
    if r is None:
        r = _WLAYOUT[key] = int(_lib.lib().p3d_conv_weight_layout(*key))
    return r


def conv_weights_to_f16(weight, split=False, layout=0):
    """[O,I,k,k] f32 -> the [O,k*k,I] f16 operand copy of the f16-operand convolution (made once per layer); split: the
    [2,O,k*k,I] hi / lo pair of the two-term variant (hi = f16(w), lo = f16(w - hi)).  layout (split, 3x3): conv_weight_layout(...) of
    the layer the copy is for — non-zero: a WeightOperand in that image layout (same bytes, the pipelined kernels' own order)."""
    weight = _chk(weight, "weight")
    O, I, kh, kw = weight.shape
    wh = torch.empty((2, O, kh * kw, I) if split else (O, kh * kw, I), dtype=torch.float16, device=weight.device)
    if layout:
        if not split:
            raise RuntimeError("conv_weights_to_f16: the image layouts are two-term layouts (split=True)")
        with _on(weight.device):
            _lib.check(_lib.lib().p3d_conv_weights_to_f16x2_layout(_p(weight), O, I, kh, int(layout), _p(wh), _stream()), "p3d_conv_weights_to_f16x2_layout")
        wh = wh.as_subclass(WeightOperand)
        wh.p3d_layout = int(layout)
        return wh
    with _on(weight.device):
        if split:
            _lib.check(_lib.lib().p3d_conv_weights_to_f16x2(_p(weight), O, I, kh, _p(wh), _stream()), "p3d_conv_weights_to_f16x2")
        else:
            _lib.check(_lib.lib().p3d_conv_weights_to_f16(_p(weight), O, I, kh, _p(wh), _stream()), "p3d_conv_weights_to_f16")
    return wh


def conv_domain_flag(device):
    """A caller-owned flag word for the two-term f16 convolutions (`saturated` argument of p3d_modconv2d_f16x2mma_f32, ABI 5): the
    kernels OR it with 1 when a modulated activation leaves the domain |s*x| <= 4094 (or is NaN).  Zeroed here."""
    return torch.zeros(1, dtype=torch.int32, device=device)


def conv_domain_violated(flag, reset=True):
    """Read a conv_domain_flag (synchronises the stream); reset: zero it again."""
    hit = bool(flag.item() != 0)
    if hit and reset:
        flag.zero_()
    return hit


def demod_coefs(w2_all, styles_all, table, L, N, total_waves, out):
    """Demodulation coefficients of L layers in one launch (p3d_demod_coefs_f32); see stylegan2.StylePlan."""
    with _on(out.device):
        rc = _lib.lib().p3d_demod_coefs_f32(_p(w2_all), _p(styles_all), _p(table), int(L), int(N), int(total_waves), _p(out), _stream())
    _lib.check(rc, "p3d_demod_coefs_f32")
    return out


_CONV_SCRATCH = {}  # (device index, stream handle) -> workspace tensor
_WSB = {}           # (N, I, O, H, W, up) -> p3d_modconv2d_workspace_bytes (a pure function of the shape)


def _conv_scratch(device, nbytes):
    """The convolution workspace (demodulation coefficients, the transposed-conv intermediate, split-K partial sums), per
    (device, stream): launches on one stream are ordered, so consecutive convolutions share ONE workspace instead of allocating
    one per call (the batch-1 backbone is host-bound in its 4^2 .. 32^2 layers)."""
    idx = device.index if device.index is not None else _cur_device()
    key = (idx, _raw_stream(idx))
    ws = _CONV_SCRATCH.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty((int(nbytes * 1.25) + 4096,), dtype=torch.uint8, device=device)
        _CONV_SCRATCH[key] = ws
    return ws


class ActImage:
    """The operand of a plain 3x3 two-term layer prepared by the layer in front of it (include/panic3d_hip.h, "activation IMAGE"):
    data = float16 [2 (hi | lo), N, C/8, H, W, 8] holding split(16 * styles * x) for the CONSUMER's styles.  Produced by an
    up-sampling modulated_conv2d called with next_styles=... (or act_to_image); consumed by modulated_conv2d in place of x."""
    __slots__ = ("data", "shape")

    def __init__(self, data, shape):
        self.data, self.shape = data, tuple(shape)

    @staticmethod
    def empty(N, C, H, W, device):
        if C % 8:
            raise RuntimeError("ActImage: C % 8 != 0")
        return ActImage(torch.empty((2, N, C // 8, H, W, 8), dtype=torch.float16, device=device), (N, C, H, W))

    @property
    def device(self):
        return self.data.device

    def float(self):
        """(hi + lo) / 16 as fp32 [N,C,H,W]: the MODULATED activation styles * x (tests)."""
        N, C, H, W = self.shape
        v = (self.data[0].float() + self.data[1].float()) * (1.0 / 16.0)
        return v.permute(0, 1, 4, 2, 3).reshape(N, C, H, W).contiguous()


_TAKES_IMAGE = {}


def takes_image(I, O, W, up):
    """The LIBRARY's rule for when a 3x3 layer with these sizes stages its input from an ActImage (p3d_conv_takes_image: k_modconv_w3 for
    up = 1, k_modconv_up3 for up = 2) — asked, not restated, so that the host's choice of hand-overs and the library's acceptance of
    x_img cannot drift apart; fixed for the life of the process, hence memoised."""
    key = (int(I), int(O), int(W), int(up))
    r = _TAKES_IMAGE.get(key)
    if r is None:
        r = _TAKES_IMAGE[key] = bool(_lib.lib().p3d_conv_takes_image(*key))
    return r


def takes_image_up(I, O, W):
    """An up-sampling 3x3 layer with these sizes stages its input from an ActImage (k_modconv_up3)."""
    return takes_image(I, O, W, 2)


def act_to_image(x, styles=None, saturated=None):
    """p3d_act_to_image_f32: fp32 [N,C,H,W] (* styles [N,C]) -> ActImage."""
    x = _chk(x, "x")
    N, C, H, W = x.shape
    if styles is not None:
        styles = _chk(styles, "styles")
        if tuple(styles.shape) != (N, C):
            raise RuntimeError("act_to_image: styles [N,C]")
    img = ActImage.empty(N, C, H, W, x.device)
    with _on(x.device):
        _lib.check(_lib.lib().p3d_act_to_image_f32(_p(x), _p(styles), N, C, H, W, _p(img.data), _p(saturated), _stream()), "p3d_act_to_image_f32")
    return img


def act_to_image_add(x, styles, add, c0, saturated=None):
    """p3d_act_to_image_add_f32: `x[:, c0 : c0 + add.shape[1]] += add` IN PLACE (add [1 or N, Ca, H, W]) and the ActImage of the updated
    x for a consumer with `styles` — the conditioning add between two synthesis blocks and the conversion pass in one launch."""
    if not (isinstance(x, torch.Tensor) and x.is_contiguous()):
        raise RuntimeError("act_to_image_add: x is updated in place and must be contiguous")
    x, styles, add = _chk(x, "x"), _chk(styles, "styles"), _chk(add, "add")
    N, C, H, W = x.shape
    if add.ndim != 4 or tuple(add.shape[2:]) != (H, W) or add.shape[0] not in (1, N) or tuple(styles.shape) != (N, C):
        raise RuntimeError("act_to_image_add: x [N,C,H,W], styles [N,C], add [1|N,Ca,H,W]")
    img = ActImage.empty(N, C, H, W, x.device)
    with _on(x.device):
        _lib.check(_lib.lib().p3d_act_to_image_add_f32(_p(x), _p(styles), N, C, H, W, _p(add), int(c0), int(add.shape[1]), int(add.shape[0]),
                                                       _p(img.data), _p(saturated), _stream()), "p3d_act_to_image_add_f32")
    return img


def modulated_conv2d(x, weight, styles, noise=None, up=1, padding=0, resample_filter=None, demodulate=True,
                     bias=None, act="linear", gain=None, clamp=None, weight_f16=None, dcoef=None, saturated=None,
                     next_styles=None, rgb_weight=None, rgb_styles=None, want_y=True):
    """modulated_conv2d (networks_stylegan2.py:40-97) FUSED with the bias_act that follows it in SynthesisLayer.forward
    (:350-352) / ToRGBLayer.forward (:379).  Supported shapes are the generator's: 3x3 / padding 1 / up 1 or 2, and 1x1.
    noise: None, [H,W] (noise_const * strength) or [N,1,H,W] (random * strength).
    weight_f16 (from conv_weights_to_f16): run the matrix cores on f16 operands (fp32 accumulate, fp32 in/out) — what the
    reference's fp16 super-resolution blocks do on the GPU, with less rounding; needs I % 16 == 0.  A [2,O,k*k,I] tensor
    (conv_weights_to_f16(split=True)) selects the two-term variant: fp32-class results on the f16 matrix cores; `saturated`: an
    int32 [1] device tensor of the caller (conv_domain_flag) that the two-term kernels OR with 1 when |s*x| > 4094.
    x may be an ActImage (3x3, up 1, two-term weights, dcoef given: styles are already in it).  next_styles [N,O] (up 2 only):
    return the result as the ActImage of a following layer with those styles instead of an fp32 tensor.
    rgb_weight [R<=4,O] + rgb_styles [N,O] (plain 3x3 layer with an ActImage input, only where conv_fuses_torgb(...) says so): the
    block's ToRGB layer rides on this launch — the call returns (y, image or None, partial [O/64,N,R,H,W]) and torgb_combine(partial,
    ...) finishes the ToRGB layer; want_y=False: y is not written (None is returned in its place)."""
    ximg = x if isinstance(x, ActImage) else None
    if ximg is not None:
        if up == 2 and not takes_image_up(ximg.shape[1], weight.shape[0], ximg.shape[3]):
            raise RuntimeError("modulated_conv2d: the library does not stage this up-sampling layer from an ActImage (p3d_conv_takes_image: I % 16 == 0, O % 32 == 0, W >= P3D_UP3_MIN_W)")
        if weight_f16 is None or weight_f16.ndim != 4 or (demodulate and dcoef is None):
            raise RuntimeError("modulated_conv2d: an ActImage input needs two-term weights (conv_weights_to_f16(split=True)) and precomputed dcoef")
        x = None
        weight = _chk(weight, "weight")
        N, I, H, W = ximg.shape
    else:
        x, weight, styles = _chk(x, "x"), _chk(weight, "weight"), _chk(styles, "styles")
        N, I, H, W = x.shape
    O, I2, kh, kw = weight.shape
    if I2 != I or (ximg is None and tuple(styles.shape) != (N, I)):
        raise RuntimeError("modulated_conv2d: x [N,I,H,W], weight [O,I,k,k], styles [N,I]")
    dev_ = ximg.device if ximg is not None else x.device
    if not ((kh == kw == 3 and padding == 1 and up in (1, 2)) or (kh == kw == 1 and padding == 0 and up == 1)):
        raise NotImplementedError("modulated_conv2d: only 3x3/pad 1/up 1|2 and 1x1 are on the hot path")
    idx, da, dg = _ACTS[act]
    gain = float(gain if gain is not None else dg)
    clampv = float(clamp if clamp is not None else -1)
    nps = 0
    if noise is not None:
        noise = _chk(noise, "noise")
        nps = 1 if noise.numel() == N * H * up * W * up and noise.ndim == 4 and N > 1 else 0
        if noise.numel() not in (H * up * W * up, N * H * up * W * up):
            raise RuntimeError("noise must be [H*up, W*up] or [N,1,H*up,W*up]")
    fir = None
    if up == 2:
        fir = prepared_filter(resample_filter, dev_, 4.0, False)  # upfirdn2d.py:193-196, gain = up^2
        if tuple(fir.shape) != (4, 4):
            raise NotImplementedError("resample_filter must be the 4x4 [1,3,3,1] filter")
    if bias is not None:
        bias = _chk(bias, "bias")
    if dcoef is not None:
        dcoef = _chk(dcoef, "dcoef")
        if dcoef.numel() != N * O:
            raise RuntimeError("dcoef must hold N*O demodulation coefficients")
    out_image = next_styles is not None
    if out_image:
        next_styles = _chk(next_styles, "next_styles")
        if tuple(next_styles.shape) != (N, O) or kh != 3:
            raise RuntimeError("next_styles [N,O] goes with a 3x3 layer")
    both = out_image and up == 1  # a plain layer writes the image NEXT TO the fp32 result (ToRGB reads the one, the next conv0 the other)
    rgb = rgb_weight is not None
    rgbp = None
    if rgb:
        rgb_weight, rgb_styles = _chk(rgb_weight, "rgb_weight"), _chk(rgb_styles, "rgb_styles")
        R = rgb_weight.shape[0]
        if ximg is None or up != 1 or kh != 3 or tuple(rgb_weight.shape) != (R, O) or tuple(rgb_styles.shape) != (N, O) \
                or not conv_fuses_torgb(N, I, O, H, W, R):
            raise RuntimeError("modulated_conv2d: rgb_weight [R,O] / rgb_styles [N,O] go with a plain 3x3 layer fed by an ActImage whose "
                               "shape conv_fuses_torgb accepts")
        rgbp = torch.empty((O // 64, N, R, H, W), dtype=torch.float32, device=dev_)
    y = None if ((out_image and not both) or (rgb and not want_y)) else torch.empty((N, O, H * up, W * up), dtype=torch.float32, device=dev_)
    yimg = ActImage.empty(N, O, H * up, W * up, dev_) if out_image else None
    L = _lib.lib()
    mma = _lib.P3D_CONV_MMA_F32
    if weight_f16 is not None:
        split = weight_f16.ndim == 4
        if weight_f16.dtype != torch.float16 or tuple(weight_f16.shape)[-3:] != (O, kh * kw, I) or not weight_f16.is_contiguous() \
                or (split and weight_f16.shape[0] != 2):
            raise RuntimeError("weight_f16 must be the contiguous [O,k*k,I] / [2,O,k*k,I] float16 tensor of conv_weights_to_f16")
        mma = _lib.P3D_CONV_MMA_F16X2 if split else _lib.P3D_CONV_MMA_F16
        if split and saturated is not None and (saturated.dtype != torch.int32 or saturated.numel() != 1 or saturated.device != dev_):
            raise RuntimeError("saturated must be an int32 [1] tensor on x's device (ops.conv_domain_flag)")
    with _on(dev_):
        key = (N, I, O, H, W, up)
        wsb = _WSB.get(key)
        if wsb is None:
            wsb = _WSB[key] = L.p3d_modconv2d_workspace_bytes(N, I, O, H, W, up)
        ws = _conv_scratch(dev_, wsb)
        a = _lib.ConvArgs(x.data_ptr() if x is not None else None, weight.data_ptr(), weight_f16.data_ptr() if weight_f16 is not None else None,
                          styles.data_ptr() if styles is not None else None, dcoef.data_ptr() if dcoef is not None else None,
                          noise.data_ptr() if noise is not None else None, bias.data_ptr() if bias is not None else None,
                          fir.data_ptr() if fir is not None else None, y.data_ptr() if y is not None else None, ws.data_ptr(),
                          saturated.data_ptr() if (saturated is not None and (mma == _lib.P3D_CONV_MMA_F16X2 or yimg is not None)) else None,
                          ximg.data.data_ptr() if ximg is not None else None, yimg.data.data_ptr() if yimg is not None else None,
                          next_styles.data_ptr() if yimg is not None else None,
                          rgb_weight.data_ptr() if rgb else None, rgb_styles.data_ptr() if rgb else None, rgbp.data_ptr() if rgb else None,
                          ws.numel(), N, I, H, W, O, kh, int(up), int(bool(demodulate)), nps, idx, mma, float(da), gain, clampv,
                          int(rgb_weight.shape[0]) if rgb else 0, int(getattr(weight_f16, "p3d_layout", 0)))
        _lib.check(L.p3d_modconv2d_ex_f32(C.byref(a), _stream()), "p3d_modconv2d_ex_f32")
    if rgb:
        return y, yimg, rgbp
    return (y, yimg) if both else (yimg if out_image else y)
