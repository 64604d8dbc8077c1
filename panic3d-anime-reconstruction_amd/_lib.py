"""ctypes binding of libpanic3d_hip.so — the C ABI of include/panic3d_hip.h.  No torch types cross this boundary.

The library is the product: if it is missing this module raises (there is NO CPU / PyTorch fallback path).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("P3D_LIB") or os.path.join(HERE, "libpanic3d_hip.so")  # P3D_LIB: development override

P3D_FLAG_CROP, P3D_FLAG_CULL, P3D_FLAG_BINARIZE, P3D_FLAG_FORCE_SIGMOID, P3D_FLAG_WHITE_BACK, P3D_FLAG_NO_EARLY_OUT, P3D_FLAG_SHARED_PLANES, P3D_FLAG_SKIP_CROPPED, P3D_FLAG_NO_PAIR, P3D_FLAG_FAST_COLOR, P3D_FLAG_PER_VIEW_CLAMP, P3D_FLAG_NO_STAGING, P3D_FLAG_FORCE_STAGING = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 8192
P3D_FLAG_DISPARITY = 4096
P3D_FLAG_PAIR16 = 16384
P3D_FLAG_QUAD8 = 32768
P3D_FLAG_WEIGHTS_ONLY = 65536
P3D_MAX_S = 192
P3D_ABI_VERSION = 9  # include/panic3d_hip.h; lib() refuses a library built for another version


class Opts(C.Structure):
    """p3d_opts"""
    _fields_ = [("coord_scale", C.c_float), ("ray_start", C.c_float), ("ray_end", C.c_float),
                ("depth_delta", C.c_float), ("crop_limit", C.c_float), ("cull_thresh", C.c_float),
                ("Sc", C.c_int32), ("Sf", C.c_int32), ("plane_mode", C.c_int32), ("flags", C.c_int32)]


class Dumps(C.Structure):
    """p3d_dumps"""
    _fields_ = [("depths_coarse", C.c_void_p), ("sigma_coarse", C.c_void_p), ("weights_coarse", C.c_void_p),
                ("depths_fine", C.c_void_p), ("inds", C.c_void_p), ("depths_sorted", C.c_void_p),
                ("sigma_sorted", C.c_void_p), ("depth_unclamped", C.c_void_p), ("tminmax", C.c_void_p)]


class PasteArgs(C.Structure):
    """p3d_paste_args"""
    _fields_ = [(n, C.c_void_p) for n in ("weights", "xyz", "occ", "rays_o", "rays_d", "front", "image", "out_image", "out_paste",
                                          "out_mask", "out_mask_weights", "out_mask_edges", "out_mask_occ", "out_mask_dxyz")] + \
               [(n, C.c_int32) for n in ("N", "r", "S", "front_shared", "normalize_images")] + \
               [(n, C.c_float) for n in ("thresh_weight", "thresh_edges", "thresh_occ", "thresh_dxyz", "box_warp")]


class ConvArgs(C.Structure):
    """p3d_conv_args"""
    _fields_ = [(n, C.c_void_p) for n in ("x", "w", "w_f16", "styles", "demod_coefs", "noise", "bias", "fir", "y", "workspace",
                                          "saturated", "x_img", "y_img", "y_img_styles", "rgb_w", "rgb_styles", "rgb_partial")] + \
               [("workspace_bytes", C.c_size_t)] + \
               [(n, C.c_int32) for n in ("N", "I", "H", "W", "O", "ks", "up", "demodulate", "noise_per_sample", "act", "mma")] + \
               [(n, C.c_float) for n in ("alpha", "gain", "clamp")] + [("rgb_channels", C.c_int32), ("w_f16_layout", C.c_int32)]


P3D_CONV_MMA_F32, P3D_CONV_MMA_F16, P3D_CONV_MMA_F16X2 = 0, 1, 2
P3D_WLAYOUT_OIK, P3D_WLAYOUT_PLAIN, P3D_WLAYOUT_UP = 0, 1, 2

# symbol -> (restype, argtypes); every function include/panic3d_hip.h declares
_P, _I, _L, _F, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
SIGNATURES = {
    "p3d_planes_to_nhwc_f32": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "p3d_decode_features_f32": (_I, [_P, _I, _L, _P, _P, _P, _P, _I, _P, _P, _P]),
    "p3d_struct_layout": (_I, [_I, C.POINTER(C.c_size_t), _I]),
    "p3d_triplane_decode_f32": (_I, [_P, _I, _I, _I, _P, _L, _P, _P, _P, _P, C.POINTER(Opts), _P, _P, _P]),
    "p3d_grid_density_f32": (_I, [_P, _I, _I, _I, _L, _L, _F, _F, _F, _F, _P, _P, _P, _P, C.POINTER(Opts), _P, _P, _F, _P]),
    "p3d_render_workspace_bytes": (_Z, [_I, _L, _I, _I]),
    "p3d_render_f32": (_I, [_P, _I, _I, _I, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, C.POINTER(Opts), _P, _P, _P, _P, _P,
                            _Z, C.POINTER(Dumps), _P]),
    "p3d_render_limits_f32": (_I, [_P, _I, _I, _I, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(Opts), _P, _P, _P, _P, _P,
                                   _Z, C.POINTER(Dumps), _P]),
    "p3d_render_rng_f32": (_I, [_P, _I, _I, _I, _P, _P, _L, _I, C.c_uint64, _P, _P, _P, _P, _P, _P, C.POINTER(Opts), _P, _P, _P, _P, _P,
                                _Z, C.POINTER(Dumps), _P]),
    "p3d_sample_stratified_f32": (_I, [_F, _F, _F, _I, _P, _L, _P, _P]),
    "p3d_composite_workspace_bytes": (_Z, [_L, _I, _I]),
    "p3d_depth_minmax_f32": (_I, [_P, _L, _P, _P, _Z, _P]),
    "p3d_composite_f32": (_I, [_P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P]),
    "p3d_importance_f32": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P]),
    "p3d_unify_perm_f32": (_I, [_P, _P, _L, _I, _I, _P, _P]),
    "p3d_modconv2d_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "p3d_modconv2d_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P, _P, _I, _P, _I, _I, _F, _F, _F, _P, _P, _P, _Z, _P]),
    "p3d_demod_coefs_f32": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "p3d_conv_weights_to_f16": (_I, [_P, _I, _I, _I, _P, _P]),
    "p3d_modconv2d_f16mma_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _P, _P, _I, _P, _I, _I, _F, _F, _F, _P, _P, _P, _Z, _P]),
    "p3d_conv_weights_to_f16x2": (_I, [_P, _I, _I, _I, _P, _P]),
    "p3d_modconv2d_f16x2mma_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _P, _P, _I, _P, _I, _I, _F, _F, _F, _P, _P, _P, _Z, _P, _P]),
    "p3d_modconv2d_ex_f32": (_I, [C.POINTER(ConvArgs), _P]),
    "p3d_conv_takes_image": (_I, [_I, _I, _I, _I]),
    "p3d_act_image_bytes": (_Z, [_I, _I, _I, _I]),
    "p3d_act_to_image_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "p3d_act_to_image_add_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P]),
    "p3d_conv_fuses_torgb": (_I, [_I, _I, _I, _I, _I, _I]),
    "p3d_conv_weight_layout": (_I, [_I, _I, _I, _I]),
    "p3d_conv_weights_to_f16x2_layout": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "p3d_torgb_partial_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "p3d_torgb_combine_f32": (_I, [_P, _I, _I, _I, _I, _I, _P, _F, _P, _P, _P, _P]),
    "p3d_torgb_weights_f32": (_I, [_P, _I, _I, _P, _P]),
    "p3d_torgb_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _F, _P, _P, _P, _P]),
    "p3d_upfirdn2d_f32": (_I, [_P, _L, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "p3d_upsample2d_add_f32": (_I, [_P, _L, _I, _I, _P, _P, _P, _P]),
    "p3d_bias_act_f32": (_I, [_P, _P, _L, _I, _L, _I, _F, _F, _F, _P, _P]),
    "p3d_sigma2density_f32": (_I, [_P, _P, _L, _F, _P, _P]),
    "p3d_mc_workspace_bytes": (_Z, [_I]),
    "p3d_mc_count_f32": (_I, [_P, _I, _I, _F, _P, _Z, _P, _P]),
    "p3d_mc_emit_f32": (_I, [_P, _I, _I, _F, _P, _Z, _L, _L, _P, _P, _P, _P, _P]),
    "p3d_paste_front_f32": (_I, [C.POINTER(PasteArgs), _P]),
    "p3d_build_info": (C.c_char_p, []),
    "p3d_abi_version": (_I, []),
}

_LIB = None


def lib():
    """Load the shared library (once).  Raises RuntimeError when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO):
            raise RuntimeError(f"{SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the HIP path)")
        if "P3D_LIB" not in os.environ:
            from . import _build
            if _build.needs_build():  # compiled from other sources than the ones on disk: never run stale kernels silently
                import sys
                print(f"panic3d_amd: {SO} does not match csrc/ (source hash): rebuilding", file=sys.stderr)
                _build.build()  # (not force: under torch.distributed.run the rank that gets the lock builds, the others find it done)
        L = C.CDLL(SO)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        got = L.p3d_abi_version()
        if got != P3D_ABI_VERSION:
            raise RuntimeError(f"{SO} was built for ABI version {got}, this binding is written for {P3D_ABI_VERSION}: rebuild the library")
        check_struct_layouts(L)
        _LIB = L
    return _LIB


STRUCT_MIRRORS = {0: Opts, 1: Dumps, 2: PasteArgs, 3: ConvArgs}  # P3D_STRUCT_* -> the ctypes restatement above


def check_struct_layouts(L):
    """Compare every ctypes mirror with the layout the library was compiled with (p3d_struct_layout: sizeof + the offset of
    each field in declaration order).  The mirrors are written by hand; p3d_abi_version() alone would not notice a field that
    was added to one side only, or a padding difference."""
    buf = (C.c_size_t * 64)()
    for which, cls in STRUCT_MIRRORS.items():
        n = L.p3d_struct_layout(which, buf, 64)
        if n <= 0:
            raise RuntimeError(f"p3d_struct_layout({which}) failed: {n}")
        mine = [C.sizeof(cls)] + [getattr(cls, name).offset for name, _ in cls._fields_]
        theirs = list(buf[:n])
        if mine != theirs:
            raise RuntimeError(f"{cls.__name__}: the ctypes mirror in _lib.py (size, offsets {mine}) does not match the struct "
                               f"{SO} was compiled with ({theirs}): include/panic3d_hip.h and _lib.py have diverged")


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "size out of supported range", -3: "workspace too small"}.get(rc, f"hipError {rc}")
        raise RuntimeError(f"{what} failed: {kind} (code {rc})")
