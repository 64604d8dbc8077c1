"""The C ABI from a host with no PyTorch in it: examples/render_c_abi.cpp is compiled against include/panic3d_hip.h and
libpanic3d_hip.so only.  CPU: it builds (hipcc cross-compiles).  GPU: it renders the same bytes as panic3d_amd.ops.render."""
import os, subprocess, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "panic3d-anime-reconstruction_amd")


def build_example(out, src="render_c_abi.cpp"):
    import panic3d_amd
    panic3d_amd.build()
    cmd = [panic3d_amd._build._hipcc(), "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "examples", src),
           "-I", os.path.join(ROOT, "include"), "-L", PKG, "-lpanic3d_hip", f"-Wl,-rpath,{PKG}", "-o", out]
    subprocess.check_call(cmd)
    return out


def test_example_builds_against_the_header_only(tmp_path):
    exe = build_example(str(tmp_path / "render_c_abi"))
    assert os.path.getsize(exe) > 0
    src = open(os.path.join(ROOT, "examples", "render_c_abi.cpp")).read()
    assert "torch" not in src.replace("PyTorch", "") and '#include "panic3d_hip.h"' in src
    exe2 = build_example(str(tmp_path / "synthesis_c_abi"), "synthesis_c_abi.cpp")
    assert os.path.getsize(exe2) > 0
    src2 = open(os.path.join(ROOT, "examples", "synthesis_c_abi.cpp")).read()
    assert "torch" not in src2.replace("PyTorch", "").replace("torgb", "") and '#include "panic3d_hip.h"' in src2


def test_struct_mirrors_match_the_library():
    """The four POD structs of include/panic3d_hip.h are restated by hand in _lib.py (ctypes); p3d_struct_layout reports the
    layout the library was compiled with — sizeof and every field offset in declaration order — and lib() refuses to load on any
    difference.  No GPU needed: host code of the library."""
    import ctypes as C
    import panic3d_amd
    L = panic3d_amd._lib.lib()  # (runs check_struct_layouts)
    buf = (C.c_size_t * 64)()
    for which, cls in panic3d_amd._lib.STRUCT_MIRRORS.items():
        n = L.p3d_struct_layout(which, buf, 64)
        assert n == 1 + len(cls._fields_), cls.__name__
        assert buf[0] == C.sizeof(cls)
        assert [buf[1 + i] for i in range(n - 1)] == [getattr(cls, f).offset for f, _ in cls._fields_]
    assert L.p3d_struct_layout(0, buf, 2) == -2 and L.p3d_struct_layout(7, buf, 64) == -2 and L.p3d_struct_layout(0, None, 64) == -1

    class Wrong(C.Structure):  # a mirror that lost a field is caught (what p3d_abi_version alone would not see)
        _fields_ = [f for f in panic3d_amd._lib.ConvArgs._fields_ if f[0] != "x_img"]
    saved = dict(panic3d_amd._lib.STRUCT_MIRRORS)
    try:
        panic3d_amd._lib.STRUCT_MIRRORS[3] = Wrong
        with pytest.raises(RuntimeError, match="does not match"):
            panic3d_amd._lib.check_struct_layouts(L)
    finally:
        panic3d_amd._lib.STRUCT_MIRRORS.clear()
        panic3d_amd._lib.STRUCT_MIRRORS.update(saved)


@pytest.mark.gpu
def test_cpp_host_renders_the_same_bytes(tmp_path, oracle):
    import torch
    import panic3d_amd as P
    import p3d_testing as T
    g = T.load_golden("render_32x32_16p16.npz")
    inp = T.golden_render_inputs(g)
    opts = P.ops.make_opts(inp["ro"], **inp["kw"])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    raw = [dev(x) for x in inp["raw_mlp"]]
    mlp = P.ops.prescale_mlp(*raw, inp["lr_mul"] / np.sqrt(32), inp["lr_mul"], inp["lr_mul"] / np.sqrt(64), inp["lr_mul"])
    N, _, _, H, W = inp["planes"].shape
    R = inp["rays_o"].shape[1]
    d = str(tmp_path)
    for name, a in (("planes", inp["planes"]), ("rays_o", inp["rays_o"]), ("rays_d", inp["rays_d"]), ("jitter", inp["jitter"]),
                    ("u", inp["u"]), ("w0", mlp[0].cpu().numpy()), ("b0", mlp[1].cpu().numpy()), ("w1", mlp[2].cpu().numpy()),
                    ("b1", mlp[3].cpu().numpy())):
        np.ascontiguousarray(a, dtype=np.float32).tofile(os.path.join(d, name + ".bin"))
    with open(os.path.join(d, "meta.txt"), "w") as f:
        f.write(f"{N} {H} {W} {R} 32 {opts.Sc} {opts.Sf} {opts.plane_mode} {opts.flags} " + " ".join(
            repr(float(np.float32(x))) for x in (opts.coord_scale, opts.ray_start, opts.ray_end, opts.depth_delta, opts.crop_limit, opts.cull_thresh)))
    exe = build_example(os.path.join(d, "render_c_abi"))
    out = subprocess.run([exe, d], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "gfx950" in out.stdout
    ref = P.ops.render(P.ops.planes_to_nhwc(dev(inp["planes"])), dev(inp["rays_o"]), dev(inp["rays_d"]), dev(inp["jitter"]), dev(inp["u"]),
                       mlp, opts, ray_tile_w=32)
    orc = oracle.render(inp["planes"], inp["rays_o"], inp["rays_d"], inp["jitter"], inp["u"],
                        oracle.prescale_mlp(*inp["raw_mlp"], lr_mul=inp["lr_mul"]), oracle.make_opts(inp["ro"], **inp["kw"]))
    for name, a, b in zip(("feat", "depth", "wsum", "xyz"), ref, orc):
        got = np.fromfile(os.path.join(d, name + ".bin"), dtype=np.float32).reshape(b.shape)
        assert np.array_equal(got, a.cpu().numpy()) and np.array_equal(got, b), name


@pytest.mark.gpu
@pytest.mark.parametrize("up", [1, 2])
def test_cpp_host_runs_a_synthesis_layer_and_torgb(tmp_path, up):
    """examples/synthesis_c_abi.cpp: p3d_conv_weights_to_f16x2 + p3d_modconv2d_ex_f32 (the struct entry) + p3d_torgb_weights_f32 +
    p3d_torgb_f32 from C++, on the bytes the Python host (ops.modulated_conv2d / ops.torgb, i.e. stylegan2.SynthesisLayer / ToRGBLayer)
    gets: identical outputs, and its own sizeof / offsetof of the four structs equal p3d_struct_layout."""
    import torch
    import panic3d_amd as P
    g = torch.Generator().manual_seed(70 + up)
    N, I, O, H, W, ORGB = 2, 64, 64, 32, 32, 96
    OH, OW = H * up, W * up
    x = torch.randn(N, I, H, W, generator=g)
    w = torch.randn(O, I, 3, 3, generator=g)
    styles = torch.randn(N, I, generator=g) * 0.5 + 1.0
    noise = torch.randn(OH, OW, generator=g) * 0.1
    bias = torch.randn(O, generator=g) * 0.2
    wrgb = torch.randn(ORGB, O, 1, 1, generator=g)
    srgb = (torch.randn(N, O, generator=g) * 0.5 + 1.0) / np.sqrt(O)
    brgb = torch.randn(ORGB, generator=g) * 0.2
    skip = torch.randn(N, ORGB, OH // 2, OW // 2, generator=g)
    filt = P.ops.setup_filter((1, 3, 3, 1))
    fir = P.ops.prepared_filter(filt.cuda(), torch.device("cuda"), 4.0, False).cpu()
    d = str(tmp_path)
    for name, a in (("x", x), ("w", w), ("styles", styles), ("noise", noise), ("bias", bias), ("fir", fir), ("wrgb", wrgb), ("srgb", srgb),
                    ("brgb", brgb), ("skip", skip)):
        a.contiguous().numpy().astype(np.float32).tofile(os.path.join(d, name + ".bin"))
    with open(os.path.join(d, "meta.txt"), "w") as f:
        f.write(f"{N} {I} {O} {H} {W} {up} {ORGB}")
    exe = build_example(os.path.join(d, "synthesis_c_abi"), "synthesis_c_abi.cpp")
    out = subprocess.run([exe, d], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "struct layouts agree" in out.stdout
    c = lambda t: t.cuda().contiguous()
    y = P.ops.modulated_conv2d(c(x), c(w), c(styles), noise=c(noise), up=up, padding=1, resample_filter=c(filt), demodulate=True, bias=c(bias),
                               act="lrelu", gain=float(np.sqrt(2)), weight_f16=P.ops.conv_weights_to_f16(c(w), split=True))
    img = P.ops.torgb(y, P.ops.torgb_weights(c(wrgb)), ORGB, c(srgb), bias=c(brgb), skip=c(skip), skip_filter=c(filt))
    got_y = np.fromfile(os.path.join(d, "y.bin"), dtype=np.float32).reshape(N, O, OH, OW)
    got_i = np.fromfile(os.path.join(d, "img.bin"), dtype=np.float32).reshape(N, ORGB, OH, OW)
    assert np.array_equal(got_y, y.cpu().numpy())
    assert np.array_equal(got_i, img.cpu().numpy())
    # and the layer is the reference's modulated_conv2d (networks_stylegan2.py:40-97) + bias_act, restated in plain torch fp32
    xs = c(x) * c(styles)[:, :, None, None]
    dco = ((c(w)[None] * c(styles)[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt()
    if up == 1:
        ref = torch.nn.functional.conv2d(xs, c(w), padding=1)
    else:
        ref = torch.nn.functional.conv_transpose2d(xs, c(w).transpose(0, 1), stride=2)  # flip_weight = (up == 1): networks_stylegan2.py:346
        f4 = c(fir)[None, None].repeat(O, 1, 1, 1)
        ref = torch.nn.functional.conv2d(torch.nn.functional.pad(ref, (1, 1, 1, 1)), f4, groups=O)
    ref = ref * dco[:, :, None, None] + c(noise)[None, None] + c(bias)[None, :, None, None]
    ref = torch.nn.functional.leaky_relu(ref, 0.2) * float(np.sqrt(2))
    assert float((y - ref).abs().max()) <= 3e-6 * np.sqrt(9 * I) * float(ref.abs().max())


@pytest.mark.gpu
def test_cpp_host_runs_conv1_with_torgb_riding_and_the_weight_image_layout(tmp_path):
    """examples/synthesis_c_abi.cpp `ride` (ABI 9): p3d_conv_weight_layout + p3d_conv_weights_to_f16x2_layout + p3d_act_to_image_f32 +
    p3d_modconv2d_ex_f32 with rgb_* (no fp32 activation) + p3d_torgb_combine_f32 from C++ on the bytes the Python host
    (ops.modulated_conv2d(..., rgb_weight=, want_y=False) + ops.torgb_combine, i.e. a super-resolution block's conv1 + ToRGB) gets:
    the identical image; a weight copy in the wrong image layout is refused."""
    import torch
    import panic3d_amd as P
    g = torch.Generator().manual_seed(91)
    N, I, O, H, W, R = 1, 32, 128, 256, 256, 3
    x = torch.randn(N, I, H, W, generator=g)
    w = torch.randn(O, I, 3, 3, generator=g) / np.sqrt(9 * I)
    styles = torch.randn(N, I, generator=g) * 0.5 + 1.0
    noise = torch.randn(H, W, generator=g) * 0.1
    bias = torch.randn(O, generator=g) * 0.2
    wrgb = torch.randn(R, O, generator=g)
    srgb = (torch.randn(N, O, generator=g) * 0.5 + 1.0) / np.sqrt(O)
    brgb = torch.randn(R, generator=g) * 0.2
    skip = torch.randn(N, R, H // 2, W // 2, generator=g)
    dcoef = ((w[None] * styles[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt()
    filt = P.ops.setup_filter((1, 3, 3, 1))
    fir = P.ops.prepared_filter(filt.cuda(), torch.device("cuda"), 4.0, False).cpu()
    d = str(tmp_path)
    for name, a in (("x", x), ("w", w), ("styles", styles), ("dcoef", dcoef), ("noise", noise), ("bias", bias), ("fir", fir), ("wrgb", wrgb),
                    ("srgb", srgb), ("brgb", brgb), ("skip", skip)):
        a.contiguous().numpy().astype(np.float32).tofile(os.path.join(d, name + ".bin"))
    with open(os.path.join(d, "meta.txt"), "w") as f:
        f.write(f"{N} {I} {O} {H} {W}")
    exe = build_example(os.path.join(d, "synthesis_c_abi"), "synthesis_c_abi.cpp")
    out = subprocess.run([exe, d, "ride"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ToRGB on its launch, weight layout 1" in out.stdout
    c = lambda t: t.cuda().contiguous()
    ops = P.ops
    assert ops.conv_fuses_torgb(N, I, O, H, W, R) and ops.conv_weight_layout(I, O, W, 1) == 1
    img = ops.act_to_image(c(x), c(styles))
    y, yi, part = ops.modulated_conv2d(img, c(w), None, noise=c(noise), padding=1, demodulate=True, bias=c(bias), act="lrelu", gain=float(np.sqrt(2)),
                                       dcoef=c(dcoef), weight_f16=ops.conv_weights_to_f16(c(w), split=True, layout=1), rgb_weight=c(wrgb),
                                       rgb_styles=c(srgb), want_y=False)
    ref = ops.torgb_combine(part, bias=c(brgb), skip=c(skip), skip_filter=c(filt))
    got = np.fromfile(os.path.join(d, "rgb.bin"), dtype=np.float32).reshape(N, R, H, W)
    assert y is None and np.array_equal(got, ref.cpu().numpy())
    # and it is the block's conv1 + ToRGB + skip connection in plain torch fp32 (networks_stylegan2.py:334-380, 476-478)
    xs = c(x) * c(styles)[:, :, None, None]
    t = torch.nn.functional.conv2d(xs, c(w), padding=1) * c(dcoef)[:, :, None, None] + c(noise)[None, None] + c(bias)[None, :, None, None]
    t = torch.nn.functional.leaky_relu(t, 0.2) * float(np.sqrt(2))
    rgbt = torch.einsum("ro,no,nohw->nrhw", c(wrgb), c(srgb), t) + c(brgb)[None, :, None, None] + ops.upsample2d(c(skip), c(filt))
    assert float((ref - rgbt).abs().max()) <= 2e-5 * float(rgbt.abs().max())
