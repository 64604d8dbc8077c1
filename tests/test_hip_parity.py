"""GPU: the HIP path (through the C ABI) against the CPU oracle — BIT-EXACT — and against the reference's golden outputs.

Both sides implement include/p3d_numerics.h, so every float, every searchsorted index and every merged depth must be
identical (np.array_equal), not merely close.  Agreement with the reference itself is the tolerance test of
tests/test_oracle_golden.py, repeated here on the HIP outputs.
"""
import numpy as np
import pytest
import torch

import p3d_testing as T

pytestmark = pytest.mark.gpu

TOL_FEAT, TOL_DEPTH, TOL_WEIGHT, TOL_XYZ = 1e-4, 2e-5, 3e-5, 1e-4  # same as tests/test_oracle_golden.py


@pytest.fixture(scope="module")
def hip():
    import panic3d_amd
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"
    panic3d_amd._lib.lib()  # must load: no fallback
    return panic3d_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def hip_opts(hip, ro, kw):
    return hip.ops.make_opts(ro, **kw)


def hip_mlp(hip, raw, lr_mul):
    w0, b0, w1, b1 = (dev(x) for x in raw)
    return hip.ops.prescale_mlp(w0, b0, w1, b1, lr_mul / np.sqrt(32), lr_mul, lr_mul / np.sqrt(64), lr_mul)


def run_hip_render(hip, inp, tile_w=0):
    opts = hip_opts(hip, inp["ro"], inp["kw"])
    mlp = hip_mlp(hip, inp["raw_mlp"], inp["lr_mul"])
    planes = hip.ops.planes_to_nhwc(dev(inp["planes"]))
    out = hip.ops.render(planes, dev(inp["rays_o"]), dev(inp["rays_d"]), dev(inp["jitter"]), dev(inp["u"]), mlp, opts,
                         ray_tile_w=tile_w, dumps=True,
                         ray_limits=None if inp.get("ray_limits") is None else tuple(dev(x) for x in inp["ray_limits"]))
    torch.cuda.synchronize()
    feat, depth, wsum, xyz, d = out
    return feat.cpu().numpy(), depth.cpu().numpy(), wsum.cpu().numpy(), xyz.cpu().numpy(), {k: v.cpu().numpy() for k, v in d.items()}


@pytest.mark.parametrize("name", T.RENDER_GOLDENS + T.RENDER_GOLDENS_AUTO)
@pytest.mark.parametrize("tiled", [0, 1])
def test_render_bit_exact_vs_oracle(hip, oracle, name, tiled):
    g = T.load_golden(name + ".npz")
    inp = T.golden_render_inputs(g)
    R = inp["rays_o"].shape[1]
    side = int(round(R ** 0.5))
    if tiled and (side * side != R or side % 8):
        pytest.skip("not an 8x4-tileable image")
    oo = oracle.make_opts(inp["ro"], **inp["kw"])
    om = oracle.prescale_mlp(*inp["raw_mlp"], lr_mul=inp["lr_mul"])
    of, od, ow, ox, odm = oracle.render(inp["planes"], inp["rays_o"], inp["rays_d"], inp["jitter"], inp["u"], om, oo, dumps=True,
                                        ray_limits=inp["ray_limits"])
    hf, hd, hw, hx, hdm = run_hip_render(hip, inp, tile_w=side if tiled else 0)
    # host-side parameter preparation is part of the contract
    hm = hip_mlp(hip, inp["raw_mlp"], inp["lr_mul"])
    for a, b in zip(hm, om):
        assert np.array_equal(a.cpu().numpy(), b)
    Sc, Sf = oo.Sc, oo.Sf
    assert np.array_equal(hdm["depths_coarse"], odm["depths_coarse"])
    if Sf > 0:
        assert np.array_equal(hdm["sigma_coarse"], odm["sigma_coarse"])
        assert np.array_equal(hdm["weights_coarse"], odm["weights_coarse"])
        assert np.array_equal(hdm["depths_fine"], odm["depths_fine"])
        assert np.array_equal(hdm["inds"], odm["inds"])  # "ray hit indices"
        all_d = np.concatenate([odm["depths_coarse"], odm["depths_fine"]], axis=1)
        all_s = np.concatenate([odm["sigma_coarse"], odm["sigma_fine"]], axis=1)
        assert np.array_equal(hdm["depths_sorted"], np.take_along_axis(all_d, odm["perm"], axis=1))
        assert np.array_equal(hdm["sigma_sorted"], np.take_along_axis(all_s, odm["perm"], axis=1))
        # the stand-alone sort operator reproduces the oracle's permutation exactly
        perm = hip.ops.unify_perm(dev(odm["depths_coarse"]), dev(odm["depths_fine"])).cpu().numpy()
        assert np.array_equal(perm, odm["perm"])
    assert np.array_equal(hdm["tminmax"], odm["tminmax"])
    assert np.array_equal(hdm["depth_unclamped"], odm["depth_unclamped"], equal_nan=True)
    assert np.array_equal(hf, of)
    assert np.array_equal(hd, od)
    assert np.array_equal(hw, ow)
    assert np.array_equal(hx, ox)
    # and the HIP outputs agree with the REFERENCE's own outputs within the fp32 tolerance
    assert np.abs(hf - g["feat"]).max() <= TOL_FEAT
    assert np.abs(hd - g["depth"]).max() <= TOL_DEPTH
    assert np.abs(hw - g["wsum"]).max() <= TOL_WEIGHT
    assert np.abs(hx - g["xyz"]).max() <= TOL_XYZ
    if "inds" in g:
        assert int((hdm["inds"] != g["inds"]).sum()) == 0


@pytest.mark.parametrize("name", T.RENDER_GOLDENS + T.RENDER_GOLDENS_AUTO)
@pytest.mark.parametrize("early_out", [True, False])
@pytest.mark.parametrize("pair", ["quad", "pair", False])
def test_render_production_kernel_bit_exact(hip, oracle, name, early_out, pair):
    """The kernel variants that ship (no dumps; with and without the exact early-outs; the small-launch kernels — 8 rays x 4 samples
    per wave ("quad") and 16 rays x 2 samples ("pair"), forced — and the 32-rays-per-wave kernel) against the
    oracle: outputs only."""
    g = T.load_golden(name + ".npz")
    inp = T.golden_render_inputs(g)
    R = inp["rays_o"].shape[1]
    side = int(round(R ** 0.5))
    ref = oracle.render(inp["planes"], inp["rays_o"], inp["rays_d"], inp["jitter"], inp["u"],
                        oracle.prescale_mlp(*inp["raw_mlp"], lr_mul=inp["lr_mul"]), oracle.make_opts(inp["ro"], **inp["kw"]),
                        ray_limits=inp["ray_limits"])
    opts = hip.ops.make_opts(inp["ro"], early_out=early_out, small_launch_kernel=pair, **inp["kw"])
    planes = hip.ops.planes_to_nhwc(dev(inp["planes"]))
    st = {}
    out = hip.ops.render(planes, dev(inp["rays_o"]), dev(inp["rays_d"]), dev(inp["jitter"]), dev(inp["u"]),
                         hip_mlp(hip, inp["raw_mlp"], inp["lr_mul"]), opts,
                         ray_tile_w=side if (side * side == R and side % 8 == 0) else 0, stats=st,
                         ray_limits=None if inp["ray_limits"] is None else tuple(dev(x) for x in inp["ray_limits"]))
    for name_, a, b in zip(("feat", "depth", "wsum", "xyz"), out, ref):
        assert np.array_equal(a.cpu().numpy(), b), name_
    assert st["small_launch_kernel"] == bool(pair) and st["small_launch_kind"] == (pair or None)
    assert 0 < st["decode_steps"] <= st["decode_steps_full"]
    if not early_out:
        assert st["decode_steps"] == st["decode_steps_full"]


def test_importance_renderer_auto_ray_limits(hip, oracle):
    """ImportanceRenderer.forward with rendering_options ray_start = ray_end = 'auto' (renderer.py:165-171): the host restates
    get_ray_limits_box + the patching of the rays that miss the box (bit-identical to the reference's limits in the fixture,
    21 of 400 rays miss), the kernel math_utils.linspace + the per-ray depth_delta; exact mode equals the oracle bit for bit
    and the reference within the fp32 tolerances; the renderer's default (tolerance) mode stays within 2e-5 of it."""
    g = T.load_golden("render_auto_limits.npz")
    inp = T.golden_render_inputs(g)
    assert inp["ro"]["ray_start"] == "auto"
    rs, re = hip.cameras.patch_ray_limits(*hip.cameras.ray_limits_box(dev(inp["rays_o"]), dev(inp["rays_d"]), inp["ro"]["box_warp"]))
    assert np.array_equal(rs.reshape(1, -1).cpu().numpy(), g["ray_start"]) and np.array_equal(re.reshape(1, -1).cpu().numpy(), g["ray_end"])
    assert int((g["ray_end"] <= g["ray_start"]).sum()) == 0 and float(g["ray_start"].min()) > 0  # patched limits
    ref = oracle.render(inp["planes"], inp["rays_o"], inp["rays_d"], inp["jitter"], inp["u"],
                        oracle.prescale_mlp(*inp["raw_mlp"], lr_mul=inp["lr_mul"]), oracle.make_opts(inp["ro"], **inp["kw"]),
                        ray_limits=inp["ray_limits"])
    rend = hip.ImportanceRenderer(use_triplane=bool(inp["ro"]["use_triplane"]))

    class FC:
        def __init__(self, w, b, i):
            self.weight, self.bias, self.weight_gain, self.bias_gain = w, b, inp["lr_mul"] / np.sqrt(i), inp["lr_mul"]

    class Dec:
        force_sigmoid = bool(inp["kw"]["force_sigmoid"])
    raw = [dev(x) for x in inp["raw_mlp"]]
    Dec.net = [FC(raw[0], raw[1], 32), None, FC(raw[2], raw[3], 64)]
    dec = Dec()
    kw = {k: v for k, v in inp["kw"].items() if k != "force_sigmoid"}
    common = dict(jitter=dev(inp["jitter"]), u=dev(inp["u"]), **kw)
    exact = rend(dev(inp["planes"]), dec, dev(inp["rays_o"]), dev(inp["rays_d"]), inp["ro"], exact=True, **common)
    for name, a, b in zip(("feat", "depth", "wsum", "xyz"), exact, ref):
        assert np.array_equal(a.cpu().numpy(), b), name
    for a, key, tol in zip(exact, ("feat", "depth", "wsum", "xyz"), (TOL_FEAT, TOL_DEPTH, TOL_WEIGHT, TOL_XYZ)):
        assert np.abs(a.cpu().numpy() - g[key]).max() <= tol, key
    fast = rend(dev(inp["planes"]), dec, dev(inp["rays_o"]), dev(inp["rays_d"]), inp["ro"], **common)
    for a, b in zip(fast, exact):
        assert float((a - b).abs().max()) <= 2e-5
    # without injected draws the host draws in the reference's memory order ([Sc,N,R,1] viewed as [N,R,Sc,1])
    torch.manual_seed(int(inp["meta"]["seed"]) + 2)
    drawn = rend(dev(inp["planes"]), dec, dev(inp["rays_o"]), dev(inp["rays_d"]), inp["ro"], exact=True, **kw)
    assert all(torch.isfinite(t).all() for t in drawn)


@pytest.mark.parametrize("ut", [0, 1])
def test_decode_points(hip, oracle, ut):
    g = T.load_golden(f"decode_points_ut{ut}.npz")
    seed = int(g["meta_seed"])
    planes = T.make_planes(seed, 2, 64, 96)
    pts = T.make_points(seed + 2, 2, 4096, extent=0.45)
    raw = T.make_decoder_params(seed + 1)
    om = oracle.prescale_mlp(*raw)
    osig, orgb = oracle.decode(planes, pts, om, 0.7, plane_mode=ut, flags=0)
    ro = dict(T.RENDERING_KWARGS, use_triplane=ut)
    opts = hip.ops.make_opts(ro, force_sigmoid=False)
    pl = hip.ops.planes_to_nhwc(dev(planes))
    assert np.array_equal(pl.cpu().numpy(), np.ascontiguousarray(planes.transpose(0, 1, 3, 4, 2)))
    hs, hr = hip.ops.triplane_decode(pl, dev(pts), hip_mlp(hip, raw, 1.0), opts)
    assert np.array_equal(hs.cpu().numpy(), osig)
    assert np.array_equal(hr.cpu().numpy(), orgb)
    assert np.abs(hs.cpu().numpy() - g["sigma"]).max() <= 1e-5  # vs the reference
    assert np.abs(hr.cpu().numpy() - g["rgb"]).max() <= 2e-6
    # density-only entry point (get_eg3d_volume) and a ragged point count (M % 32 != 0)
    hs2, none = hip.ops.triplane_decode(pl, dev(pts[:, :1000]), hip_mlp(hip, raw, 1.0), opts, density_only=True)
    assert none is None and np.array_equal(hs2.cpu().numpy(), osig[:, :1000])
    # masks through the flags (crop + cull), as ImportanceRenderer.forward applies them
    ok = dict(triplane_crop=0.1, cull_clouds=0.5)
    oo = oracle.make_opts(ro, **ok)
    osig3, _ = oracle.decode(planes, pts, om, 0.7, plane_mode=ut, flags=oo.flags, crop_limit=oo.crop_limit, cull_thresh=oo.cull_thresh)
    hs3, _ = hip.ops.triplane_decode(pl, dev(pts), hip_mlp(hip, raw, 1.0), hip.ops.make_opts(ro, **ok))
    assert np.array_equal(hs3.cpu().numpy(), osig3) and (osig3 == -1000).any()


@pytest.mark.parametrize("wb", [0, 1])
def test_composite_op(hip, oracle, wb):
    g = T.load_golden(f"marcher_wb{wb}.npz")
    orgb, odepth, ow = oracle.composite(g["colors"], g["sigma"], g["depths"], white_back=bool(wb))
    hrgb, hdepth, hw = hip.ops.composite(dev(g["colors"]), dev(g["sigma"]), dev(g["depths"]), white_back=bool(wb))
    assert np.array_equal(hrgb.cpu().numpy().reshape(orgb.shape), orgb)
    assert np.array_equal(hdepth.cpu().numpy().reshape(odepth.shape), odepth)
    assert np.array_equal(hw.cpu().numpy().reshape(ow.shape), ow)
    assert np.abs(hrgb.cpu().numpy() - g["rgb"]).max() <= TOL_FEAT


def test_importance_op(hip, oracle):
    g = T.load_golden("importance.npz")
    ofine, oinds = oracle.importance(g["depths"], g["weights"], g["u"])
    hfine, hinds = hip.ops.importance(dev(g["depths"]), dev(g["weights"]), dev(g["u"]), return_inds=True)
    assert np.array_equal(hinds.cpu().numpy(), oinds) and np.array_equal(hinds.cpu().numpy(), g["inds"])
    assert np.array_equal(hfine.cpu().numpy().reshape(ofine.shape), ofine)


@pytest.mark.parametrize("S", [48, 96, 17])
def test_stratified_op(hip, S):
    g = T.load_golden(f"stratified_{S}.npz")
    out = hip.ops.sample_stratified(float(g["start"]), float(g["end"]), S, dev(g["jitter"]))
    assert np.array_equal(out.cpu().numpy().reshape(g["depths"].shape), g["depths"])  # bit-exact vs the REFERENCE


def test_cpu_tensors_raise(hip):
    with pytest.raises(RuntimeError):
        hip.ops.planes_to_nhwc(torch.zeros(1, 3, 32, 8, 8))


def test_error_behaviour(hip):
    """Argument / range errors surface as RuntimeError (the reference's TORCH_CHECK convention, bias_act.cpp:39-55)."""
    ro = dict(T.RENDERING_KWARGS)
    planes = torch.zeros(1, 3, 8, 8, 32, device="cuda")
    o = torch.zeros(1, 64, 3, device="cuda")
    d = torch.ones(1, 64, 3, device="cuda")
    mlp = (torch.zeros(64, 32, device="cuda"), torch.zeros(64, device="cuda"), torch.zeros(33, 64, device="cuda"), torch.zeros(33, device="cuda"))
    opts = hip.ops.make_opts(ro)
    good_j, good_u = torch.rand(1, 64, 48, device="cuda"), torch.rand(64, 48, device="cuda")
    with pytest.raises(RuntimeError):  # wrong jitter size
        hip.ops.render(planes, o, d, torch.rand(1, 64, 47, device="cuda"), good_u, mlp, opts)
    with pytest.raises(RuntimeError):  # Sc below the supported range
        hip.ops.render(planes, o, d, torch.rand(1, 64, 3, device="cuda"), good_u, mlp, hip.ops.make_opts(dict(ro, depth_resolution=3)))
    with pytest.raises(RuntimeError):  # Sf above the supported range
        hip.ops.render(planes, o, d, good_j, torch.rand(64, 500, device="cuda"), mlp, hip.ops.make_opts(dict(ro, depth_resolution_importance=500)))
    with pytest.raises(RuntimeError):  # decoder of the wrong shape
        hip.ops.render(planes, o, d, good_j, good_u, (torch.zeros(64, 16, device="cuda"),) + mlp[1:], opts)
    with pytest.raises(RuntimeError):  # fp64 input
        hip.ops.planes_to_nhwc(torch.zeros(1, 3, 32, 8, 8, device="cuda", dtype=torch.float64))
    with pytest.raises(ValueError):  # 'auto' limits come in pairs (renderer.py:165)
        hip.ops.make_opts(dict(ro, ray_start="auto"))
    with pytest.raises(NotImplementedError):  # the one combination of sampling options that is not built
        hip.ops.make_opts(dict(ro, ray_start="auto", ray_end="auto", disparity_space_sampling=True))
    with pytest.raises(RuntimeError):  # 'auto' options without the per-ray limits
        hip.ops.render(planes, o, d, good_j, good_u, mlp, hip.ops.make_opts(dict(ro, ray_start="auto", ray_end="auto")))
    with pytest.raises(RuntimeError):  # per-ray limits of the wrong size
        hip.ops.render(planes, o, d, good_j, good_u, mlp, hip.ops.make_opts(dict(ro, ray_start="auto", ray_end="auto")),
                       ray_limits=(torch.zeros(3, device="cuda"), torch.zeros(3, device="cuda")))
    # a valid call on the same inputs still works afterwards (no sticky error state), incl. the single-pass branch
    out = hip.ops.render(planes, o, d, good_j, good_u, mlp, opts)
    out0 = hip.ops.render(planes, o, d, good_j, None, mlp, hip.ops.make_opts(dict(ro, depth_resolution_importance=0)))
    torch.cuda.synchronize()
    assert torch.isfinite(out[0]).all() and torch.isfinite(out0[0]).all()
    # large sample counts take the generic (LDS-sorted) path: 160 + 160
    big = hip.ops.make_opts(dict(ro, depth_resolution=160, depth_resolution_importance=160))
    outb = hip.ops.render(planes, o, d, torch.rand(1, 64, 160, device="cuda"), torch.rand(64, 160, device="cuda"), mlp, big)
    torch.cuda.synchronize()
    assert torch.isfinite(outb[0]).all()


def test_generic_sort_path_bit_exact(hip, oracle):
    """Sf > 128 uses the LDS insertion-sort path (k_render<0>): check it against the oracle too."""
    g = T.load_golden("render_32x32_16p16.npz")
    inp = T.golden_render_inputs(g)
    Sc, Sf = 20, 136
    ro = dict(inp["ro"], depth_resolution=Sc, depth_resolution_importance=Sf)
    R = 96
    jit, u = T.make_random_draws(5, 2, R, Sc, Sf)
    o_, d_ = inp["rays_o"][:, :R].copy(), inp["rays_d"][:, :R].copy()
    ref = oracle.render(inp["planes"], o_, d_, jit, u, oracle.prescale_mlp(*inp["raw_mlp"], lr_mul=inp["lr_mul"]), oracle.make_opts(ro, **inp["kw"]))
    out = hip.ops.render(hip.ops.planes_to_nhwc(dev(inp["planes"])), dev(o_), dev(d_), dev(jit), dev(u),
                         hip_mlp(hip, inp["raw_mlp"], inp["lr_mul"]), hip.ops.make_opts(ro, **inp["kw"]))
    for a, b in zip(out, ref):
        assert np.array_equal(a.cpu().numpy(), b)


def psnr(a, b, peak=1.0):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 200.0 if mse == 0 else 10.0 * np.log10(peak * peak / mse)


@pytest.mark.parametrize("name", T.RENDER_GOLDENS)
def test_psnr_vs_reference_render(hip, name):
    """BASELINE.json's quality half ("PSNR vs ref"): image_raw = first 3 feature channels mapped to [0,1]
    (training/triplane.py:223, 0.5x+0.5) of the HIP render against the REFERENCE's own render of the same planes / rays /
    draws.  (The README's 16.91 dB is PSNR against ground-truth renders and needs the checkpoint + AnimeRecon data.)"""
    g = T.load_golden(name + ".npz")
    inp = T.golden_render_inputs(g)
    opts = hip.ops.make_opts(inp["ro"], **inp["kw"])
    out = hip.ops.render(hip.ops.planes_to_nhwc(dev(inp["planes"])), dev(inp["rays_o"]), dev(inp["rays_d"]), dev(inp["jitter"]),
                         dev(inp["u"]), hip_mlp(hip, inp["raw_mlp"], inp["lr_mul"]), opts)
    img = out[0][..., :3].cpu().numpy() * 0.5 + 0.5
    ref = g["feat"][..., :3] * 0.5 + 0.5
    assert psnr(img, ref) > 80.0, psnr(img, ref)


def test_renderer_plane_cache_is_not_fooled_by_reused_storage(hip):
    """ImportanceRenderer caches the channels-last copy of the planes per tensor OBJECT; a new planes tensor that the caching
    allocator places at the same address must not hit the cache."""
    r = hip.ImportanceRenderer(use_triplane=True)

    class FC:
        def __init__(self, o, i, seed):
            g = torch.Generator().manual_seed(seed)
            self.weight, self.bias = torch.randn(o, i, generator=g).cuda(), torch.randn(o, generator=g).cuda()
            self.weight_gain, self.bias_gain = 1 / np.sqrt(i), 1

    class Dec:
        force_sigmoid = True
        net = [FC(64, 32, 1), None, FC(33, 64, 2)]

    pts = torch.rand(1, 1000, 3).cuda() * 0.5 - 0.25
    outs = []
    for seed in (1, 2):
        planes = torch.randn(1, 3, 32, 64, 64, generator=torch.Generator().manual_seed(seed)).cuda()
        outs.append((r.run_model(planes, Dec(), pts, None, T.RENDERING_KWARGS)["sigma"].clone(), planes.data_ptr()))
        same = r.run_model(planes, Dec(), pts, None, T.RENDERING_KWARGS)["sigma"]  # same object again: cache hit, same result
        assert torch.equal(same, outs[-1][0])
        del planes
    assert not torch.equal(outs[0][0], outs[1][0])


def test_shared_planes_many_views_one_launch(hip, oracle):
    """Many views of ONE subject in one launch (planes [1,...] shared by all ray batches, or planes.expand(N,...) through the
    renderer class) == the same views rendered one by one, up to the depth clamp (global min/max is per CALL, ray_marcher.py:50)."""
    g = T.load_golden("render_32x32_16p16.npz")
    inp = T.golden_render_inputs(g)
    planes1 = inp["planes"][:1]
    opts = hip.ops.make_opts(inp["ro"], **inp["kw"])
    mlp = hip_mlp(hip, inp["raw_mlp"], inp["lr_mul"])
    nhwc = hip.ops.planes_to_nhwc(dev(planes1))
    o, d, jit, u = dev(inp["rays_o"]), dev(inp["rays_d"]), dev(inp["jitter"]), dev(inp["u"])
    both = hip.ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=32)
    ref = oracle.render(np.concatenate([planes1, planes1]), inp["rays_o"], inp["rays_d"], inp["jitter"], inp["u"],
                        oracle.prescale_mlp(*inp["raw_mlp"], lr_mul=inp["lr_mul"]), oracle.make_opts(inp["ro"], **inp["kw"]))
    for a, b in zip(both, ref):
        assert np.array_equal(a.cpu().numpy(), b)
    # through the class with planes.expand (what training/triplane.py would pass for several views of one subject)
    class FC:
        def __init__(self, w, b, i):
            self.weight, self.bias, self.weight_gain, self.bias_gain = w, b, inp["lr_mul"] / np.sqrt(i), inp["lr_mul"]
    class Dec:
        force_sigmoid = bool(inp["kw"]["force_sigmoid"])
    raw = [dev(x) for x in inp["raw_mlp"]]
    Dec.net = [FC(raw[0], raw[1], 32), None, FC(raw[2], raw[3], 64)]
    r = hip.ImportanceRenderer(use_triplane=bool(inp["ro"]["use_triplane"]))
    kw = {k: v for k, v in inp["kw"].items() if k != "force_sigmoid"}
    out = r(dev(planes1).expand(2, -1, -1, -1, -1), Dec(), o, d, inp["ro"], jitter=jit, u=u, exact=True, **kw)
    for a, b in zip(out, ref):
        assert np.array_equal(a.cpu().numpy(), b)


def _random_config(seed):
    rng = np.random.default_rng(seed)
    N = int(rng.integers(1, 4))
    H, W = int(rng.integers(8, 97)), int(rng.integers(8, 97))  # non-square, odd sizes
    side = int(rng.choice([0, 8, 16, 24]))  # 0: a ragged ray list; else an image of side x (4k) rays (tiled lane order)
    R = int(rng.integers(1, 150)) if side == 0 else side * int(rng.choice([4, 8, 12]))
    Sc = int(rng.integers(4, 71))
    Sf = int(rng.choice([0, int(rng.integers(1, 71)), int(rng.integers(129, 150))]))
    rs = float(rng.uniform(0.3, 0.7)); re = rs + float(rng.uniform(0.5, 1.2))
    ro = dict(T.RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf, ray_start=rs, ray_end=re,
              box_warp=float(rng.choice([0.7, 1.0, 2.0])), white_back=bool(rng.integers(0, 2)), use_triplane=int(rng.integers(0, 2)))
    kw = dict(triplane_crop=[None, 0.1, 0.05][int(rng.integers(0, 3))], force_sigmoid=bool(rng.integers(0, 2)))
    m = int(rng.integers(0, 3))
    kw["cull_clouds"] = 0.5 if m == 1 else None
    kw["binarize_clouds"] = 0.4 if m == 2 else None
    planes = T.make_planes(seed, N, H, W, scale=float(rng.uniform(0.5, 4.0)), smooth=int(rng.choice([0, 4, 8])))
    raw = T.make_decoder_params(seed + 1, float(rng.choice([1.0, 0.5])), float(rng.choice([1.0, 10.0, 30.0])))
    # cameras on a sphere of radius ~1 looking at the box, plus a few rays that miss it entirely
    o = rng.standard_normal((N, R, 3)); o /= np.linalg.norm(o, axis=-1, keepdims=True)
    tgt = rng.uniform(-0.3, 0.3, (N, R, 3)) * ro["box_warp"]
    d = tgt - o; d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d[:, ::7] = -d[:, ::7]
    jit, u = T.make_random_draws(seed + 2, N, R, Sc, Sf)
    return dict(ro=ro, kw=kw, planes=planes, raw=raw, lr_mul=1.0, o=o.astype(np.float32), d=d.astype(np.float32), jit=jit, u=u,
                tile_w=(side if side else 0), early_out=bool(rng.integers(0, 2)))


@pytest.mark.parametrize("branch", ["auto_limits", "disparity", "plain"])
@pytest.mark.parametrize("fast", [False, True])
def test_96p96_large_launch_kernel_honours_ray_limits_and_disparity(hip, oracle, branch, fast):
    """the production 96+96 kernel of LARGE launches keeps no coarse-depth column (TCG) and recomputes a
    stratified depth from the fixed ray_start / ray_end spacing; with per-ray limits (ray_start = 'auto', renderer.py:165-171,
    317-319) or disparity spacing (:309-316) it must therefore not be launched — until round 4 it was, silently.  The
    32-rays-per-wave kernel forced on a small image (what a > 512-tile launch gets), early-outs on, no dumps: exact mode bit for
    bit against the oracle, tolerance mode within its stated bound."""
    seed, res, Sc, Sf = 4100, 16, 96, 96
    ro = dict(T.RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf)
    limits = None
    label = hip.cameras.camera_label(10.0, 35.0, 1.0, 30.0)
    o, d = hip.cameras.rays_from_label(label[None], res)
    if branch == "auto_limits":
        ro["ray_start"] = ro["ray_end"] = "auto"
        rs, re = hip.cameras.patch_ray_limits(*hip.cameras.ray_limits_box(o.cuda(), d.cuda(), ro["box_warp"]))
        limits = (rs.reshape(1, -1).cpu().numpy(), re.reshape(1, -1).cpu().numpy())
    elif branch == "disparity":
        ro["disparity_space_sampling"] = True
    planes = T.make_planes(seed, 1, 64, 64, scale=4.0, smooth=8)
    raw = T.make_decoder_params(seed + 1, 1.0, 30.0)
    jit, u = T.make_random_draws(seed + 2, 1, res * res, Sc, Sf, auto_limits=branch == "auto_limits")
    kw = dict(triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True)
    ref = oracle.render(planes, o.numpy(), d.numpy(), jit, u, oracle.prescale_mlp(*raw), oracle.make_opts(ro, **kw), ray_limits=limits)
    st = {}
    out = hip.ops.render(hip.ops.planes_to_nhwc(dev(planes)), o.cuda(), d.cuda(), dev(jit), dev(u), hip_mlp(hip, raw, 1.0),
                         hip.ops.make_opts(ro, small_launch_kernel=False, early_out=True, fast_color=fast, **kw), ray_tile_w=res,
                         stats=st, ray_limits=None if limits is None else tuple(dev(x) for x in limits))
    assert st["small_launch_kernel"] is False
    for name, a, b, tol in zip(("feat", "depth", "wsum", "xyz"), out, ref, (1e-5, 2e-5, 1e-5, 1e-5)):
        a = a.cpu().numpy()
        if fast:
            assert float(np.abs(a - b).max()) <= tol, (name, branch)
        else:
            assert np.array_equal(a, b), (name, branch, float(np.abs(a - b).max()))
    assert float(ref[2].mean()) > 0.05  # the scene has surfaces: the depths matter


@pytest.mark.parametrize("pair", ["quad", "pair", False])
@pytest.mark.parametrize("seed", range(100, 124))
def test_render_random_configs_bit_exact(hip, oracle, seed, pair):
    """Randomised sweep over what the golden fixtures do not enumerate: N 1-3, non-square planes of odd sizes, ragged ray
    counts (incl. fewer than one wavefront), 4 <= Sc <= 70, Sf in {0, 1..70 (sorting network), 129..149 (LDS sort)}, every
    mask mode, both plane conventions, white/black background, early-outs on/off, rays that miss the volume, and both render
    kernels (the small-launch kernel these sizes would get, and the 32-rays-per-wave kernel forced).  Bit-exact."""
    c = _random_config(seed)
    mlp = hip_mlp(hip, c["raw"], c["lr_mul"])
    # pair: small launches go to k_render_pair (16 rays x 2 samples per wave); not pair: the 32-rays-per-wave kernel
    opts = hip.ops.make_opts(c["ro"], early_out=c["early_out"], small_launch_kernel=pair, **c["kw"])
    out = hip.ops.render(hip.ops.planes_to_nhwc(dev(c["planes"])), dev(c["o"]), dev(c["d"]), dev(c["jit"]), dev(c["u"]), mlp, opts,
                         ray_tile_w=c["tile_w"])
    ref = oracle.render(c["planes"], c["o"], c["d"], c["jit"], c["u"], oracle.prescale_mlp(*c["raw"], lr_mul=c["lr_mul"]),
                        oracle.make_opts(c["ro"], **c["kw"]))
    for name, a, b in zip(("feat", "depth", "wsum", "xyz"), out, ref):
        a = a.cpu().numpy()
        assert np.array_equal(a, b, equal_nan=True), (name, seed, float(np.nanmax(np.abs(a - b))))


def test_sigma2density_bit_exact(hip, oracle):
    """get_eg3d_volume's activation + crop / cull masks in one pass (p3d_sigma2density_f32) against the oracle, bit for bit,
    incl. the quirk that the cull mask is evaluated on the densities; and against the torch formulation to 1e-6."""
    rng = np.random.default_rng(3)
    sigma = (rng.standard_normal(100003) * 8).astype(np.float32)
    sigma[:5] = [-1000.0, 1000.0, 0.0, 1.0, -0.0]
    crop = rng.integers(0, 2, sigma.size).astype(np.uint8)
    for cm, cull in ((None, None), (crop, None), (None, 0.5), (crop, 0.37)):
        got = hip.ops.sigma2density(dev(sigma), None if cm is None else dev(cm).bool(), cull).cpu().numpy()
        assert np.array_equal(got, oracle.sigma2density(sigma, cm, cull))
    plain = hip.ops.sigma2density(dev(sigma)).cpu()
    ref = 1 - torch.exp(-torch.nn.functional.softplus(torch.from_numpy(sigma) - 1))
    assert (plain - ref).abs().max() < 1e-6


# ---- P3D_FLAG_FAST_COLOR: tolerance mode of the final pass (include/panic3d_hip.h) ---------------------------------------
def _psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return np.inf if mse == 0 else 10 * np.log10(1.0 / mse)


# the stated bound of the tolerance mode against the exact contract, every ray, every output
FAST_MAX = dict(feat=1e-5, depth=2e-5, wsum=1e-5, xyz=1e-5)


@pytest.mark.parametrize("name", [n for n in T.RENDER_GOLDENS if n != "render_c1_64x64_s32"])
def test_fast_color_keeps_the_coarse_pass_exact_and_the_outputs_close(hip, oracle, name):
    """The opt-in tolerance mode may only touch the FINAL pass: everything the importance resampling produces — coarse
    sigma / weights, the inverse-CDF bin indices ("ray hit indices"), the fine depths and the merged depth order — stays
    bit-exact vs the oracle; feat / depth / wsum / xyz of EVERY ray stay within FAST_MAX of it (round 2 allowed 0.1 % of the rays
    to flip a cull mask; the exact mask guard removed those), and the PSNR against the REFERENCE's own render moves by less
    than 0.01 dB."""
    g = T.load_golden(name + ".npz")
    inp = T.golden_render_inputs(g)
    oo = oracle.make_opts(inp["ro"], **inp["kw"])
    om = oracle.prescale_mlp(*inp["raw_mlp"], lr_mul=inp["lr_mul"])
    ref = oracle.render(inp["planes"], inp["rays_o"], inp["rays_d"], inp["jitter"], inp["u"], om, oo, dumps=True)
    odm = ref[4]
    mlp = hip_mlp(hip, inp["raw_mlp"], inp["lr_mul"])
    planes = hip.ops.planes_to_nhwc(dev(inp["planes"]))
    args = (planes, dev(inp["rays_o"]), dev(inp["rays_d"]), dev(inp["jitter"]), dev(inp["u"]), mlp)
    fast = hip.ops.make_opts(inp["ro"], fast_color=True, **inp["kw"])
    out = hip.ops.render(*args, fast, dumps=True)
    hdm = {k: v.cpu().numpy() for k, v in out[4].items()}
    for k in ("depths_coarse", "sigma_coarse", "weights_coarse", "depths_fine", "inds"):
        assert np.array_equal(hdm[k], odm[k]), k
    all_d = np.concatenate([odm["depths_coarse"], odm["depths_fine"]], axis=1)
    assert np.array_equal(hdm["depths_sorted"], np.take_along_axis(all_d, odm["perm"], axis=1))
    assert np.array_equal(hdm["tminmax"], odm["tminmax"])
    # production launches (no dumps, early-outs on) against the oracle: the 32-rays-per-wave kernel (small_launch_kernel=False —
    # these fixtures are small launches) and the small-launch kernel (16 rays x 2 samples), which has its own tolerance variant
    prod = hip.ops.render(*args, hip.ops.make_opts(inp["ro"], fast_color=True, small_launch_kernel=False, **inp["kw"]))
    st = {}
    prod_pair = hip.ops.render(*args, hip.ops.make_opts(inp["ro"], fast_color=True, **inp["kw"]), stats=st)
    assert st["small_launch_kernel"]
    exact_pair = hip.ops.render(*args, hip.ops.make_opts(inp["ro"], **inp["kw"]))
    assert not all(torch.equal(a, b) for a, b in zip(prod_pair, exact_pair)) or float(exact_pair[2].max()) == 0.0  # it really is another decoder
    R = inp["rays_o"].shape[0] * inp["rays_o"].shape[1]
    for nm, a, b in zip(("feat", "depth", "wsum", "xyz"), prod_pair, ref[:4]):
        err = np.abs(a.cpu().numpy() - b).reshape(R, -1).max(axis=1)
        print(f"{name} fast-vs-oracle, small-launch kernel {nm}: max {err.max():.2e} median {np.median(err):.2e}")
        assert np.median(err) <= 2e-6 and err.max() <= FAST_MAX[nm], (nm, float(err.max()), int((err > FAST_MAX[nm]).sum()))
    # HARD bound (round 3): the exact mask guard (p3d_decode.hpp, P3D_FAST_MASK_BAND) makes the tolerance mode take the same
    # crop / cull decisions as the exact contract, so what is left is arithmetic round-off (two-term f16 products, hardware
    # exp2 / log2 / rcp, ray termination at Td < 2e-6): EVERY ray within FAST_MAX of the oracle, no allowance for flipped rays.
    for nm, a, b in zip(("feat", "depth", "wsum", "xyz"), prod, ref[:4]):
        a = a.cpu().numpy()
        err = np.abs(a - b).reshape(R, -1).max(axis=1)
        print(f"{name} fast-vs-oracle {nm}: max {err.max():.2e} median {np.median(err):.2e}")
        assert np.median(err) <= 2e-6, (nm, float(np.median(err)))
        assert err.max() <= FAST_MAX[nm], (nm, float(err.max()), int((err > FAST_MAX[nm]).sum()))
    # PSNR against the reference's own image_raw (first three feature channels mapped to [0,1]) must not move
    exact = hip.ops.render(*args, hip.ops.make_opts(inp["ro"], **inp["kw"]))
    img_ref = g["feat"][..., :3] * 0.5 + 0.5 if "feat" in g else None  # the reference's own output (tests/golden/make_golden.py)
    if img_ref is not None:
        p_fast = _psnr(prod[0].cpu().numpy()[..., :3] * 0.5 + 0.5, img_ref.reshape(prod[0].shape[0], -1, 3))
        p_exact = _psnr(exact[0].cpu().numpy()[..., :3] * 0.5 + 0.5, img_ref.reshape(prod[0].shape[0], -1, 3))
        assert abs(p_fast - p_exact) < 0.01 or min(p_fast, p_exact) > 100, (p_fast, p_exact)
    assert _psnr(prod[0].cpu().numpy()[..., :3] * 0.5 + 0.5, exact[0].cpu().numpy()[..., :3] * 0.5 + 0.5) > 80


def test_fast_color_flag_is_ignored_without_an_importance_pass(hip, oracle):
    """Sf == 0: the single pass feeds nothing but the output, yet it is the reference's 'coarse' pass — it stays exact."""
    g = T.load_golden("render_c1_64x64_s32.npz")
    inp = T.golden_render_inputs(g)
    mlp = hip_mlp(hip, inp["raw_mlp"], inp["lr_mul"])
    args = (hip.ops.planes_to_nhwc(dev(inp["planes"])), dev(inp["rays_o"]), dev(inp["rays_d"]), dev(inp["jitter"]), dev(inp["u"]), mlp)
    a = hip.ops.render(*args, hip.ops.make_opts(inp["ro"], fast_color=True, **inp["kw"]))
    b = hip.ops.render(*args, hip.ops.make_opts(inp["ro"], **inp["kw"]))
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("n,H,W,use_triplane,crop", [(100, 64, 96, 1, None), (64, 256, 256, 0, 0.1), (37, 33, 47, 1, 0.05), (160, 128, 128, 1, None)])
def test_grid_density_staged_equals_direct_gather(hip, oracle, n, H, W, use_triplane, crop):
    """The density query can stage each wave's texel boxes through LDS (p3d_stage_plan / _commit / p3d_gather_features_boxed;
    the default with the tolerance-mode decoder, forced here with the exact one); tiles whose taps do not fit — grids whose
    rows are not multiples of 32 points, so that tiles straddle rows — fall back to direct gathers.  Either way the results are
    the direct path's, bit for bit (same values, same arithmetic), and the oracle's on a subset; the tolerance-mode decoder stays within 2e-5 (relative to the sigma scale) of the exact one."""
    from panic3d_amd import volume
    planes = T.make_planes(300 + n, 1, H, W, scale=3.0, smooth=8)
    raw = T.make_decoder_params(301 + n, 1.0, 10.0)
    ro = dict(T.RENDERING_KWARGS, use_triplane=use_triplane)
    opts = hip.ops.make_opts(ro, force_sigmoid=True)
    mlp = hip_mlp(hip, raw, 1.0)
    pl = hip.ops.planes_to_nhwc(dev(planes))
    bw = 0.7
    vs, org = bw / (n - 1), -bw / 2
    lim = None if crop is None else bw / 2 - crop
    for lo, hi in ((0, n ** 3), (n * n * 3 + 5, n ** 3 - 7)):  # also a slab that starts / ends in the middle of a row
        kw = dict(crop_limit=lim, skip_cropped=crop is not None)
        a = hip.ops.grid_density(pl, n, lo, hi, vs, (org, org, org), mlp, opts, staged=True, **kw)      # exact decoder, staged boxes
        b = hip.ops.grid_density(pl, n, lo, hi, vs, (org, org, org), mlp, opts, staged=False, **kw)  # direct gathers
        c = hip.ops.grid_density(pl, n, lo, hi, vs, (org, org, org), mlp, opts, **kw)                   # the default launch
        f = hip.ops.grid_density(pl, n, lo, hi, vs, (org, org, org), mlp, opts, fast=True, **kw)        # tolerance decoder + staging
        f2 = hip.ops.grid_density(pl, n, lo, hi, vs, (org, org, org), mlp, opts, fast=True, staged=False, **kw)
        if crop is None:
            a, b, c, f, f2 = (a, None), (b, None), (c, None), (f, None), (f2, None)
        assert torch.equal(a[0], b[0]) and torch.equal(c[0], b[0]) and torch.equal(f[0], f2[0])
        if crop is not None:
            assert torch.equal(a[1], b[1]) and torch.equal(c[1], b[1]) and torch.equal(f[1], b[1])
        scale = float(b[0][b[0] > -999].abs().max()) if bool((b[0] > -999).any()) else 1.0
        assert float((f[0] - b[0]).abs().max()) <= 2e-5 * max(scale, 1.0)
        a = a if crop is not None else a[0]
    sig = a if crop is None else a[0]
    idx = torch.arange(lo, hi, 61)
    pts, _, _ = volume.create_samples(n, (0, 0, 0), bw, idx=idx)
    osig, _ = oracle.decode(planes, pts.numpy(), oracle.prescale_mlp(*raw), bw, plane_mode=use_triplane, flags=opts.flags, density_only=True)
    got = sig[0, (idx - lo).cuda(), 0].cpu().numpy()
    if crop is None:
        assert np.array_equal(got, osig[0, :, 0])
    else:
        cropped = ((pts[0, :, 0].abs() > np.float32(lim)) | (pts[0, :, 2].abs() > np.float32(lim))).numpy()
        assert np.array_equal(got[~cropped], osig[0, ~cropped, 0]) and np.all(got[cropped] == -1000.0)


@pytest.mark.parametrize("Sc,Sf,res", [(48, 48, 32), (96, 96, 16), (12, 12, 24), (32, 0, 16)])
def test_in_kernel_draws_match_the_restated_generator(hip, oracle, Sc, Sf, res):
    """p3d_render_rng_f32 (round 3, opt-in): the two random draws are made inside the kernel by the counter-based generator of
    include/p3d_numerics.h instead of being read from 200 MB of torch.rand output.  The oracle restates the generator
    (oracle.device_draws) and renders with those arrays: every output bit-identical, on both render kernels and with the
    tolerance-mode final pass within its stated bound.  Also: the draws are in [0, 1) on torch.rand's 2^-24 grid, and two seeds
    give different images."""
    ro = dict(T.RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf)
    planes = T.make_planes(61, 2, 64, 64, scale=4.0, smooth=8)
    raw = T.make_decoder_params(62, 1.0, 30.0)
    lab = torch.stack([hip.cameras.camera_label(0.0, 20.0, 1.0, 30.0), hip.cameras.camera_label(10.0, 200.0, 1.0, 30.0)])
    o, d = hip.cameras.rays_from_label(lab, res)
    kw = dict(triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True)
    seed = 0x1234_5678_9ABC_DEF1
    jit, u = oracle.device_draws(seed, 2, res * res, Sc, Sf)
    assert jit.min() >= 0 and jit.max() < 1 and np.all(jit * 2 ** 24 == np.floor(jit * 2 ** 24)) and abs(float(jit.mean()) - 0.5) < 0.02
    ref = oracle.render(planes, o.numpy(), d.numpy(), jit, u if Sf > 0 else None, oracle.prescale_mlp(*raw), oracle.make_opts(ro, **kw))
    mlp = hip_mlp(hip, raw, 1.0)
    nhwc = hip.ops.planes_to_nhwc(dev(planes))
    for small in (True, False):
        out = hip.ops.render(nhwc, o.cuda(), d.cuda(), None, None, mlp, hip.ops.make_opts(ro, small_launch_kernel=small, **kw), ray_tile_w=res,
                             rng_seed=seed)
        for nm, a, b in zip(("feat", "depth", "wsum", "xyz"), out, ref):
            assert np.array_equal(a.cpu().numpy(), b), (nm, small)
    if Sf > 0:
        fast = hip.ops.render(nhwc, o.cuda(), d.cuda(), None, None, mlp,
                              hip.ops.make_opts(ro, small_launch_kernel=False, fast_color=True, **kw), ray_tile_w=res, rng_seed=seed)
        for nm, a, b in zip(("feat", "depth", "wsum", "xyz"), fast, ref):
            assert float(np.abs(a.cpu().numpy() - b).max()) <= FAST_MAX[nm], nm
    other = hip.ops.render(nhwc, o.cuda(), d.cuda(), None, None, mlp, hip.ops.make_opts(ro, **kw), ray_tile_w=res, rng_seed=seed + 1)
    assert not torch.equal(other[0], out[0])


@pytest.mark.parametrize("Sc,Sf", [(96, 96), (64, 96), (48, 48)])
def test_stratified_depths_out_of_order_are_sorted_like_the_reference(hip, oracle, Sc, Sf):
    """unify_samples sorts the concatenated depths (renderer.py:289-301), so a stratified row that is NOT ascending still has a
    defined result.  With jitter from torch.rand_like only rounding can swap two neighbours; this test forces both kinds of
    disorder with doctored jitter — neighbour swaps (jitter 1.5 next to 0.1) and far moves (3.7, -2.2: outside anything the
    reference draws) — on the large-launch kernels.  The 96-key production kernel holds no coarse-depth column at all and
    resolves sorted ranks on the fly (csrc/p3d_kernels.hip, P3D_TCG): its three access paths must all match the oracle bit
    for bit, in both precision modes' depth handling."""
    res = 24
    ro = dict(T.RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf)
    planes = T.make_planes(71, 1, 64, 64, scale=4.0, smooth=8)
    raw = T.make_decoder_params(72, 1.0, 30.0)
    o, d = hip.cameras.rays_from_label(hip.cameras.camera_label(5.0, 30.0, 1.0, 30.0)[None], res)
    kw = dict(triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True)
    jit, u = T.make_random_draws(73, 1, res * res, Sc, Sf)
    jit = jit.copy()
    rng = np.random.default_rng(74)
    rays = rng.permutation(res * res)
    for r in rays[:120]:  # neighbour swaps, some rays twice, also at both ends of the row
        for i in rng.choice(Sc - 1, size=rng.integers(1, 3), replace=False):
            jit.reshape(res * res, Sc)[r, i] = 1.5
            jit.reshape(res * res, Sc)[r, i + 1] = 0.1
    for r in rays[120:150]:  # far moves: the general selection
        jit.reshape(res * res, Sc)[r, rng.integers(0, Sc)] = 3.7
        jit.reshape(res * res, Sc)[r, rng.integers(0, Sc)] = -2.2
    ref = oracle.render(planes, o.numpy(), d.numpy(), jit, u, oracle.prescale_mlp(*raw), oracle.make_opts(ro, **kw))
    mlp = hip_mlp(hip, raw, 1.0)
    nhwc = hip.ops.planes_to_nhwc(dev(planes))
    for flags in (dict(), dict(early_out=False)):
        out = hip.ops.render(nhwc, o.cuda(), d.cuda(), dev(jit), dev(u), mlp,
                             hip.ops.make_opts(ro, small_launch_kernel=False, **flags, **kw), ray_tile_w=res)
        for nm, a, b in zip(("feat", "depth", "wsum", "xyz"), out, ref):
            assert np.array_equal(a.cpu().numpy(), b), (nm, flags)
    fast = hip.ops.render(nhwc, o.cuda(), d.cuda(), dev(jit), dev(u), mlp,
                          hip.ops.make_opts(ro, small_launch_kernel=False, fast_color=True, **kw), ray_tile_w=res)
    for nm, a, b in zip(("feat", "depth", "wsum", "xyz"), fast, ref):
        assert float(np.abs(a.cpu().numpy() - b).max()) <= FAST_MAX[nm], nm


# ---- the staged path (ImportanceRenderer.forward_staged): the reference's structure on the stand-alone stage kernels ------------
def _stub_decoder(raw, lr_mul, force_sigmoid):
    class FC:
        def __init__(self, w, b, i):
            self.weight, self.bias, self.weight_gain, self.bias_gain = w, b, lr_mul / np.sqrt(i), lr_mul

    class Dec:
        pass
    d = Dec()
    d.force_sigmoid = bool(force_sigmoid)
    r = [dev(x) for x in raw]
    d.net = [FC(r[0], r[1], 32), None, FC(r[2], r[3], 64)]
    return d


def _within(a, b, tol):
    """fraction of rays whose every component is within tol, and the largest difference"""
    diff = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).reshape(-1, a.shape[-1]).max(axis=1)
    return float((diff <= tol).mean()), float(diff.max())


@pytest.mark.parametrize("name", ["render_32x32_16p16", "render_variant_b", "render_c1_64x64_s32", "render_12x12_96p96"])
def test_staged_path_equals_the_fused_kernel_without_noise(hip, name):
    """forward_staged — stratified depths, decode, masks, marcher, importance, decode, masks, unify, marcher as separate launches with
    the reference's torch glue in between — computes what the fused kernel computes: against the fused exact render and against the
    REFERENCE's outputs of the fixture, within the fp32 tolerances (the masks are torch's softplus / exp here, the contract's
    polynomials there: a sample within rounding of the cull threshold may flip, so a 0.5 % allowance on the rays)."""
    g = T.load_golden(name + ".npz")
    inp = T.golden_render_inputs(g)
    rend = hip.ImportanceRenderer(use_triplane=bool(inp["ro"]["use_triplane"]))
    dec = _stub_decoder(inp["raw_mlp"], inp["lr_mul"], inp["kw"]["force_sigmoid"])
    kw = {k: v for k, v in inp["kw"].items() if k != "force_sigmoid"}
    args = (dev(inp["planes"]), dec, dev(inp["rays_o"]), dev(inp["rays_d"]), inp["ro"])
    common = dict(jitter=dev(inp["jitter"]), u=dev(inp["u"]) if inp["ro"]["depth_resolution_importance"] > 0 else None, **kw)
    with torch.no_grad():
        fused = rend(*args, exact=True, **common)
        staged = rend.forward_staged(*args, **common)
    for key, a, b, tol in zip(("feat", "depth", "wsum", "xyz"), staged, fused, (TOL_FEAT, TOL_DEPTH, TOL_WEIGHT, TOL_XYZ)):
        assert a.shape == b.shape, key
        frac, worst = _within(a.cpu().numpy(), b.cpu().numpy(), tol)
        assert frac >= 0.995, (key, "vs fused", frac, worst)
        frac, worst = _within(a.cpu().numpy(), g[key], tol)
        assert frac >= 0.995, (key, "vs reference", frac, worst)


def test_density_noise_vs_reference(hip):
    """rendering_options['density_noise'] > 0 (renderer.py:276-277: `sigma += randn_like(sigma) * density_noise` inside run_model, before
    the masks, in both passes) — ImportanceRenderer.forward routes it to the staged path.  Against the REFERENCE's own render with its
    four draws (jitter, coarse noise, u, fine noise) captured by tests/golden/make_golden.py; the noise really changes the frame; and
    with no draws given the renderer makes its own on the device."""
    g = T.load_golden("render_density_noise.npz")
    seed, res, Sc, Sf, dn = int(g["meta_seed"]), int(g["meta_res"]), int(g["meta_Sc"]), int(g["meta_Sf"]), float(g["meta_density_noise"])
    planes = T.make_planes(seed, 1, 256, 256, scale=4.0, smooth=8)
    assert T.checksum(planes) == str(g["planes_checksum"])
    raw = T.make_decoder_params(seed + 1, 1.0, float(g["meta_sigma_gain"]))
    ro = dict(T.RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf, density_noise=dn)
    rend = hip.ImportanceRenderer(use_triplane=True)
    dec = _stub_decoder(raw, 1.0, True)
    args = (dev(planes), dec, dev(g["rays_o"]), dev(g["rays_d"]))
    kw = dict(triplane_crop=0.1, cull_clouds=0.5, jitter=dev(g["jitter"]), u=dev(g["u"]))
    with torch.no_grad():
        out = rend(*args, ro, density_noise_draws=(dev(g["noise_coarse"]), dev(g["noise_fine"])), **kw)
        quiet = rend.forward_staged(*args, dict(ro, density_noise=0), **kw)
        own = rend(*args, ro, triplane_crop=0.1, cull_clouds=0.5)
    for key, a, tol in zip(("feat", "depth", "wsum", "xyz"), out, (TOL_FEAT, TOL_DEPTH, TOL_WEIGHT, TOL_XYZ)):
        frac, worst = _within(a.cpu().numpy(), g[key], tol)
        assert frac >= 0.995, (key, frac, worst)
    assert float((out[0] - quiet[0]).abs().max()) > 1e-2  # the noise matters in this fixture
    assert all(torch.isfinite(t).all() for t in own) and own[0].shape == (1, res * res, 32)
    with pytest.raises(NotImplementedError):
        hip.ops.make_opts(ro)  # the fused kernel itself does not take the option


@pytest.mark.parametrize("seed", range(300, 312))
def test_staged_path_random_configs(hip, seed):
    """The staged path against the fused kernel over the randomised sweep of test_render_random_configs_bit_exact (batches, non-square
    planes, ragged ray lists, 4..70 coarse samples, 0 / 1..70 / 129..149 fine samples, every mask mode, both plane conventions, both
    backgrounds, rays that miss the volume): two implementations of the same forward() that share no kernel except the decoder's
    arithmetic.  fp32 tolerances on >= 99 % of the rays (torch's softplus / exp decide the cull mask in the staged path)."""
    c = _random_config(seed)
    rend = hip.ImportanceRenderer(use_triplane=bool(c["ro"]["use_triplane"]))
    dec = _stub_decoder(c["raw"], c["lr_mul"], c["kw"]["force_sigmoid"])
    kw = {k: v for k, v in c["kw"].items() if k != "force_sigmoid"}
    Sf = c["ro"]["depth_resolution_importance"]
    args = (dev(c["planes"]), dec, dev(c["o"]), dev(c["d"]), c["ro"])
    common = dict(jitter=dev(c["jit"]), u=dev(c["u"]) if Sf > 0 else None, **kw)
    with torch.no_grad():
        fused = rend(*args, exact=True, ray_tile_w=c["tile_w"], **common)
        staged = rend.forward_staged(*args, **common)
    for key, a, b, tol in zip(("feat", "depth", "wsum", "xyz"), staged, fused, (TOL_FEAT, TOL_DEPTH, TOL_WEIGHT, TOL_XYZ)):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert a.shape == b.shape and np.isfinite(a).all() == np.isfinite(b).all(), key
        ok = np.isfinite(a) & np.isfinite(b)
        frac, worst = _within(np.where(ok, a, 0), np.where(ok, b, 0), tol)
        assert frac >= 0.99, (key, seed, frac, worst)


@pytest.mark.parametrize("Sc,Sf,res", [(96, 96, 32), (48, 48, 32), (48, 48, 128), (64, 64, 32), (48, 48, 256)])
def test_weights_only_launch_equals_the_full_launch(hip, Sc, Sf, res):
    """P3D_FLAG_WEIGHTS_ONLY (round 5): paste_front's occlusion pass reads `image_weights` of its second render and nothing else
    (training/triplane.py:565-578).  A ray's weights depend on depths and densities only, so the weights-only launch (tolerance mode,
    small launches, 48 / 96 fine samples: k_render_quad<NF, true, true>) decodes no colours — and wsum / depth must be BIT-IDENTICAL to
    the full tolerance-mode launch's; where the library has no such instantiation (64 fine samples; a large launch; the exact mode) the
    hint is ignored and the same bits come from the full kernel."""
    ro = dict(T.RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf)
    planes = T.make_planes(71, 1, 64, 64, scale=4.0, smooth=8)
    raw = T.make_decoder_params(72, 1.0, 30.0)
    o, d = hip.cameras.rays_from_label(hip.cameras.camera_label(5.0, 30.0, 1.0, 30.0)[None], res)
    jit, u = T.make_random_draws(73, 1, res * res, Sc, Sf)
    kw = dict(triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True)
    mlp, nhwc = hip_mlp(hip, raw, 1.0), hip.ops.planes_to_nhwc(dev(planes))
    for fast in (True, False):
        opts = hip.ops.make_opts(ro, fast_color=fast, **kw)
        full = hip.ops.render(nhwc, o.cuda(), d.cuda(), dev(jit), dev(u), mlp, opts, ray_tile_w=res)
        st = {}
        wo = hip.ops.render(nhwc, o.cuda(), d.cuda(), dev(jit), dev(u), mlp, opts, ray_tile_w=res, weights_only=True, stats=st)
        assert wo[0] is None and wo[3] is None
        assert torch.equal(wo[2], full[2]) and torch.equal(wo[1], full[1]), (fast, Sc, res)
        assert float(full[2].mean()) > 0.05  # a volume with surfaces: the weights are not trivially zero
