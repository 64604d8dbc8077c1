"""GPU: BASELINE.json's full-size configurations.

c2 (4 x 256^2 rays, 48+48, planes from a random-init StyleGAN2-256 backbone run on the HIP synthesis path) and
c3 (512^2 rays, 48+48, synthetic planes) are compared BIT-EXACT with the CPU oracle (OpenMP over the GPU box's host
cores: seconds), plus size-independent properties.  c5 (density grid) is checked on slabs."""
import numpy as np
import pytest
import torch

import p3d_testing as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import panic3d_amd
    assert torch.cuda.is_available()
    panic3d_amd._lib.lib()
    return panic3d_amd


RO = dict(T.RENDERING_KWARGS)
KW = dict(triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True)


def check_properties(feat, depth, wsum, xyz, ro):
    assert np.isfinite(feat).all() and np.isfinite(depth).all() and np.isfinite(xyz).all()
    assert (wsum >= 0).all() and (wsum <= 1 + 1e-5).all()                       # alpha compositing weights
    delta = (ro["ray_end"] - ro["ray_start"]) / (ro["depth_resolution"] - 1)
    assert depth.min() >= ro["ray_start"] - 1e-6 and depth.max() <= ro["ray_end"] + delta + 1e-6
    assert feat.min() >= -1 - 1e-5 and feat.max() <= 1 + 1e-5                    # white_back + 2x-1 of sigmoid colours
    empty = wsum[..., 0] < 1e-7
    if empty.any():                                                              # empty rays: white, depth = global max
        assert np.abs(feat[empty] - 1).max() < 1e-5 and np.all(depth[empty, 0] == depth.max())


def test_c2_backbone_planes_batch4_256(hip, oracle):
    from panic3d_amd import stylegan2 as sg
    torch.manual_seed(0)
    G = sg.Generator(z_dim=512, c_dim=25, w_dim=512, img_resolution=256, img_channels=96, cond_mode="none",
                     mapping_kwargs={"num_layers": 2}, channel_base=32768, channel_max=512, num_fp16_res=0,
                     conv_clamp=None).cuda().eval()
    N, res, Sc, Sf = 4, 256, 48, 48
    with torch.no_grad():
        ws = G.mapping(torch.randn(N, 512).cuda(), torch.zeros(N, 25).cuda(), {})
        planes = G.synthesis(ws, {}, noise_mode="const").view(N, 3, 32, 256, 256) * 4.0  # random-init planes are ~N(0,1): scale up
    assert torch.isfinite(planes).all() and planes.std() > 0.5
    raw = T.make_decoder_params(21, 1.0, 30.0)
    labels = torch.stack([hip.cameras.camera_label(0.0, a, 1.0, 30.0) for a in (0.0, 90.0, 180.0, 270.0)])
    o, d = hip.cameras.rays_from_label(labels, res)
    jit, u = T.make_random_draws(22, N, res * res, Sc, Sf)
    opts = hip.ops.make_opts(RO, **KW)
    mlp = hip.ops.prescale_mlp(*(torch.from_numpy(x).cuda() for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
    out = hip.ops.render(hip.ops.planes_to_nhwc(planes.contiguous()), o.cuda(), d.cuda(), torch.from_numpy(jit).cuda(),
                         torch.from_numpy(u).cuda(), mlp, opts, ray_tile_w=res)
    feat, depth, wsum, xyz = (t.cpu().numpy() for t in out)
    check_properties(feat, depth, wsum, xyz, RO)
    ref = oracle.render(planes.cpu().numpy(), o.numpy(), d.numpy(), jit, u, oracle.prescale_mlp(*raw), oracle.make_opts(RO, **KW))
    for name, a, b in zip(("feat", "depth", "wsum", "xyz"), (feat, depth, wsum, xyz), ref):
        assert np.array_equal(a, b), name
    assert 0.05 < wsum.mean() < 0.999  # the scene has both hit and empty rays


def test_c3_512x512_96_samples(hip, oracle):
    res, Sc, Sf = 512, 48, 48
    planes = T.make_planes(31, 1, 256, 256, scale=4.0, smooth=16)
    raw = T.make_decoder_params(32, 1.0, 30.0)
    o, d = hip.cameras.rays_from_label(hip.cameras.camera_label(5.0, 20.0, 1.0, 30.0)[None], res)
    jit, u = T.make_random_draws(33, 1, res * res, Sc, Sf)
    opts = hip.ops.make_opts(RO, **KW)
    mlp = hip.ops.prescale_mlp(*(torch.from_numpy(x).cuda() for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
    pl = hip.ops.planes_to_nhwc(torch.from_numpy(planes).cuda())
    args = (o.cuda(), d.cuda(), torch.from_numpy(jit).cuda(), torch.from_numpy(u).cuda(), mlp, opts)
    out = hip.ops.render(pl, *args, ray_tile_w=res)
    feat, depth, wsum, xyz = (t.cpu().numpy() for t in out)
    check_properties(feat, depth, wsum, xyz, RO)
    ref = oracle.render(planes, o.numpy(), d.numpy(), jit, u, oracle.prescale_mlp(*raw), oracle.make_opts(RO, **KW))
    for name, a, b in zip(("feat", "depth", "wsum", "xyz"), (feat, depth, wsum, xyz), ref):
        assert np.array_equal(a, b), name
    # idempotence / layout independence: the unstructured ray-list path gives the same bits as the 8x4-tiled image path
    out2 = hip.ops.render(pl, *args, ray_tile_w=0)
    for a, b in zip(out, out2):
        assert torch.equal(a, b)
    # rays are independent (only the depth clamp is global): a 128x128 block of the same rays, small enough for the
    # small-launch kernel (16 rays x 2 samples per wave), gives the same bits as the 512^2 launch on the 32-ray kernel
    sub = (torch.arange(128)[:, None] * res + torch.arange(128)[None, :] + 200 * res + 170).reshape(-1).cuda()
    o2, d2, j2 = args[0][:, sub].contiguous(), args[1][:, sub].contiguous(), args[2][:, sub].contiguous()
    st = {}
    out3 = hip.ops.render(pl, o2, d2, j2, args[3][sub].contiguous(), mlp, opts, ray_tile_w=128, stats=st)
    assert st["small_launch_kernel"]
    for k in (0, 2, 3):
        assert torch.equal(out[k][:, sub], out3[k])


def test_c5_density_grid_slabs(hip, oracle):
    """get_eg3d_volume's query (_util/eg3d_metrics3d.py:94-183): density-only decode of a regular grid, in slabs."""
    Ngrid = 128
    planes = T.make_planes(41, 1, 256, 256, scale=4.0, smooth=16)
    raw = T.make_decoder_params(42, 1.0, 30.0)
    opts = hip.ops.make_opts(RO, force_sigmoid=True)
    mlp = hip.ops.prescale_mlp(*(torch.from_numpy(x).cuda() for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
    pl = hip.ops.planes_to_nhwc(torch.from_numpy(planes).cuda())
    lin = (torch.arange(Ngrid, dtype=torch.float32) / (Ngrid - 1) - 0.5) * 0.7
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    pts = torch.stack([xx, yy, zz], -1).reshape(1, -1, 3)
    sig = torch.cat([hip.ops.triplane_decode(pl, pts[:, a:a + Ngrid ** 3 // 8].cuda().contiguous(), mlp, opts, density_only=True)[0]
                     for a in range(0, Ngrid ** 3, Ngrid ** 3 // 8)], dim=1)  # 8 slabs, as 8 GPUs would split them
    whole, _ = hip.ops.triplane_decode(pl, pts.cuda(), mlp, opts, density_only=True)
    assert torch.equal(sig, whole)
    sub = slice(0, Ngrid ** 3, 97)
    osig, _ = oracle.decode(planes, pts[:, sub].numpy(), oracle.prescale_mlp(*raw), 0.7, plane_mode=1, flags=opts.flags,
                            density_only=True)
    assert np.array_equal(whole[:, sub].cpu().numpy(), osig)


@pytest.mark.parametrize("Sc,Sf", [(96, 96), (64, 64)])
def test_large_launch_other_sampling_rates(hip, oracle, Sc, Sf):
    """The eval-faithful 96+96 (eg3dc_v0.py:30-31) and 64+64 at 256^2 rays: large launches whose LDS rows make the host pick
    other workgroup shapes (1 x 4 resp. 3 x 2 waves per CU: it fills the CU's 160 KB) — bit-exact with the oracle like the 4-wave 48+48 configuration."""
    res = 256
    ro = dict(RO, depth_resolution=Sc, depth_resolution_importance=Sf)
    planes = T.make_planes(51, 1, 128, 128, scale=4.0, smooth=16)
    raw = T.make_decoder_params(52, 1.0, 30.0)
    o, d = hip.cameras.rays_from_label(hip.cameras.camera_label(-5.0, 140.0, 1.0, 30.0)[None], res)
    jit, u = T.make_random_draws(53, 1, res * res, Sc, Sf)
    mlp = hip.ops.prescale_mlp(*(torch.from_numpy(x).cuda() for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
    st = {}
    out = hip.ops.render(hip.ops.planes_to_nhwc(torch.from_numpy(planes).cuda()), o.cuda(), d.cuda(), torch.from_numpy(jit).cuda(),
                         torch.from_numpy(u).cuda(), mlp, hip.ops.make_opts(ro, **KW), ray_tile_w=res, stats=st)
    assert not st["small_launch_kernel"]
    ref = oracle.render(planes, o.numpy(), d.numpy(), jit, u, oracle.prescale_mlp(*raw), oracle.make_opts(ro, **KW))
    for name, a, b in zip(("feat", "depth", "wsum", "xyz"), out, ref):
        assert np.array_equal(a.cpu().numpy(), b), name
    check_properties(*(t.cpu().numpy() for t in out), ro)
    # the tolerance-mode kernel of the same instantiation (96+96: the 128-key variant; quad gathers, colour on demand, run jump)
    fast = hip.ops.render(hip.ops.planes_to_nhwc(torch.from_numpy(planes).cuda()), o.cuda(), d.cuda(), torch.from_numpy(jit).cuda(),
                          torch.from_numpy(u).cuda(), mlp, hip.ops.make_opts(ro, fast_color=True, **KW), ray_tile_w=res)
    for name, a, b in zip(("feat", "depth", "wsum", "xyz"), fast, out):
        err = (a - b).abs().reshape(res * res, -1).amax(dim=-1)
        print(f"{Sc}+{Sf} tolerance vs exact {name}: max {float(err.max()):.2e}")
        assert float(err.median()) <= 2e-6 and float(err.max()) <= FAST_MAX[name], (name, float(err.max()))  # every ray


FAST_MAX = dict(feat=1e-5, depth=2e-5, wsum=1e-5, xyz=1e-5)  # the tolerance mode's stated bound vs the exact contract


# ---- full-size holes closed in round 2 (VERDICT r01 "What's weak" 1-2) ---------------------------------------------------
@pytest.mark.parametrize("scene", ["canonical", "surface"])
def test_bench_scene_512_frame_vs_oracle(hip, oracle, scene):
    """The frame bench.py times: a 128^2 block of the 512^2 launch equals the same rays rendered as their own launch of the
    same kernel, and that equals the CPU oracle bit for bit (the check bench.py itself runs outside its timed region)."""
    import bench
    res, Sc, Sf = 512, 48, 48
    ro = T.bench_rendering_kwargs(Sc, Sf)
    w = bench.Workload(scene, torch.device("cuda"), res, 20.0, 7)
    for early in (True, False):
        v, frame, (jit, u) = w.verify_block(Sc, Sf, False, early=early, return_frame=True)
        assert v["ok"] and all(v[k]["bit_exact_vs_oracle"] for k in ("feat", "depth", "wsum", "xyz")), v
    nhwc, o, d, mlp = hip.ops.planes_to_nhwc(w.planes), w.o, w.d, w.mlp
    feat, depth, wsum, xyz = (t.cpu().numpy() for t in frame)
    check_properties(feat, depth, wsum, xyz, ro)
    if scene == "surface":
        assert 0.3 < (wsum > 0.5).mean() < 0.9
        fast = hip.ops.render(nhwc, o, d, jit, u, mlp, hip.ops.make_opts(ro, fast_color=True, **T.BENCH_KW), ray_tile_w=res)
        # tolerance mode at full size: the stated hard bound on every ray of every output, and PSNR >= 120 dB (measured 138.8)
        for nm, a, b in zip(("feat", "depth", "wsum", "xyz"), fast, frame):
            err = (a - b).abs().reshape(res * res, -1).amax(dim=-1)
            print(f"surface 512^2 tolerance vs exact {nm}: max {float(err.max()):.2e}")
            assert float(err.median()) < 2e-6 and float(err.max()) <= FAST_MAX[nm], (nm, float(err.max()), int((err > FAST_MAX[nm]).sum()))
        mse = float((((fast[0][..., :3] - frame[0][..., :3]) * 0.5).double() ** 2).mean())
        assert 10 * np.log10(1.0 / max(mse, 1e-30)) >= 120.0
    else:
        assert wsum.max() == 0.0  # SURVEY 8(d)'s decoder never clears cull_clouds = 0.5: an empty volume


@pytest.mark.parametrize("skip_cropped", [False, True])
def test_c5_grid_entry_point_at_512(hip, oracle, skip_cropped):
    """p3d_grid_density_f32 at N = 512 (in-kernel create_samples, 32-bit index arithmetic and p3d_fmod_pos at magnitudes no
    small grid reaches, with and without P3D_FLAG_SKIP_CROPPED): both ends of the index range and a stride-997 subset of the
    134 M points against oracle.decode on the reference's create_samples points, bit for bit; and the crop mask."""
    from panic3d_amd import volume
    N = 512
    planes = T.make_planes(61, 1, 256, 256, scale=4.0, smooth=16)
    raw = T.make_decoder_params(62, 1.0, 30.0)
    opts = hip.ops.make_opts(RO, force_sigmoid=True)
    mlp = hip.ops.prescale_mlp(*(torch.from_numpy(x).cuda() for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
    pl = hip.ops.planes_to_nhwc(torch.from_numpy(planes).cuda())
    vs, org, lim = 0.7 / (N - 1), -0.35, 0.35 - 0.1
    sig, msk = hip.ops.grid_density(pl, N, 0, N ** 3, vs, (org, org, org), mlp, opts, crop_limit=lim, skip_cropped=skip_cropped)
    idx = torch.cat([torch.arange(0, 4096), torch.arange(0, N ** 3, 997), torch.arange(N ** 3 - 4096, N ** 3)]).unique()
    pts, _, _ = volume.create_samples(N, (0, 0, 0), 0.7, idx=idx)
    osig, _ = oracle.decode(planes, pts.numpy(), oracle.prescale_mlp(*raw), 0.7, plane_mode=1, flags=opts.flags, density_only=True)
    cropped = ((pts[0, :, 0].abs() > np.float32(lim)) | (pts[0, :, 2].abs() > np.float32(lim))).numpy()
    got = sig[0, idx.cuda(), 0].cpu().numpy()
    assert np.array_equal(msk[0, idx.cuda(), 0].cpu().numpy(), cropped)
    if skip_cropped:
        assert np.array_equal(got[~cropped], osig[0, ~cropped, 0]) and np.all(got[cropped] == -1000.0)
    else:
        assert np.array_equal(got, osig[0, :, 0])
    assert 0.2 < cropped.mean() < 0.8


def test_c4_shared_planes_four_views_512(hip, oracle):
    """c4's batched launch (P3D_FLAG_SHARED_PLANES): V = 4 views of one subject at 512^2 rays in ONE launch equal four single
    launches bit for bit: feat / wsum / xyz always; depth too with P3D_FLAG_PER_VIEW_CLAMP (each view keeps its own clamp
    range, as four calls of the reference would), and without it once both are clamped to the batch's common range (the
    reference's scope for ONE call with a batch of V cameras: torch.min/max over everything, ray_marcher.py:49-50)."""
    res, Sc, Sf, V = 512, 48, 48, 4
    planes_np, raw = T.make_bench_scene("surface")
    ro = T.bench_rendering_kwargs(Sc, Sf)
    labels = torch.stack([hip.cameras.camera_label(0.0, a, 1.0, 30.0) for a in (0.0, 90.0, 180.0, 270.0)])
    o, d = hip.cameras.rays_from_label(labels, res)
    o, d = o.cuda(), d.cuda()
    gen = torch.Generator(device="cuda").manual_seed(11)
    jit = torch.rand((V, res * res, Sc, 1), device="cuda", generator=gen)
    u = torch.rand((V * res * res, Sf), device="cuda", generator=gen)
    mlp = hip.ops.prescale_mlp(*(torch.from_numpy(x).cuda() for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
    nhwc = hip.ops.planes_to_nhwc(torch.from_numpy(planes_np).cuda())  # [1,...]: shared by the V views
    opts = hip.ops.make_opts(ro, **T.BENCH_KW)
    batched = hip.ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res)
    singles = [hip.ops.render(nhwc, o[v:v + 1], d[v:v + 1], jit[v:v + 1], u[v * res * res:(v + 1) * res * res], mlp, opts, ray_tile_w=res)
               for v in range(V)]
    for k in (0, 2, 3):
        assert torch.equal(batched[k], torch.cat([s[k] for s in singles]))
    hit = batched[2] > 0  # a ray with weight has its depth inside its own sample range: no clamp touches it in either launch
    assert torch.equal(batched[1][hit], torch.cat([s[1] for s in singles])[hit])
    assert bool((batched[1][~hit] == batched[1].max()).all())  # empty rays: +inf clamped to the CALL's far end (all views)
    per_view = hip.ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res, per_view_clamp=True)
    for k in range(4):
        assert torch.equal(per_view[k], torch.cat([s[k] for s in singles])), k
