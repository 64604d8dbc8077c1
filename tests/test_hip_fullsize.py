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
