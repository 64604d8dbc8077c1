"""GPU: StyleGAN2 synthesis operators / networks on the HIP path vs (a) the REFERENCE's own outputs (tests/golden/syn_*.npz,
produced by tests/golden/make_golden_synthesis.py) and (b) a plain PyTorch fp32 restatement of the same op on CPU at
hot-path shapes.  Floating-point convolutions: tolerance, stated per test (different summation order: the reference sums
in oneDNN/cuDNN order, the HIP kernel in MFMA k-order; K up to 4608 terms)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import p3d_testing as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import panic3d_amd
    assert torch.cuda.is_available()
    panic3d_amd._lib.lib()
    return panic3d_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def load_sd(mod, g, prefix):
    sd = {k[len(prefix):].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}
    missing, unexpected = mod.load_state_dict(sd, strict=True), None
    return mod.cuda().eval()


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


@pytest.mark.parametrize("tag", ["conv1", "conv0", "conv0b"])
def test_synthesis_layer_vs_reference(hip, tag):
    from panic3d_amd import stylegan2 as sg
    g = T.load_golden("syn_layers.npz")
    cin, cout, res, up = (int(v) for v in g[f"{tag}_cfg"])
    lay = load_sd(sg.SynthesisLayer(cin, cout, w_dim=32, resolution=res, up=up, conv_clamp=None), g, f"{tag}_sd_")
    with torch.no_grad():
        y = lay(dev(g[f"{tag}_x"]), dev(g[f"{tag}_w"]), noise_mode="const").cpu().numpy()
        assert rel_err(y, g[f"{tag}_y"]) < 2e-6 * np.sqrt(cin * 9) + 1e-6  # fp32 dot products of 9*cin terms
        if tag == "conv1":
            lay.conv_clamp = 0.8
            yc = lay(dev(g[f"{tag}_x"]), dev(g[f"{tag}_w"]), noise_mode="const", gain=0.5).cpu().numpy()
            assert np.abs(yc - g[f"{tag}_y_clamp"]).max() < 1e-5 and np.abs(yc).max() <= 0.4 + 1e-7
            lay.conv_clamp = None
            yn = lay(dev(g[f"{tag}_x"]), dev(g[f"{tag}_w"]), noise_mode="none").cpu().numpy()
            assert rel_err(yn, g[f"{tag}_y_nonoise"]) < 3e-5


def test_torgb_vs_reference(hip):
    from panic3d_amd import stylegan2 as sg
    g = T.load_golden("syn_layers.npz")
    rgb = load_sd(sg.ToRGBLayer(16, 96, w_dim=32, conv_clamp=None), g, "torgb_sd_")
    with torch.no_grad():
        y = rgb(dev(g["torgb_x"]), dev(g["torgb_w"])).cpu().numpy()
    assert rel_err(y, g["torgb_y"]) < 1e-5


def test_fir_and_bias_act_vs_reference(hip):
    g = T.load_golden("syn_layers.npz")
    f = dev(g["fir_f"])
    assert np.array_equal(hip.ops.setup_filter([1, 3, 3, 1]).numpy(), g["fir_f"])
    x = dev(g["up_x"])
    assert np.abs(hip.ops.upsample2d(x, f).cpu().numpy() - g["up_y"]).max() < 2e-6
    assert np.abs(hip.ops.upfirdn2d(x, f, up=1, padding=[1, 1, 1, 1], gain=4).cpu().numpy() - g["ufd_y"]).max() < 2e-6
    y2 = hip.ops.upfirdn2d(x, f, up=2, down=1, padding=[3, 0, 1, 2], flip_filter=True, gain=2).cpu().numpy()
    assert y2.shape == g["ufd2_y"].shape and np.abs(y2 - g["ufd2_y"]).max() < 2e-6
    xb, bb = dev(g["ba_x"]), dev(g["ba_b"])
    assert np.array_equal(hip.ops.bias_act(xb, bb, act="lrelu").cpu().numpy(), g["ba_lrelu"])  # same fp32 ops: exact
    assert np.array_equal(hip.ops.bias_act(xb, bb, act="linear", gain=2.0, clamp=1.5).cpu().numpy(), g["ba_lin_clamp"])
    assert np.array_equal(hip.ops.bias_act(xb.reshape(3, -1)[:, :7].contiguous(), bb, act="lrelu").cpu().numpy(), g["ba_fc"])


def torch_modconv_ref(x, w, s, noise, up, demod, bias, f):
    """Plain PyTorch fp32 (CPU) restatement of modulated_conv2d + bias/lrelu: per-sample weights, grouped conv."""
    N, I, H, W = x.shape
    O, _, k, _ = w.shape
    ww = w.unsqueeze(0) * s.reshape(N, 1, I, 1, 1)
    if demod:
        ww = ww * (ww.square().sum(dim=[2, 3, 4], keepdim=True) + 1e-8).rsqrt()
    ys = []
    for n in range(N):
        if up == 1:
            y = F.conv2d(x[n:n + 1], ww[n], padding=k // 2)
        else:
            y = F.conv_transpose2d(x[n:n + 1], ww[n].transpose(0, 1), stride=2)
            ff = (f * 4).flip([0, 1])[None, None].repeat(O, 1, 1, 1)
            y = F.conv2d(F.pad(y, [1, 1, 1, 1]), ff, groups=O)
        ys.append(y)
    y = torch.cat(ys)
    if noise is not None:
        y = y + noise
    if bias is not None:
        y = y + bias.reshape(1, -1, 1, 1)
    return y


@pytest.mark.parametrize("I,O,H,up,ks", [(512, 512, 16, 1, 3), (512, 512, 8, 2, 3), (256, 128, 32, 2, 3), (128, 96, 64, 1, 1),
                                         (40, 72, 13, 1, 3),
                                         # channel tails (I not a multiple of the 8-channel K chunk; e.g. conditioning channels
                                         # concatenated to the feature map), ragged tiles, O < 64, split-K with a partial slice
                                         (20, 70, 13, 1, 3), (35, 10, 9, 2, 3), (19, 5, 11, 1, 1), (515, 64, 8, 1, 3), (131, 64, 16, 2, 3)])
def test_modconv_hot_path_shapes_vs_torch_fp32(hip, I, O, H, up, ks):
    g = torch.Generator().manual_seed(I + O + H)
    N = 2
    x = torch.randn(N, I, H, H, generator=g)
    w = torch.randn(O, I, ks, ks, generator=g)
    s = torch.randn(N, I, generator=g) * 0.5 + 1.0
    noise = torch.randn(H * up, H * up, generator=g) * 0.3
    bias = torch.randn(O, generator=g) * 0.2
    f = hip.ops.setup_filter([1, 3, 3, 1])
    demod = ks == 3
    ref = torch_modconv_ref(x, w, s, noise, up, demod, bias, f)
    ref = F.leaky_relu(ref, 0.2) * np.sqrt(2) if ks == 3 else ref
    y = hip.ops.modulated_conv2d(x.cuda(), w.cuda(), s.cuda(), noise=noise.cuda(), up=up, padding=ks // 2,
                                 resample_filter=f.cuda(), demodulate=demod, bias=bias.cuda(),
                                 act="lrelu" if ks == 3 else "linear")
    err = rel_err(y.cpu().numpy(), ref.numpy())
    assert err < 3e-6 * np.sqrt(I * ks * ks), err  # ~2e-4 at K = 4608


@pytest.mark.parametrize("tag", ["none", "cond", "cond2", "cond3", "cond4"])  # cond*: every branch of the conditioning glue
def test_generator_vs_reference(hip, tag):
    from panic3d_amd import stylegan2 as sg
    g = T.load_golden(f"syn_generator_{tag}.npz")
    kw = dict(z_dim=64, c_dim=25, w_dim=64, img_resolution=32, img_channels=96, mapping_kwargs={"num_layers": 2},
              channel_base=2048, channel_max=64, num_fp16_res=0, conv_clamp=None, fused_modconv_default="inference_only")
    G = load_sd(sg.Generator(cond_mode=str(g["cond_mode"]), **kw), g, "sd_")
    cond = {k[5:]: dev(v) for k, v in g.items() if k.startswith("cond_") and k != "cond_mode"}
    with torch.no_grad():
        ws = G.mapping(dev(g["z"]), dev(g["c"]), cond, truncation_psi=0.7, truncation_cutoff=4)
        assert np.abs(ws.cpu().numpy() - g["ws"]).max() < 1e-5
        assert np.abs(G.mapping(dev(g["z"]), dev(g["c"]), cond).cpu().numpy() - g["ws_psi1"]).max() < 1e-5
        img = G.synthesis(dev(g["ws"]), cond, noise_mode="const").cpu().numpy()
    assert img.shape == g["img"].shape
    assert rel_err(img, g["img"]) < 1e-4


TRI_RK = {"image_resolution": 512, "disparity_space_sampling": False, "clamp_mode": "softplus",
          "superresolution_module": "training.superresolution.SuperresolutionHybrid8XDC", "c_gen_conditioning_zero": False,
          "gpc_reg_prob": 0.5, "c_scale": 1.0, "superresolution_noise_mode": "none", "density_reg": 0.25,
          "density_reg_p_dist": 0.004, "reg_type": "l1", "decoder_lr_mul": 1.0, "sr_antialias": True, "white_back": True,
          "triplane_depth": 1, "use_triplane": 1, "tanh_rgb_output": False, "box_warp": 0.7, "ray_start": 0.5, "ray_end": 1.5,
          "depth_resolution": 12, "depth_resolution_importance": 12, "avg_camera_radius": 1.0, "avg_camera_pivot": [0, 0, 0]}
TRI_KW = dict(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, sr_num_fp16_res=0,
              mapping_kwargs={"num_layers": 2}, rendering_kwargs=TRI_RK,
              sr_kwargs={"channel_base": 32768, "channel_max": 512, "fused_modconv_default": "inference_only"},
              cond_mode="none", triplane_width=32, sr_channels_hidden=16, backbone_resolution=32, channel_base=1024,
              channel_max=32, fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None)


def test_triplane_generator_f_vs_reference(hip):
    """TriPlaneGenerator.f end to end (seeds -> z -> ws -> planes -> fused renderer -> super-resolution) against the
    reference's own G.f on CPU, one perspective and one orthographic view, with the reference's random draws injected."""
    from panic3d_amd.generator import TriPlaneGenerator
    g = T.load_golden("syn_triplane_f.npz")
    G = load_sd(TriPlaneGenerator(**TRI_KW), g, "sd_")
    G.set_force_sigmoid(True)
    G._inject_draws = (dev(g["jitter"]), dev(g["u"]))
    x = dict(elevations=torch.tensor([0.0, 10.0]).cuda(), azimuths=torch.tensor([20.0, 200.0]).cuda(),
             fovs=torch.tensor([30.0, -1.0]).cuda(), seeds=[3, 4], cond={}, triplane_crop=0.1, cull_clouds=0.5,
             neural_rendering_resolution=16)
    with torch.no_grad():
        out = G.f(x)
        assert np.abs(x["camera_params"].cpu().numpy() - g["camera_params"]).max() < 1e-6
        assert np.abs(x["ws"].cpu().numpy() - g["ws"]).max() < 1e-5
        assert rel_err(out["triplane"].cpu().numpy(), g["triplane"]) < 1e-4
        # the renderer consumes the HIP planes (not the golden ones), so differences of ~1e-5 in the planes move a few
        # importance samples; compare at image level with a loose bound and a tight mean bound
        for k, tol in (("image_raw", 2e-3), ("image_weights", 2e-3), ("image_xyz", 2e-3)):
            d = np.abs(out[k].cpu().numpy() - g[k])
            assert d.max() < 20 * tol and d.mean() < tol, (k, d.max(), d.mean())
        d = np.abs(out["image"][..., ::4, ::4].cpu().numpy() - g["image_sub4"])
        assert out["image"].shape == (2, 3, 512, 512) and d.mean() < 2e-3 and d.max() < 0.1, (d.mean(), d.max())
        # PSNR of the final 512^2 image against the reference's G.f (images in [-1,1]: peak-to-peak 2)
        mse = float(np.mean((out["image"][..., ::4, ::4].cpu().numpy().astype(np.float64) - g["image_sub4"]) ** 2))
        assert 10 * np.log10(4.0 / mse) > 50.0, 10 * np.log10(4.0 / mse)
        sm = G.sample_mixed(dev(g["sm_pts"]), None, dev(g["ws"]), {}, noise_mode="const")
        assert rel_err(sm["sigma"].cpu().numpy(), g["sm_sigma"]) < 1e-3 and np.abs(sm["rgb"].cpu().numpy() - g["sm_rgb"]).max() < 1e-3


def test_paste_front_vs_reference(hip):
    """f() with paste_params (generate.py:55-66): masks, the extra front-occlusion render pass, grid_sample of the input
    illustration and the final lerp, against the reference's paste_front (with kornia's Sobel restated, see paste.py)."""
    from panic3d_amd.generator import TriPlaneGenerator
    g = T.load_golden("syn_triplane_f.npz")
    G = load_sd(TriPlaneGenerator(**TRI_KW), g, "sd_")
    G.set_force_sigmoid(True)
    G._inject_draws = [(dev(g["paste_draw0"]), dev(g["paste_draw1"])), (dev(g["paste_draw2"]), dev(g["paste_draw3"]))]
    front = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(11))
    xp = dict(elevations=torch.tensor([0.0]).cuda(), azimuths=torch.tensor([0.0]).cuda(), fovs=torch.tensor([-1.0]).cuda(),
              seeds=[3], cond={"image_ortho_front": front.cuda()}, triplane_crop=0.1, cull_clouds=0.5,
              neural_rendering_resolution=16,
              paste_params={"mode": "default", "thresh_weight": 0.5, "thresh_edges": 0.2, "thresh_occ": 0.5,
                            "offset_occ": 0.01, "thresh_dxyz": 0.05})
    with torch.no_grad():
        out = G.f(xp)
    assert G._inject_draws == []  # both renderer passes ran
    for k in ("mask", "mask_weights", "mask_edges", "mask_occ", "mask_dxyz"):
        a, b = out["paste"][k].cpu().numpy(), g["paste_" + k]
        assert a.shape == b.shape and (np.abs(a - b) > 1e-3).mean() < 0.01, k  # thresholded masks: <1 % boundary pixels
    m = g["paste_mask"]
    assert 0.01 < m.mean() < 0.99  # the fixture exercises both pasted and kept pixels
    sub = lambda t: t[..., ::4, ::4].cpu().numpy()
    assert np.abs(sub(out["paste"]["paste"]) - g["paste_paste_sub4"]).mean() < 2e-3
    assert np.abs(sub(out["image_prepaste"]) - g["paste_prepaste_sub4"]).mean() < 2e-3
    assert np.abs(sub(out["image"]) - g["paste_image_sub4"]).mean() < 5e-3


def test_paste_front_fused_kernel_vs_torch_formulation(hip):
    """p3d_paste_front_f32 (one launch) against the same post-process written as the reference writes it — torch resizes,
    Sobel on shifted slices, grid_sample, lerp (paste.paste_front_torch): masks identical except for pixels whose value sits
    on a threshold (< 0.1 %), paste / image to fp32 round-off.  Perspective and orthographic view, one and three views."""
    from panic3d_amd.generator import TriPlaneGenerator
    from panic3d_amd import paste
    g = T.load_golden("syn_triplane_f.npz")
    G = load_sd(TriPlaneGenerator(**TRI_KW), g, "sd_")
    G.set_force_sigmoid(True)
    gen = torch.Generator().manual_seed(21)
    front = torch.rand(1, 3, 512, 512, generator=gen).cuda()
    pp = {"mode": "default", "thresh_weight": 0.5, "thresh_edges": 0.2, "thresh_occ": 0.5, "offset_occ": 0.01, "thresh_dxyz": 0.05}
    for res, el, az, fv in ((16, [0.0], [0.0], [-1.0]), (32, [5.0], [30.0], [30.0])):
        R, S = res * res, 12
        draws = [(torch.rand(1, R, S, 1, generator=gen).cuda(), torch.rand(R, S, generator=gen).cuda()) for _ in range(2)]
        x = dict(elevations=torch.tensor(el).cuda(), azimuths=torch.tensor(az).cuda(), fovs=torch.tensor(fv).cuda(), seeds=[3],
                 cond={"image_ortho_front": front}, triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=res,
                 noise_mode="const", paste_params=pp)
        with torch.no_grad():
            G._inject_draws = [tuple(d) for d in draws]
            out = G.f(x)
            ret = {k: out[k] for k in ("image_raw", "image_depth", "image_weights", "triplane", "image_xyz", "normalize_images")}
            ret["image"] = out["image_prepaste"]
            G._inject_draws = [tuple(draws[1])]  # the occlusion pass of the comparator consumes the same draws
            ref = paste.paste_front_torch(G, x, ret, **pp)
        fused = out["paste"]
        assert 0.005 < float(ref["mask"].mean()) < 0.995
        for k in ("mask", "mask_weights", "mask_edges", "mask_occ", "mask_dxyz"):
            assert fused[k].shape == ref[k].shape
            assert float(((fused[k] - ref[k]).abs() > 1e-4).float().mean()) < 1e-3, k
        same = (fused["mask"] - ref["mask"]).abs() <= 1e-4
        # the illustration is white noise (gradient ~1 per texel) sampled at coordinates near 512, whose fp32 spacing is 6e-5
        assert float((fused["paste"] - ref["paste"]).abs().max()) < 2e-4 and float((fused["paste"] - ref["paste"]).abs().mean()) < 1e-5
        assert float(((fused["image"] - ref["image"]).abs() * same).max()) < 2e-4
    G._inject_draws = None


def test_density_grid(hip):
    """volume.density_grid == get_eg3d_volume's loop (sample_mixed per chunk + sigma2density + crop/cull on densities)."""
    from panic3d_amd.generator import TriPlaneGenerator
    g = T.load_golden("syn_triplane_f.npz")
    G = load_sd(TriPlaneGenerator(**TRI_KW), g, "sd_")
    G.set_force_sigmoid(True)
    ws = dev(g["ws"])[:1]
    N = 24
    with torch.no_grad():
        out = hip.volume.density_grid(G, ws, {}, resolution=N, triplane_crop=0.1, cull_clouds=0.5)
        pts = hip.volume.create_samples(N, cube_length=0.7)[0].cuda()  # the reference builds the grid on the CPU (eg3d_metrics3d.py:111)
        ref = G.sample_mixed(pts.contiguous(), None, ws, {}, noise_mode="const")["sigma"]
        assert torch.equal(out["sigmas"], ref)
        dens = hip.volume.sigma2density(ref)  # the reference's torch formulation
        dens[(pts[..., 0].abs() > 0.25) | (pts[..., 2].abs() > 0.25)] = -1e3
        dens[hip.volume.sigma2density(dens) < 0.5] = -1e3
        got = out["densities"]
        same = (got == -1e3) == (dens == -1e3)  # thresholded: a voxel exactly on the cull threshold may flip
        assert same.float().mean() > 0.999 and (dens == -1e3).any()  # (the quirk culls all but saturated voxels)
        assert (got - dens)[same].abs().max() < 1e-6
        vol = hip.volume.to_volume(out["densities"], N)
        assert vol.shape == (1, 1, N, N, N)
        # skip_cropped: masked points are not decoded — the densities are identical, the sigmas of masked points read -1000
        sk = hip.volume.density_grid(G, ws, {}, resolution=N, triplane_crop=0.1, cull_clouds=0.5, skip_cropped=True)
        assert torch.equal(sk["densities"], out["densities"])
        cropped = (pts[..., 0].abs() > 0.25) | (pts[..., 2].abs() > 0.25)
        assert 0.3 < cropped.float().mean() < 0.7
        assert torch.equal(sk["sigmas"][~cropped], out["sigmas"][~cropped]) and (sk["sigmas"][cropped] == -1000).all()
        half = hip.volume.density_grid(G, ws, {}, resolution=N, lo=0, hi=N ** 3 // 2)
        assert torch.equal(half["sigmas"], out["sigmas"][:, :N ** 3 // 2])


def test_f_many_views_of_one_subject_in_one_call(hip):
    """Extension of the dict API: ws / cond of batch 1 with V cameras renders V views from ONE backbone pass and ONE fused
    renderer launch (shared planes).  Equal to V separate f() calls on the same planes and random draws: renderer outputs
    bit for bit (rays are independent; the depth clamp is kept per view), super-resolved / pasted images to fp32 round-off
    (the batch size changes the split-K choice of the small conv layers)."""
    from panic3d_amd.generator import TriPlaneGenerator
    g = T.load_golden("syn_triplane_f.npz")
    G = load_sd(TriPlaneGenerator(**TRI_KW), g, "sd_")
    G.set_force_sigmoid(True)
    V, res, S = 3, 16, 12
    R = res * res
    gen = torch.Generator().manual_seed(5)
    draws = [(torch.rand(V, R, S, 1, generator=gen).cuda(), torch.rand(V * R, S, generator=gen).cuda()) for _ in range(2)]
    front = torch.rand(1, 3, 512, 512, generator=gen).cuda()
    el, az, fv = torch.tensor([0.0, 10.0, -5.0]).cuda(), torch.tensor([0.0, 40.0, 200.0]).cuda(), torch.tensor([-1.0, 30.0, 30.0]).cuda()
    common = dict(seeds=[3], cond={"image_ortho_front": front}, triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=res,
                  noise_mode="const", paste_params={"mode": "default", "thresh_weight": 0.5, "thresh_edges": 0.2, "thresh_occ": 0.5,
                                                    "offset_occ": 0.01, "thresh_dxyz": 0.05})
    with torch.no_grad():
        with pytest.raises(RuntimeError):  # this fixture's generator is pose-conditioned: ws would differ per view
            G.f(dict(common, elevations=el, azimuths=az, fovs=fv))
        x0 = dict(common, elevations=el[:1], azimuths=az[:1], fovs=fv[:1], paste_params=None)
        G.f(x0)
        common["ws"] = x0["ws"]  # one subject: the same ws for every view
        G._inject_draws = [tuple(d) for d in draws]
        both = G.f(dict(common, elevations=el, azimuths=az, fovs=fv))
        assert G._inject_draws == [] and both["image"].shape == (V, 3, 512, 512) and both["triplane"].shape[0] == V
        for v in range(V):
            G._inject_draws = [(j[v:v + 1].contiguous(), u[v * R:(v + 1) * R].contiguous()) for j, u in draws]
            one = G.f(dict(common, elevations=el[v:v + 1], azimuths=az[v:v + 1], fovs=fv[v:v + 1]))
            for k in ("image_raw", "image_weights", "image_xyz", "image_depth"):  # depth too: one clamp range per view
                assert torch.equal(both[k][v:v + 1], one[k]), (k, v)
            assert (both["image_prepaste"][v:v + 1] - one["image_prepaste"]).abs().max() < 1e-4
            assert ((both["paste"]["mask"][v:v + 1] - one["paste"]["mask"]).abs() > 1e-3).float().mean() < 1e-3
            assert (both["image"][v:v + 1] - one["image"]).abs().mean() < 1e-4
    G._inject_draws = None


@pytest.mark.parametrize("I,O,H,up,ks,N", [(32, 64, 24, 1, 3, 2), (64, 40, 20, 2, 3, 1), (48, 3, 33, 1, 1, 2), (256, 128, 64, 2, 3, 1),
                                           (128, 128, 96, 1, 3, 1)])
def test_modconv_f16_operands_vs_fp32(hip, I, O, H, up, ks, N):
    """The opt-in f16-operand convolution (fp32 accumulate) against (a) the exact-fp32 HIP convolution: error at the level of
    the f16 rounding of the operands (2^-11 relative per operand, averaged over K terms); (b) the same fp32 convolution fed
    with operands pre-rounded to f16 by torch: only fp32 summation-order differences remain."""
    ops = hip.ops
    g = torch.Generator().manual_seed(I + O)
    x = torch.randn(N, I, H, H, generator=g).cuda(); w = torch.randn(O, I, ks, ks, generator=g).cuda()
    s = (torch.randn(N, I, generator=g) * 0.5 + 1).cuda(); b = torch.randn(O, generator=g).cuda()
    f = ops.setup_filter([1, 3, 3, 1]).cuda()
    kw = dict(up=up, padding=ks // 2, resample_filter=f, demodulate=ks == 3, bias=b, act="lrelu" if ks == 3 else "linear")
    wh = ops.conv_weights_to_f16(w)
    assert wh.shape == (O, ks * ks, I) and torch.equal(wh, w.reshape(O, I, ks * ks).permute(0, 2, 1).half())
    y32 = ops.modulated_conv2d(x, w, s, **kw)
    y16 = ops.modulated_conv2d(x, w, s, weight_f16=wh, **kw)
    scale = y32.abs().mean()
    assert (y16 - y32).abs().mean() < 1e-3 * scale and (y16 - y32).abs().max() < 2e-2 * y32.abs().max()
    if ks == 1:  # (b) without demodulation the rounded-operand convolution can be stated exactly: x' = f16(s*x) with s = 1
        xs = (x * s[:, :, None, None]).half().float()
        ref = ops.modulated_conv2d(xs, w.half().float(), torch.ones_like(s), **kw)
        assert (y16 - ref).abs().max() < 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("I,O,H,up,ks,N", [(32, 64, 24, 1, 3, 2), (64, 40, 20, 2, 3, 1), (48, 3, 33, 1, 1, 2), (256, 128, 64, 2, 3, 1),
                                           (128, 128, 96, 1, 3, 1), (512, 512, 16, 1, 3, 1)])
def test_modconv_f16x2_operands_are_fp32_class(hip, I, O, H, up, ks, N):
    """The two-term f16-operand convolution (hi + lo operands, fp32 accumulate) measured against a float64 evaluation of the same
    layer, next to the exact-fp32 HIP convolution: its error has to be of the size of the fp32 kernel's own rounding /
    summation-order error, three orders of magnitude below the one-term f16 variant."""
    ops = hip.ops
    g = torch.Generator().manual_seed(I + O)
    x = torch.randn(N, I, H, H, generator=g).cuda(); w = torch.randn(O, I, ks, ks, generator=g).cuda()
    s = (torch.randn(N, I, generator=g) * 0.5 + 1).cuda(); b = torch.randn(O, generator=g).cuda()
    f = ops.setup_filter([1, 3, 3, 1]).cuda()
    kw = dict(up=up, padding=ks // 2, resample_filter=f, demodulate=ks == 3, bias=b, act="lrelu" if ks == 3 else "linear")
    w2 = ops.conv_weights_to_f16(w, split=True)
    wsc = w.reshape(O, I, ks * ks).permute(0, 2, 1) * 64.0  # stored scaled by 2^6 (the matrix cores flush f16 subnormals)
    hi = wsc.half()
    assert w2.shape == (2, O, ks * ks, I) and torch.equal(w2[0], hi)
    assert torch.equal(w2[1], (wsc - hi.float()).half())
    ref = torch_modconv_ref(x.double().cpu(), w.double().cpu(), s.double().cpu(), None, up, ks == 3, b.double().cpu(), f.double().cpu())
    ref = (F.leaky_relu(ref, 0.2) * np.sqrt(2) if ks == 3 else ref).cuda()
    y32 = ops.modulated_conv2d(x, w, s, **kw).double()
    yx2 = ops.modulated_conv2d(x, w, s, weight_f16=w2, **kw).double()
    y16 = ops.modulated_conv2d(x, w, s, weight_f16=ops.conv_weights_to_f16(w), **kw).double()
    e32, ex2, e16 = ((y - ref).abs().mean().item() for y in (y32, yx2, y16))
    m32, mx2 = ((y - ref).abs().max().item() for y in (y32, yx2))
    scale = ref.abs().mean().item()
    print(f"I={I} O={O} H={H} up={up} ks={ks}: mean |err| fp32 {e32:.3e}  f16x2 {ex2:.3e}  f16 {e16:.3e} | max fp32 {m32:.3e}  f16x2 {mx2:.3e} | mean |y| {scale:.3f}")
    assert e32 < 1e-6 * scale, (e32, scale)          # the fp32 kernel itself
    # operands carry 22 significant bits (fp32: 24): the mean error stays at the fp32 kernel's (summation dominates), single
    # outputs can be off by a few times more
    assert ex2 < 3 * e32 and mx2 < 8 * m32, (ex2, e32, mx2, m32)
    assert ex2 < 1e-2 * e16, (ex2, e16)


def test_modconv_f16x2_saturation_is_reported(hip):
    """Outside its domain (|s*x| > 4094 = 65504 / 16) the two-term convolution saturates the operand — finite, wrong — and says so
    in the CALLER's flag word (ABI 5: no flag lives in the library); inside it the result is fp32-class.  ADVICE r02: the band
    4094 < |s*x| <= 8188 used to be clamped silently — probed at 6000."""
    ops = hip.ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 32, 16, 16, generator=g).cuda(); w = torch.randn(64, 32, 3, 3, generator=g).cuda()
    s = torch.ones(1, 32).cuda()
    w2 = ops.conv_weights_to_f16(w, split=True)
    flag, other = ops.conv_domain_flag(x.device), ops.conv_domain_flag(x.device)
    y = ops.modulated_conv2d(x * 100, w, s, padding=1, weight_f16=w2, saturated=flag)
    assert not ops.conv_domain_violated(flag) and torch.isfinite(y).all()
    y32 = ops.modulated_conv2d(x * 100, w, s, padding=1)
    assert (y - y32).abs().max() < 1e-5 * y32.abs().max()
    xe = x.clone(); xe[0, 3, 5, 5] = 4000.0  # the edge of the domain: still exact to fp32 class
    ye = ops.modulated_conv2d(xe, w, s, padding=1, weight_f16=w2, saturated=flag)
    ye32 = ops.modulated_conv2d(xe, w, s, padding=1)
    assert not ops.conv_domain_violated(flag) and (ye - ye32).abs().max() < 1e-5 * ye32.abs().max()
    for big in (6000.0, 9000.0, float("nan")):
        xb = x.clone(); xb[0, 3, 5, 5] = big
        y = ops.modulated_conv2d(xb, w, s, padding=1, weight_f16=w2, saturated=flag)
        assert torch.isfinite(y).all()
        assert ops.conv_domain_violated(flag, reset=True) and not ops.conv_domain_violated(flag), big
        ops.modulated_conv2d(xb, w, s, padding=1, weight_f16=w2)  # no flag passed: nothing to report to, must not fault
    assert not ops.conv_domain_violated(other)  # a flag of another caller never moves


def test_generator_conv_mma_modes_agree(hip):
    """TriPlaneGenerator.set_conv_mma: the two-term f16 convolutions reproduce the fp32-operand image to fp32-class accuracy
    through the whole backbone + renderer + super-resolution stack (PSNR > 100 dB on [-1,1] images), the one-term ones do not."""
    from panic3d_amd.generator import TriPlaneGenerator
    g = T.load_golden("syn_triplane_f.npz")
    G = load_sd(TriPlaneGenerator(**TRI_KW), g, "sd_")
    G.set_force_sigmoid(True)
    x = dict(elevations=torch.tensor([0.0]).cuda(), azimuths=torch.tensor([20.0]).cuda(), fovs=torch.tensor([30.0]).cuda(),
             seeds=[3], cond={}, triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=16, noise_mode="const")
    out = {}
    with torch.no_grad():
        G.watch_conv_domain()
        for mode in ("f32", "x2", "f16"):
            G.set_conv_mma(mode)
            G._inject_draws = (dev(g["jitter"])[:1], dev(g["u"])[:256])
            out[mode] = G.f(dict(x))
    G._inject_draws = None
    G.set_conv_mma(None)
    assert not G.conv_domain_violated()
    used = [m for m in list(G.backbone.modules()) + list(G.superresolution.modules()) if getattr(m, "_wh", None) is not None]
    assert len(used) >= 6
    psnr = lambda a, b: 10 * np.log10(4.0 / max(float(((a - b).double() ** 2).mean()), 1e-30))
    p_x2, p_16 = psnr(out["x2"]["image"], out["f32"]["image"]), psnr(out["f16"]["image"], out["f32"]["image"])
    print(f"PSNR vs fp32 operands: two-term {p_x2:.1f} dB, one-term {p_16:.1f} dB; planes max |diff| "
          f"{(out['x2']['triplane'] - out['f32']['triplane']).abs().max().item():.2e}")
    assert p_x2 > 100 and p_16 < p_x2 - 30


def test_sr_f16_operands_image_quality(hip):
    """TriPlaneGenerator with set_sr_mma_f16: the 512^2 image stays within f16-operand rounding of the fp32 image
    (PSNR > 50 dB on [-1,1] images; the reference's own fp16 blocks round activations as well)."""
    from panic3d_amd.generator import TriPlaneGenerator
    g = T.load_golden("syn_triplane_f.npz")
    G = load_sd(TriPlaneGenerator(**TRI_KW), g, "sd_")
    G.set_force_sigmoid(True)
    x = dict(elevations=torch.tensor([0.0]).cuda(), azimuths=torch.tensor([20.0]).cuda(), fovs=torch.tensor([30.0]).cuda(),
             seeds=[3], cond={}, triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=16, noise_mode="const")
    with torch.no_grad():
        G._inject_draws = (dev(g["jitter"])[:1], dev(g["u"])[:256])
        a = G.f(dict(x))["image"]
        G.set_sr_mma_f16(True)
        G._inject_draws = (dev(g["jitter"])[:1], dev(g["u"])[:256])
        b = G.f(dict(x))["image"]
    G._inject_draws = None
    used = [m for m in G.superresolution.modules() if getattr(m, "_wh", None) is not None]
    assert len(used) >= 3  # the 16-channel-aligned layers really took the f16 path
    mse = float(((a - b).double() ** 2).mean())
    assert mse > 0 and 10 * np.log10(4.0 / mse) > 50.0, 10 * np.log10(4.0 / max(mse, 1e-30))


def test_f_rays_from_the_view_cache_equal_caller_given_rays(hip):
    """G.f hands synthesis() the cached view's rays already in the renderer's [N,R,3] layout; a caller who passes the same rays as
    x['force_rays'] ([N,3,res,res], the reference's layout) goes through the permute + copy — both must render the same bits, for one
    view (a plain view of the cached tensor) and for several views in one call (stacked), perspective and orthographic."""
    from panic3d_amd.generator import TriPlaneGenerator
    g = T.load_golden("syn_triplane_f.npz")
    G = load_sd(TriPlaneGenerator(**dict(TRI_KW, rendering_kwargs=dict(TRI_RK, c_gen_conditioning_zero=True))), g, "sd_")
    G.set_force_sigmoid(True)
    for elev, azim, fov in (([0.0], [20.0], [30.0]), ([0.0], [90.0], [-1.0]), ([0.0, 10.0], [20.0, -40.0], [30.0, 30.0])):
        V = len(elev)
        x = dict(elevations=torch.tensor(elev).cuda(), azimuths=torch.tensor(azim).cuda(), fovs=torch.tensor(fov).cuda(), seeds=[3], cond={},
                 triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=16, noise_mode="const")
        draws = lambda: (torch.rand(V, 256, TRI_RK["depth_resolution"], 1, generator=torch.Generator().manual_seed(1)).cuda(),
                         torch.rand(V * 256, TRI_RK["depth_resolution_importance"], generator=torch.Generator().manual_seed(2)).cuda())
        with torch.no_grad():
            G._inject_draws = draws()
            xa = dict(x)
            a = G.f(xa)
            G._inject_draws = draws()
            xb = dict(x, force_rays={k: v.clone() for k, v in xa["force_rays"].items()}, camera_params=xa["camera_params"].clone())
            b = G.f(xb)
        G._inject_draws = None
        for k in ("image", "image_raw", "image_depth", "image_weights", "image_xyz"):
            assert torch.equal(a[k], b[k]), (k, fov)
        assert a["image_xyz"].shape == (V, 3, 16, 16) and a["image"].shape == (V, 3, 512, 512)


def test_raw_stream_handle_follows_the_current_stream(hip):
    """ops._stream() (torch._C._cuda_getCurrentRawStream: the C entry point torch.cuda.current_stream() ends in) is the stream the
    C ABI launches on: it must follow torch.cuda.stream(...) contexts, and work issued under one must be ordered on that stream."""
    ops = hip.ops
    s = torch.cuda.Stream()
    assert (ops._stream().value or 0) == torch.cuda.current_stream().cuda_stream
    with torch.cuda.stream(s):
        assert (ops._stream().value or 0) == s.cuda_stream
        x = torch.randn(1, 16, 8, 8, device="cuda")
        y = ops.bias_act(x, torch.ones(16, device="cuda"), act="lrelu")
    assert (ops._stream().value or 0) == torch.cuda.current_stream().cuda_stream
    s.synchronize()
    assert torch.allclose(y, torch.nn.functional.leaky_relu(x + 1, 0.2) * np.sqrt(2), atol=1e-6)


def test_triplane_generator_f_conditioned_vs_reference(hip):
    """G.f of a conditioned generator (front illustration + resnet features + resnet 'chonk', pose conditioning zeroed: what
    _scripts/eval/generate.py:88-96 feeds the released model) against the reference's own G.f on CPU: mapping_zplus with
    resnet features, the conditioned backbone, fused renderer, super-resolution."""
    from panic3d_amd.generator import TriPlaneGenerator
    g = T.load_golden("syn_triplane_f_cond.npz")
    kw = dict(TRI_KW, cond_mode=str(g["cond_mode"]), rendering_kwargs=dict(TRI_RK, c_gen_conditioning_zero=True))
    G = load_sd(TriPlaneGenerator(**kw), g, "sd_")
    G.set_force_sigmoid(True)
    G._inject_draws = (dev(g["jitter"]), dev(g["u"]))
    cond = {k[5:]: dev(v) for k, v in g.items() if k.startswith("cond_") and k != "cond_mode"}
    x = dict(elevations=torch.tensor([5.0]).cuda(), azimuths=torch.tensor([-30.0]).cuda(), fovs=torch.tensor([30.0]).cuda(), seeds=[7],
             cond=cond, triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=16, noise_mode="const")
    with torch.no_grad():
        out = G.f(x)
    G._inject_draws = None
    assert np.abs(x["ws"].cpu().numpy() - g["ws"]).max() < 1e-5
    assert rel_err(out["triplane"].cpu().numpy(), g["triplane"]) < 1e-4
    for k, tol in (("image_raw", 2e-3), ("image_weights", 2e-3), ("image_xyz", 2e-3)):
        d = np.abs(out[k].cpu().numpy() - g[k])
        assert d.max() < 20 * tol and d.mean() < tol, (k, d.max(), d.mean())
    d = np.abs(out["image"][..., ::4, ::4].cpu().numpy() - g["image_sub4"])
    assert d.mean() < 2e-3 and d.max() < 0.1, (d.mean(), d.max())
    assert 0.05 < float(out["image_weights"].mean()) < 0.999  # the fixture shows both surface and background


def test_density_grid_vs_the_references_get_eg3d_volume(hip):
    """volume.density_grid / to_volume / create_samples against the output of the reference's OWN get_eg3d_volume
    (_util/eg3d_metrics3d.py:94-183, executed from its source text by tests/golden/make_golden_synthesis.py volume_case):
    sigmas, densities with and without the crop / cull masks, the rgb grid and the coordinate grid, in the reference's final
    [1,C,N,N,N] layout (first axis flipped)."""
    from panic3d_amd.generator import TriPlaneGenerator
    gv, gt = T.load_golden("volume_reference.npz"), T.load_golden("syn_triplane_f.npz")
    G = load_sd(TriPlaneGenerator(**TRI_KW), gt, "sd_")
    G.set_force_sigmoid(True)
    N = int(gv["resolution"])
    vol = hip.volume
    with torch.no_grad():
        x = dict(elevations=torch.zeros(1).cuda(), azimuths=torch.zeros(1).cuda(), seeds=[3], cond={}, neural_rendering_resolution=8)
        G.f(x)  # the reference obtains ws the same way (eg3d_metrics3d.py:101-109)
        ws = x["ws"]
        plain = vol.density_grid(G, ws, {}, resolution=N)
        masked = vol.density_grid(G, ws, {}, resolution=N, triplane_crop=0.1, cull_clouds=0.5)
        pts = vol.create_samples(N, cube_length=0.7)[0]
        rgb = G.sample_mixed(pts.cuda().contiguous(), None, ws, {}, noise_mode="const")["rgb"]
    assert np.array_equal(vol.to_volume(pts, N).numpy(), gv["plain_coordinates"])  # same float ops on the CPU
    sig = vol.to_volume(plain["sigmas"], N).cpu().numpy()
    assert sig.shape == gv["plain_sigmas"].shape == (1, 1, N, N, N) and rel_err(sig, gv["plain_sigmas"]) < 1e-3
    assert np.abs(vol.to_volume(plain["densities"], N).cpu().numpy() - gv["plain_densities"]).max() < 2e-4
    assert np.abs(vol.to_volume(rgb, N)[:, :3].cpu().numpy() - gv["plain_rgb3"]).max() < 2e-3
    dm, ref = vol.to_volume(masked["densities"], N).cpu().numpy(), gv["masked_densities"]
    same = (dm == -1e3) == (ref == -1e3)
    assert same.mean() > 0.995 and (ref == -1e3).mean() > 0.5  # thresholded masks: only boundary voxels may flip
    both = same & (ref != -1e3)
    assert both.sum() > 10 and np.abs(dm[both] - ref[both]).max() < 2e-4
    # and the mesh of generate.py:98-103 from the reference's own volume runs through the device extractor
    mc = vol.marching_cubes(torch.from_numpy(gv["plain_densities"][0, 0]).cuda(), torch.from_numpy(gv["plain_rgb3"][0]).cuda(), 0.7, level=0.5)
    assert len(mc["faces"]) > 50 and mc["colors"].shape == (len(mc["verts"]), 3) and np.abs(mc["verts"]).max() <= 0.35 + 1e-6


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("conv", ["f32", "x2"])
def test_fullsize_generator_vs_reference(hip, conv, exact):
    """The FULL-WIDTH generator (30 M parameters: StyleGAN2-256 backbone with 512-channel layers, 96-channel planes,
    SuperresolutionHybrid8XDC with 256 hidden channels — the released model's constructor kwargs) against the
    reference's own run on CPU (tests/golden/make_golden_fullsize.py).  No checkpoint exists here, so both sides replay the
    same seeded parameter recipe (p3d_testing.fill_generator_params).  Both convolution operand modes (fp32 / two-term f16) and
    both renderer modes (exact / tolerance).  VERDICT r02 item 7: G.f parity used to exist at toy widths only."""
    from panic3d_amd.generator import TriPlaneGenerator
    g = T.load_golden("fullsize_generator.npz")
    G = T.fill_generator_params(TriPlaneGenerator(**T.FULL_KW), int(g["seed"])).cuda().eval()
    assert sum(p.numel() for p in G.parameters()) == 30662136
    G.set_force_sigmoid(True)
    G.set_conv_mma(conv)
    G.set_render_exact(exact)
    nrr, R = int(g["nrr"]), int(g["nrr"]) ** 2
    jit, u = T.make_random_draws(int(g["draw_seed"]), 1, R, 48, 48)
    G._inject_draws = (dev(jit), dev(u))
    c = dev(g["camera_params"])
    psnr = lambda a, b: 10 * np.log10(4.0 / max(float(np.mean((a.astype(np.float64) - b) ** 2)), 1e-30))  # images in [-1, 1]
    with torch.no_grad():
        G.watch_conv_domain()
        ws = G.mapping(dev(g["z"]), c, {})
        assert rel_err(ws.cpu().numpy(), g["ws"]) < 1e-5
        out = G.synthesis(dev(g["ws"]), c, {}, neural_rendering_resolution=nrr, noise_mode="const", triplane_crop=0.1, cull_clouds=0.5)
    assert not G.conv_domain_violated()
    planes = out["triplane"][..., ::8, ::8].cpu().numpy()
    pe = rel_err(planes, g["planes_sub8"])
    p_img, p_raw = psnr(out["image"][..., ::4, ::4].cpu().numpy(), g["image_sub4"]), psnr(out["image_raw"].cpu().numpy(), g["image_raw"])
    dw = np.abs(out["image_weights"].cpu().numpy() - g["image_weights"])
    dd = np.abs(out["image_depth"].cpu().numpy() - g["image_depth"])
    print(f"full-size generator [{conv}, {'exact' if exact else 'tolerance'}]: planes rel err {pe:.2e}, PSNR image {p_img:.1f} dB, "
          f"image_raw {p_raw:.1f} dB, weights mean abs {dw.mean():.2e}, depth mean abs {dd.mean():.2e}")
    assert out["image"].shape == (1, 3, 512, 512) and pe < 2e-5
    assert p_img >= 100.0 and p_raw >= 100.0, (p_img, p_raw)  # measured 118-123 dB (VERDICT asked for >= 60)
    assert dw.mean() < 1e-3 and dd.mean() < 1e-3


@pytest.mark.parametrize("mma", ["f32", "x2"])
def test_up_conv_fir_sums_shallow_splitk_partials(hip, mma):
    """Round 3: for shallow split-K (<= 8 slices) of the up-sampling layer the FIR pass sums the partial slices itself (no separate
    reduce launch).  Same slice-ordered sum: the result must not depend on whether a layer's split is shallow (FIR sums) or deep
    (k_splitk_reduce) — checked against the torch fp32 formulation at both kinds of shape, and for run-to-run determinism."""
    ops = hip.ops
    g = torch.Generator().manual_seed(3)
    rn = lambda *s: torch.randn(*s, generator=g)
    filt = ops.setup_filter([1, 3, 3, 1])
    for (N, I, O, H) in ((1, 512, 512, 4), (1, 512, 256, 64), (2, 256, 128, 32), (1, 256, 128, 128)):  # 64 / 8 / 4 / 2 slices at batch 1
        x, w3, s = rn(N, I, H, H), rn(O, I, 3, 3), rn(N, I) * 0.2 + 1.0
        b, noise = rn(O) * 0.1, rn(2 * H, 2 * H) * 0.1
        wh = ops.conv_weights_to_f16(w3.cuda(), split=True) if mma == "x2" else None
        kw = dict(noise=noise.cuda(), up=2, padding=1, resample_filter=filt.cuda(), bias=b.cuda(), act="lrelu", weight_f16=wh)
        a = ops.modulated_conv2d(x.cuda(), w3.cuda(), s.cuda(), **kw)
        c = ops.modulated_conv2d(x.cuda(), w3.cuda(), s.cuda(), **kw)
        assert torch.equal(a, c)
        ref = F.leaky_relu(torch_modconv_ref(x, w3, s, noise, 2, True, b, filt), 0.2) * np.sqrt(2)
        assert rel_err(a.cpu().numpy(), ref.numpy()) < 3e-6 * np.sqrt(I * 9), (N, I, O, H)


@pytest.mark.parametrize("N,I,O,H", [(1, 128, 96, 256), (1, 512, 96, 64), (2, 256, 96, 32), (1, 512, 96, 4), (1, 128, 3, 512), (3, 64, 96, 16), (1, 72, 40, 18)])
def test_torgb_gemm_kernel_vs_the_three_launch_form(hip, N, I, O, H):
    """p3d_torgb_f32 (round 3: ToRGB as a GEMM that reads its activation once, the skip image added in the same launch) against
    the form it replaces — p3d_modconv2d_f32 (1x1) [+ k_splitk_reduce] + p3d_upsample2d_add_f32.  Large maps (no split-K in the
    old kernel, no K split across waves in the new one) must agree BIT FOR BIT; small maps sum their K slices in another order:
    fp32 tolerance.  Channel tails (I = 72), O < 32 (the super-resolution's RGB), batch > 1, bias, clamp."""
    ops = hip.ops
    g = torch.Generator().manual_seed(N * 1000 + I + O + H)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    x, w, s, b = rn(N, I, H, H), rn(O, I, 1, 1), rn(N, I) * 0.3 + 1.0, rn(O) * 0.2
    prev, filt = rn(N, O, H // 2, H // 2), ops.setup_filter([1, 3, 3, 1]).cuda()
    wt = ops.torgb_weights(w)
    for clamp in (None, 2.0):
        y_old = ops.modulated_conv2d(x, w, s, demodulate=False, bias=b, clamp=clamp)
        old = ops.upsample2d_add(prev, filt, y_old)
        new = ops.torgb(x, wt, O, s, bias=b, clamp=clamp, skip=prev, skip_filter=filt)
        new_noskip = ops.torgb(x, wt, O, s, bias=b, clamp=clamp)
        scale = float(y_old.abs().max())
        if N * ((H * H + 127) // 128) >= 512:  # PX shape and the old kernel without split-K: the same fma chains
            assert torch.equal(new_noskip, y_old) and torch.equal(new, old), (float((new - old).abs().max()), scale)
        else:
            assert float((new_noskip - y_old).abs().max()) < 3e-6 * np.sqrt(I) * scale and float((new - old).abs().max()) < 3e-6 * np.sqrt(I) * scale
        assert torch.equal(new - 0, ops.torgb(x, wt, O, s, bias=b, clamp=clamp, skip=prev, skip_filter=filt))  # deterministic


@pytest.mark.parametrize("N,I,O,H", [(1, 256, 128, 128), (2, 512, 512, 16), (1, 32, 256, 64), (3, 64, 48, 32), (1, 128, 64, 20), (1, 64, 32, 8),
                                     (2, 24, 32, 32),  # (I = 24: conv0 itself runs on fp32 operands, its FIR pass still writes the image)
                                     (2, 16, 64, 40), (1, 16, 32, 72)])  # (unsplit launches: the FIR pass runs inside k_modconv_up3<true>; 80^2 / 144^2 outputs: ragged 12 x 60 tiles)
def test_activation_image_between_conv0_and_conv1_is_bit_identical(hip, N, I, O, H):
    """Round 3: an up-sampling layer called with next_styles writes the following layer's two-term operand
    (ops.ActImage: split(16 * s1 * y), the pieces k_modconv_w2 would build itself) from its FIR pass instead of the fp32 tensor.
    conv1 on the image must equal conv1 on the fp32 tensor BIT FOR BIT — all split-K depths of conv0 (FIR sums / separate reduce),
    batch > 1, clamp, per-sample noise, a map that is not a multiple of the tile (40^2), channel counts with a tail in the 64-wide
    output tile (O = 48) — and the image itself must be what act_to_image makes of the fp32 result.  Also: the generator's blocks
    really take this path (P3D_CONV_IMG default), and the domain flag trips through the image writer."""
    ops = hip.ops
    g = torch.Generator().manual_seed(N * 7 + I + O + H)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    filt = ops.setup_filter([1, 3, 3, 1]).cuda()
    x, w0, w1 = rn(N, I, H, H), rn(O, I, 3, 3), rn(O, O, 3, 3)
    s0, s1 = rn(N, I) * 0.4 + 1.0, rn(N, O) * 0.4 + 1.0
    d0 = ((w0[None] * s0[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt().contiguous()
    d1 = ((w1[None] * s1[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt().contiguous()
    wf0 = ops.conv_weights_to_f16(w0, split=True) if I % 16 == 0 else None
    wf1 = ops.conv_weights_to_f16(w1, split=True)
    for clamp, per_sample_noise in ((None, False), (1.5, True)):
        nz0 = rn(N, 1, 2 * H, 2 * H) * 0.1 if (per_sample_noise and N > 1) else rn(2 * H, 2 * H) * 0.1
        k0 = dict(up=2, padding=1, resample_filter=filt, demodulate=True, bias=rn(O) * 0.1, act="lrelu", dcoef=d0, noise=nz0, weight_f16=wf0, clamp=clamp)
        k1 = dict(up=1, padding=1, demodulate=True, bias=rn(O) * 0.1, act="lrelu", dcoef=d1, noise=rn(2 * H, 2 * H) * 0.1, weight_f16=wf1)
        mid = ops.modulated_conv2d(x, w0, s0, **k0)
        img = ops.modulated_conv2d(x, w0, s0, next_styles=s1, **k0)
        assert isinstance(img, ops.ActImage) and img.shape == (N, O, 2 * H, 2 * H)
        assert torch.equal(img.data, ops.act_to_image(mid, s1).data)
        if 2 * H >= 32:
            assert torch.equal(ops.modulated_conv2d(img, w1, None, **k1), ops.modulated_conv2d(mid, w1, s1, **k1))
        else:
            with pytest.raises(RuntimeError):  # maps narrower than the wide tile do not stage from images
                ops.modulated_conv2d(img, w1, None, **k1)
    # the domain flag trips through the image WRITER too: conv0 on fp32 operands (it raises nothing itself), no clamp, a result whose
    # 16 * s1 * y leaves the f16 range
    flag = ops.conv_domain_flag(x.device)
    kz = dict(k0, clamp=None, weight_f16=None)
    ops.modulated_conv2d(x, w0, s0, next_styles=s1, saturated=flag, **kz)
    assert not ops.conv_domain_violated(flag)
    ops.modulated_conv2d(x * 3e4, w0, s0, next_styles=s1, saturated=flag, **kz)
    assert ops.conv_domain_violated(flag)
    # round 4: a PLAIN layer called with next_styles returns (fp32 result, its image for the layer that follows) — the next block's
    # up-sampling conv0 stages from the image, ToRGB reads the tensor: same bits as the result alone and as act_to_image of it
    if 2 * H >= 32:
        y_alone = ops.modulated_conv2d(mid, w1, s1, **k1)
        y_both, img_next = ops.modulated_conv2d(mid, w1, s1, next_styles=s0.new_ones(N, O) * 0.75, **k1)
        assert torch.equal(y_both, y_alone) and isinstance(img_next, ops.ActImage)
        assert torch.equal(img_next.data, ops.act_to_image(y_alone, s0.new_ones(N, O) * 0.75).data)
        y_b2, img_n2 = ops.modulated_conv2d(img, w1, None, next_styles=s1, **k1)  # image in, image out (the generator's case)
        assert torch.equal(y_b2, y_alone) and torch.equal(img_n2.data, ops.act_to_image(y_alone, s1).data)
    with pytest.raises(RuntimeError):  # next_styles of the wrong shape
        ops.modulated_conv2d(mid, w1, s1, next_styles=s1[:, :1], **k1)


def test_up_conv_with_the_fir_pass_inside_is_bit_identical(hip, monkeypatch):
    """k_modconv_up3<true> (the FIR pass and the epilogue inside the transposed convolution; chosen for unsplit layers of <= 64 input
    channels, forced here by P3D_UP3_FUSED) writes the image the two-pass path writes, bit for bit: a 16-chunk K loop, batch 2 with
    per-sample noise and a clamp, ragged 12 x 60 output tiles (176^2), and the domain flag."""
    ops = hip.ops
    filt = ops.setup_filter([1, 3, 3, 1]).cuda()
    for N, I, O, H in ((1, 256, 128, 128), (2, 64, 128, 88), (1, 32, 256, 128)):
        g = torch.Generator().manual_seed(I + O + H)
        rn = lambda *s: torch.randn(*s, generator=g).cuda()
        x, w0 = rn(N, I, H, H), rn(O, I, 3, 3)
        s0, s1 = rn(N, I) * 0.4 + 1.0, rn(N, O) * 0.4 + 1.0
        d0 = ((w0[None] * s0[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt().contiguous()
        wf0 = ops.conv_weights_to_f16(w0, split=True)
        nz = rn(N, 1, 2 * H, 2 * H) * 0.1 if N > 1 else rn(2 * H, 2 * H) * 0.1
        k0 = dict(up=2, padding=1, resample_filter=filt, demodulate=True, bias=rn(O) * 0.1, act="lrelu", dcoef=d0, noise=nz, weight_f16=wf0,
                  clamp=1.5 if N > 1 else None, next_styles=s1)
        out = {}
        monkeypatch.setenv("P3D_UP4", "0")  # (round 6: k_modconv_up4 takes these shapes by default; this test pins the round-4 kernel)
        for mode in ("0", "1"):
            monkeypatch.setenv("P3D_UP3_FUSED", mode)
            flag = ops.conv_domain_flag(x.device)
            out[mode] = ops.modulated_conv2d(x, w0, s0, saturated=flag, **k0).data.clone()
            assert not ops.conv_domain_violated(flag)
        assert torch.equal(out["0"], out["1"]), (N, I, O, H)
        monkeypatch.setenv("P3D_UP3_FUSED", "1")
        flag = ops.conv_domain_flag(x.device)
        ops.modulated_conv2d(x * 3e4, w0, s0, saturated=flag, **dict(k0, clamp=None))
        assert ops.conv_domain_violated(flag)


@pytest.mark.parametrize("N,I,O,H", [(1, 256, 128, 128), (2, 64, 128, 88), (1, 32, 256, 128), (2, 128, 256, 64), (2, 16, 64, 40), (1, 16, 32, 72)])
@pytest.mark.parametrize("rpw", ["0", "2"])
def test_up4_one_launch_up_layer_equals_the_two_pass_form(hip, monkeypatch, N, I, O, H, rpw):
    """k_modconv_up4 (round 6: transposed convolution + FIR pass + epilogue in one launch; 16 x 32 grid points per tile on eight waves
    or 8 x 32 on four) against the round-5 form it replaces (k_modconv_up3 + k_fir4x4_img / k_fir4x4_tiled, P3D_UP4=0; shapes whose
    round-5 launch is unsplit, so that both sum in the same order): the activation image AND the fp32 tensor BIT FOR BIT — long and
    short K loops (1, 2, 16 chunks), batch 2 with per-sample noise and a clamp, ragged tiles (80^2, 144^2, 176^2 outputs), both tile
    heights forced, the domain flag, fp32 and image inputs."""
    ops = hip.ops
    filt = ops.setup_filter([1, 3, 3, 1]).cuda()
    g = torch.Generator().manual_seed(I + O + H)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    x, w0 = rn(N, I, H, H), rn(O, I, 3, 3)
    s0, s1 = rn(N, I) * 0.4 + 1.0, rn(N, O) * 0.4 + 1.0
    d0 = ((w0[None] * s0[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt().contiguous()
    wf0 = ops.conv_weights_to_f16(w0, split=True)
    nz = rn(N, 1, 2 * H, 2 * H) * 0.1 if N > 1 else rn(2 * H, 2 * H) * 0.1
    k0 = dict(up=2, padding=1, resample_filter=filt, demodulate=True, bias=rn(O) * 0.1, act="lrelu", dcoef=d0, noise=nz, weight_f16=wf0,
              clamp=1.5 if N > 1 else None)
    ximg = ops.act_to_image(x, s0)
    monkeypatch.setenv("P3D_UP4_RPW", rpw)
    monkeypatch.setenv("P3D_UP4_MIN_WGS", "0")
    monkeypatch.setenv("P3D_UP4_MIN_I", "0")
    monkeypatch.setenv("P3D_UP3_FUSED", "0")
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("P3D_UP4", mode)
        flag = ops.conv_domain_flag(x.device)
        out[mode] = (ops.modulated_conv2d(x, w0, s0, saturated=flag, next_styles=s1, **k0).data.clone(),
                     ops.modulated_conv2d(x, w0, s0, saturated=flag, **k0).clone(),
                     ops.modulated_conv2d(ximg, w0, None, saturated=flag, next_styles=s1, **k0).data.clone())
        assert not ops.conv_domain_violated(flag)
    for a, b, what in zip(out["0"], out["1"], ("image out", "fp32 out", "image in, image out")):
        assert torch.equal(a, b), (what, N, I, O, H, rpw, float((a.float() - b.float()).abs().max()))
    assert torch.equal(out["1"][0], out["1"][2])
    monkeypatch.setenv("P3D_UP4", "1")
    flag = ops.conv_domain_flag(x.device)
    ops.modulated_conv2d(x * 3e4, w0, s0, saturated=flag, next_styles=s1, **dict(k0, clamp=None))
    assert ops.conv_domain_violated(flag)
    ops.modulated_conv2d(x, w0, s0, next_styles=s1, **dict(k0, noise=None, bias=None))  # (no noise, no bias: runs)


def test_generator_blocks_use_the_activation_image(hip, monkeypatch):
    """SynthesisBlock hands conv1 an ops.ActImage from res 32 on (StylePlan styles), conv1 hands the next block's up-sampling conv0 one
    (every map, nothing editing x in between), and the planes do not change by a bit when the path is switched off."""
    sg = hip.stylegan2
    torch.manual_seed(5)
    net = sg.SynthesisNetwork(w_dim=512, img_resolution=64, img_channels=96, cond_mode="none", channel_base=8192, channel_max=128, num_fp16_res=0).cuda()
    ws = torch.randn(2, net.num_ws, 512, device="cuda")
    seen = []
    orig = hip.ops.modulated_conv2d

    def spy(x, *a, **k):
        seen.append(isinstance(x, hip.ops.ActImage))
        return orig(x, *a, **k)

    monkeypatch.setattr(hip.ops, "modulated_conv2d", spy)
    with torch.no_grad():
        a = net(ws, {}, noise_mode="const")
        # conv1 of b32 and b64 (from their conv0), and conv0 of b8 .. b64: every block's conv1 hands the next block's conv0 its operand
        # next to the fp32 tensor (round 4: from the pipelined kernel's epilogue; round 6: also from the launch that sums split-K slices)
        assert sum(seen) == 6
        monkeypatch.setattr(sg, "CONV_IMG", False)
        seen.clear()
        b = net(ws, {}, noise_mode="const")
    assert sum(seen) == 0 and torch.equal(a, b)


def test_random_noise_from_one_draw_per_pass(hip, monkeypatch):
    """noise_mode='random' (what generate.py's G.f calls run): the pass draws all layers' noise in one randn call (NoisePool).  With
    the layers' noise_const buffers set to the SAME draws, noise_mode='const' must give the same planes bit for bit; the
    call-for-call path (P3D_NOISE_POOL=0, the reference's sequence of randn calls) still runs and differs only in its random values."""
    sg = hip.stylegan2
    torch.manual_seed(11)
    net = sg.SynthesisNetwork(w_dim=512, img_resolution=64, img_channels=96, cond_mode="none", channel_base=8192, channel_max=128, num_fp16_res=0).cuda()
    blocks = [getattr(net, f"b{r}") for r in net.block_resolutions]
    layers = [l for b in blocks for l in ([b.conv1] if b.in_channels == 0 else [b.conv0, b.conv1])]
    with torch.no_grad():
        for k, l in enumerate(layers):
            l.noise_strength.fill_(0.05 * (k + 1))
    ws = torch.randn(1, net.num_ws, 512, device="cuda")
    total = sum(l.resolution ** 2 for l in layers)
    with torch.no_grad():
        torch.manual_seed(3)
        y_rand = net(ws, {}, noise_mode="random")
        torch.manual_seed(3)
        raw = torch.randn([total], device="cuda")
        o = 0
        for l in layers:
            l.noise_const.copy_(raw[o:o + l.resolution ** 2].view(l.resolution, l.resolution))
            o += l.resolution ** 2
        y_const = net(ws, {}, noise_mode="const")
        assert torch.equal(y_rand, y_const)
        y_none = net(ws, {}, noise_mode="none")
        assert not torch.equal(y_rand, y_none)  # (the noise is really applied)
        torch.manual_seed(3)
        y_again = net(ws, {}, noise_mode="random")
        assert torch.equal(y_again, y_rand)
        monkeypatch.setattr(sg, "NOISE_POOL", False)
        torch.manual_seed(3)
        y_calls = net(ws, {}, noise_mode="random")  # one randn per layer, as networks_stylegan2.py:342
        assert torch.isfinite(y_calls).all() and not torch.equal(y_calls, y_none)
        # ... and it IS the reference's sequence: one torch.randn([N,1,res,res]) per layer in execution order (seed compatibility
        # with a call-for-call run on the same device); the per-network switch does the same with the process default back on
        torch.manual_seed(3)
        for l in layers:
            l.noise_const.copy_(torch.randn([1, 1, l.resolution, l.resolution], device="cuda")[0, 0])
        assert torch.equal(net(ws, {}, noise_mode="const"), y_calls) and not torch.equal(y_calls, y_rand)
        monkeypatch.setattr(sg, "NOISE_POOL", True)
        net.noise_pool = False
        torch.manual_seed(3)
        assert torch.equal(net(ws, {}, noise_mode="random"), y_calls)
        net.noise_pool = None
        monkeypatch.setattr(sg, "NOISE_POOL", False)
        assert float((y_calls - y_none).abs().mean()) == pytest.approx(float((y_rand - y_none).abs().mean()), rel=0.5)
        # batch 2: per-sample noise, every sample its own slice
        ws2 = torch.randn(2, net.num_ws, 512, device="cuda")
        monkeypatch.setattr(sg, "NOISE_POOL", True)
        y2 = net(ws2, {}, noise_mode="random")
        assert torch.isfinite(y2).all() and y2.shape[0] == 2


# ---- the memo layers in front of a G.f call: the cases live in tests/p3d_shared_cases.py and run here on the HIP kernels, and in
# tests/test_host_cpu.py on CPU stand-ins of the device operators
import p3d_shared_cases as MC  # noqa: E402


def test_prepared_conditioning_follows_the_conditioning_tensors(hip):
    MC.prepared_conditioning_follows_the_conditioning_tensors(hip, "cuda")


def test_f_memoises_ws_only_while_nothing_the_mapping_reads_has_changed(hip):
    MC.f_memoises_ws_only_while_nothing_the_mapping_reads_has_changed(hip, "cuda")


def test_style_plan_memo_follows_ws_and_parameters(hip):
    MC.style_plan_memo_follows_ws_and_parameters(hip, "cuda")


@pytest.mark.parametrize("how", ["data", "dlpack"])
def test_one_switch_turns_every_memo_layer_off_and_hidden_writes_are_then_seen(hip, how):
    MC.one_switch_turns_every_memo_layer_off_and_hidden_writes_are_then_seen(hip, how, "cuda")


@pytest.mark.parametrize("tag", ["none", "cond"])
def test_latent_injection_and_stop_level_vs_reference(hip, tag):
    MC.latent_injection_and_stop_level_vs_reference(hip, tag, "cuda")


def test_f_options_vs_reference(hip):
    MC.f_options_vs_reference(hip, "cuda")


def _memo_generator():
    return MC.memo_generator("cuda")


def _memo_call(G, cond, seed=4, azim=0.0):
    return MC.memo_call(G, cond, "cuda", seed=seed, azim=azim)


def test_generator_moved_between_devices_keeps_its_domain_watch_and_drops_device_state(hip):
    """the out-of-domain flag of the two-term convolutions is per DEVICE (created where a layer runs), never pickled
    or deep-copied; prepared conditioning terms are dropped when the network is moved (they live on the old device)."""
    import copy, pickle
    G = _memo_generator()
    G.watch_conv_domain()
    cond = {"image_ortho_front": torch.rand(1, 3, 32, 32, device="cuda"), "resnet_feats": torch.randn(1, 16, device="cuda")}
    a = _memo_call(G, cond)
    assert not G.conv_domain_violated()
    flags = G.__dict__["_conv_domain_flag"]
    assert list(flags.words) == [torch.device("cuda", torch.cuda.current_device())] or list(flags.words) == [torch.device("cuda:0")]
    assert G.backbone.synthesis.__dict__.get("_cond_cache")
    G2 = copy.deepcopy(G)
    assert G2.__dict__.get("_conv_domain_flag") is None
    assert all(not hasattr(m, "conv_domain_flag") for m in G2.modules())
    G3 = pickle.loads(pickle.dumps(G))
    assert all(not hasattr(m, "conv_domain_flag") for m in G3.modules())
    G.cpu()
    assert not G.backbone.synthesis.__dict__.get("_cond_cache")
    G.cuda()
    assert torch.equal(_memo_call(G, cond), a) and not G.conv_domain_violated()  # (the forward after a move used to raise)
    G.backbone.synthesis.clear_cond_cache()
    assert not G.backbone.synthesis.__dict__.get("_cond_cache")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_osg_decoder_forward_is_the_contract_decoder(hip, oracle, tag):
    """OSGDecoder.forward (triplane.py:528-544) — the module called like the reference's, on sampled features [N,3,M,32]: bit for
    bit the oracle's decoder (the arithmetic of every fused kernel), within fp32 tolerance of the reference's own module output
    (tests/golden/decoder_forward_*.npz), and equal to run_model's kernel fed with planes that reproduce those features."""
    from test_oracle_golden import decoder_forward_inputs
    from panic3d_amd.generator import OSGDecoder
    g, feats, raw, lr_mul, fs = decoder_forward_inputs(tag)
    dec = OSGDecoder(32, {"decoder_lr_mul": lr_mul, "decoder_output_dim": 32}).cuda()
    with torch.no_grad():
        for p, v in zip((dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias), raw):
            p.copy_(dev(v))
    dec.set_force_sigmoid(fs)
    out = dec(dev(feats), None)
    assert set(out) == {"rgb", "sigma"} and out["sigma"].shape == (2, 777, 1) and out["rgb"].shape == (2, 777, 32)
    os_, or_ = oracle.decode_features(feats, oracle.prescale_mlp(*raw, lr_mul=lr_mul), force_sigmoid=fs)
    assert np.array_equal(out["sigma"].cpu().numpy(), os_) and np.array_equal(out["rgb"].cpu().numpy(), or_)
    assert np.abs(out["sigma"].cpu().numpy() - g["sigma"]).max() <= 5e-5 and np.abs(out["rgb"].cpu().numpy() - g["rgb"]).max() <= 2e-6
    if not fs:  # the per-call override of the reference's signature
        forced = dec(dev(feats), None, force_sigmoid=True)
        assert np.array_equal(forced["rgb"].cpu().numpy(), oracle.decode_features(feats, oracle.prescale_mlp(*raw, lr_mul=lr_mul), force_sigmoid=True)[1])
    with pytest.raises(RuntimeError):
        dec(torch.from_numpy(feats), None)  # CPU tensors: no fallback


def _view(G, cond, azim, noise, seed=4, paste=False):
    x = dict(seeds=[seed], cond=cond, elevations=torch.zeros(1, device="cuda"), azimuths=torch.full((1,), float(azim), device="cuda"),
             neural_rendering_resolution=16, noise_mode=noise, triplane_crop=0.1, cull_clouds=0.5)
    with torch.no_grad():
        out = G.f(x)
    return {k: out[k].clone() for k in ("image", "image_raw", "image_depth", "image_weights", "image_xyz", "triplane")}


@pytest.mark.parametrize("noise", ["const", "random"])
def test_view_replay_is_bit_identical_to_eager_calls(hip, noise):
    """Round 6: the second call of a kind is captured into a hipGraph and later ones replay it
    (TriPlaneGenerator._replay_view).  Two subjects x four views, conditioned generator: every output of every call equals the eager
    generator's BIT FOR BIT — also under noise_mode='random' with the same torch seed (the device generator is registered with the
    capture) —, the replays really happen (views 2 .. 4 of both subjects; the first view of subject B refreshes the prepared
    conditioning terms in place), outputs of earlier views are not overwritten by later replays, and the switches turn it off."""
    import p3d_testing as T
    G = T.fill_generator_params(MC.memo_generator("cuda"), 3)  # (a volume with surfaces: the views differ)
    G.set_render_exact(None)
    conds = [{"image_ortho_front": torch.rand(1, 3, 32, 32, device="cuda"), "resnet_feats": torch.randn(1, 16, device="cuda")} for _ in range(2)]
    plan = [(s, c, 30.0 * v) for s, c in ((4, conds[0]), (9, conds[1])) for v in range(4)]

    def run(replay):
        G.clear_memo()
        G.set_view_replay(replay)
        torch.manual_seed(123)
        return [_view(G, c, az, noise, seed=s) for s, c, az in plan]

    eager = run(False)
    assert not G.__dict__.get("_view_graphs")
    rep = run(True)
    ents = list(G.__dict__["_view_graphs"]["entries"].values())
    assert len(ents) == 1 and ents[0]["graph"] is not None and ents[0]["replays"] == 6, [(e["graph"] is not None, e.get("replays")) for e in ents]
    for i, (a, b) in enumerate(zip(eager, rep)):
        for k in a:
            assert torch.equal(a[k], b[k]), (i, k, float((a[k] - b[k]).abs().max()))
    assert not torch.equal(rep[2]["image_depth"], rep[3]["image_depth"])  # (views differ, and view 3's tensors survived view 4's replay)
    # in-place parameter writes drop the capture (the derived operands move); the memo switch forbids replays altogether
    with torch.no_grad():
        G.backbone.synthesis.b8.conv1.weight.mul_(1.25)
    a = _view(G, conds[0], 0.0, noise)
    assert all(e["graph"] is None for e in G.__dict__["_view_graphs"]["entries"].values())
    prev = hip.memo.set_enabled(False)
    try:
        for _ in range(3):
            _view(G, conds[0], 0.0, noise)
        assert all(e["graph"] is None for e in (G.__dict__.get("_view_graphs") or {"entries": {}})["entries"].values())
    finally:
        hip.memo.set_enabled(prev)


def test_conv_domain_is_checked_once_per_set_of_weights(hip):
    """Round 6: nothing bounds a real checkpoint's activations (conv_clamp=None), and the default two-term f16
    convolutions SATURATE beyond |s*x| = 4094.  A generator whose weights drive an activation out of the domain says so on the FIRST
    call after the weights arrived (RuntimeWarning + G.conv_domain_was_violated), without any environment variable; fp32 operands
    (set_conv_mma('f32')) run clean; in-domain weights are checked once and never again (no per-call synchronisation)."""
    import warnings
    G = MC.memo_generator("cuda")
    G.set_render_exact(None)
    cond = {"image_ortho_front": torch.rand(1, 3, 32, 32, device="cuda"), "resnet_feats": torch.randn(1, 16, device="cuda")}
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _view(G, cond, 0.0, "const")          # in-domain weights: checked, silent
    flags = G.__dict__["_conv_domain_flag"]
    assert flags.dirty is False and not G.__dict__.get("conv_domain_was_violated")
    state = {k: v.clone() for k, v in G.state_dict().items()}
    big = {k: (v * 3e3 if k.endswith("b16.conv0.weight") or k == "backbone.synthesis.b4.const" else v) for k, v in state.items()}
    G.load_state_dict(big)                    # "a checkpoint": one layer's input now leaves the domain
    with pytest.warns(RuntimeWarning, match="saturated"):
        _view(G, cond, 0.0, "const")
    assert G.__dict__.get("conv_domain_was_violated") and flags.dirty is False
    G.set_conv_mma("f32")
    G.__dict__["conv_domain_was_violated"] = False
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out = _view(G, cond, 0.0, "const")
    assert torch.isfinite(out["image"]).all() and not G.__dict__.get("conv_domain_was_violated")


@pytest.mark.parametrize("N,I,O,H,R,hand", [(1, 32, 128, 256, 3, False), (1, 64, 256, 128, 3, True), (2, 16, 64, 256, 4, True), (1, 16, 64, 256, 1, False), (1, 48, 128, 200, 2, True)])
def test_torgb_riding_on_conv1_equals_the_stand_alone_torgb(hip, N, I, O, H, R, hand):
    """Round 6: a block of <= 4 image channels computes its ToRGB sums in conv1's epilogue (k_modconv_w3<true>) and finishes them with
    p3d_torgb_combine_f32.  Against the stand-alone ToRGB launch on conv1's fp32 result: the same products, another summation order
    (fp32-class: 1e-5 of the image's scale); conv1's own outputs (fp32 tensor, handed-over image) must not change by a bit, and the
    launch that is told nobody reads the fp32 tensor must return the same sums."""
    ops = hip.ops
    torch.manual_seed(5)
    d = torch.device("cuda")
    x = torch.randn(N, I, H, H, device=d)
    w = torch.randn(O, I, 3, 3, device=d) / np.sqrt(9 * I)
    s, s2 = torch.randn(N, I, device=d) * 0.3 + 1.0, torch.randn(N, O, device=d) * 0.3 + 1.0
    b, nz = torch.randn(O, device=d) * 0.1, torch.randn(H, H, device=d) * 0.05
    dco = ((w[None] * s[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt().contiguous()
    tw, ts, tb = torch.randn(R, O, 1, 1, device=d), (torch.randn(N, O, device=d) * 0.3 + 1.0) / np.sqrt(O), torch.randn(R, device=d)
    skip = torch.randn(N, R, H // 2, H // 2, device=d)
    f = ops.setup_filter((1, 3, 3, 1)).to(d)
    assert ops.conv_fuses_torgb(N, I, O, H, H, R), "an unsplit launch of the pipelined kernel was expected for this shape"
    kw = dict(padding=1, demodulate=True, bias=b, act="lrelu", dcoef=dco, noise=nz, weight_f16=ops.conv_weights_to_f16(w, split=True))
    img = ops.act_to_image(x, s)
    if hand:
        y0, yi0 = ops.modulated_conv2d(img, w, None, next_styles=s2, **kw)
    else:
        y0, yi0 = ops.modulated_conv2d(img, w, None, **kw), None
    ref = ops.torgb(y0, ops.torgb_weights(tw), R, ts, bias=tb, skip=skip, skip_filter=f)
    y1, yi1, part = ops.modulated_conv2d(img, w, None, next_styles=s2 if hand else None, rgb_weight=tw.reshape(R, O), rgb_styles=ts, **kw)
    got = ops.torgb_combine(part, bias=tb, skip=skip, skip_filter=f)
    assert torch.equal(y1, y0) and (yi0 is None or torch.equal(yi1.data, yi0.data))
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    print(f"ToRGB on conv1 [{N}x{I}->{O}@{H}^2, {R} channels]: max abs diff {err:.2e} on a scale of {scale:.2f}")
    assert err <= 1e-5 * scale
    y2, yi2, part2 = ops.modulated_conv2d(img, w, None, next_styles=s2 if hand else None, rgb_weight=tw.reshape(R, O), rgb_styles=ts, want_y=False, **kw)
    assert y2 is None and torch.equal(part2, part) and (yi0 is None or torch.equal(yi2.data, yi0.data))
    # the double-precision statement of the layer bounds both forms
    exact = torch.einsum("ro,no,nohw->nrhw", tw.reshape(R, O).double(), ts.double(), y0.double()) + tb.double()[None, :, None, None]
    exact = exact + ops.upsample2d(skip, f).double()
    assert float((got.double() - exact).abs().max()) <= 2e-5 * scale and float((ref.double() - exact).abs().max()) <= 2e-5 * scale


def test_superresolution_with_torgb_on_conv1_matches_the_separate_launches(hip, monkeypatch):
    """The full-width super-resolution module (32 -> 256 -> 128 channels, 128^2 -> 512^2) with and without the ToRGB layers riding
    on conv1: images agree to fp32 round-off (PSNR >= 120 dB on [-1, 1]-scale images), and the riding form launches no k_torgb."""
    from panic3d_amd import stylegan2 as sg
    from panic3d_amd.generator import TriPlaneGenerator
    G = T.fill_generator_params(TriPlaneGenerator(**T.FULL_KW), 11).cuda().eval()
    torch.manual_seed(2)
    rgb, feat = torch.randn(1, 3, 128, 128, device="cuda"), torch.randn(1, 32, 128, 128, device="cuda")
    ws = torch.randn(1, 14, 512, device="cuda")
    seen = []
    real = hip.ops.torgb
    monkeypatch.setattr(hip.ops, "torgb", lambda *a, **k: (seen.append(1), real(*a, **k))[1])
    with torch.no_grad():
        a = G.superresolution(rgb, feat, ws, noise_mode="const")
        assert not seen, "the super-resolution blocks should not launch the stand-alone ToRGB kernel"
        monkeypatch.setattr(sg, "TORGB_RIDES", False)
        b = G.superresolution(rgb, feat, ws, noise_mode="const")
        assert len(seen) == 2
    mse = float(((a - b).double() ** 2).mean())
    assert a.shape == (1, 3, 512, 512) and 10 * np.log10(4.0 / max(mse, 1e-30)) >= 120.0, mse


@pytest.mark.parametrize("N,I,O,H,up", [(1, 256, 256, 64, 1), (2, 64, 128, 40, 1), (1, 512, 512, 32, 1), (1, 256, 128, 128, 2), (1, 512, 512, 8, 2),
                                        (2, 64, 96, 24, 2), (1, 32, 256, 64, 2), (1, 512, 256, 64, 2)])
def test_weight_image_layouts_are_bit_identical_to_the_oik_layout(hip, N, I, O, H, up):
    """Round 6: the two-term weight copy in the consuming kernel's own LDS image order (P3D_WLAYOUT_PLAIN for k_modconv_w3, _UP for
    k_modconv_up3 / _up4: 1 KB of consecutive memory per staging request instead of 64 scattered 16-byte pieces).  Another element
    order of the same values: every path (unsplit, split-K, the one-launch up layer, image in / image out) must give the same bits
    as the [hi|lo][O][9][I] copy; a copy in the wrong image layout is refused."""
    ops = hip.ops
    torch.manual_seed(3)
    d = torch.device("cuda")
    x = torch.randn(N, I, H, H, device=d)
    w = torch.randn(O, I, 3, 3, device=d) / np.sqrt(9 * I)
    s, s2 = torch.randn(N, I, device=d) * 0.3 + 1.0, torch.randn(N, O, device=d) * 0.3 + 1.0
    b, nz = torch.randn(O, device=d) * 0.1, torch.randn(H * up, H * up, device=d) * 0.05
    dco = ((w[None] * s[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt().contiguous()
    f = ops.setup_filter((1, 3, 3, 1)).to(d)
    lay = ops.conv_weight_layout(I, O, H, up)
    assert lay == (1 if up == 1 else 2)
    w0, w1 = ops.conv_weights_to_f16(w, split=True), ops.conv_weights_to_f16(w, split=True, layout=lay)
    assert w1.p3d_layout == lay and w1.shape == w0.shape and not torch.equal(w0, w1) and torch.equal(w0.flatten().sort()[0], w1.flatten().sort()[0])
    kw = dict(up=up, padding=1, resample_filter=f, demodulate=True, bias=b, act="lrelu", dcoef=dco, noise=nz)
    img = ops.act_to_image(x, s)
    for inp, st in ((x, s), (img, None)):
        a = ops.modulated_conv2d(inp, w, st, weight_f16=w0, **kw)
        c = ops.modulated_conv2d(inp, w, st, weight_f16=w1, **kw)
        assert torch.equal(a, c)
        a = ops.modulated_conv2d(inp, w, st, weight_f16=w0, next_styles=s2, **kw)
        c = ops.modulated_conv2d(inp, w, st, weight_f16=w1, next_styles=s2, **kw)
        a, c = (a, c) if up == 2 else (a[1], c[1])
        assert torch.equal(a.data, c.data)
    wrong = ops.conv_weights_to_f16(w, split=True, layout=3 - lay) if O % 64 == 0 else None
    if wrong is not None:
        with pytest.raises(RuntimeError):
            ops.modulated_conv2d(img, w, None, weight_f16=wrong, **kw)


@pytest.mark.parametrize("N,I,O,H", [(1, 512, 512, 16), (2, 64, 128, 44), (1, 128, 64, 64), (1, 32, 40, 20)])
def test_fir_pass_on_64_column_tiles_is_bit_identical(hip, monkeypatch, N, I, O, H):
    """Round 6: k_fir4x4_img2 (the FIR pass that ends an up-sampling layer and writes the next layer's image, on conflict-free
    64-column LDS rows; chosen for launches too small for k_fir4x4_img's 32 x 32 tiles) against k_fir4x4_img: both tile heights, split-K
    partials summed inside (shallow splits), batch 2 with per-sample noise and a clamp, ragged maps (88^2, 40^2): the same image bits
    and the same domain flag."""
    ops = hip.ops
    filt = ops.setup_filter([1, 3, 3, 1]).cuda()
    g = torch.Generator().manual_seed(I * 7 + H)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    x, w0 = rn(N, I, H, H), rn(O, I, 3, 3)
    s0, s1 = rn(N, I) * 0.4 + 1.0, rn(N, O) * 0.4 + 1.0
    d0 = ((w0[None] * s0[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt().contiguous()
    nz = rn(N, 1, 2 * H, 2 * H) * 0.1 if N > 1 else rn(2 * H, 2 * H) * 0.1
    kw = dict(up=2, padding=1, resample_filter=filt, demodulate=True, bias=rn(O) * 0.1, act="lrelu", dcoef=d0, noise=nz,
              weight_f16=ops.conv_weights_to_f16(w0, split=True), clamp=1.5 if N > 1 else None, next_styles=s1)
    monkeypatch.setenv("P3D_UP4", "0")        # the two-pass form: transposed convolution (+ reduction), then the FIR pass
    monkeypatch.setenv("P3D_UP3_FUSED", "0")
    out = {}
    for mode in ("0", "8", "32"):
        monkeypatch.setenv("P3D_FIR_IMG2", mode)
        flag = ops.conv_domain_flag(x.device)
        out[mode] = ops.modulated_conv2d(x, w0, s0, saturated=flag, **kw).data.clone()
        assert not ops.conv_domain_violated(flag)
    assert torch.equal(out["0"], out["8"]) and torch.equal(out["0"], out["32"])
    monkeypatch.setenv("P3D_FIR_IMG2", "8")
    flag = ops.conv_domain_flag(x.device)
    ops.modulated_conv2d(x * 3e4, w0, s0, saturated=flag, **dict(kw, clamp=None))
    assert ops.conv_domain_violated(flag)


@pytest.mark.parametrize("N,I,O,H", [(1, 512, 512, 32), (1, 512, 256, 64), (2, 64, 96, 24), (1, 512, 512, 8), (1, 48, 32, 33), (3, 32, 64, 16), (1, 512, 512, 4)])
@pytest.mark.parametrize("layout", [0, 2])
def test_up5_deep_prefetch_transposed_conv_equals_up3(hip, monkeypatch, N, I, O, H, layout):
    """Round 6: k_modconv_up5 (the split-K / raw-store transposed convolution for under-filled launches: one workgroup per CU, a ring of
    four patches, operands of the next chunk prefetched into registers) against k_modconv_up3<false> (P3D_UP5=0): the same image and
    the same fp32 tensor, bit for bit — full tiles and the light tiles of the last grid row / column (H + 1 = 33, 34, 65 ...), one-chunk
    and odd chunk counts, batch > 1, both weight layouts, deep and shallow splits."""
    ops = hip.ops
    filt = ops.setup_filter([1, 3, 3, 1]).cuda()
    g = torch.Generator().manual_seed(I * 3 + O + H)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    x, w0 = rn(N, I, H, H), rn(O, I, 3, 3)
    s0, s1 = rn(N, I) * 0.4 + 1.0, rn(N, O) * 0.4 + 1.0
    d0 = ((w0[None] * s0[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt().contiguous()
    nz = rn(N, 1, 2 * H, 2 * H) * 0.1 if N > 1 else rn(2 * H, 2 * H) * 0.1
    if layout and ops.conv_weight_layout(I, O, H, 2) != layout:
        pytest.skip("the library keeps this shape on the OIK layout")
    kw = dict(up=2, padding=1, resample_filter=filt, demodulate=True, bias=rn(O) * 0.1, act="lrelu", dcoef=d0, noise=nz,
              weight_f16=ops.conv_weights_to_f16(w0, split=True, layout=layout))
    monkeypatch.setenv("P3D_UP4", "0")
    monkeypatch.setenv("P3D_UP3_FUSED", "0")
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("P3D_UP5", mode)
        out[mode] = (ops.modulated_conv2d(x, w0, s0, **kw).clone(), ops.modulated_conv2d(x, w0, s0, next_styles=s1, **kw).data.clone())
    assert torch.equal(out["0"][0], out["1"][0]) and torch.equal(out["0"][1], out["1"][1])


def test_conditioning_add_with_the_next_image_in_one_launch(hip, monkeypatch):
    """Round 6: a conditioned SynthesisNetwork (the released model's mode: resnet chonk at 8^2, `add_shuffle2_4` elsewhere) edits x between
    two blocks; ops.act_to_image_add does that in-place add and the next conv0's activation image in one launch.  (1) the op against
    `x[:, c0:c0+Ca] += t` followed by act_to_image: the same x and the same image, bit for bit, shared and per-sample terms; (2) a
    conditioned network takes it at every level and its planes do not change by a bit when the hand-over is switched off."""
    ops, sg = hip.ops, hip.stylegan2
    torch.manual_seed(7)
    d = torch.device("cuda")
    for N, C, H, c0, Ca, Na in ((2, 64, 16, 48, 16, 2), (1, 512, 8, 0, 64, 1), (3, 32, 20, 24, 8, 1)):
        x, s, t = torch.randn(N, C, H, H, device=d), torch.randn(N, C, device=d), torch.randn(Na, Ca, H, H, device=d)
        x1 = x.clone()
        x1[:, c0:c0 + Ca].add_(t)
        ref = ops.act_to_image(x1, s)
        x2 = x.clone()
        got = ops.act_to_image_add(x2, s, t, c0)
        assert torch.equal(x2, x1) and torch.equal(got.data, ref.data)
    with pytest.raises(RuntimeError):
        ops.act_to_image_add(torch.randn(1, 16, 8, 8, device=d)[:, :, ::2], torch.randn(1, 16, device=d), torch.randn(1, 8, 4, 8, device=d), 0)
    net = sg.SynthesisNetwork(w_dim=512, img_resolution=64, img_channels=96, cond_mode="ortho_front.add_shuffle2_4.inj_6b_4.reschonk_add_64",
                              channel_base=8192, channel_max=128, num_fp16_res=0).cuda()
    ws = torch.randn(1, net.num_ws, 512, device=d)
    cond = {"image_ortho_front": torch.rand(1, 4, 64, 64, device=d), "resnet_chonk": torch.randn(1, 128, 8, 8, device=d)}
    calls = []
    real = ops.act_to_image_add
    monkeypatch.setattr(ops, "act_to_image_add", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        a = net(ws, cond, noise_mode="const")
        n_fused = len(calls)
        monkeypatch.setattr(sg, "CONV_IMG", False)
        net.clear_cond_cache()
        b = net(ws, cond, noise_mode="const")
    assert n_fused >= 3 and len(calls) == n_fused and torch.equal(a, b), n_fused
