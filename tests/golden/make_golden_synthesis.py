#!/usr/bin/env python3
"""Golden vectors for the StyleGAN2 synthesis operators / networks, produced by the REFERENCE ITSELF on CPU
(/root/reference imported unmodified; only runs in the build container).  Outputs: tests/golden/syn_*.npz.

    python tests/golden/make_golden_synthesis.py
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("PROJECT_DN", "/root/reference")
os.environ.setdefault("PROJECT_NAME", "x")
sys.path[:0] = ["/root/reference"]
sys.path.append("/root/reference/_train/eg3dc/src")
sys.modules.setdefault("kornia", types.ModuleType("kornia"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from training import networks_stylegan2 as ns  # noqa: E402
from torch_utils.ops import upfirdn2d, bias_act  # noqa: E402

torch.set_grad_enabled(False)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def sd_np(mod, prefix=""):
    return {prefix + k.replace(".", "__"): v.numpy() for k, v in mod.state_dict().items()}


def layer_cases():
    g = torch.Generator().manual_seed(1234)
    rn = lambda *s: torch.randn(*s, generator=g)
    out = {}
    # SynthesisLayer up=1 (conv1): 16 -> 24 channels at 20x20 (tile-ragged sizes), batch 2
    for tag, (cin, cout, res, up) in {"conv1": (16, 24, 20, 1), "conv0": (16, 40, 24, 2), "conv0b": (8, 72, 34, 2)}.items():
        lay = ns.SynthesisLayer(cin, cout, w_dim=32, resolution=res, up=up, conv_clamp=None).eval()
        lay.weight.copy_(rn(*lay.weight.shape))
        lay.bias.copy_(rn(cout) * 0.3)
        lay.noise_strength.copy_(torch.tensor(0.7))
        lay.noise_const.copy_(rn(res, res))
        lay.affine.weight.copy_(rn(*lay.affine.weight.shape))
        x = rn(2, cin, res // up, res // up)
        w = rn(2, 32)
        y = lay(x, w, noise_mode="const", fused_modconv=True)
        y_nf = lay(x, w, noise_mode="const", fused_modconv=False)
        assert (y - y_nf).abs().max() < 1e-3
        out.update({f"{tag}_x": x.numpy(), f"{tag}_w": w.numpy(), f"{tag}_y": y.numpy(), f"{tag}_ynf": y_nf.numpy(),
                    f"{tag}_cfg": np.array([cin, cout, res, up])})
        out.update(sd_np(lay, f"{tag}_sd_"))
        if tag == "conv1":  # clamp + gain variant (SR blocks use conv_clamp 256; use a small clamp so that it bites)
            lay.conv_clamp = 0.8
            out[f"{tag}_y_clamp"] = lay(x, w, noise_mode="const", gain=0.5).numpy()
            lay.conv_clamp = None
            out[f"{tag}_y_nonoise"] = lay(x, w, noise_mode="none").numpy()
    rgb = ns.ToRGBLayer(16, 96, w_dim=32, conv_clamp=None).eval()
    rgb.weight.copy_(rn(*rgb.weight.shape))
    rgb.bias.copy_(rn(96) * 0.3)
    rgb.affine.weight.copy_(rn(*rgb.affine.weight.shape))
    x = rn(2, 16, 20, 20)
    w = rn(2, 32)
    out.update({"torgb_x": x.numpy(), "torgb_w": w.numpy(), "torgb_y": rgb(x, w).numpy()})
    out.update(sd_np(rgb, "torgb_sd_"))
    # upsample2d of the skip image, bias_act, upfirdn2d with odd padding
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    img = rn(2, 5, 9, 11)
    out.update({"fir_f": f.numpy(), "up_x": img.numpy(), "up_y": upfirdn2d.upsample2d(img, f).numpy(),
                "ufd_y": upfirdn2d.upfirdn2d(img, f, up=1, padding=[1, 1, 1, 1], gain=4).numpy(),
                "ufd2_y": upfirdn2d.upfirdn2d(img, f, up=2, down=1, padding=[3, 0, 1, 2], flip_filter=True, gain=2).numpy()})
    xb = rn(3, 7, 4, 5)
    bb = rn(7)
    out.update({"ba_x": xb.numpy(), "ba_b": bb.numpy(), "ba_lrelu": bias_act.bias_act(xb, bb, act="lrelu").numpy(),
                "ba_lin_clamp": bias_act.bias_act(xb, bb, act="linear", gain=2.0, clamp=1.5).numpy(),
                "ba_fc": bias_act.bias_act(xb.reshape(3, -1)[:, :7].contiguous(), bb, act="lrelu").numpy()})
    save("syn_layers.npz", **out)


GEN_KW = dict(z_dim=64, c_dim=25, w_dim=64, img_resolution=32, img_channels=96, mapping_kwargs={"num_layers": 2},
              channel_base=2048, channel_max=64, num_fp16_res=0, conv_clamp=None, fused_modconv_default="inference_only")


# every branch of the conditioning glue (networks_stylegan2.py:551-694) appears in at least one mode; the image-channel
# counts limit which branches can be combined (the tiled condition must fit a quarter of the 64 feature channels)
COND_MODES = (("none", "none"), ("cond", "ortho_front.add_shuffle2_4.inj_6b_4.crossavg_4.reschonk_add_8.resnetcond_16"),
              ("cond2", "ortho_front.gt_sides.cond_img_norm_4.add_4.concatfront.crossavgt_38"),
              ("cond3", "ortho_front.mult_shuffle2_4.crossavg_4"), ("cond4", "ortho_front.dorthoA.add_4"))


def generator_cases(only=None):
    for tag, cond_mode in COND_MODES:
        if only and tag not in only:
            continue
        torch.manual_seed(77)
        G = ns.Generator(cond_mode=cond_mode, **GEN_KW).eval()
        gg = torch.Generator().manual_seed(78)
        for n, p in G.named_parameters():  # make every path matter: non-zero biases / noise strengths
            if n.endswith("noise_strength"):
                p.copy_(torch.rand((), generator=gg) * 0.5)
            elif n.endswith(".bias") and "affine" not in n:
                p.copy_(torch.randn(p.shape, generator=gg) * 0.2)
        z = torch.randn(2, 64, generator=gg)
        c = torch.randn(2, 25, generator=gg)
        cond = {"image_ortho_front": torch.rand(2, 4, 32, 32, generator=gg), "resnet_feats": torch.randn(2, 32, generator=gg),
                "resnet_chonk": torch.randn(2, 8, 8, 8, generator=gg)}
        if "gt_sides" in cond_mode:
            cond.update(image_ortho_left=torch.rand(2, 4, 32, 32, generator=gg), image_ortho_right=torch.rand(2, 4, 32, 32, generator=gg))
        if "dorthoA" in cond_mode:
            cond.update(image_dorthoA_left=torch.rand(2, 4, 32, 32, generator=gg), image_dorthoA_right=torch.rand(2, 4, 32, 32, generator=gg))
        ws = G.mapping(z, c, cond, truncation_psi=0.7, truncation_cutoff=4)
        img = G.synthesis(ws, cond, noise_mode="const")
        ws1 = G.mapping(z, c, cond)
        out = {"z": z.numpy(), "c": c.numpy(), "ws": ws.numpy(), "ws_psi1": ws1.numpy(), "img": img.numpy(),
               "cond_mode": np.array(cond_mode)}
        out.update({"cond_" + k: v.numpy() for k, v in cond.items()})
        out.update(sd_np(G, "sd_"))
        save(f"syn_generator_{tag}.npz", **out)




# ---- TriPlaneGenerator.f end to end (tiny backbone / SR widths; released rendering_kwargs) --------------------------
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


def triplane_case():
    from training.triplane import TriPlaneGenerator
    torch.manual_seed(5)
    G = TriPlaneGenerator(**TRI_KW).eval()
    gg = torch.Generator().manual_seed(6)
    for n, p in G.named_parameters():
        if n.endswith(".bias") and "affine" not in n:
            p.copy_(torch.randn(p.shape, generator=gg) * 0.2)
    for n, p in G.backbone.synthesis.named_parameters():  # random-init ToRGB gives planes ~0: scale them to O(1) features
        if n.endswith("torgb.weight"):
            p.mul_(30.0)
    G.decoder.net[2].weight[0] *= 20.0  # solid-ish density field
    G.decoder.net[2].bias[0] = 25.0
    G.set_force_sigmoid(True)
    rec = {}
    o_rl, o_r = torch.rand_like, torch.rand

    def rand_like(t, *a, **k):
        r = o_rl(t, *a, **k)
        rec.setdefault("jitter", r.clone())
        return r

    def rand(*a, **k):
        r = o_r(*a, **k)
        rec.setdefault("u", r.clone())
        return r

    x = dict(elevations=torch.tensor([0.0, 10.0]), azimuths=torch.tensor([20.0, 200.0]), fovs=torch.tensor([30.0, -1.0]),
             seeds=[3, 4], cond={}, triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=16)
    torch.rand_like, torch.rand = rand_like, rand
    try:
        torch.manual_seed(9)
        out = G.f(x)
    finally:
        torch.rand_like, torch.rand = o_rl, o_r
    arrs = {k: out[k].numpy() for k in ("image_raw", "image_depth", "image_weights", "image_xyz", "triplane")}
    arrs["image_sub4"] = out["image"][..., ::4, ::4].contiguous().numpy()  # every 4th pixel of the 512^2 SR image (fixture size)
    arrs.update(jitter=rec["jitter"].numpy(), u=rec["u"].numpy(), ws=x["ws"].numpy(), camera_params=x["camera_params"].numpy())
    # ---- the same call with the front-view paste (generate.py:55-66).  kornia is not installed: its Sobel filter is
    # restated in the product (paste.sobel_magnitude, kornia 0.6.5 semantics) and injected here under kornia's name, so this
    # fixture pins the paste GLUE (masks, second render pass, grid_sample, lerp), not kornia itself.
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import panic3d_amd.paste as my_paste
    import kornia
    kornia.filters = types.SimpleNamespace(sobel=my_paste.sobel_magnitude)
    front = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(11))
    xp = dict(elevations=torch.tensor([0.0]), azimuths=torch.tensor([0.0]), fovs=torch.tensor([-1.0]), seeds=[3],
              cond={"image_ortho_front": front}, triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=16,
              paste_params={"mode": "default", "thresh_weight": 0.5, "thresh_edges": 0.2, "thresh_occ": 0.5,
                            "offset_occ": 0.01, "thresh_dxyz": 0.05})
    rec.clear()
    draws = []
    def rand_like2(t, *a, **k):
        r = o_rl(t, *a, **k); draws.append(r.clone()); return r
    def rand2(*a, **k):
        r = o_r(*a, **k); draws.append(r.clone()); return r
    torch.rand_like, torch.rand = rand_like2, rand2
    try:
        torch.manual_seed(10)
        outp = G.f(xp)
    finally:
        torch.rand_like, torch.rand = o_rl, o_r
    assert len(draws) == 4  # two renderer passes (view + front-occlusion), two draws each
    print("paste fixture: weights mean %.3f; mask means:" % float(outp["image_weights"].mean()),
          {k: round(float(v.mean()), 3) for k, v in outp["paste"].items() if k.startswith("mask")})
    arrs.update({"paste_" + k: outp["paste"][k].numpy() for k in ("mask", "mask_weights", "mask_edges", "mask_occ", "mask_dxyz")})
    arrs["paste_image_sub4"] = outp["image"][..., ::4, ::4].contiguous().numpy()
    arrs["paste_prepaste_sub4"] = outp["image_prepaste"][..., ::4, ::4].contiguous().numpy()
    arrs["paste_paste_sub4"] = outp["paste"]["paste"][..., ::4, ::4].contiguous().numpy()
    for i, dr in enumerate(draws):
        arrs[f"paste_draw{i}"] = dr.numpy()
    pts = (torch.rand(2, 500, 3, generator=gg) - 0.5) * 0.6
    sm = G.sample_mixed(pts, None, x["ws"], {}, noise_mode="const")
    arrs.update(sm_pts=pts.numpy(), sm_sigma=sm["sigma"].numpy(), sm_rgb=sm["rgb"].numpy())
    arrs.update(sd_np(G, "sd_"))
    save("syn_triplane_f.npz", **arrs)


def triplane_cond_case():
    """G.f of a CONDITIONED generator (image + resnet conditioning as _scripts/eval/generate.py:88-96 passes them, pose
    conditioning zeroed like PAniC-3D's trainer default): pins mapping_zplus with resnet features (triplane.py:124-186),
    c_gen_conditioning_zero and the conditioned backbone inside the full f() flow."""
    from training.triplane import TriPlaneGenerator
    torch.manual_seed(15)
    kw = dict(TRI_KW, cond_mode="ortho_front.concatfront.inj_6b_4.crossavg_4.reschonk_add_8.resnetcond_16",
              rendering_kwargs=dict(TRI_RK, c_gen_conditioning_zero=True))
    G = TriPlaneGenerator(**kw).eval()
    gg = torch.Generator().manual_seed(16)
    for n, p in G.named_parameters():
        if n.endswith(".bias") and "affine" not in n:
            p.copy_(torch.randn(p.shape, generator=gg) * 0.2)
    for n, p in G.backbone.synthesis.named_parameters():
        if n.endswith("torgb.weight"):
            p.mul_(30.0)
    G.decoder.net[2].weight[0] *= float(os.environ.get('P3D_COND_SIGMA_GAIN', '1.0'))
    G.decoder.net[2].bias[0] = float(os.environ.get('P3D_COND_SIGMA_BIAS', '-20.0'))  # mixes surface and background
    G.set_force_sigmoid(True)
    cond = {"image_ortho_front": torch.rand(1, 3, 64, 64, generator=gg), "resnet_feats": torch.randn(1, 32, generator=gg),
            "resnet_chonk": torch.randn(1, 8, 8, 8, generator=gg)}
    rec = {}
    o_rl, o_r = torch.rand_like, torch.rand

    def rand_like(t, *a, **k):
        r = o_rl(t, *a, **k); rec.setdefault("jitter", r.clone()); return r

    def rand(*a, **k):
        r = o_r(*a, **k); rec.setdefault("u", r.clone()); return r

    x = dict(elevations=torch.tensor([5.0]), azimuths=torch.tensor([-30.0]), fovs=torch.tensor([30.0]), seeds=[7], cond=cond,
             triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=16, noise_mode="const")
    torch.rand_like, torch.rand = rand_like, rand
    try:
        torch.manual_seed(19)
        out = G.f(x)
    finally:
        torch.rand_like, torch.rand = o_rl, o_r
    print("conditioned fixture: image_weights mean %.3f" % float(out["image_weights"].mean()))
    arrs = {k: out[k].numpy() for k in ("image_raw", "image_weights", "image_xyz", "triplane")}
    arrs["image_sub4"] = out["image"][..., ::4, ::4].contiguous().numpy()
    arrs.update(jitter=rec["jitter"].numpy(), u=rec["u"].numpy(), ws=x["ws"].numpy(), cond_mode=np.array(kw["cond_mode"]))
    arrs.update({"cond_" + k: v.numpy() for k, v in cond.items()})
    arrs.update(sd_np(G, "sd_"))
    save("syn_triplane_f_cond.npz", **arrs)


def volume_case():
    """The reference's OWN get_eg3d_volume (_util/eg3d_metrics3d.py:94-183) on the syn_triplane_f generator.  The module
    cannot be imported here (it pulls the whole _util / _databacks tree: cv2, trimesh, ...), so the three functions are
    compiled from the reference's source text, unmodified, into a namespace that supplies their module-level names."""
    import ast
    from training.triplane import TriPlaneGenerator
    from training.volumetric_rendering.renderer import triplane_crop_mask, cull_clouds_mask
    g = np.load(os.path.join(HERE, "syn_triplane_f.npz"))
    G = TriPlaneGenerator(**TRI_KW).eval()
    G.load_state_dict({k[3:].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")})
    G.set_force_sigmoid(True)

    class Dict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    path = "/root/reference/_util/eg3d_metrics3d.py"
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("sigma2density", "create_samples", "get_eg3d_volume")]
    assert len(keep) == 3
    ns_ = {"torch": torch, "nn": torch.nn, "np": np, "Dict": Dict, "device": torch.device("cpu"),
           "triplane_crop_mask": triplane_crop_mask, "cull_clouds_mask": cull_clouds_mask}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns_)
    N = 14
    out = {}
    for tag, extra in (("plain", {}), ("masked", {"triplane_crop": 0.1, "cull_clouds": 0.5})):
        xin = {"cond": {}, "seeds": [3], "neural_rendering_resolution": 8, **extra}
        torch.manual_seed(21)
        vol = ns_["get_eg3d_volume"](G, xin, resolution=N, max_batch=1000)
        out[f"{tag}_densities"] = vol["densities"].contiguous().numpy()
        out[f"{tag}_sigmas"] = vol["sigmas"].contiguous().numpy()
        out[f"{tag}_rgb3"] = vol["rgbs"][:, :3].contiguous().numpy()
        out[f"{tag}_coordinates"] = vol["coordinates"].contiguous().numpy()
    print("volume fixture: density > 0.5 fraction %.3f, culled fraction %.3f" % (
        float((out["plain_densities"] > 0.5).mean()), float((out["masked_densities"] == -1e3).mean())))
    save("volume_reference.npz", resolution=np.array(N), **out)


def triplane_options_case():
    """TriPlaneGenerator.f with the options the other fixtures leave at their defaults, on the generator of syn_triplane_f.npz (its
    weights are read from that fixture): one z PER w slot (x['zs']), truncation with a cutoff, latent injection given both ways (the
    argument and x['latent_injection'], merged: dw / dws on the ws, da_<lvl> / db_<lvl> inside the backbone), stop_level,
    binarize_clouds instead of cull_clouds, normalize_images=True.  The two draws of the renderer are captured and stored."""
    from training.triplane import TriPlaneGenerator
    g = dict(np.load(os.path.join(HERE, "syn_triplane_f.npz")))
    G = TriPlaneGenerator(**TRI_KW).eval()
    G.load_state_dict({k[3:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("sd_")}, strict=True)
    G.set_force_sigmoid(True)
    gg = torch.Generator().manual_seed(31)
    n = G.backbone.num_ws
    zs = torch.randn(1, n, 512, generator=gg)
    _, loc = G.backbone.synthesis(torch.zeros(1, n, 512), {}, noise_mode="const", return_more=True)
    x1, img2 = loc["ximgs"][1][0], loc["ximgs"][2][1]
    inj_arg = {"dw": torch.randn(1, 1, 512, generator=gg) * 0.2, "da_1": torch.randn(x1.shape, generator=gg) * 0.3}
    inj_x = {"dws": torch.randn(1, n, 512, generator=gg) * 0.1, "db_2": torch.randn(img2.shape, generator=gg) * 0.3}
    rec = {}
    o_rl, o_r = torch.rand_like, torch.rand

    def rand_like(t, *a, **k):
        r = o_rl(t, *a, **k)
        rec.setdefault("jitter", r.clone())
        return r

    def rand(*a, **k):
        r = o_r(*a, **k)
        rec.setdefault("u", r.clone())
        return r

    x = dict(elevations=torch.tensor([12.0]), azimuths=torch.tensor([-50.0]), fovs=torch.tensor([25.0]), distances=torch.tensor([1.1]),
             zs=zs, cond={}, triplane_crop=0.08, binarize_clouds=0.4, neural_rendering_resolution=16, normalize_images=True,
             latent_injection=inj_x)
    torch.rand_like, torch.rand = rand_like, rand
    try:
        torch.manual_seed(12)
        out = G.f(x, truncation_psi=0.7, truncation_cutoff=6, latent_injection=inj_arg, stop_level=2)
    finally:
        torch.rand_like, torch.rand = o_rl, o_r
    arrs = {k: out[k].numpy() for k in ("image_raw", "image_depth", "image_weights", "image_xyz")}
    arrs["triplane_sub"] = out["triplane"][:, :, ::4].contiguous().numpy()
    arrs["image_sub4"] = out["image"][..., ::4, ::4].contiguous().numpy()
    arrs.update(jitter=rec["jitter"].numpy(), u=rec["u"].numpy(), ws=x["ws"].numpy(), camera_params=x["camera_params"].numpy(), zs=zs.numpy())
    arrs.update({"injarg_" + k: v.numpy() for k, v in inj_arg.items()})
    arrs.update({"injx_" + k: v.numpy() for k, v in inj_x.items()})
    print("options fixture: weights mean %.3f, image range [%.2f, %.2f]" % (float(out["image_weights"].mean()), float(out["image"].min()), float(out["image"].max())))
    save("syn_triplane_f_options.npz", **arrs)


def injection_case():
    """SynthesisNetwork.forward with latent_injection (da_<lvl> added to x, db_<lvl> to img after a block and its conditioning,
    networks_stylegan2.py:700-705) and with stop_level (the image of an inner level up-sampled to the output size, :707-714), on
    the generators of syn_generator_none.npz and syn_generator_cond.npz (weights and inputs are read from those fixtures; this one
    stores the injected tensors and the reference's outputs)."""
    out = {}
    for tag, cond_mode in COND_MODES[:2]:
        g = dict(np.load(os.path.join(HERE, f"syn_generator_{tag}.npz")))
        G = ns.Generator(cond_mode=cond_mode, **GEN_KW).eval()
        G.load_state_dict({k[3:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("sd_")}, strict=True)
        cond = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("cond_") and k != "cond_mode"}
        ws = torch.from_numpy(g["ws"])
        assert np.abs(G.synthesis(ws, cond, noise_mode="const").numpy() - g["img"]).max() < 1e-5  # the fixture's generator, rebuilt
        gg = torch.Generator().manual_seed(91)
        _, loc = G.synthesis(ws, cond, noise_mode="const", return_more=True)
        inj = {}
        for lvl, (x, img) in enumerate(loc["ximgs"]):
            if lvl in (0, 2):
                inj[f"da_{lvl}"] = torch.randn(x.shape, generator=gg) * 0.3
            if lvl in (1, 2):
                inj[f"db_{lvl}"] = torch.randn(img.shape, generator=gg) * 0.3
        sub = lambda t: t[:, ::4, ::2, ::2].contiguous().numpy()  # (a quarter of the channels, every second pixel: 49 KB per output)
        out[f"{tag}_inj_checksum"] = np.array([float(sum(v.double().sum() for v in inj.values()))])  # the test re-draws them from the seed
        out[f"{tag}_img_inj"] = sub(G.synthesis(ws, cond, latent_injection=inj, noise_mode="const"))
        for sl in (0, 2):
            out[f"{tag}_img_stop{sl}"] = sub(G.synthesis(ws, cond, stop_level=sl, noise_mode="const"))
        out[f"{tag}_img_inj_stop1"] = sub(G.synthesis(ws, cond, latent_injection=inj, stop_level=1, noise_mode="const"))
    save("syn_generator_inject.npz", **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if sys.argv[1:] == ["inject"]:  # only the fixtures added at the end of round 4
        injection_case()
        sys.exit(0)
    if sys.argv[1:] == ["options"]:
        triplane_options_case()
        sys.exit(0)
    which = sys.argv[1:] or ["layers", "generator", "triplane"]
    if "layers" in which:
        layer_cases()
    if "generator" in which:
        generator_cases()
    if any(w.startswith("cond") for w in which):  # e.g. `make_golden_synthesis.py cond2 cond3 cond4`
        generator_cases(only=[w for w in which if w.startswith("cond")])
    if "triplane" in which:
        triplane_case()
    if "triplane_cond" in which or not sys.argv[1:]:
        triplane_cond_case()
    if "volume" in which or not sys.argv[1:]:
        volume_case()
