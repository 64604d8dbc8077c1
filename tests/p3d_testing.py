"""Shared deterministic input builders for the parity tests and the golden-vector generator.

Everything here is plain torch-CPU / numpy so that the GPU box (same image, same torch build)
regenerates bit-identical inputs from the seeds stored in tests/golden/*.npz; every fixture also
stores a checksum of the regenerated tensors, asserted on load.
"""
import hashlib
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# rendering_kwargs of the released configuration (trainers/train_eclustrousC.py:409-440) + generate.py:56-57
RENDERING_KWARGS = dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=48, depth_resolution_importance=48,
                        disparity_space_sampling=False, clamp_mode="softplus", white_back=True, use_triplane=1)


def checksum(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def make_planes(seed, N=1, H=256, W=256, C=32, scale=1.0, smooth=0):
    """Synthetic triplanes [N,3,C,H,W].  smooth=0: white noise (a fog).  smooth=k>0: a k x k noise grid bilinearly
    upsampled to HxW plus 10% white noise — spatially coherent blobs, so rays meet 'surfaces' (saturating weights,
    empty rays) like a trained model's planes do."""
    g = torch.Generator().manual_seed(int(seed))
    if not smooth:
        return (torch.randn(N, 3, C, H, W, generator=g) * scale).numpy()
    low = torch.randn(N * 3, C, int(smooth), int(smooth), generator=g)
    up = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=False)
    up = up + 0.1 * torch.randn(N * 3, C, H, W, generator=g)
    return (up.reshape(N, 3, C, H, W) * scale).contiguous().numpy()


def make_decoder_params(seed, lr_mul=1.0, sigma_gain=1.0):
    """Raw (un-scaled) OSGDecoder parameters: net.0.weight [64,32], net.0.bias [64], net.2.weight [33,64], net.2.bias [33]
    (triplane.py:521-526).  Biases are drawn non-zero so the bias path is exercised; sigma bias +1 gives a mix of
    occupied and empty space under cull_clouds=0.5.  Row 0 of net.2 (sigma) is scaled by sigma_gain."""
    g = torch.Generator().manual_seed(int(seed))
    w0 = torch.randn(64, 32, generator=g) / lr_mul
    b0 = torch.randn(64, generator=g) * 0.5
    w1 = torch.randn(33, 64, generator=g) / lr_mul
    b1 = torch.randn(33, generator=g) * 0.5
    b1[0] += 1.0
    w1[0] *= sigma_gain  # sigma_gain >> 1: solid objects (weights saturate) like a trained model
    return w0.numpy(), b0.numpy(), w1.numpy(), b1.numpy()


def make_random_draws(seed, N, R, Sc, Sf, auto_limits=False):
    """The two draws of ImportanceRenderer.forward in the reference's order (renderer.py:324 then :371):
    torch.manual_seed(seed); rand_like([N,R,Sc,1]); rand(N*R, Sf).  auto_limits: with per-ray limits the depths come from
    math_utils.linspace(...).permute(1,2,0,3) (renderer.py:317) — a [Sc,N,R,1] tensor viewed as [N,R,Sc,1] — and rand_like
    fills it in MEMORY order."""
    torch.manual_seed(int(seed))
    jitter = torch.rand(Sc, N, R, 1).permute(1, 2, 0, 3).contiguous() if auto_limits else torch.rand(N, R, Sc, 1)
    u = torch.rand(N * R, Sf) if Sf > 0 else torch.zeros(N * R, 0)
    return jitter.numpy(), u.numpy()


def make_points(seed, N, M, extent=0.5):
    """Query points for run_model: mostly inside the box, some outside (zeros-padding path)."""
    g = torch.Generator().manual_seed(int(seed))
    return ((torch.rand(N, M, 3, generator=g) * 2 - 1) * extent).numpy()


def load_golden(name):
    path = os.path.join(GOLDEN_DIR, name)
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


RENDER_GOLDENS = ["render_c1_64x64_s32", "render_32x32_16p16", "render_24x24_48p48_ortho", "render_12x12_96p96",
                  "render_variant_a", "render_variant_b"]


# the other branches of sample_stratified: ray_start = ray_end = 'auto' (renderer.py:165-171, per-ray limits in the fixture) and
# disparity_space_sampling (renderer.py:309-316)
RENDER_GOLDENS_AUTO = ["render_auto_limits", "render_disparity"]


def golden_render_inputs(g):
    """Rebuild the inputs of a render_*.npz fixture from its meta_* fields (see tests/golden/make_golden.py)."""
    m = {k[5:]: g[k].item() for k in g if k.startswith("meta_")}
    ro = dict(RENDERING_KWARGS, depth_resolution=int(m["Sc"]), depth_resolution_importance=int(m["Sf"]),
              use_triplane=int(m["use_triplane"]), white_back=bool(m["white_back"]), ray_start=float(m["ray_start"]),
              ray_end=float(m["ray_end"]), box_warp=float(m["box_warp"]))
    if int(m.get("auto_limits", 0)):  # renderer.py:165-171; the fixture carries the reference's per-ray limits as ray_start / ray_end
        ro["ray_start"] = ro["ray_end"] = "auto"
    if int(m.get("disparity", 0)):
        ro["disparity_space_sampling"] = True
    planes = make_planes(m["seed"], m["N"], m["H"], m["W"], scale=float(m["plane_scale"]), smooth=int(m["smooth"]))
    assert checksum(planes) == str(g["planes_checksum"]), "regenerated planes differ from the fixture's"
    raw = make_decoder_params(m["seed"] + 1, float(m["lr_mul"]), float(m["sigma_gain"]))
    R = g["rays_o"].shape[1]
    jitter, u = make_random_draws(m["seed"] + 2, m["N"], R, int(m["Sc"]), int(m["Sf"]), auto_limits=bool(int(m.get("auto_limits", 0))))
    kw = dict(triplane_crop=float(m["crop"]) or None, cull_clouds=float(m["cull"]) or None,
              binarize_clouds=float(m["binarize"]) or None, force_sigmoid=bool(m["force_sigmoid"]))
    limits = (g["ray_start"], g["ray_end"]) if int(m.get("auto_limits", 0)) else None
    return dict(ro=ro, planes=planes, raw_mlp=raw, lr_mul=float(m["lr_mul"]), rays_o=g["rays_o"], rays_d=g["rays_d"],
                jitter=jitter, u=u, kw=kw, meta=m, ray_limits=limits)


# ---- bench scenes (bench.py, tools/cpu_baseline_reference.py, tests/test_hip_fullsize.py share these builders) ---------------
BENCH_KW = dict(triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True)  # generate.py:56-57; OSGDecoder force_sigmoid


def make_bench_scene(scene="canonical"):
    """(planes [1,3,32,256,256] f32 numpy, raw decoder params) of a bench scene.

    'canonical' = SURVEY.md §8(d) exactly: planes torch.randn(1,3,32,256,256, generator=seed 0); decoder = the reference's
        OSGDecoder(32, {decoder_lr_mul: 1, decoder_output_dim: 32}) constructed under torch.manual_seed(0) — its
        FullyConnectedLayers draw torch.randn([64,32]) then torch.randn([33,64]) and zero biases (networks_stylegan2.py:110-115,
        triplane.py:521-526); tools/cpu_baseline_reference.py asserts that this restatement equals the reference constructor's
        parameters bit for bit.  A fog: no ray ever turns opaque.
    'surface' = the round-1 bench scene: 16x16 noise bilinearly upsampled + 10 % white noise, scale 4, sigma row x30, sigma
        bias -45 — about half of the rays hit an opaque surface (character-like coverage), the others see empty space."""
    if scene == "canonical":
        planes = torch.randn(1, 3, 32, 256, 256, generator=torch.Generator().manual_seed(0)).numpy()
        state = torch.random.get_rng_state()
        torch.manual_seed(0)
        w0 = torch.randn(64, 32)
        w1 = torch.randn(33, 64)
        torch.random.set_rng_state(state)
        return planes, (w0.numpy(), np.zeros(64, np.float32), w1.numpy(), np.zeros(33, np.float32))
    if scene == "surface":
        g = torch.Generator().manual_seed(0)
        low = torch.randn(3, 32, 16, 16, generator=g)
        planes = torch.nn.functional.interpolate(low, size=(256, 256), mode="bilinear", align_corners=False)
        planes = ((planes + 0.1 * torch.randn(3, 32, 256, 256, generator=g)) * 4.0).reshape(1, 3, 32, 256, 256).contiguous()
        w0 = torch.randn(64, 32, generator=g)
        b0 = torch.randn(64, generator=g) * 0.5
        w1 = torch.randn(33, 64, generator=g)
        b1 = torch.randn(33, generator=g) * 0.5
        w1[0] *= 30.0
        b1[0] = -45.0
        return planes.numpy(), (w0.numpy(), b0.numpy(), w1.numpy(), b1.numpy())
    raise ValueError(f"unknown bench scene {scene!r}")


def bench_rendering_kwargs(Sc=48, Sf=48):
    return dict(RENDERING_KWARGS, depth_resolution=int(Sc), depth_resolution_importance=int(Sf))


# ---- the full-size generator without a checkpoint (tests/golden/make_golden_fullsize.py, tests/test_hip_synthesis.py) ----------
# rendering_kwargs of the released configuration (trainers/train_eclustrousC.py:409-440) at the trainer's 48+48 samples
FULL_RK = {"image_resolution": 512, "disparity_space_sampling": False, "clamp_mode": "softplus",
           "superresolution_module": "training.superresolution.SuperresolutionHybrid8XDC", "c_gen_conditioning_zero": False,
           "gpc_reg_prob": 0.5, "c_scale": 1.0, "superresolution_noise_mode": "none", "density_reg": 0.25,
           "density_reg_p_dist": 0.004, "reg_type": "l1", "decoder_lr_mul": 1.0, "sr_antialias": True, "white_back": True,
           "triplane_depth": 1, "use_triplane": 1, "tanh_rgb_output": False, "box_warp": 0.7, "ray_start": 0.5, "ray_end": 1.5,
           "depth_resolution": 48, "depth_resolution_importance": 48, "avg_camera_radius": 1.0, "avg_camera_pivot": [0, 0, 0]}
# constructor kwargs that mirror the released model: 30 M parameters, StyleGAN2-256 backbone (512 channels up to
# 64^2), 96-channel planes, SuperresolutionHybrid8XDC with 256 hidden channels.  sr_num_fp16_res = 0: the reference runs its
# super-resolution in fp32 on the CPU whatever this says (networks_stylegan2.py:444-445), and the CPU run is what the fixture holds.
FULL_KW = dict(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, sr_num_fp16_res=0,
               mapping_kwargs={"num_layers": 2}, rendering_kwargs=FULL_RK,
               sr_kwargs={"channel_base": 32768, "channel_max": 512, "fused_modconv_default": "inference_only"},
               cond_mode="none", triplane_width=32, sr_channels_hidden=256, backbone_resolution=256, channel_base=32768,
               channel_max=512, fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None)


def fill_generator_params(G, seed):
    """Deterministic parameters for a TriPlaneGenerator of ANY implementation (the reference's or ours: same parameter / buffer
    names by construction), independent of the order in which a constructor draws its own random numbers: every parameter and
    every `noise_const` buffer, in sorted-name order, from one seeded CPU generator.  Weights N(0,1) (StyleGAN2's init; the
    layers apply their own 1/sqrt(fan_in) gains), affine biases 1 (their init), other biases N(0, 0.2^2), noise strengths 0.1,
    ToRGB weights x8 and a strong sigma row so that the planes are O(1) features and the volume has surfaces."""
    g = torch.Generator().manual_seed(int(seed))
    with torch.no_grad():
        named = dict(G.named_parameters())
        named.update({n: b for n, b in G.named_buffers() if n.endswith("noise_const")})
        for name in sorted(named):
            p = named[name]
            if name.endswith("noise_strength"):
                p.fill_(0.1)
            elif name.endswith("affine.bias"):
                p.fill_(1.0)
            elif name.endswith(".bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
            else:
                p.copy_(torch.randn(p.shape, generator=g))
            if name.endswith("torgb.weight") and name.startswith("backbone."):
                p.mul_(8.0)
        G.decoder.net[2].weight[0] *= 20.0
        G.decoder.net[2].bias[0] = 25.0
    return G


# ---- index-level parity against the REFERENCE at BASELINE scale (tests/golden/bench_reference_block.npz) -------------------------
REF_TOL = dict(feat=1e-4, depth=2e-5, wsum=3e-5, xyz=1e-4)  # the tolerances of tests/ vs the reference


def reference_index_report(z, key, ours, outputs, tol=REF_TOL, max_detail=16):
    """"`inds` and sort permutation: exact-match count = 100 % or reported mismatch count with cause".

    z = the loaded bench_reference_block.npz (the REFERENCE's own searchsorted indices, sort permutation, mask bits, fine depths and
    its lists of draws / samples that sit next to a decision boundary: tests/golden/make_golden_bench.py); key = 'surface' (48+48)
    or 'surface96'; ours = dict(inds, perm, depths_coarse, depths_fine, sigma_coarse, sigma_fine) of THIS implementation on the
    same rays and draws (numpy; sigma after the masks); outputs = (feat, depth, wsum, xyz).

    Every mismatch is attributed to a cause measured on the reference's side:
      inds      the draw's u lies within `near_ulps` float32 ulps of an edge of the reference's own cdf (two fp32 evaluation
                orders of the same cdf round an edge differently);
      perm      the two depths the reference ordered differ by no more than this implementation's deviation from the
                reference's fine depths (or tie exactly: torch.sort is not stable);
      mask      the reference's own opacity is within `near_thr_abs` of the cull threshold, or |x|,|z| within `near_ulps` ulps of
                the crop limit.
    and every ray whose outputs differ by more than the tolerance must contain an attributed inds mismatch or mask flip."""
    p = key + "_"
    Sc, Sf = int(z[p + "Sc"]), int(z[p + "Sf"])
    R = int(z["side"]) ** 2
    rep = {"rays": R, "Sc": Sc, "Sf": Sf}
    # -- searchsorted indices
    ref_inds = z[p + "inds"].astype(np.int64)
    my_inds = np.asarray(ours["inds"]).reshape(R, Sf).astype(np.int64)
    near = {int(i): (int(k), float(u)) for i, k, u in zip(z[p + "near_edge_index"], z[p + "near_edge_k"], z[p + "near_edge_ulps"])}
    bad = np.argwhere(my_inds != ref_inds)
    det, bad_rays, unexplained = [], set(), 0
    for r, i in bad:
        ent = near.get(int(r) * Sf + int(i))
        ok = ent is not None and abs(int(my_inds[r, i]) - int(ref_inds[r, i])) == 1
        unexplained += not ok
        if ok:
            bad_rays.add(int(r))
        if len(det) < max_detail:
            det.append({"ray": int(r), "draw": int(i), "ours": int(my_inds[r, i]), "reference": int(ref_inds[r, i]),
                        "u_ulps_to_reference_cdf_edge": None if ent is None else abs(ent[1]), "explained": bool(ok)})
    rep.update(inds_total=R * Sf, inds_mismatch=int(len(bad)), inds_unexplained=int(unexplained), inds_mismatch_detail=det)
    # -- sort permutation
    ref_perm = z[p + "perm"].astype(np.int64)
    my_perm = np.asarray(ours["perm"]).reshape(R, Sc + Sf).astype(np.int64)
    dc = np.asarray(ours["depths_coarse"], np.float32).reshape(R, Sc)
    d_ref = np.concatenate([dc, z[p + "depths_fine"]], 1)  # coarse depths are bit-identical to the reference's (test_hip_parity.py)
    d_my = np.concatenate([dc, np.asarray(ours["depths_fine"], np.float32).reshape(R, Sf)], 1)
    dev = np.abs(d_my.astype(np.float64) - d_ref)
    pb = np.argwhere(my_perm != ref_perm)
    worst, punexp = 0.0, 0
    for r, j in pb:
        a, b = ref_perm[r, j], my_perm[r, j]
        gap = abs(float(d_ref[r, a]) - float(d_ref[r, b]))
        worst = max(worst, gap / float(np.spacing(np.float32(d_ref[r, a]))))
        punexp += not (gap <= dev[r, a] + dev[r, b] + 1e-12)
    rep.update(perm_total=R * (Sc + Sf), perm_mismatch=int(len(pb)), perm_mismatch_rays=int(len({int(r) for r, _ in pb})),
               perm_unexplained=int(punexp), perm_max_gap_ulps=worst,
               fine_depth_max_abs_dev=float(np.abs(d_my[:, Sc:].astype(np.float64) - d_ref[:, Sc:]).max()) if Sf else 0.0)
    # -- crop / cull decisions
    thr, lim = float(z[p + "cull_thresh"]), float(z[p + "crop_limit"])
    near_thr = {(int(ps), int(i)): (float(a), float(s)) for ps, i, a, s in
                zip(z[p + "near_thr_pass"], z[p + "near_thr_index"], z[p + "near_thr_alpha"], z[p + "near_thr_sigma"])}
    near_crop = {(int(ps), int(i)): [float(x) for x in v] for ps, i, v in zip(z[p + "near_crop_pass"], z[p + "near_crop_index"], z[p + "near_crop_abs"])}
    flips, fdet, funexp = 0, [], 0
    for ps, (name, S) in enumerate((("coarse", Sc), ("fine", Sf))):
        ref_m = np.unpackbits(z[p + "masked_" + name], axis=1)[:, :S].astype(bool)
        my_m = np.asarray(ours["sigma_" + name]).reshape(R, S) == -1000.0
        for r, i in np.argwhere(ref_m != my_m):
            flips += 1
            k = (ps, int(r) * S + int(i))
            e = {"ray": int(r), "pass": name, "sample": int(i), "masked_here": bool(my_m[r, i]), "masked_in_reference": bool(ref_m[r, i])}
            if k in near_thr:
                a, s = near_thr[k]
                e.update(cause="cull threshold", reference_alpha=a, reference_sigma=s, alpha_abs_to_threshold=abs(a - thr),
                         alpha_ulps_to_threshold=abs(a - thr) / float(np.spacing(np.float32(thr))))
            elif k in near_crop:
                e.update(cause="crop limit", reference_abs_xz=near_crop[k],
                         ulps_to_limit=min(abs(x - lim) for x in near_crop[k]) / float(np.spacing(np.float32(lim))))
            else:
                e.update(cause=None)
                funexp += 1
            if e["cause"]:
                bad_rays.add(int(r))
            if len(fdet) < max_detail:
                fdet.append(e)
    rep.update(mask_total=R * (Sc + Sf), mask_flips=int(flips), mask_unexplained=int(funexp), mask_flip_detail=fdet)
    # -- outputs
    beyond = np.zeros(R, bool)
    for name, g in zip(("feat", "depth", "wsum", "xyz"), outputs):
        want = z[p + name]
        err = np.abs(np.asarray(g).reshape(want.shape) - want).reshape(R, -1).max(-1)
        rep["max_abs_" + name] = float(err.max())
        rep["median_abs_" + name] = float(np.median(err))
        beyond |= err > tol[name]
    rep["rays_beyond_tolerance"] = int(beyond.sum())
    rep["rays_beyond_tolerance_unexplained"] = int(sum(int(r) not in bad_rays for r in np.flatnonzero(beyond)))
    rep["rays_with_attributed_cause"] = sorted(bad_rays)[:256]
    rep["all_explained"] = bool(unexplained == 0 and punexp == 0 and funexp == 0 and rep["rays_beyond_tolerance_unexplained"] == 0)
    return rep
