#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (imported unmodified from /root/reference, CPU).

This script only runs in the build container (the GPU box has no /root/reference); its outputs are committed.
Nothing here is imported by the product or by the tests — the tests read the .npz files.

    python tests/golden/make_golden.py            # regenerates every fixture

What is pinned: ImportanceRenderer.forward with per-stage captures (depths, sigma, weights, searchsorted indices,
sort permutation), run_model, MipRayMarcher2, sample_importance, RaySampler, get_rays_ortho, camera labels.
Randomness: the reference draws torch.rand_like / torch.rand internally; we seed torch, capture the actual draws and
check that tests/p3d_testing.make_random_draws(seed) reproduces them bit-for-bit, so fixtures store only the seed.
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

os.environ.setdefault("PROJECT_DN", "/root/reference")
os.environ.setdefault("PROJECT_NAME", "x")
sys.path[:0] = ["/root/reference"]
sys.path.append("/root/reference/_train/eg3dc/src")
sys.modules.setdefault("kornia", types.ModuleType("kornia"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import p3d_testing as T  # noqa: E402
from training.volumetric_rendering import renderer as ref_renderer  # noqa: E402
from training.volumetric_rendering.renderer import ImportanceRenderer  # noqa: E402
from training.volumetric_rendering.ray_marcher import MipRayMarcher2  # noqa: E402
from training.volumetric_rendering.ray_sampler import RaySampler  # noqa: E402
from training.triplane import OSGDecoder  # noqa: E402
import _databacks.lustrous_renders_v1 as dklustr  # noqa: E402

torch.set_grad_enabled(False)


def ref_decoder(seed, lr_mul=1.0, force_sigmoid=True, sigma_gain=1.0):
    dec = OSGDecoder(32, {"decoder_lr_mul": lr_mul, "decoder_output_dim": 32})
    w0, b0, w1, b1 = T.make_decoder_params(seed, lr_mul, sigma_gain)
    dec.net[0].weight.copy_(torch.from_numpy(w0))
    dec.net[0].bias.copy_(torch.from_numpy(b0))
    dec.net[2].weight.copy_(torch.from_numpy(w1))
    dec.net[2].bias.copy_(torch.from_numpy(b1))
    dec.set_force_sigmoid(force_sigmoid)
    return dec


def persp_rays(elev, azim, fov, res, dist=1.0):
    cp = dklustr.camera_params_to_matrix("eg3d_lustrousB", elev=elev, azim=azim, dist=dist, fov=fov)
    label = cp["camera_label"][None]
    o, d = RaySampler()(label[:, :16].view(-1, 4, 4), label[:, 16:25].view(-1, 3, 3), res)
    return o, d, label


def ortho_rays(elev, azim, res, bw=0.7, dist=1.0):
    fr = dklustr.get_rays_ortho(elev, azim, dist, bw, res)
    o = fr["ray_origins"].permute(0, 2, 3, 1).reshape(1, res * res, 3)  # triplane.py:181-182 rearrange
    d = fr["ray_directions"].permute(0, 2, 3, 1).reshape(1, res * res, 3)
    return o.contiguous(), d.contiguous()


class Capture:
    """Monkey-patch hooks that record the reference's intermediates during one forward()."""

    def __init__(self, rend):
        self.rend = rend
        self.rec = {}

    def __enter__(self):
        rec, rend = self.rec, self.rend
        self._orig = dict(rand_like=torch.rand_like, rand=torch.rand, searchsorted=torch.searchsorted, sort=torch.sort,
                          run_model=rend.run_model, marcher=rend.ray_marcher.run_forward,
                          strat=rend.sample_stratified, imp=rend.sample_importance)
        o = self._orig

        def rand_like(x, *a, **k):
            r = o["rand_like"](x, *a, **k)
            rec.setdefault("jitter", r.clone())
            return r

        def rand(*a, **k):
            r = o["rand"](*a, **k)
            rec.setdefault("u", r.clone())
            return r

        def searchsorted(cdf, u, **k):
            r = o["searchsorted"](cdf, u, **k)
            rec["inds"] = r.clone()
            rec["cdf"] = cdf.clone()
            return r

        def sort(x, **k):
            r = o["sort"](x, **k)
            rec["perm"] = r[1].clone()
            return r

        def run_model(*a, **k):
            out = o["run_model"](*a, **k)
            rec.setdefault("run_model", []).append(out)  # tensors are later modified in place by the masks
            return out

        def marcher(colors, densities, depths, opts):
            r = o["marcher"](colors, densities, depths, opts)
            rec.setdefault("marcher", []).append(tuple(x.clone() for x in r))
            return r

        def strat(*a, **k):
            r = o["strat"](*a, **k)
            rec["depths_coarse_ref"] = r  # jitter is added in place before return
            if isinstance(a[1], torch.Tensor):  # ray_start = ray_end = 'auto' (renderer.py:165-171): the per-ray limits, already patched
                rec["ray_start"], rec["ray_end"] = a[1].clone(), a[2].clone()
            return r

        def imp(*a, **k):
            r = o["imp"](*a, **k)
            rec["depths_fine"] = r.clone()
            return r

        torch.rand_like, torch.rand, torch.searchsorted, torch.sort = rand_like, rand, searchsorted, sort
        rend.run_model, rend.ray_marcher.run_forward = run_model, marcher
        rend.sample_stratified, rend.sample_importance = strat, imp
        return self

    def __exit__(self, *exc):
        o = self._orig
        torch.rand_like, torch.rand, torch.searchsorted, torch.sort = o["rand_like"], o["rand"], o["searchsorted"], o["sort"]
        for k in ("run_model", "sample_stratified", "sample_importance"):
            self.rend.__dict__.pop(k, None)
        self.rend.ray_marcher.__dict__.pop("run_forward", None)


def run_reference(planes, dec, rays_o, rays_d, ro, seed, crop, cull, binarize):
    rend = ImportanceRenderer(use_triplane=bool(ro.get("use_triplane", False)))
    N, R = rays_o.shape[:2]
    Sc, Sf = ro["depth_resolution"], ro["depth_resolution_importance"]
    with Capture(rend) as cap:
        torch.manual_seed(seed)
        feat, depth, wsum, xyz = rend(torch.from_numpy(planes), dec, rays_o, rays_d, ro, triplane_crop=crop,
                                      cull_clouds=cull, binarize_clouds=binarize)
    rec = cap.rec
    jit, u = T.make_random_draws(seed, N, R, Sc, Sf, auto_limits=(ro["ray_start"] == "auto"))
    assert np.array_equal(rec["jitter"].numpy(), jit), "rand_like draw not reproduced from the seed"
    if Sf > 0:
        assert np.array_equal(rec["u"].numpy(), u), "rand draw not reproduced from the seed"
    out = dict(feat=feat.numpy(), depth=depth.numpy(), wsum=wsum.numpy(), xyz=xyz.numpy())
    out["depths_coarse"] = rec["depths_coarse_ref"].reshape(N * R, Sc).numpy()
    if "ray_start" in rec:
        out["ray_start"], out["ray_end"] = rec["ray_start"].reshape(N, R).numpy(), rec["ray_end"].reshape(N, R).numpy()
    rm = rec["run_model"]
    out["sigma_coarse"] = rm[0]["sigma"].reshape(N * R, Sc).numpy()
    out["rgb_coarse"] = rm[0]["rgb"].reshape(N * R, Sc, 32).numpy()
    if Sf > 0:
        out["weights_coarse"] = rec["marcher"][0][2].reshape(N * R, Sc - 1).numpy()
        out["depths_fine"] = rec["depths_fine"].reshape(N * R, Sf).numpy()
        out["inds"] = rec["inds"].reshape(N * R, Sf).numpy().astype(np.int32)
        out["sigma_fine"] = rm[1]["sigma"].reshape(N * R, Sf).numpy()
        out["perm"] = rec["perm"].reshape(N * R, Sc + Sf).numpy().astype(np.int32)
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def render_case(name, *, res, Sc, Sf, N=1, H=256, W=256, seed=0, views=((0.0, 20.0, 30.0),), ortho=False, crop=0.1, cull=0.5,
                binarize=None, use_triplane=1, white_back=True, force_sigmoid=True, lr_mul=1.0, plane_scale=1.0,
                smooth=0, sigma_gain=1.0, ray_start=0.5, ray_end=1.5, disparity=False, keep=("feat", "depth", "wsum", "xyz", "inds", "perm", "depths_fine",
                                                  "weights_coarse", "sigma_coarse", "sigma_fine", "depths_coarse")):
    auto = ray_start == "auto" and ray_end == "auto"
    ro = dict(T.RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf, use_triplane=use_triplane,
              white_back=white_back, ray_start=ray_start, ray_end=ray_end, disparity_space_sampling=bool(disparity))
    planes = T.make_planes(seed, N, H, W, scale=plane_scale, smooth=smooth)
    dec = ref_decoder(seed + 1, lr_mul, force_sigmoid, sigma_gain)
    os_, ds_ = [], []
    for (elev, azim, fov) in views:
        if ortho:
            o, d = ortho_rays(elev, azim, res)
        else:
            o, d, _ = persp_rays(elev, azim, fov, res)
        os_.append(o)
        ds_.append(d)
    rays_o, rays_d = torch.cat(os_), torch.cat(ds_)
    assert rays_o.shape[0] == N
    out = run_reference(planes, dec, rays_o, rays_d, ro, seed + 2, crop, cull, binarize)
    meta = dict(res=res, Sc=Sc, Sf=Sf, N=N, H=H, W=W, seed=seed, crop=crop or 0.0, cull=cull or 0.0, binarize=binarize or 0.0,
                use_triplane=use_triplane, white_back=int(white_back), force_sigmoid=int(force_sigmoid), lr_mul=lr_mul,
                plane_scale=plane_scale, smooth=smooth, sigma_gain=sigma_gain, ray_start=0.0 if auto else ray_start,
                ray_end=0.0 if auto else ray_end, auto_limits=int(auto), disparity=int(bool(disparity)), box_warp=ro["box_warp"])
    arrs = {k: v for k, v in out.items() if k in keep or (auto and k in ("ray_start", "ray_end"))}
    arrs["rays_o"] = rays_o.numpy()
    arrs["rays_d"] = rays_d.numpy()
    arrs["planes_checksum"] = np.array(T.checksum(planes))
    for k, v in meta.items():
        arrs["meta_" + k] = np.array(v)
    save(name, **arrs)


def decode_case():
    seed = 10
    planes = T.make_planes(seed, 2, 64, 96)  # non-square, batch 2
    dec = ref_decoder(seed + 1, 1.0, False)  # force_sigmoid off -> *1.002-0.001 branch
    pts = T.make_points(seed + 2, 2, 4096, extent=0.45)
    ro = dict(T.RENDERING_KWARGS)
    for ut in (0, 1):
        rend = ImportanceRenderer(use_triplane=bool(ut))
        out = rend.run_model(torch.from_numpy(planes), dec, torch.from_numpy(pts), None, ro)
        save(f"decode_points_ut{ut}.npz", sigma=out["sigma"].numpy(), rgb=out["rgb"].numpy(), meta_seed=np.array(seed),
             planes_checksum=np.array(T.checksum(planes)), pts_checksum=np.array(T.checksum(pts)))


def decoder_forward_case():
    """OSGDecoder.forward itself (training/triplane.py:528-544) on sampled features [N,3,M,32] — the module's own call surface,
    which our OSGDecoder.forward / p3d_decode_features_f32 mirror: both sigmoid branches, lr_mul 1 and 0.5."""
    seed = 40
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(2, 3, 777, 32, generator=g) * 2.0
    for tag, lr_mul, fs in (("a", 1.0, True), ("b", 0.5, False)):
        dec = ref_decoder(seed + 1, lr_mul, fs, sigma_gain=5.0)
        out = dec(feats, None)
        save(f"decoder_forward_{tag}.npz", sigma=out["sigma"].numpy(), rgb=out["rgb"].numpy(), meta_seed=np.array(seed),
             meta_lr_mul=np.array(lr_mul), meta_force_sigmoid=np.array(int(fs)), feats_checksum=np.array(T.checksum(feats.numpy())))


def stage_cases():
    g = torch.Generator().manual_seed(20)
    NR, S, K = 256, 24, 35
    depths = torch.sort(torch.rand(2, NR // 2, S, 1, generator=g) + 0.5, dim=-2)[0]
    colors = torch.rand(2, NR // 2, S, K, generator=g)
    sigma = torch.randn(2, NR // 2, S, 1, generator=g) * 4
    sigma[0, :8] = -1000.0  # empty rays -> NaN depth -> +inf -> clamp to max
    m = MipRayMarcher2()
    for wb in (0, 1):
        rgb, depth, w = m(colors, sigma, depths, dict(clamp_mode="softplus", white_back=bool(wb)))
        save(f"marcher_wb{wb}.npz", colors=colors.numpy(), sigma=sigma.numpy(), depths=depths.numpy(), rgb=rgb.numpy(),
             depth=depth.numpy(), weights=w.numpy())
    # importance: weights from the marcher above, coarse depths = sorted depths
    rend = ImportanceRenderer()
    Sf = 20
    with Capture(rend) as cap:
        torch.manual_seed(21)
        fine = rend.sample_importance(depths, w, Sf)
    save("importance.npz", depths=depths.numpy(), weights=w.numpy(), u=cap.rec["u"].numpy(), fine=fine.numpy(),
         inds=cap.rec["inds"].numpy().astype(np.int32), cdf=cap.rec["cdf"].numpy())
    # stratified depths with awkward limits (linspace rounding)
    for (a, b, S_) in ((0.5, 1.5, 48), (2.25, 3.3, 96), (0.88, 1.12, 17)):
        torch.manual_seed(22)
        d = rend.sample_stratified(torch.zeros(1, 64, 3), a, b, S_, False)
        torch.manual_seed(22)
        j = torch.rand(1, 64, S_, 1)
        save(f"stratified_{S_}.npz", depths=d.numpy(), jitter=j.numpy(), start=np.array(a), end=np.array(b))


def ray_cases():
    arrs = {}
    for i, (elev, azim, fov) in enumerate(((0.0, 0.0, 30.0), (10.0, 135.0, 12.0), (-20.0, -60.0, 45.0))):
        o, d, label = persp_rays(elev, azim, fov, 16)
        arrs[f"persp{i}_o"], arrs[f"persp{i}_d"], arrs[f"persp{i}_label"] = o.numpy(), d.numpy(), label.numpy()
        arrs[f"persp{i}_cam"] = np.array([elev, azim, fov], np.float64)
        fr = dklustr.get_rays_ortho(elev, azim, 1.0, 0.7, 16)
        arrs[f"ortho{i}_o"], arrs[f"ortho{i}_d"] = fr["ray_origins"].numpy(), fr["ray_directions"].numpy()
    save("rays.npz", **arrs)


def main():
    torch.set_num_threads(8)
    # BASELINE config 1: single 64x64 image, 32 samples/ray, single pass
    render_case("render_c1_64x64_s32.npz", res=64, Sc=32, Sf=0, seed=100, keep=("feat", "depth", "wsum", "xyz"))
    # two-pass with every stage dumped, batch of 2 views
    render_case("render_32x32_16p16.npz", res=32, Sc=16, Sf=16, N=2, seed=301, plane_scale=4.0, smooth=8, sigma_gain=60.0,
                views=((0.0, 20.0, 30.0), (15.0, 200.0, 30.0)))
    # trainer default 48+48 (BASELINE config 2 sampling), ortho rays like generate.py's front views
    render_case("render_24x24_48p48_ortho.npz", res=24, Sc=48, Sf=48, seed=303, plane_scale=4.0, smooth=8, sigma_gain=60.0, ortho=True, views=((0.0, 0.0, -1.0),),
                keep=("feat", "depth", "wsum", "xyz", "inds", "perm", "depths_fine"))
    # eval-faithful 96+96 (eg3dc_v0.py:30-31)
    render_case("render_12x12_96p96.npz", res=12, Sc=96, Sf=96, seed=307, plane_scale=4.0, smooth=8, sigma_gain=60.0,
                keep=("feat", "depth", "wsum", "xyz", "inds", "perm", "depths_fine"))
    # option variants: EG3D-stock plane orientation, no masks, black background, mipnerf sigmoid, small planes, lr_mul
    render_case("render_variant_a.npz", res=16, Sc=12, Sf=10, seed=500, use_triplane=0, crop=None, cull=None, white_back=False,
                force_sigmoid=False, H=64, W=48, lr_mul=0.5, plane_scale=2.0)
    render_case("render_variant_b.npz", res=16, Sc=20, Sf=8, seed=600, binarize=0.4, cull=None, crop=0.05, plane_scale=4.0, smooth=8, sigma_gain=20.0,
                ray_start=0.6, ray_end=1.4)
    decode_case()
    decoder_forward_case()
    stage_cases()
    ray_cases()
    auto_case()
    density_noise_case()


def auto_case():
    """ray_start = ray_end = 'auto' (renderer.py:165-171): per-ray limits from the box intersection; a wide field of view so that
    part of the rays miss the box and take the patched limits."""
    render_case("render_auto_limits.npz", res=20, Sc=24, Sf=16, seed=700, plane_scale=4.0, smooth=8, sigma_gain=40.0,
                views=((10.0, 35.0, 50.0),), ray_start="auto", ray_end="auto",
                keep=("feat", "depth", "wsum", "xyz", "inds", "perm", "depths_fine", "depths_coarse", "weights_coarse"))
    # disparity_space_sampling (renderer.py:309-316): coarse samples uniform in 1 / depth
    render_case("render_disparity.npz", res=16, Sc=32, Sf=24, seed=710, plane_scale=4.0, smooth=8, sigma_gain=40.0,
                views=((5.0, 320.0, 30.0),), ray_start=0.55, ray_end=1.45, disparity=True,
                keep=("feat", "depth", "wsum", "xyz", "inds", "perm", "depths_fine", "depths_coarse", "weights_coarse"))


def density_noise_case():
    """rendering_options['density_noise'] > 0 (renderer.py:276-277): run_model adds randn_like(sigma) * density_noise BEFORE the crop /
    cull masks, in both passes.  The reference's draws in call order — rand_like (jitter), randn_like (coarse noise), rand (u), randn_like
    (fine noise) — are captured and stored (the two uniform draws no longer follow each other in the generator's stream, so
    make_random_draws(seed) does not reproduce u).  Also stored: the same view WITHOUT noise through the same fixture planes."""
    seed, res, Sc, Sf, dn = 720, 16, 16, 12, 0.75
    ro = dict(T.RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf, density_noise=dn)
    planes = T.make_planes(seed, 1, 256, 256, scale=4.0, smooth=8)
    dec = ref_decoder(seed + 1, 1.0, True, 40.0)
    o, d, _ = persp_rays(5.0, 25.0, 30.0, res)
    rend = ImportanceRenderer(use_triplane=True)
    rec = {"noise": []}
    orig = dict(rand_like=torch.rand_like, rand=torch.rand, randn_like=torch.randn_like)

    def rand_like(x, *a, **k):
        r = orig["rand_like"](x, *a, **k)
        rec.setdefault("jitter", r.clone())
        return r

    def rand(*a, **k):
        r = orig["rand"](*a, **k)
        rec.setdefault("u", r.clone())
        return r

    def randn_like(x, *a, **k):
        r = orig["randn_like"](x, *a, **k)
        rec["noise"].append(r.clone())
        return r

    torch.rand_like, torch.rand, torch.randn_like = rand_like, rand, randn_like
    try:
        torch.manual_seed(seed + 2)
        feat, depth, wsum, xyz = rend(torch.from_numpy(planes), dec, o, d, ro, triplane_crop=0.1, cull_clouds=0.5)
    finally:
        torch.rand_like, torch.rand, torch.randn_like = orig["rand_like"], orig["rand"], orig["randn_like"]
    assert len(rec["noise"]) == 2 and rec["noise"][0].shape == (1, res * res * Sc, 1) and rec["noise"][1].shape == (1, res * res * Sf, 1)
    save("render_density_noise.npz", feat=feat.numpy(), depth=depth.numpy(), wsum=wsum.numpy(), xyz=xyz.numpy(),
         jitter=rec["jitter"].numpy(), u=rec["u"].numpy(), noise_coarse=rec["noise"][0].numpy(), noise_fine=rec["noise"][1].numpy(),
         rays_o=o.numpy(), rays_d=d.numpy(), planes_checksum=np.array(T.checksum(planes)), meta_seed=np.array(seed), meta_res=np.array(res),
         meta_Sc=np.array(Sc), meta_Sf=np.array(Sf), meta_density_noise=np.array(dn), meta_sigma_gain=np.array(40.0))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "density_noise":  # only the fixture added at the end of round 4
        torch.set_num_threads(8)
        density_noise_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "auto":  # only the fixture added in round 2 (the others are unchanged)
        torch.set_num_threads(8)
        auto_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "decoder":  # only the fixtures added in round 4
        decoder_forward_case()
    else:
        main()
