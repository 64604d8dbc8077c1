#!/usr/bin/env python3
"""tests/golden/bench_reference_block.npz — the REFERENCE renderer's output on a block of bench.py's frame.

BUILD CONTAINER ONLY (imports the unmodified reference from /root/reference on CPU; the GPU box has no reference tree).
bench.py's `verify.reference_block` re-renders exactly these rays with exactly these draws on the HIP path and prints
max-abs and PSNR of `image_raw` against this file: the "PSNR vs ref" half of BASELINE.json's metric, measured inside the
driver's own run against the real reference (not the port).

What is stored, per bench scene with a non-empty volume ('surface' at 48+48, 'surface96' at 96+96): the centre SIDE x SIDE rays
of bench.py's 512^2 view (azimuth 20, fov 30), the reference's four outputs, seed of the two draws
(p3d_testing.make_random_draws reproduces the reference's rand_like / rand bit for bit — asserted here), and — SURVEY 8(d)
"`inds` and sort permutation: exact-match count = 100 % or reported mismatch count with cause" at BASELINE scale:

  <p>inds  u8 [R,Sf]      torch.searchsorted's result inside sample_pdf (renderer.py:374)
  <p>perm  u8 [R,Sc+Sf]   torch.sort's permutation inside unify_samples (renderer.py:295)
  <p>depths_fine f32 [R,Sf]  sample_importance's return value (renderer.py:213)
  <p>masked_coarse / masked_fine  bit-packed [R,Sc] / [R,Sf]: the density the reference marches is -1e3 (crop or cull mask)
  <p>near_edge_*    the draws whose u is within NEAR_ULPS float32 ulps of an edge of the REFERENCE's own cdf (flat index r*Sf+i,
                    k, signed distance in ulps of u to the nearest edge): the only draws whose bin can legitimately differ between
                    two fp32 evaluation orders
  <p>near_thr_*     the samples whose opacity 1 - exp(-softplus(sigma - 1)) — the reference's own float, captured inside
                    cull_clouds_mask (renderer.py:150-153) — is within NEAR_THR_ABS of the cull threshold (pass 0 = coarse,
                    1 = fine; flat index; alpha; the raw sigma it came from): the only samples whose cull decision can
                    legitimately flip.  The window is absolute, not a few ulps of 0.5: the surface scene's sigma row is
                    30 x weights with bias -45, so a sigma near the threshold (sigma = 1) is the difference of two numbers of
                    magnitude 45 and carries the rounding of that magnitude (1 ulp of 45 = 3.8e-6 in sigma ~ 1e-6 in alpha per
                    operation of a 64-term sum, plus the sample position's own ulp times the field's slope)
  <p>near_crop_*    the same for triplane_crop_mask's comparison |x|, |z| <= bw/2 - tc (renderer.py:138-149)

    python tests/golden/make_golden_bench.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets up the reference import path; its Capture records the stage tensors)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import p3d_testing as T  # noqa: E402
from training.volumetric_rendering import renderer as ref_renderer  # noqa: E402
from training.volumetric_rendering.renderer import ImportanceRenderer  # noqa: E402
from training.volumetric_rendering.ray_sampler import RaySampler  # noqa: E402
from training.triplane import OSGDecoder  # noqa: E402
import _databacks.lustrous_renders_v1 as dklustr  # noqa: E402

torch.set_grad_enabled(False)
RES, SIDE, SEED = 512, 64, 4242
NEAR_ULPS = 64.0
NEAR_THR_ABS = 2e-3  # the window inside which the tolerance mode re-decodes a sample exactly


def block_rays(res=RES, side=SIDE, azim=20.0):
    cp = dklustr.camera_params_to_matrix("eg3d_lustrousB", elev=0.0, azim=azim, dist=1.0, fov=30.0)
    label = cp["camera_label"][None]
    o, d = RaySampler()(label[:, :16].view(-1, 4, 4), label[:, 16:25].view(-1, 3, 3), res)
    a = (res - side) // 2
    idx = (torch.arange(a, a + side)[:, None] * res + torch.arange(a, a + side)[None, :]).reshape(-1)
    return o[:, idx].contiguous(), d[:, idx].contiguous()


class MaskCapture:
    """Records what the reference's two mask functions computed, call by call (coarse pass, then fine pass)."""

    def __init__(self):
        self.alpha, self.sigma, self.thr, self.crop_abs, self.crop_lim = [], [], [], [], []

    def __enter__(self):
        self._cull, self._crop = ref_renderer.cull_clouds_mask, ref_renderer.triplane_crop_mask
        me = self

        def cull(densities, thresh):  # renderer.py:150-153, the same three operations, alpha kept
            alpha = 1 - torch.exp(-torch.nn.functional.softplus(densities - 1))
            me.alpha.append(alpha.clone())
            me.sigma.append(densities.clone())
            me.thr.append(float(thresh))
            got = me._cull(densities, thresh)
            assert torch.equal(got, alpha < thresh)
            return got

        def crop(xyz, thresh, boxwarp, allow_bottom=True):
            me.crop_abs.append(xyz[:, :, [0, 2]].abs().clone())  # renderer.py:143 (the sign flip does not change abs)
            me.crop_lim.append(boxwarp / 2 - thresh)
            return me._crop(xyz, thresh, boxwarp, allow_bottom)

        ref_renderer.cull_clouds_mask, ref_renderer.triplane_crop_mask = cull, crop
        return self

    def __exit__(self, *exc):
        ref_renderer.cull_clouds_mask, ref_renderer.triplane_crop_mask = self._cull, self._crop


def ulps32(x):
    return np.spacing(np.abs(np.asarray(x, np.float32)).astype(np.float32))


def main():
    out = {}
    for scene, (Sc, Sf) in (("surface", (48, 48)), ("surface96", (96, 96))):
        planes_np, raw = T.make_bench_scene("surface")
        dec = OSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32})
        for prm, want in zip((dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias), raw):
            prm.copy_(torch.from_numpy(want))
        dec.set_force_sigmoid(True)
        o, d = block_rays()
        R = SIDE * SIDE
        ro = T.bench_rendering_kwargs(Sc, Sf)
        # the draws the reference is about to make, captured, must be the ones make_random_draws regenerates
        jit, u = T.make_random_draws(SEED, 1, R, Sc, Sf)
        rend = ImportanceRenderer(use_triplane=True)
        with MG.Capture(rend) as cap, MaskCapture() as mc:
            torch.manual_seed(SEED)
            feat, depth, wsum, xyz = rend(torch.from_numpy(planes_np), dec, o, d, ro, **{k: v for k, v in T.BENCH_KW.items() if k != "force_sigmoid"})
        rec = cap.rec
        assert np.array_equal(rec["jitter"].numpy(), jit) and np.array_equal(rec["u"].numpy(), u)
        print(scene, "wsum mean", float(wsum.mean()), "hit", float((wsum > 0.5).float().mean()))
        p = scene + "_"
        out.update({p + "feat": feat.numpy(), p + "depth": depth.numpy(), p + "wsum": wsum.numpy(), p + "xyz": xyz.numpy(),
                    p + "rays_o": o.numpy(), p + "rays_d": d.numpy(), p + "Sc": np.int32(Sc), p + "Sf": np.int32(Sf)})
        # ---- the index-level record
        inds = rec["inds"].reshape(R, Sf).numpy()
        perm = rec["perm"].reshape(R, Sc + Sf).numpy()
        assert inds.max() < 256 and perm.max() < 256
        rm = rec["run_model"]  # the sigma tensors were masked in place after run_model returned them
        mcoarse = rm[0]["sigma"].reshape(R, Sc).numpy() == -1000.0
        mfine = rm[1]["sigma"].reshape(R, Sf).numpy() == -1000.0
        out.update({p + "inds": inds.astype(np.uint8), p + "perm": perm.astype(np.uint8),
                    p + "depths_fine": rec["depths_fine"].reshape(R, Sf).numpy(),
                    p + "masked_coarse": np.packbits(mcoarse, axis=1), p + "masked_fine": np.packbits(mfine, axis=1)})
        # draws within NEAR_ULPS ulps (of u) of an edge of the reference's cdf
        cdf = rec["cdf"].reshape(R, -1).numpy()  # [R, Sc-2]
        uu = rec["u"].numpy()
        k = inds
        lo = np.take_along_axis(cdf, np.clip(k - 1, 0, cdf.shape[1] - 1), 1)
        hi = np.take_along_axis(cdf, np.clip(k, 0, cdf.shape[1] - 1), 1)
        d_lo = (uu.astype(np.float64) - lo) / ulps32(uu)      # >= 0: cdf[k-1] <= u
        d_hi = np.where(k < cdf.shape[1], (hi.astype(np.float64) - uu) / ulps32(uu), np.inf)  # > 0: u < cdf[k]
        near = np.minimum(d_lo, d_hi) <= NEAR_ULPS
        flat = np.flatnonzero(near)
        out.update({p + "near_edge_index": flat.astype(np.int32), p + "near_edge_k": k.reshape(-1)[flat].astype(np.uint8),
                    p + "near_edge_ulps": np.where(d_lo <= d_hi, -d_lo, d_hi).reshape(-1)[flat].astype(np.float32)})
        # samples within NEAR_ULPS ulps of the cull threshold / the crop limit, per pass
        ni, npass, na, nsg = [], [], [], []
        ci, cpass, cv = [], [], []
        for ps in range(2):
            a, thr = mc.alpha[ps].reshape(-1).numpy(), np.float32(mc.thr[ps])
            sel = np.flatnonzero(np.abs(a.astype(np.float64) - float(thr)) <= NEAR_THR_ABS)
            ni.append(sel); npass.append(np.full(sel.size, ps)); na.append(a[sel]); nsg.append(mc.sigma[ps].reshape(-1).numpy()[sel])
            ab, lim = mc.crop_abs[ps].reshape(-1, 2).numpy(), np.float32(mc.crop_lim[ps])
            dist = np.abs(ab.astype(np.float64) - float(lim)) / float(ulps32(lim))
            sel = np.flatnonzero(dist.min(1) <= NEAR_ULPS)
            ci.append(sel); cpass.append(np.full(sel.size, ps)); cv.append(ab[sel])
        out.update({p + "near_thr_index": np.concatenate(ni).astype(np.int32), p + "near_thr_pass": np.concatenate(npass).astype(np.uint8),
                    p + "near_thr_alpha": np.concatenate(na).astype(np.float32), p + "near_thr_sigma": np.concatenate(nsg).astype(np.float32), p + "cull_thresh": np.float32(mc.thr[0]),
                    p + "near_crop_index": np.concatenate(ci).astype(np.int32), p + "near_crop_pass": np.concatenate(cpass).astype(np.uint8),
                    p + "near_crop_abs": np.concatenate(cv).astype(np.float32).reshape(-1, 2), p + "crop_limit": np.float32(mc.crop_lim[0])})
        print(scene, "near-edge draws", flat.size, "of", uu.size, "; near-threshold samples", out[p + "near_thr_index"].size,
              "; near-crop samples", out[p + "near_crop_index"].size, "; masked coarse/fine", mcoarse.mean(), mfine.mean())
    out.update(seed=np.int32(SEED), res=np.int32(RES), side=np.int32(SIDE), near_ulps=np.float32(NEAR_ULPS), near_thr_abs=np.float32(NEAR_THR_ABS),
               planes_checksum=np.str_(T.checksum(T.make_bench_scene("surface")[0])), torch_version=np.str_(torch.__version__))
    path = os.path.join(HERE, "bench_reference_block.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
