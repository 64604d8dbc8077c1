"""Host-side mirror of the reference's volumetric renderer call surface, on the HIP path.

Same class name, method names, argument order and return values as
training/volumetric_rendering/renderer.py:156-387 (ImportanceRenderer) so that
training/triplane.py:209 (`self.renderer(planes, self.decoder, ray_origins, ray_directions, rendering_kwargs, ...)`)
and :292 (`self.renderer.run_model(...)`) keep working when the class is swapped in.

Differences that are deliberate:
  * the two random draws (renderer.py:324 `torch.rand_like`, :371 `torch.rand`) are made on the device with the same
    call order and shapes, or injected via `jitter=` / `u=` (parity tests, CPU-generated draws);
  * the decoder module is read for its parameters only (it is evaluated inside the fused kernel), so it must be an
    OSGDecoder-shaped module: net[0] 32->64, Softplus, net[2] 64->33 (training/triplane.py:516-544);
  * CPU tensors raise: the product path has no CPU fallback.
"""
import os
import weakref

import torch

from . import memo, ops


# The final pass of the fused renderer may run in its tolerance mode (P3D_FLAG_FAST_COLOR: f16 two-term MFMA + hardware
# transcendentals; 15 % faster at 512^2 x 96, PSNR > 90 dB against the exact path, inverse-CDF indices untouched — DESIGN.md
# §4.6).  The reference itself only promises "minor hardware variations" between GPUs (readme.md:74).  Set to False (or pass
# exact=True to forward) for results that are bit-identical to the arithmetic contract (include/p3d_numerics.h).
# How to pin the exact contract: environment P3D_EXACT=1 (process-wide default), ImportanceRenderer(..., exact=True) /
# renderer.exact = True (per instance; TriPlaneGenerator.set_render_exact), or forward(..., exact=True) (per call).
DEFAULT_FAST_COLOR = os.environ.get("P3D_EXACT", "0") != "1"


def decoder_params(decoder):
    """(w0, b0, w1, b1) pre-scaled exactly like FullyConnectedLayer.forward (networks_stylegan2.py:121-127)."""
    l0, l2 = decoder.net[0], decoder.net[2]
    if hasattr(l0, "_scaled") and hasattr(l2, "_scaled"):
        # this package's FullyConnectedLayer keeps `w * weight_gain`, `b * bias_gain` per parameter version (the same two
        # multiplications, so the same bits): two launches and ~20 us of host time less per render
        (w0, b0), (w1, b1) = l0._scaled(torch.float32), l2._scaled(torch.float32)
        return w0, b0, w1, b1
    return ops.prescale_mlp(l0.weight, l0.bias, l2.weight, l2.bias, l0.weight_gain, l0.bias_gain, l2.weight_gain,
                            l2.bias_gain)


class ImportanceRenderer(torch.nn.Module):
    def __init__(self, use_triplane=False, exact=None):
        super().__init__()
        self.use_triplane = bool(use_triplane)  # generate_planes(use_triplane): renderer.py:26-50
        self.exact = exact  # None: the module default (tolerance mode unless P3D_EXACT=1); True / False: this instance's final pass
        self._planes_cache = (None, None, None)

    def _nhwc(self, planes):
        # planes arrive NCHW [N,3,32,H,W] (training/triplane.py:200-206).  Reuse the channels-last copy only while the very
        # same tensor OBJECT (weak reference) at the same version is rendered again — generate.py renders 16 views per
        # subject.  (Keying on data_ptr would alias a freed tensor whose storage the caching allocator handed out again.)
        ref, version, cached = self._planes_cache
        if memo.enabled() and ref is not None and ref() is planes and version == planes._version:
            return cached
        if planes.dim() == 5 and planes.shape[0] > 1 and planes.stride(0) == 0:
            planes_src = planes[:1]  # planes.expand(N, ...): one subject, N views -> one shared channels-last copy
        else:
            planes_src = planes
        cached = ops.planes_to_nhwc(planes_src)

        def _drop(_ref, self_ref=weakref.ref(self)):  # the planes tensor died: free its 25 MB channels-last copy too
            me = self_ref()
            if me is not None and me._planes_cache[0] is _ref:
                me._planes_cache = (None, None, None)

        self._planes_cache = (weakref.ref(planes, _drop), planes._version, cached)
        return cached

    def __getstate__(self):
        """The reference pickles / deep-copies G (snapshots, eg3dc_v0 loader): the cache (a weakref and a device tensor) stays out."""
        state = dict(self.__dict__)
        state["_planes_cache"] = (None, None, None)
        return state

    def _opts(self, rendering_options, decoder, **kw):
        ro = dict(rendering_options)
        ro["use_triplane"] = self.use_triplane
        return ops.make_opts(ro, force_sigmoid=bool(getattr(decoder, "force_sigmoid", False)), **kw)

    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options, triplane_crop=None,
                cull_clouds=None, binarize_clouds=None, jitter=None, u=None, ray_tile_w=None, return_dumps=False,
                per_view_clamp=False, exact=None, density_noise_draws=None, weights_only=False):
        # weights_only (an extension): the caller reads `weights.sum(2)` and the depth only — returns (None, depth, wsum, None); the
        # colours are not decoded where the library has a weights-only kernel (ops.render)
        if (rendering_options.get("density_noise", 0) or 0) > 0:  # renderer.py:276-277 (a training-time option): the staged path
            if return_dumps or per_view_clamp:
                raise NotImplementedError("density_noise > 0 runs the staged path: no per-stage dumps, no per-view clamp")
            return self.forward_staged(planes, decoder, ray_origins, ray_directions, rendering_options, triplane_crop=triplane_crop,
                                       cull_clouds=cull_clouds, binarize_clouds=binarize_clouds, jitter=jitter, u=u,
                                       density_noise_draws=density_noise_draws)
        if exact is None:
            exact = getattr(self, "exact", None)
        fast = DEFAULT_FAST_COLOR if exact is None else not exact
        opts = self._opts(rendering_options, decoder, triplane_crop=triplane_crop, cull_clouds=cull_clouds,
                          binarize_clouds=binarize_clouds, fast_color=fast)
        N, R, _ = ray_origins.shape
        dev = ray_origins.device
        limits = None
        if rendering_options.get("ray_start") == "auto" and rendering_options.get("ray_end") == "auto":  # renderer.py:165-171
            from . import cameras
            limits = cameras.patch_ray_limits(*cameras.ray_limits_box(ray_origins.float(), ray_directions.float(),
                                                                        rendering_options["box_warp"]))
            if jitter is None:  # rand_like of linspace(...).permute(1,2,0,3) fills in memory order [Sc,N,R,1] (renderer.py:317-319)
                jitter = torch.rand((opts.Sc, N, R, 1), dtype=torch.float32, device=dev).permute(1, 2, 0, 3).contiguous()
        if jitter is None:  # renderer.py:324
            jitter = torch.rand((N, R, opts.Sc, 1), dtype=torch.float32, device=dev)
        if u is None and opts.Sf > 0:  # renderer.py:371
            u = torch.rand((N * R, opts.Sf), dtype=torch.float32, device=dev)
        if ray_tile_w is None:  # square images are the reference's only use (training/triplane.py:222-226)
            side = int(round(R ** 0.5))
            ray_tile_w = side if side * side == R else 0
        out = ops.render(self._nhwc(planes), ray_origins.float(), ray_directions.float(), jitter, u,
                         decoder_params(decoder), opts, ray_tile_w=ray_tile_w, dumps=return_dumps, per_view_clamp=per_view_clamp,
                         ray_limits=limits, weights_only=weights_only)
        return out  # rgb_final, depth_final, weights.sum(2), xyz_final  (renderer.py:264)

    # ---- the reference's own structure, stage by stage ---------------------------------------------------------------------
    def forward_staged(self, planes, decoder, ray_origins, ray_directions, rendering_options, triplane_crop=None, cull_clouds=None,
                       binarize_clouds=None, jitter=None, u=None, density_noise_draws=None):
        """ImportanceRenderer.forward (renderer.py:162-264) as the reference structures it — stratified depths, run_model, masks, ray
        marcher, importance resampling, run_model, masks, unify, ray marcher — with every stage on its stand-alone HIP kernel
        (p3d_sample_stratified_f32, p3d_triplane_decode_f32, p3d_composite_f32, p3d_importance_f32, p3d_unify_perm_f32) and the
        reference's element-wise glue (sample points, masks, gathers) as torch ops on the device; intermediates ([N,R*S,32] colours)
        are materialised like the reference's.  It exists for the one option the fused kernel does not take: `density_noise > 0`
        (renderer.py:276-277, `sigma += randn_like(sigma) * density_noise` inside run_model, i.e. BEFORE the crop / cull masks), which
        needs a value per decoded sample that the final pass of the fused kernel, re-decoding the coarse samples, would have to see
        twice.  Without noise it computes what forward() computes (tested), far slower.  Fixed ray limits, linear depth spacing.
        density_noise_draws: (noise of the coarse pass [N,R*Sc,1], of the fine pass [N,R*Sf,1]) — parity tests; default: torch.randn
        on the device, drawn in the reference's order (jitter, coarse noise, u, fine noise)."""
        ro = dict(rendering_options)
        dn = float(ro.pop("density_noise", 0) or 0)
        if ro.get("ray_start") == "auto" or ro.get("ray_end") == "auto" or ro.get("disparity_space_sampling", False):
            raise NotImplementedError("forward_staged: fixed ray limits and linear depth spacing only")
        opts = self._opts(ro, decoder)  # run_model's options: no masks inside the decode (they follow the noise)
        N, R, _ = ray_origins.shape
        dev = ray_origins.device
        Sc, Sf = opts.Sc, opts.Sf
        o, d = ray_origins.float().contiguous(), ray_directions.float().contiguous()
        planes_nhwc, mlp = self._nhwc(planes), decoder_params(decoder)
        bw = float(ro["box_warp"])
        nc, nf = density_noise_draws if density_noise_draws is not None else (None, None)

        def masks(sigma, xyz):  # renderer.py:187-198 with triplane_crop_mask / cull_clouds_mask (:138-153), same torch ops
            if triplane_crop:
                inside = (xyz[:, :, [0, 2]].abs() <= (bw / 2 - triplane_crop)).all(dim=-1, keepdim=True)
                sigma = torch.where(inside, sigma, torch.full_like(sigma, -1e3))  # (the 'bottom' clause never changes the result)
            thr = binarize_clouds or cull_clouds
            if thr:
                alpha = 1 - torch.exp(-torch.nn.functional.softplus(sigma - 1))
                low = alpha < thr
                sigma = torch.where(low, torch.full_like(sigma, -1e3), torch.full_like(sigma, 1e3) if binarize_clouds else sigma)
            return sigma

        def decode(depths, S, noise):  # run_model on the sample points of `depths` [N,R,S,1] (renderer.py:179-183, 266-280)
            xyz = (o.unsqueeze(-2) + depths * d.unsqueeze(-2)).reshape(N, R * S, 3)
            sigma, rgb = ops.triplane_decode(planes_nhwc, xyz, mlp, opts)
            if dn > 0:
                noise = torch.randn_like(sigma) if noise is None else noise.to(dev, torch.float32).reshape(sigma.shape)
                sigma = sigma + noise * dn
            sigma = masks(sigma, xyz)
            return rgb.reshape(N, R, S, 32), sigma.reshape(N, R, S, 1), xyz.reshape(N, R, S, 3)

        if jitter is None:  # renderer.py:324
            jitter = torch.rand((N, R, Sc, 1), dtype=torch.float32, device=dev)
        depths_c = ops.sample_stratified(float(ro["ray_start"]), float(ro["ray_end"]), Sc, jitter.to(dev, torch.float32).reshape(N, R, Sc, 1))
        rgb_c, sig_c, xyz_c = decode(depths_c, Sc, nc)
        wb = bool(ro.get("white_back", False))
        if Sf > 0:
            _, _, w = ops.composite(rgb_c, sig_c, depths_c, white_back=wb)
            if u is None:  # renderer.py:371
                u = torch.rand((N * R, Sf), dtype=torch.float32, device=dev)
            depths_f = ops.importance(depths_c, w, u.to(dev, torch.float32))
            rgb_f, sig_f, xyz_f = decode(depths_f, Sf, nf)
            perm = ops.unify_perm(depths_c.reshape(N * R, Sc), depths_f.reshape(N * R, Sf)).to(torch.int64).reshape(N, R, Sc + Sf, 1)
            take = lambda a, b: torch.gather(torch.cat([a, b], dim=-2), -2, perm.expand(-1, -1, -1, a.shape[-1]))
            depths, colors, sigma = take(depths_c, depths_f), take(torch.cat([rgb_c, xyz_c], -1), torch.cat([rgb_f, xyz_f], -1)), take(sig_c, sig_f)
        else:
            depths, colors, sigma = depths_c, torch.cat([rgb_c, xyz_c], -1), sig_c
        out, depth, w = ops.composite(colors.contiguous(), sigma.contiguous(), depths.contiguous(), white_back=wb)
        return out[..., :-3].contiguous(), depth, w.sum(2), out[..., -3:].contiguous()

    def run_model(self, planes, decoder, sample_coordinates, sample_directions, options):
        """renderer.py:266-280 — sample_directions are ignored, as the reference's decoder ignores them
        (training/triplane.py:528-531)."""
        dn = float(options.get("density_noise", 0) or 0)
        if dn > 0:  # renderer.py:276-277: `out['sigma'] += torch.randn_like(out['sigma']) * options['density_noise']` after the decode
            options = {k: v for k, v in options.items() if k != "density_noise"}
        opts = self._opts(options, decoder)
        sigma, rgb = ops.triplane_decode(self._nhwc(planes), sample_coordinates.float(), decoder_params(decoder), opts)
        if dn > 0:
            sigma = sigma + torch.randn_like(sigma) * dn
        return {"rgb": rgb, "sigma": sigma, "xyz": sample_coordinates}

    def run_model_density(self, planes, decoder, sample_coordinates, options):
        """Density-only variant for get_eg3d_volume (_util/eg3d_metrics3d.py:140-150 keeps only sigma and rgb[:3])."""
        opts = self._opts(options, decoder)
        sigma, _ = ops.triplane_decode(self._nhwc(planes), sample_coordinates.float(), decoder_params(decoder), opts,
                                       density_only=True)
        return sigma
