"""TriPlaneGenerator on the MI355X path — same constructor arguments, sub-module / parameter names, method signatures and
return dictionaries as training/triplane.py:30-513 (so `_train/eg3dc/util/eg3dc_v0.py:47-52` can re-instantiate it from a
checkpoint's init kwargs and copy the parameters by name), with the backbone on the HIP synthesis operators
(stylegan2.py), the volumetric renderer on the fused HIP kernel (renderer.py) and the super-resolution blocks on the same
modulated-conv kernel.

Inference only.  Not mirrored: `sample` (broken in the reference, triplane.py:254-271).  The `paste_front`
post-process (triplane.py:555-691) lives in paste.py.
"""
import os

import numpy as np
import torch

from . import cameras, memo, stylegan2
from .renderer import ImportanceRenderer


def _capturing():
    """A hipGraph capture is underway on the current stream (False on a box without a device)."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class OSGDecoder(torch.nn.Module):
    """The tiny decoder MLP (triplane.py:516-547).  The renderer evaluates it INSIDE the fused kernels (it reads the parameters
    only); `forward` keeps the module callable like the reference's, on the same device arithmetic (p3d_decode_features_f32)."""

    def __init__(self, n_features, options):
        super().__init__()
        self.hidden_dim = 64
        self.force_sigmoid = False
        lr = options["decoder_lr_mul"]
        self.net = torch.nn.Sequential(stylegan2.FullyConnectedLayer(n_features, self.hidden_dim, lr_multiplier=lr),
                                       torch.nn.Softplus(),
                                       stylegan2.FullyConnectedLayer(self.hidden_dim, 1 + options["decoder_output_dim"], lr_multiplier=lr))

    def forward(self, sampled_features, ray_directions, force_sigmoid=None):
        """triplane.py:528-544: sampled_features [N,3,M,C] -> {'rgb': [N,M,32], 'sigma': [N,M,1]}; ray_directions are ignored, as
        the reference's decoder ignores them."""
        from . import ops
        from .renderer import decoder_params
        force_sigmoid = force_sigmoid or self.force_sigmoid
        sigma, rgb = ops.decode_features(sampled_features.float().contiguous(), decoder_params(self), force_sigmoid=force_sigmoid)
        return {"rgb": rgb, "sigma": sigma}

    def set_force_sigmoid(self, state):
        self.force_sigmoid = state
        return self.force_sigmoid


class SuperresolutionHybrid8XDC(torch.nn.Module):
    """128^2 x 32ch -> 512^2 RGB with two StyleGAN2 blocks (superresolution.py:264-293), fp32 on the HIP conv kernel."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, channels_hidden=256, num_fp16_res=4,
                 conv_clamp=None, channel_base=None, channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 512
        self.input_resolution = 128
        self.sr_antialias = sr_antialias
        clamp = 256 if sr_num_fp16_res > 0 else None  # the reference clamps only its fp16 blocks (superresolution.py:277-280)
        self.block0 = stylegan2.SynthesisBlock(channels, channels_hidden, w_dim=512, resolution=256, img_channels=3,
                                               is_last=False, conv_clamp=clamp, **block_kwargs)
        self.block1 = stylegan2.SynthesisBlock(channels_hidden, channels_hidden // 2, w_dim=512, resolution=512,
                                               img_channels=3, is_last=True, conv_clamp=clamp, **block_kwargs)

    def forward(self, rgb, x, ws, **block_kwargs):
        ws_in = ws
        ws = ws[:, -1:, :].expand(-1, 3, -1)  # (`.repeat(1, 3, 1)` of superresolution.py:283 without the copy: the layers only read it)
        if x.shape[-1] != self.input_resolution:
            size = (self.input_resolution, self.input_resolution)
            x = torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=False, antialias=self.sr_antialias)
            rgb = torch.nn.functional.interpolate(rgb, size=size, mode="bilinear", align_corners=False, antialias=self.sr_antialias)
        # the six affine layers and the four demodulations of the two blocks in one StylePlan (one GEMM + four small launches
        # instead of six addmm + two scalings + four k_demod: networks_stylegan2.py:342,377,70-73), as the backbone does
        plan = self.__dict__.get("_style_plan")
        if plan is None:
            plan = stylegan2.StylePlan(stylegan2.plan_entries([("block0", self.block0), ("block1", self.block1)], [0, 0]))
            self.__dict__["_style_plan"] = plan
        pre = plan(ws, memo_of=ws_in)
        # block0.conv1 hands block1.conv0 its operand (an activation image next to the fp32 tensor ToRGB reads): no conversion pass
        ns = stylegan2._next_conv0_styles(self.block1, pre["block1"], self.block0.resolution)
        # (neither block's fp32 activation is read by anybody once block1 takes the image and ToRGB rides on conv1: need_x=False)
        out = self.block0(x.contiguous(), rgb.contiguous(), ws, pre=pre["block0"], next_styles=ns, need_x=ns is None, **block_kwargs)
        x, rgb, x_image = out if ns is not None else (out[0], out[1], None)
        x, rgb = self.block1(x, rgb, ws, pre=pre["block1"], x_image=x_image, need_x=False, **block_kwargs)
        return rgb

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_style_plan", None)  # derived tensors stay out of pickles / deep copies
        return state


_SR_MODULES = {"training.superresolution.SuperresolutionHybrid8XDC": SuperresolutionHybrid8XDC}


class TriPlaneGenerator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, sr_num_fp16_res=0, mapping_kwargs={},
                 rendering_kwargs={}, sr_kwargs={}, cond_mode=None, triplane_width=32, sr_channels_hidden=256,
                 backbone_resolution=256, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution, self.img_channels = img_resolution, img_channels
        if rendering_kwargs.get("triplane_depth", 1) != 1:
            raise NotImplementedError("triplane_depth != 1")
        self.renderer = ImportanceRenderer(use_triplane=rendering_kwargs.get("use_triplane", False))
        self.triplane_width, self.backbone_resolution = triplane_width, backbone_resolution
        self.backbone = stylegan2.Generator(z_dim, c_dim, w_dim, img_resolution=backbone_resolution,
                                            img_channels=triplane_width * 3, cond_mode=cond_mode,
                                            mapping_kwargs=mapping_kwargs, **synthesis_kwargs)
        sr_name = rendering_kwargs["superresolution_module"]
        if sr_name not in _SR_MODULES:
            raise NotImplementedError(f"{sr_name}: only the 512^2 module of the released model is mirrored")
        self.superresolution = _SR_MODULES[sr_name](channels=32, channels_hidden=sr_channels_hidden,
                                                    img_resolution=img_resolution, sr_num_fp16_res=sr_num_fp16_res,
                                                    sr_antialias=rendering_kwargs["sr_antialias"], **sr_kwargs)
        self.decoder = OSGDecoder(triplane_width, {"decoder_lr_mul": rendering_kwargs.get("decoder_lr_mul", 1),
                                                   "decoder_output_dim": 32})
        self.neural_rendering_resolution = 64
        self.rendering_kwargs = rendering_kwargs
        self.cond_mode = cond_mode
        self._last_planes = None
        self._inject_draws = None  # tests: (jitter, u) for the renderer instead of device RNG

    def __getstate__(self):
        """Pickling / deep copies (the reference snapshots G): derived tensors and per-call state stay out."""
        state = dict(self.__dict__)
        for k in ("_sign_cache", "_last_planes", "_inject_draws", "_sr_plan", "_ws_memo", "_conv_domain_flag", "_mapping_dicts", "_occ_consts",
                  "_view_graphs", "_state_list"):
            if k in state:
                state[k] = None
        return state

    def _sign(self, device, scale=1.0):
        """scale * (-1, 1, -1) as [1,3,1,1] on `device`, created once (a host->device copy waits for everything queued on the stream
        and is not allowed inside a hipGraph capture); a plain attribute, not a buffer: the module's state_dict must stay identical
        to the reference's."""
        c = self.__dict__.get("_sign_cache")
        if c is None or c[0] != device:
            t = torch.tensor([-1.0, 1.0, -1.0], device=device)[None, :, None, None]
            c = self.__dict__["_sign_cache"] = (device, {1.0: t, 0.5: t * 0.5})
        return c[1][scale]

    # ---- latents ------------------------------------------------------------------------------------------------
    def mapping(self, z, c, cond, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        rk = self.rendering_kwargs
        if rk["c_gen_conditioning_zero"]:
            c = torch.zeros_like(c)
        if rk.get("c_gen_conditioning_force_ffhq", False):
            raise NotImplementedError("c_gen_conditioning_force_ffhq (fine-tuning hack, triplane.py:97-121)")
        return self.backbone.mapping(z, c * rk.get("c_scale", 0), cond, truncation_psi=truncation_psi,
                                     truncation_cutoff=truncation_cutoff)

    def mapping_zplus(self, zs, c, cond, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        """One z per w slot (triplane.py:123-143): map every z, keep slot i of the i-th z's broadcast ws."""
        bs, n, dim = zs.shape
        if zs.stride(1) == 0:  # f() expands ONE z to all w slots (triplane.py:356): every slot maps the same z -> map it once
            return self.mapping(zs[:, 0], c, cond, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        c_new = c[:, None, :].repeat(1, n, 1).reshape(bs * n, -1)
        cond_new = cond
        if "resnet_feats" in cond:
            cond_new = {**cond, "resnet_feats": cond["resnet_feats"][:, None, :].repeat(1, n, 1).reshape(bs * n, -1)}
        ans = self.mapping(zs.reshape(bs * n, dim), c_new, cond_new, truncation_psi=truncation_psi,
                           truncation_cutoff=truncation_cutoff)
        ans = ans.view(bs, n, n, -1)
        idx = torch.arange(n, device=ans.device)
        return ans[:, idx, idx]

    # ---- planes -------------------------------------------------------------------------------------------------
    def _planes(self, ws, cond, latent_injection=None, stop_level=None, **synthesis_kwargs):
        planes = self.backbone.synthesis(ws, cond, latent_injection=latent_injection, stop_level=stop_level, **synthesis_kwargs)
        return planes.view(len(planes), 3, self.triplane_width, planes.shape[-2], planes.shape[-1])

    def synthesis(self, ws, c, cond, neural_rendering_resolution=None, update_emas=False, cache_backbone=False,
                  use_cached_backbone=False, latent_injection=None, stop_level=None, force_rays=None, triplane_crop=None,
                  cull_clouds=None, binarize_clouds=None, normalize_images=True, return_more=False, _rays_flat=None, **synthesis_kwargs):
        if neural_rendering_resolution is None:
            neural_rendering_resolution = self.neural_rendering_resolution
        else:
            self.neural_rendering_resolution = neural_rendering_resolution
        res = neural_rendering_resolution
        if force_rays is None:
            ray_origins, ray_directions = cameras.perspective_rays(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), res)
        elif isinstance(force_rays, dict):
            ro, rd = force_rays["ray_origins"], force_rays["ray_directions"]
            assert ro.shape == rd.shape and ro.shape[1:] == (3, res, res) and (len(ro) == len(ws) or len(ws) == 1)
            if _rays_flat is not None and _rays_flat[0].shape == (len(ro), res * res, 3):  # f(): the view cache's own [N,R,3] copies of these rays
                ray_origins, ray_directions = _rays_flat
            else:
                ray_origins = ro.permute(0, 2, 3, 1).reshape(len(ro), res * res, 3)
                ray_directions = rd.permute(0, 2, 3, 1).reshape(len(ro), res * res, 3)
        else:
            assert False, "force_rays not understood"
        opts = dict(cache_backbone=cache_backbone, use_cached_backbone=use_cached_backbone, latent_injection=latent_injection,
                    stop_level=stop_level, triplane_crop=triplane_crop, cull_clouds=cull_clouds, binarize_clouds=binarize_clouds,
                    normalize_images=normalize_images)
        ray_origins, ray_directions = ray_origins.contiguous(), ray_directions.contiguous()
        self._domain_begin(ws.device)
        ans = self._replay_view(ws, cond, ray_origins, ray_directions, res, opts, synthesis_kwargs)
        if ans is None:
            ans = self._synthesis_impl(ws, cond, ray_origins, ray_directions, res, **opts, **synthesis_kwargs)
            self._domain_end()
        return ans

    def _synthesis_impl(self, ws, cond, ray_origins, ray_directions, res, cache_backbone=False, use_cached_backbone=False,
                        latent_injection=None, stop_level=None, triplane_crop=None, cull_clouds=None, binarize_clouds=None,
                        normalize_images=True, **synthesis_kwargs):
        """Everything of `synthesis` that launches device work, from flat rays [N,R,3]: backbone -> renderer -> super-resolution.
        No host read of a device value in here: the whole call can be captured into a hipGraph (`_replay_view`)."""
        N = ray_origins.shape[0]
        if use_cached_backbone and self._last_planes is not None:
            planes = self._last_planes
        else:
            planes = self._planes(ws, cond, latent_injection, stop_level, **synthesis_kwargs)
        if cache_backbone:
            self._last_planes = planes
        many_views = len(planes) == 1 and N > 1
        if many_views:
            # extension: V views of ONE subject in one call (ws / cond of batch 1, V cameras).  The planes are synthesised once
            # and shared by the V ray batches of a single renderer launch (P3D_FLAG_SHARED_PLANES); the reference would need
            # ws and cond repeated V times and would run the backbone on V copies.  The V views stand for V calls of the
            # reference (generate.py's view loop), so every view keeps its own depth-clamp range (P3D_FLAG_PER_VIEW_CLAMP).
            planes = planes.expand(N, -1, -1, -1, -1)
            ws = ws.expand(N, -1, -1)
        draws = self._inject_draws or (None, None)
        if isinstance(draws, list):  # tests: one (jitter, u) pair per renderer pass, consumed in call order
            draws = draws.pop(0)
        feat, depth, wsum, xyz = self.renderer(planes, self.decoder, ray_origins, ray_directions,
                                               self.rendering_kwargs, triplane_crop=triplane_crop, cull_clouds=cull_clouds,
                                               binarize_clouds=binarize_clouds, jitter=draws[0], u=draws[1],
                                               per_view_clamp=many_views)
        H = W = res
        feature_image = feat.permute(0, 2, 1).reshape(N, feat.shape[-1], H, W).contiguous()
        xyz_image = xyz.permute(0, 2, 1).reshape(N, 3, H, W).contiguous()
        depth_image = depth.permute(0, 2, 1).reshape(N, 1, H, W)
        weights_image = wsum.permute(0, 2, 1).reshape(N, 1, H, W)
        # 0.5 * (xyz + 1) * (-1, 1, -1)  (triplane.py:229) in one launch: h + xyz * h with h = (-0.5, 0.5, -0.5) — xyz * h is exact, so
        # the single rounding is the reference's (x + 1) rounded, then halved and signed exactly: the same bits
        half_sign = self._sign(xyz_image.device, 0.5)
        xyz_image = torch.addcmul(half_sign, xyz_image, half_sign)
        rgb_image = feature_image[:, :3]
        sr_kw = {k: v for k, v in synthesis_kwargs.items() if k != "noise_mode"}
        sr_image = self.superresolution(rgb_image, feature_image, ws,
                                        noise_mode=self.rendering_kwargs["superresolution_noise_mode"], **sr_kw)
        ans = {"image": sr_image, "image_raw": rgb_image, "image_depth": depth_image, "triplane": planes,
               "image_weights": weights_image, "image_xyz": xyz_image}
        if self.rendering_kwargs.get("tanh_rgb_output", False):
            ans["image"], ans["image_raw"] = torch.tanh(ans["image"]), torch.tanh(ans["image_raw"])
        if not normalize_images:  # 0.5 * image + 0.5, one launch each (0.5 * image is exact: the same single rounding)
            half = self._sign(sr_image.device, 0.5)[0, 1, 0, 0]
            ans["image"], ans["image_raw"] = torch.add(half, ans["image"], alpha=0.5), torch.add(half, ans["image_raw"], alpha=0.5)
        return ans

    # ---- launch replay of a view (round 6) ------------------------------------------------------------------------
    # A view is ~100 launches that Python needs ~2.2 ms to issue while the GPU runs them in ~2.1 ms: the interpreter is the
    # co-bound of `generate.py`'s view loop.  Views 2 .. 16 of a subject repeat view 1's launches with other rays (and, for the
    # next subjects, other latents and conditioning images): the second call of a kind is captured into a hipGraph and every
    # later one replays it —
    #   * inputs that change per call (ws, rays) are copied into the capture's own buffers;
    #   * everything the launches read besides — parameters, parameter-derived operands, the prepared conditioning terms —
    #     keeps its ADDRESS while its VALUES may change: derived operands are rebuilt only when a parameter version moves (then
    #     the captures are dropped), prepared conditioning terms are updated in place (stylegan2._cond_prepared); the first view
    #     of a subject therefore runs eagerly (it refreshes those terms) and its further views replay;
    #   * outputs are CLONES of the capture's buffers: a caller may keep view i while view i + 1 replays.
    # Same launches, same arithmetic, same random stream as the eager call (torch registers the device generator with the capture):
    # bit-identical images (tests/test_hip_synthesis.py).  Off: memo.set_enabled(False) / P3D_NO_MEMO=1 (no memo layer, no replay),
    # P3D_VIEW_REPLAY=0, or G.set_view_replay(False).  Not replayed: calls under autograd, with injected draws, latent injections,
    # cache_backbone / use_cached_backbone, return_more.
    _REPLAY_MAX = 4  # captures kept per generator (each owns the intermediates of a whole view)

    def set_view_replay(self, state):
        """True / False: allow / forbid the launch replay of this generator's views; None: the process default (P3D_VIEW_REPLAY)."""
        self.__dict__["_view_replay"] = state
        if state is False:
            self.__dict__["_view_graphs"] = None
        return state

    def _state_tensors(self):
        st = self.__dict__.get("_state_list")
        if st is None:
            st = self.__dict__["_state_list"] = [t for t in list(self.parameters()) + list(self.buffers())]
        return st

    def _apply(self, fn):  # .to() / .cuda() / .float(): every capture holds addresses of the old storage
        self.__dict__["_view_graphs"] = None
        self.__dict__["_state_list"] = None
        return super()._apply(fn)

    def _replay_view(self, ws, cond, ray_origins, ray_directions, res, opts, synthesis_kwargs):
        """The view from a captured launch sequence, or None (the caller then runs it eagerly)."""
        allowed = self.__dict__.get("_view_replay")
        if allowed is None:
            allowed = os.environ.get("P3D_VIEW_REPLAY", "1") != "0"
        if not (allowed and memo.enabled() and ws.is_cuda and not torch.is_grad_enabled()) or self._inject_draws is not None \
                or opts["latent_injection"] is not None or opts["cache_backbone"] or opts["use_cached_backbone"] \
                or _capturing():
            return None
        flags = self.__dict__.get("_conv_domain_flag")
        if flags is not None and flags.dirty:  # the first call on new weights runs eagerly and reads the domain flag
            return None
        try:
            ctens = sorted((k, v) for k, v in cond.items() if torch.is_tensor(v))
            if len(ctens) != len(cond):
                return None
            from operator import attrgetter
            pver = sum(map(attrgetter("_version"), self._state_tensors()))
            rk = self.rendering_kwargs
            key = (tuple(ws.shape), ws.dtype, ws.device, tuple(ray_origins.shape), ray_origins.dtype, res, opts["stop_level"], opts["triplane_crop"],
                   opts["cull_clouds"], opts["binarize_clouds"], opts["normalize_images"], tuple(sorted(synthesis_kwargs.items())),
                   tuple((k, tuple(v.shape), v.dtype) for k, v in ctens), pver, self.renderer.exact, bool(self.decoder.force_sigmoid),
                   tuple(sorted((k, v) for k, v in rk.items() if isinstance(v, (int, float, str, bool, type(None))))), self.cond_mode)
            hash(key)
        except TypeError:  # an option that cannot be compared by value
            return None
        graphs = self.__dict__.get("_view_graphs")
        if graphs is None or graphs.get("pver") != pver:  # new parameter values: derived operands move
            graphs = self.__dict__["_view_graphs"] = {"pver": pver, "entries": {}}
        ent = graphs["entries"].get(key)
        sig = tuple((k, id(v), v._version) for k, v in ctens)
        syn = self.backbone.synthesis
        if ent is None:  # first call of this kind: eager (it also creates every lazily made constant)
            if len(graphs["entries"]) >= self._REPLAY_MAX:
                graphs["entries"].pop(next(iter(graphs["entries"])))
            graphs["entries"][key] = {"graph": None, "sig": sig, "cond": [v for _, v in ctens], "failed": False}
            return None
        if ent["failed"]:
            return None
        if ent["sig"] != sig or ent.get("gen") not in (None, syn.__dict__.get("_cond_gen", 0)):
            # another subject (or conditioning tensors written to): this call runs eagerly and refreshes the prepared terms in place;
            # a term that had to be REPLACED invalidates the capture
            if ent.get("gen") not in (None, syn.__dict__.get("_cond_gen", 0)):
                ent["graph"] = None
            ent["sig"], ent["cond"], ent["gen"] = sig, [v for _, v in ctens], None
            return None
        if ent["graph"] is None:
            try:
                self._capture_view(ent, ws, cond, ray_origins, ray_directions, res, opts, synthesis_kwargs)
            except Exception as e:  # noqa: BLE001 — anything the capture cannot hold: stay eager for this kind of call
                ent["failed"], ent["graph"] = True, None
                import warnings
                warnings.warn(f"launch replay of TriPlaneGenerator.synthesis disabled for this call signature: {type(e).__name__}: {e}", RuntimeWarning)
                torch.cuda.synchronize()
                return None
            ent["gen"] = syn.__dict__.get("_cond_gen", 0)
        ent["ws"].copy_(ws)
        ent["ro"].copy_(ray_origins)
        ent["rd"].copy_(ray_directions)
        ent["graph"].replay()
        ent["replays"] = ent.get("replays", 0) + 1
        return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in ent["out"].items()}

    def _capture_view(self, ent, ws, cond, ray_origins, ray_directions, res, opts, synthesis_kwargs):
        ent["ws"], ent["ro"], ent["rd"] = ws.clone(), ray_origins.clone(), ray_directions.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self._synthesis_impl(ent["ws"], cond, ent["ro"], ent["rd"], res, **opts, **synthesis_kwargs)
        ent["graph"], ent["out"] = g, out
        # what the launches read beyond the inputs and the modules' own state: the prepared conditioning terms (kept alive here)
        ent["keep"] = list((self.backbone.synthesis.__dict__.get("_cond_cache") or {}).values())

    # ---- the domain of the two-term convolutions, checked once per set of weights -----------------------------------
    def _domain_begin(self, device):
        if self.__dict__.get("_conv_domain_flag") is None and device.type == "cuda":
            self.watch_conv_domain(device)

    def _domain_end(self):
        """The default convolution path (two-term f16 operands) is exact to fp32 class only for |s * x| <= 4094 and SATURATES beyond
        (stylegan2.DEFAULT_CONV_MMA).  Nothing bounds a real checkpoint's activations (conv_clamp=None, train_eclustrousC.py:554), so
        the first synthesis after the operands were derived from the weights — construction, load_state_dict,
        copy_params_and_buffers — reads the generator's flag word back (one 4-byte copy, once per set of weights) and says so."""
        flags = self.__dict__.get("_conv_domain_flag")
        if flags is None or not (flags.dirty or os.environ.get("P3D_CHECK_CONV_DOMAIN")) or _capturing():
            return
        flags.dirty = False
        if self.conv_domain_violated():
            self.__dict__["conv_domain_was_violated"] = True
            import warnings
            warnings.warn("a two-term f16 convolution met |s*x| > 4094 and saturated it: the image is WRONG — call "
                          "G.set_conv_mma('f32') for this model (INTEGRATION.md, 'convolution operand domain')", RuntimeWarning)

    def sample_mixed(self, coordinates, directions, ws, cond, truncation_psi=1, truncation_cutoff=None, update_emas=False,
                     **synthesis_kwargs):
        """triplane.py:273-298 (the density-grid query of _util/eg3d_metrics3d.py:140).  The reference re-runs the backbone
        on every call; pass use_cached_backbone=True to reuse the planes of the previous call with the same ws."""
        reuse = synthesis_kwargs.pop("use_cached_backbone", False)
        if reuse and self._last_planes is not None:
            planes = self._last_planes
        else:
            self._domain_begin(ws.device)
            planes = self._planes(ws, cond, **synthesis_kwargs)
            self._domain_end()
            self._last_planes = planes if reuse else self._last_planes
        return self.renderer.run_model(planes, self.decoder, coordinates, directions, self.rendering_kwargs)

    def forward(self, z, c, cond, truncation_psi=1, truncation_cutoff=None, neural_rendering_resolution=None,
                update_emas=False, cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        ws = self.mapping(z, c, cond, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return self.synthesis(ws, c, cond, neural_rendering_resolution=neural_rendering_resolution,
                              cache_backbone=cache_backbone, use_cached_backbone=use_cached_backbone, **synthesis_kwargs)

    # ---- the dict-in / dict-out API of PAniC-3D (triplane.py:313-508) -----------------------------------------
    def f(self, x, truncation_psi=1, truncation_cutoff=None, latent_injection=None, force_rays=None, stop_level=None,
          normalize_images=False, return_more=False):
        emb = self.backbone.mapping.embed.weight
        device, dtype = emb.device, emb.dtype
        for key in ("ws", "camera_params", "elevations", "zs", "z"):
            if key in x:
                device = x[key].device
                break
        if "latent_injection" in x:
            latent_injection = {**(latent_injection or {}), **x["latent_injection"]}
        ws_key = None
        if "zs" not in x and "ws" not in x:
            if "z" not in x:
                # generate.py renders its 16 views of a subject with the same seeds, conditioning tensors and (PAniC-3D's default,
                # c_gen_conditioning_zero) a mapping that does not see the camera: the latents and the ws of the previous call are
                # then the ws of this one.  Memoised on everything the mapping reads — seeds, truncation, the mapping network's
                # parameter versions, the conditioning tensors it uses (object + version, strong references) — one entry deep.
                rc = self.backbone.mapping.resnet_cond
                pose_free = bool(self.rendering_kwargs.get("c_gen_conditioning_zero", False)) or self.c_dim == 0
                if pose_free and latent_injection is None and memo.enabled():
                    ct = [x["cond"]["resnet_feats"]] if rc > 0 else []
                    # (the mapping network's tensors through its modules' own dicts: parameters() / buffers() walk the module tree
                    # through generators, ~25 us at the head of every call)
                    md = self.__dict__.get("_mapping_dicts")
                    if md is None or md[0] is not self.backbone.mapping:
                        mp = self.backbone.mapping
                        md = self.__dict__["_mapping_dicts"] = (mp, [d for m in mp.modules() for d in (m._parameters, m._buffers)])
                    ws_key = (tuple(int(s) for s in x["seeds"]), float(truncation_psi), truncation_cutoff, device, dtype,
                              tuple([(id(t), t._version) for t in ct]),
                              tuple([(q.data_ptr(), q._version) for d in md[1] for q in d.values() if q is not None]))
                    hit = self.__dict__.get("_ws_memo")
                    if hit is not None and hit[0] == ws_key and all(a is b for a, b in zip(hit[1], ct)) \
                            and hit[2]._version == hit[4][0] and hit[3]._version == hit[4][1]:  # (nobody wrote into the memoised tensors)
                        x["z"], x["ws"] = hit[2], hit[3]
                        ws_key = None
                    else:
                        ws_key = (ws_key, ct)
                if "z" not in x:
                    x["z"] = torch.tensor(np.stack([np.random.RandomState(s).randn(self.z_dim) for s in x["seeds"]]),
                                          device=device, dtype=dtype)
            x["zs"] = x["z"][:, None, :].expand(-1, self.backbone.num_ws, -1)
        force_rays = (x["force_rays"] if "force_rays" in x else None) or force_rays
        rays_flat = None
        res = x["neural_rendering_resolution"] if "neural_rendering_resolution" in x else self.neural_rendering_resolution
        if "camera_params" not in x:
            # one device -> host copy for the view parameters the caller gave (the defaults — distance 1, fov 30, triplane.py:360-363 —
            # are known on the host), then labels and rays memoised per view (cameras.cached_view).  Every launch saved here is a
            # launch the GPU does not wait for: a call starts with an empty queue.
            given = [k for k in ("elevations", "azimuths", "distances", "fovs") if k in x]
            host = dict(zip(given, torch.stack([torch.as_tensor(x[k]).reshape(-1) for k in given]).cpu().double().tolist()))
            nv = len(host["elevations"])
            if "distances" not in x:
                x["distances"] = torch.ones_like(x["elevations"])
                host["distances"] = [1.0] * nv
            if "fovs" not in x:
                x["fovs"] = torch.full_like(x["elevations"], 30)
                host["fovs"] = [30.0] * nv
            vals = [host[k] for k in ("elevations", "azimuths", "distances", "fovs")]
            view_of = cameras.cached_view if memo.enabled() else cameras.make_view
            views = [view_of(e, a, d, fv, res, self.rendering_kwargs["box_warp"], device, dtype) for e, a, d, fv in zip(*vals)]
            x["camera_params"] = torch.stack([v[0] for v in views])
            if force_rays is None:
                x["force_rays"] = force_rays = {"ray_origins": torch.stack([v[1] for v in views]),
                                                "ray_directions": torch.stack([v[2] for v in views])}
                # the same rays in the renderer's [N,R,3] layout, straight from the view cache (never handed to the caller, never
                # written: for one view a plain view of the cached tensor) — synthesis() would permute + copy the dict's tensors
                rays_flat = (views[0][3][None], views[0][4][None]) if len(views) == 1 else \
                            (torch.stack([v[3] for v in views]), torch.stack([v[4] for v in views]))
        if force_rays is None:
            cp = x["camera_params"]
            intr = cp[:, 16:25].view(-1, 3, 3)
            ro, rd = cameras.perspective_rays(cp[:, :16].view(-1, 4, 4), intr, res)
            ro = ro.reshape(len(cp), res, res, 3).permute(0, 3, 1, 2).contiguous()
            rd = rd.reshape(len(cp), res, res, 3).permute(0, 3, 1, 2).contiguous()
            for i in range(len(cp)):  # negative fov = orthographic view (triplane.py:402-414)
                if intr[i, 0, 0] < 0:
                    r = cameras.ortho_rays(x["elevations"][i], x["azimuths"][i], x["distances"][i],
                                           self.rendering_kwargs["box_warp"], res, device=device)
                    ro[i], rd[i] = r["ray_origins"], r["ray_directions"]
            x["force_rays"] = force_rays = {"ray_origins": ro, "ray_directions": rd}
        x["conditioning_params"] = x["camera_params"]
        if "ws" not in x:
            cpm = x["conditioning_params"]
            if len(x["zs"]) == 1 and len(cpm) > 1:  # V views of one subject in one call (see synthesis)
                if not self.rendering_kwargs["c_gen_conditioning_zero"]:
                    raise RuntimeError("many views per call need ws that do not depend on the camera: pass x['ws'] or use a "
                                       "generator without pose conditioning (c_gen_conditioning_zero, PAniC-3D's default)")
                cpm = cpm[:1]
            x["ws"] = self.mapping_zplus(x["zs"], cpm, x["cond"], truncation_psi=truncation_psi,
                                         truncation_cutoff=truncation_cutoff)
            if ws_key is not None:
                self.__dict__["_ws_memo"] = (ws_key[0], ws_key[1], x["z"], x["ws"], (x["z"]._version, x["ws"]._version))
        _ws = x["ws"]
        if latent_injection is not None:
            for k in ("dw", "dws"):
                if k in latent_injection:
                    _ws = _ws + latent_injection[k]
        normalize_images = x["normalize_images"] if "normalize_images" in x else normalize_images
        synth = self.synthesis(_ws, x["camera_params"], x["cond"], latent_injection=latent_injection,
                               triplane_crop=x.get("triplane_crop"), cull_clouds=x.get("cull_clouds"),
                               binarize_clouds=x.get("binarize_clouds"), force_rays=force_rays, stop_level=stop_level,
                               normalize_images=normalize_images, neural_rendering_resolution=res,
                               # extensions of the dict API (absent keys = the reference's behaviour): reuse the planes of
                               # the previous call for further views of the same subject, deterministic backbone noise
                               cache_backbone=bool(x.get("cache_backbone", False)),
                               use_cached_backbone=bool(x.get("use_cached_backbone", False)), _rays_flat=rays_flat,
                               **({"noise_mode": x["noise_mode"]} if "noise_mode" in x else {}))
        ret = {k: synth[k] for k in ("image", "image_raw", "image_depth", "image_weights", "triplane", "image_xyz")}
        ret["normalize_images"] = normalize_images
        x.update(ret)
        if x.get("paste_params") is not None:  # front-view paste post-process (triplane.py:497-502)
            from .paste import paste_front
            ret["image_prepaste"] = ret["image"]
            ret["paste"] = paste_front(self, x, ret, **x["paste_params"])
            ret["image"] = ret["paste"]["image"]
        return ret

    def watch_conv_domain(self, device=None):
        """Give every modulated 3x3 layer of THIS generator one flag word (owned by the generator, on its device) that the two-term
        f16 convolutions raise when a modulated activation leaves their domain |s*x| <= 4094.  Per generator, not per process:
        two generators on two streams do not share it (the C ABI keeps no such state, include/panic3d_hip.h ABI 5)."""
        device = device if device is not None else next(self.parameters()).device
        flags = stylegan2.DomainFlags()  # one word per device, created where a layer runs (the generator may be moved later)
        for m in list(self.backbone.modules()) + list(self.superresolution.modules()):
            if isinstance(m, (stylegan2.SynthesisLayer, stylegan2.ToRGBLayer)):
                m.conv_domain_flag = flags
        self.__dict__["_conv_domain_flag"] = flags
        return flags.get(device)

    def conv_domain_violated(self, reset=True):
        """True if a two-term convolution of this generator saturated an operand since the last reset (synchronises).  The flag is
        created on first use: convolutions that ran before `watch_conv_domain()` were not watched."""
        from . import ops
        flags = self.__dict__.get("_conv_domain_flag")
        if flags is None:
            self.watch_conv_domain()
            return False
        return any([ops.conv_domain_violated(w, reset) for w in list(flags.words.values())])  # (a list: every device's word is read and reset)

    def set_noise_pool(self, state):
        """noise_mode='random' of THIS generator's backbone and super-resolution: True = one pooled draw per pass, False = the reference's
        call-for-call torch.randn sequence (seed-compatible with it), None = the process default (stylegan2.NOISE_POOL)."""
        for net in (self.backbone.synthesis, self.superresolution):
            for m in net.modules():
                if isinstance(m, stylegan2.SynthesisNetwork):
                    m.__dict__["noise_pool"] = None if state is None else bool(state)
        return state

    def clear_memo(self):
        """Drop everything this generator remembers between calls — the memoised latents, both StylePlans (styles and the
        parameter-derived tables), the prepared conditioning, the cached planes and their channels-last copy, every layer's derived
        weights — and the process-wide view cache.  For callers that wrote into parameters or conditioning tensors behind the
        version counter (`.data`, DLPack aliases; memo.py), and for servers that want the last subject's tensors released."""
        self.__dict__["_ws_memo"] = None
        self.__dict__["_view_graphs"] = None
        self.__dict__["_state_list"] = None
        self.__dict__["_mapping_dicts"] = None
        self.__dict__["_occ_consts"] = None
        self._last_planes = None
        self.renderer._planes_cache = (None, None, None)
        for m in self.modules():
            for k in stylegan2._CACHE_ATTRS:
                if k != "conv_domain_flag":
                    m.__dict__.pop(k, None)
            m.__dict__.pop("_style_plan", None)
        cameras.cached_view_clear()

    def set_force_sigmoid(self, state):
        return self.decoder.set_force_sigmoid(state)

    def set_render_exact(self, state=True):
        """True: every render of this generator runs the exact fp32 contract (bit-identical to the arithmetic contract); False: the
        tolerance mode of the final pass (P3D_FLAG_FAST_COLOR); None: the package default (tolerance unless P3D_EXACT=1)."""
        self.renderer.exact = state
        return state

    def set_conv_mma(self, mode):
        """How the 3x3 convolutions of the backbone and of the super-resolution feed the matrix cores: "f32" (fp32 operands,
        v_mfma_f32_32x32x2_f32), "x2" (two-term f16 operands: fp32-class results, ~2x faster; domain |s*x| <= 4094, watched by
        watch_conv_domain() / conv_domain_violated()), "f16" (one f16 term: the precision of the reference's fp16 blocks) or None (the
        package default, stylegan2.DEFAULT_CONV_MMA)."""
        val = {"f32": False, "x2": "x2", "f16": True, None: None}[mode]
        for m in list(self.backbone.modules()) + list(self.superresolution.modules()):
            if isinstance(m, stylegan2.SynthesisLayer):
                if val is None:
                    if hasattr(m, "mma_f16"):
                        del m.mma_f16
                else:
                    m.mma_f16 = val
        return mode

    def set_sr_mma_f16(self, state=True):
        """Opt-in: run the super-resolution convolutions on f16 MFMA operands (fp32 accumulate, fp32 activations in HBM).
        The reference runs these blocks in fp16 on the GPU (sr_num_fp16_res = 4, superresolution.py:277-280); the default here
        is exact fp32, which is what its CPU path (the source of the golden fixtures) computes."""
        for m in self.superresolution.modules():
            if isinstance(m, (stylegan2.SynthesisLayer, stylegan2.ToRGBLayer)):
                m.mma_f16 = bool(state)
        return bool(state)
