"""StyleGAN2 mapping + synthesis host modules of the triplane backbone, on the HIP operators.

Mirrors the module tree and parameter names of training/networks_stylegan2.py (FullyConnectedLayer :102-133,
MappingNetwork :199-296, SynthesisLayer :299-360, ToRGBLayer :363-383, SynthesisBlock :388-487, SynthesisNetwork
:492-725, Generator :729-757) so that `misc.copy_params_and_buffers(..., require_all=True)` / `load_state_dict` of a
PAniC-3D checkpoint works unchanged.  Inference only, fp32 only (the backbone runs with num_fp16_res=0:
trainers/train_eclustrousC.py:253,553).  All convolution-shaped work runs in libpanic3d_hip.so: each SynthesisLayer /
ToRGBLayer is ONE fused call (modulation, conv on the matrix cores, demodulation, noise, bias, lrelu, gain, clamp);
the tiny fully-connected layers (w -> styles, mapping) stay on torch matmul.
"""
import os

import numpy as np
import torch

from . import memo, ops

SQRT2 = float(np.sqrt(2))

# derived tensors the modules keep as plain attributes (never part of the state_dict, dropped when pickled / deep-copied)
_CACHE_ATTRS = ("_scaled_wb", "_scaled_key", "_wh", "_wh_key", "_wt", "_wt_key", "_noise_cache", "_style_plan", "_cond_cache", "conv_domain_flag", "_noise_pool")


class _CacheFree(torch.nn.Module):
    """nn.Module whose derived-tensor caches are not pickled (the reference pickles G for snapshots and deep-copies it)."""

    def __getstate__(self):
        state = dict(self.__dict__)
        for k in _CACHE_ATTRS:
            state.pop(k, None)
        return state


def normalize_2nd_moment(x, dim=1, eps=1e-8):  # networks_stylegan2.py:33-35
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


class FullyConnectedLayer(_CacheFree):
    def __init__(self, in_features, out_features, bias=True, activation="linear", lr_multiplier=1, bias_init=0):
        super().__init__()
        self.in_features, self.out_features, self.activation = in_features, out_features, activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def _scaled(self, dtype):
        """`weight * weight_gain`, `bias * bias_gain` (networks_stylegan2.py:121-127), computed once per parameter version:
        the same two multiplications the reference does on every call (inference: the parameters do not change), so the
        values are bit-identical.  Plain attributes, not buffers: the state_dict stays the reference's."""
        key = (dtype, self.weight.data_ptr(), self.weight._version, None if self.bias is None else (self.bias.data_ptr(), self.bias._version))
        if getattr(self, "_scaled_key", None) != key or not memo.enabled():
            w = self.weight.detach().to(dtype) * self.weight_gain
            b = self.bias
            if b is not None:
                b = b.detach().to(dtype)
                if self.bias_gain != 1:
                    b = b * self.bias_gain
            self._scaled_wb, self._scaled_key = (w, b), key
        return self._scaled_wb

    def forward(self, x):
        w, b = self._scaled(x.dtype)
        if self.activation == "linear" and b is not None:
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return ops.bias_act(x.matmul(w.t()).contiguous(), b, act=self.activation)


class MappingNetwork(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, num_ws, cond_mode, num_layers=8, embed_features=None, layer_features=None,
                 activation="lrelu", lr_multiplier=0.01, w_avg_beta=0.998):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws, self.num_layers = z_dim, c_dim, w_dim, num_ws, num_layers
        self.w_avg_beta, self.cond_mode = w_avg_beta, cond_mode
        self.resnet_cond = 0
        for m in cond_mode.split("."):  # 'resnetcond_<k>': first k resnet features join the camera label
            if m.startswith("resnetcond_"):
                self.resnet_cond = int(m.split("_")[-1])
                assert c_dim > 0
                break
        embed_features = (w_dim if embed_features is None else embed_features) if c_dim > 0 else 0
        layer_features = w_dim if layer_features is None else layer_features
        feats = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim + self.resnet_cond, embed_features)
        for i in range(num_layers):
            setattr(self, f"fc{i}", FullyConnectedLayer(feats[i], feats[i + 1], activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer("w_avg", torch.zeros([w_dim]))

    def forward(self, z, c, cond, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        assert not update_emas, "inference only"
        x = None
        if self.z_dim > 0:
            x = normalize_2nd_moment(z.to(torch.float32))
        if self.c_dim > 0:
            if self.resnet_cond > 0:
                c = torch.cat([c, cond["resnet_feats"][:, :self.resnet_cond]], dim=1)
            y = normalize_2nd_moment(self.embed(c.to(torch.float32)))
            x = torch.cat([x, y], dim=1) if x is not None else y
        for i in range(self.num_layers):
            x = getattr(self, f"fc{i}")(x)
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


# How the 3x3 convolutions feed the matrix cores when a layer does not say (`layer.mma_f16`): "f32" = v_mfma_f32_32x32x2_f32,
# "x2" = two-term f16 operands on v_mfma_f32_32x32x16_f16 (fp32-class results: tests/test_hip_synthesis.py measures its error
# against float64 next to the f32 kernel's; ~1.8x faster).  The environment variable is for A/B runs.
DEFAULT_CONV_MMA = os.environ.get("P3D_CONV_MMA", "x2")
DEFAULT_CONV_MMA_1X1 = os.environ.get("P3D_CONV_MMA_1X1", "f32")  # the 1x1 ToRGB layers (A/B knob, measured in round 3)


# The activation IMAGE (ops.ActImage; csrc/p3d_synthesis.hip "activation IMAGE"): in a block whose map is >= IMG_MIN_RES^2, conv0's
# last pass writes conv1's two-term operand directly (conv1's styles applied, hi / lo split done) instead of an fp32 tensor that
# conv1 would modulate and split again in each of its channel-tile workgroups.  Bit-identical results; P3D_CONV_IMG=0 for A/B runs.
CONV_IMG = os.environ.get("P3D_CONV_IMG", "1") != "0"
# StylePlan returns the previous call's result for the same ws object (views of one subject).  Off: every pass computes its styles
# (the pass timings of tools/bench_backbone.py, graph_backbone.py, profile_backbone.py are taken that way).
STYLE_MEMO = os.environ.get("P3D_STYLE_MEMO", "1") != "0"
IMG_MIN_RES = int(os.environ.get("P3D_W3_MIN_W", "32"))  # (the library's own rule is asked as well: ops.takes_image)
# A block of <= 4 image channels (the super-resolution's) computes its ToRGB sums in conv1's epilogue (ops.conv_fuses_torgb): the
# activation is not read back — and not written when nobody else reads it.  fp32-class agreement with the stand-alone ToRGB launch
# (another summation order); P3D_TORGB_RIDES=0 for A/B runs.
TORGB_RIDES = os.environ.get("P3D_TORGB_RIDES", "1") != "0"
# noise_mode='random': one draw per pass for all layers (NoisePool) instead of one per layer.  The pool consumes the device generator
# in ONE randn call per pass, so under a fixed torch seed the noise VALUES differ from the reference's call-for-call sequence (same
# distribution; INTEGRATION.md "seed compatibility").  Switches, all read at call time: P3D_NOISE_POOL=0 in the environment (the
# default for the process), set_noise_pool(False) (the process, at run time), `net.noise_pool = False` on a SynthesisNetwork /
# TriPlaneGenerator.set_noise_pool(False) (one network) — then every layer calls torch.randn itself, in the reference's order
# (networks_stylegan2.py:342), and a seeded run reproduces the reference's draws (tests/test_host_cpu.py).
NOISE_POOL = os.environ.get("P3D_NOISE_POOL", "1") != "0"


def set_noise_pool(state):
    """Process-wide default of the pooled random noise (see NOISE_POOL); returns the previous value."""
    global NOISE_POOL
    prev, NOISE_POOL = NOISE_POOL, bool(state)
    return prev


def _takes_image(layer, res):
    """A plain 3x3 layer that can stage its input from an activation image: two-term operands, 16-channel K chunks, a map of at
    least IMG_MIN_RES columns (the wide-tile kernel)."""
    mode = layer.__dict__.get("mma_f16")
    if mode is None:
        mode = "x2" if DEFAULT_CONV_MMA == "x2" else False
    w = layer._parameters["weight"]
    return CONV_IMG and mode == "x2" and layer.up == 1 and w.shape[-1] == 3 and res >= IMG_MIN_RES and ops.takes_image(layer.in_channels, w.shape[0], res, 1)


def _next_conv0_styles(next_block, next_pre, res):
    """The styles of next_block.conv0 if that layer will stage its input (a res x res map) from an activation image, else None."""
    if next_block is None or next_pre is None or not CONV_IMG or next_pre.get("conv0") is None or next_pre["conv0"][1] is None:
        return None
    layer = next_block._modules["conv0"]
    mode = layer.__dict__.get("mma_f16")
    if mode is None:
        mode = "x2" if DEFAULT_CONV_MMA == "x2" else False
    if mode != "x2" or not ops.takes_image_up(layer.in_channels, layer.out_channels, res):
        return None
    return next_pre["conv0"][0]


def _hands_image(layer):
    """A plain two-term 3x3 layer fed an fp32 tensor can still write its result as the next layer's activation image (the library
    does it from the launch that finishes the layer)."""
    mode = layer.__dict__.get("mma_f16")
    if mode is None:
        mode = "x2" if DEFAULT_CONV_MMA == "x2" else False
    return CONV_IMG and mode == "x2" and layer.up == 1 and layer.in_channels % 16 == 0 and layer.out_channels % 8 == 0


def _f16_operand(layer):
    """The cached f16 operand copy of layer.weight when the layer runs on f16 MFMA operands, else None.  `layer.mma_f16`:
    False = fp32 operands, True = one f16 term ([O,k*k,I]; TriPlaneGenerator.set_sr_mma_f16), "x2" = two-term operands
    ([2,O,k*k,I]); unset = DEFAULT_CONV_MMA for the plain 3x3 layers.  A plain attribute, not a buffer: the state_dict stays
    the reference's."""
    mode = layer.__dict__.get("mma_f16")  # None | False | True | "x2"
    if mode is None:
        mode = "x2" if (DEFAULT_CONV_MMA == "x2" and (layer._parameters["weight"].shape[-1] == 3 or DEFAULT_CONV_MMA_1X1 == "x2")) else False
    if not mode or layer.in_channels % 16 != 0:
        return None
    w = layer._parameters["weight"]
    key = (w.data_ptr(), w._version, mode)
    d = layer.__dict__
    if d.get("_wh_key") != key or not memo.enabled():
        # (two-term 3x3 layers: in the layout the library's dispatch of THIS layer consumes — the pipelined kernels' own LDS image)
        up = layer.__dict__.get("up", 1)
        lay = ops.conv_weight_layout(layer.in_channels, w.shape[0], layer.__dict__.get("resolution", 0) // up, up) if (mode == "x2" and w.shape[-1] == 3 and "resolution" in layer.__dict__) else 0
        d["_wh"], d["_wh_key"] = ops.conv_weights_to_f16(w.detach(), split=(mode == "x2"), layout=lay), key
        f = d.get("conv_domain_flag")
        if isinstance(f, DomainFlags):  # new weights (load_state_dict, copy_params_and_buffers, an optimiser step): check their domain once
            f.dirty = True
    return d["_wh"]


class DomainFlags:
    """The out-of-domain flag words of one generator's two-term convolutions (TriPlaneGenerator.watch_conv_domain): one int32 word
    PER DEVICE, created when a layer first runs there — a generator moved with .to(device) keeps being watched (one
    tensor on the device of the moment made every forward after G.to(other) raise).  Shared by the generator's layers; never
    pickled (the attribute is in _CACHE_ATTRS) — a copy of the generator is unwatched until it asks."""

    def __init__(self):
        self.words = {}
        self.dirty = True  # operands were (re-)derived from the weights since the flag was last read: TriPlaneGenerator reads it once

    def get(self, device):
        device = torch.device(device)
        w = self.words.get(device)
        if w is None:
            w = self.words[device] = ops.conv_domain_flag(device)
        return w


def _domain_flag(layer, device):
    f = layer.__dict__.get("conv_domain_flag")
    return f.get(device) if isinstance(f, DomainFlags) else f  # (a bare int32 tensor set by hand is still honoured)


class SynthesisLayer(_CacheFree):
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True,
                 activation="lrelu", resample_filter=(1, 3, 3, 1), conv_clamp=None, channels_last=False):
        super().__init__()
        self.in_channels, self.out_channels, self.w_dim, self.resolution = in_channels, out_channels, w_dim, resolution
        self.up, self.use_noise, self.activation, self.conv_clamp = up, use_noise, activation, conv_clamp
        self.register_buffer("resample_filter", ops.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.act_gain = SQRT2 if activation == "lrelu" else 1.0  # bias_act.py:23-33 def_gain
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        if use_noise:
            self.register_buffer("noise_const", torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def _const_noise(self):
        """`noise_const * noise_strength` (networks_stylegan2.py:346), once per parameter version instead of once per call."""
        nc, ns = self._buffers["noise_const"], self._parameters["noise_strength"]  # (dict reads: nn.Module.__getattr__ is ~10x slower)
        key = (nc.data_ptr(), nc._version, ns.data_ptr(), ns._version)
        hit = self.__dict__.get("_noise_cache")
        if hit is None or hit[0] != key or not memo.enabled():
            hit = (key, (nc * ns.detach()).contiguous())
            self._noise_cache = hit
        return hit[1]

    def forward(self, x, w, noise_mode="random", fused_modconv=True, gain=1, pre=None, next_styles=None, noise_pool=None, rgb=None,
                want_y=True):
        """pre: (styles [N,I], demodulation coefficients [N,O]) already computed by a StylePlan for this layer, or None.
        x: fp32 [N,I,H,W], or the ops.ActImage the previous layer prepared for this one (its styles are in it).
        next_styles (up-sampling layers): return the ops.ActImage of the following layer, whose styles these are.
        noise_pool: a NoisePool of the enclosing network — this layer's random noise is its next slice of ONE draw per pass.
        rgb: (ToRGB weights [R,O], ToRGB styles [N,O]) — the block's ToRGB layer rides on this launch (ops.modulated_conv2d's
        rgb_weight / rgb_styles: returns (y, image, partial)); want_y=False: the fp32 result is not written."""
        assert noise_mode in ["random", "const", "none"]
        styles, dcoef = pre if pre is not None else (self.affine(w), None)
        noise = None
        if self.use_noise and noise_mode == "random":
            if noise_pool is not None:
                noise = noise_pool.take(self, x.shape[0])
            else:  # networks_stylegan2.py:342, call for call
                noise = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device) * self.noise_strength
        if self.use_noise and noise_mode == "const":
            noise = self._const_noise()
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        prm = self._parameters
        ride = {} if rgb is None else dict(rgb_weight=rgb[0], rgb_styles=rgb[1], want_y=want_y)
        return ops.modulated_conv2d(x, prm["weight"], styles, noise=noise, up=self.up, padding=self.padding,
                                    resample_filter=self._buffers["resample_filter"], demodulate=True, bias=prm["bias"],
                                    act=self.activation, gain=self.act_gain * gain, clamp=clamp, weight_f16=_f16_operand(self),
                                    saturated=_domain_flag(self, x.device), dcoef=dcoef, next_styles=next_styles, **ride)


class NoisePool:
    """noise_mode='random' for a whole pass in two launches.  The reference draws `randn([N,1,res,res]) * noise_strength` inside every
    SynthesisLayer.forward (networks_stylegan2.py:342): 2 launches x 15 layers in the launch-bound head of a batch-1 backbone pass
    (generate.py's G.f calls leave noise_mode at 'random').  Here ONE randn call draws the values of all layers of the pass, in the layers' execution order, and ONE multiply applies every
    layer's strength (a per-element vector cached per parameter version); each layer then takes its slice.  Same distribution, same
    per-layer independence; the stream of the device generator is consumed in one call instead of fifteen, so the values differ from
    a call-for-call run under the same seed (the reference's own GPU and CPU streams differ from each other in the same way)."""

    def __init__(self, layers):
        self.layers = [l for l in layers if l.use_noise]
        self._scale = {}  # (N, device) -> (key, per-element strengths, offsets)
        self._cur = None

    def _scales(self, N, dev):
        key = tuple([(l._parameters["noise_strength"].data_ptr(), l._parameters["noise_strength"]._version) for l in self.layers])
        hit = self._scale.get((N, dev))
        if hit is None or hit[0] != key or not memo.enabled():
            offs, o = {}, 0
            for l in self.layers:
                offs[id(l)] = o
                o += N * l.resolution * l.resolution
            sc = torch.cat([l.noise_strength.detach().to(torch.float32).reshape(1).expand(N * l.resolution * l.resolution) for l in self.layers])
            hit = self._scale[(N, dev)] = (key, sc.contiguous(), offs, o)
        return hit

    def draw(self, N, dev):
        """Start a pass: draw and scale the noise of every layer for batch size N."""
        _, sc, offs, total = self._scales(N, dev)
        self._cur = (torch.randn([total], device=dev).mul_(sc), offs, N)
        return self

    def take(self, layer, N):
        buf, offs, n = self._cur
        assert n == N, "NoisePool: batch size changed inside a pass"
        o = offs[id(layer)]
        r = layer.resolution
        return buf[o:o + N * r * r].view(N, 1, r, r)


class ToRGBLayer(_CacheFree):
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        self.in_channels, self.out_channels, self.w_dim, self.conv_clamp = in_channels, out_channels, w_dim, conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))

    def forward(self, x, w, fused_modconv=True, pre=None, skip=None, skip_filter=None):
        """skip / skip_filter: the previous block's image and the block's resample filter -> `upsample2d(skip) + torgb(x)`, the
        skip connection of SynthesisBlock.forward (networks_stylegan2.py:476-478), from the same launch."""
        styles = pre[0] if pre is not None else self.affine(w) * self.weight_gain
        weight, bias, d = self._parameters["weight"], self._parameters["bias"], self.__dict__
        if weight.shape[0] <= 96 and weight.shape[-1] == 1 and not d.get("mma_f16") and x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0:
            # the dedicated GEMM kernel (p3d_torgb_f32): the activation is read once, the skip image is added in the same launch
            key = (weight.data_ptr(), weight._version)
            if d.get("_wt_key") != key or not memo.enabled():
                self._wt, self._wt_key = ops.torgb_weights(weight.detach()), key
            return ops.torgb(x, d["_wt"], weight.shape[0], styles, bias=bias, clamp=self.conv_clamp, skip=skip, skip_filter=skip_filter)
        y = ops.modulated_conv2d(x, weight, styles, demodulate=False, bias=bias, act="linear", gain=1.0,
                                 clamp=self.conv_clamp, weight_f16=_f16_operand(self), saturated=_domain_flag(self, x.device))
        return ops.upsample2d_add(skip, skip_filter, y) if skip is not None else y


def _torgb_rides(block, x, pre):
    """(weights [R,O], styles [N,O]) when `block`'s ToRGB layer can ride on its conv1 launch (ops.conv_fuses_torgb: an image-fed
    plain 3x3 layer on the pipelined kernel, <= 4 image channels — the super-resolution blocks), else None.  x: conv1's input."""
    if not (CONV_IMG and TORGB_RIDES and isinstance(x, ops.ActImage)):
        return None
    t = block._modules["torgb"]
    tw = t._parameters["weight"]
    if tw.shape[0] > 4 or tw.shape[-1] != 1 or t.__dict__.get("mma_f16") or pre.get("torgb") is None:
        return None
    N, I, H, W = x.shape
    if H % 2 or W % 2 or not ops.conv_fuses_torgb(N, I, block._modules["conv1"].out_channels, H, W, tw.shape[0]):
        return None
    return tw.detach().reshape(tw.shape[0], tw.shape[1]), pre["torgb"][0]


class StylePlan:
    """Every `affine` (w -> styles) layer of a stack of SynthesisBlocks, and the demodulation coefficients of its
    SynthesisLayers, computed for a whole forward pass in four launches instead of ~6 per layer:
        Y = ws @ W_all^T                      one GEMM  [N * num_ws, 512] x [512, sum I]   (every affine weight, pre-scaled)
        S = (Y[n, widx(col), col] + b_all) * g_all      gather + add + mul  (g = ToRGB's weight_gain, 1 elsewhere)
        d = rsqrt(W2 . S^2 + 1e-8)            one launch for all layers (ops.demod_coefs), W2 = sum over taps of w^2, cached
    The results are laid out layer-major ([N, I_l] / [N, O_l] blocks back to back), so each layer gets a contiguous view.
    The reference evaluates each affine separately (networks_stylegan2.py:342,377) and recomputes sum (w*s)^2 from the full
    weights per call (:70-73): same mathematics, different summation order (covered by the tolerance of the golden tests).
    entries: [(block name, layer name, layer module, w index)] in execution order."""

    def __init__(self, entries):
        self.entries = entries
        self._key = None
        self._per_n = {}

    def _build(self, dev):
        Ws, bs, gs, widx, w2s, table = [], [], [], [], [], []
        s_off = w2_off = o_off = 0
        self.slices = []
        for _, _, layer, wi in self.entries:
            w, b = layer.affine._scaled(torch.float32)
            I = w.shape[0]
            Ws.append(w)
            bs.append(b)
            is_rgb = isinstance(layer, ToRGBLayer)
            gs.append(torch.full((I,), float(layer.weight_gain) if is_rgb else 1.0, device=dev))
            widx.append(torch.full((I,), wi, dtype=torch.long, device=dev))
            O = layer.weight.shape[0]
            if not is_rgb:
                w2 = layer.weight.detach().float().square().sum(dim=(2, 3)).contiguous()  # [O, I]
                w2s.append(w2.reshape(-1))
                table.append([w2_off, s_off, o_off, O, I])
                w2_off += O * I
                self.slices.append((s_off, I, o_off, O))
                o_off += O
            else:
                self.slices.append((s_off, I, None, O))
            s_off += I
        self.W_all_t = torch.cat(Ws, dim=0).t().contiguous()  # [512, sum I]
        self.b_all, self.g_all, self.widx = torch.cat(bs), torch.cat(gs), torch.cat(widx)
        self.w2_all = torch.cat(w2s) if w2s else None
        self.table, self.sumI, self.sumO = table, s_off, o_off
        self._per_n = {}

    def _for_n(self, N, dev):
        hit = self._per_n.get((N, self.num_ws))
        if hit is None:
            # flat gather index into Y [N, num_ws, sum I] producing the layer-major layout, and the matching bias / gain vectors
            idx, bias, gain = [], [], []
            n = torch.arange(N, device=dev)
            for (s_off, I, _, _), (_, _, _, wi) in zip(self.slices, self.entries):
                col = torch.arange(s_off, s_off + I, device=dev)
                idx.append(((n[:, None] * self.num_ws + wi) * self.sumI + col[None, :]).reshape(-1))
                bias.append(self.b_all[s_off:s_off + I].repeat(N))
                gain.append(self.g_all[s_off:s_off + I].repeat(N))
            tab, first = [], 0
            for w2_off, s_off, o_off, O, I in self.table:
                tab.append([w2_off, s_off * N, o_off * N, O, I, first])
                first += N * O
            tab.append([0, 0, 0, 1, 1, first])
            hit = (torch.cat(idx), torch.cat(bias), torch.cat(gain),
                   torch.tensor(tab, dtype=torch.int32, device=dev).contiguous(), first)
            self._per_n[(N, self.num_ws)] = hit
        return hit

    def __call__(self, ws, memo_of=None):
        """ws [N, num_ws, 512] -> {block name: {layer name: (styles [N,I], dcoef [N,O] or None)}}
        memo_of: the tensor OBJECT this ws was derived from (or ws itself).  The result of the previous call is returned when it
        is the same object at the same version and no parameter changed: the views of one subject share their ws
        (TriPlaneGenerator.f), and these five launches sit in the launch-bound head of a call."""
        dev = ws.device
        # the parameters the plan's tables derive from, read through the modules' own `_parameters` dicts (the dict objects live
        # as long as the modules; `layer.affine.weight` goes through nn.Module.__getattr__ twice — 224 such lookups per call were
        # 0.15 ms of the launch-bound head of a G.f call, tools/host_profile.py)
        pd = self.__dict__.get("_param_dicts")
        if pd is None:
            pd = self._param_dicts = [(l.affine._parameters, l._parameters) for _, _, l, _ in self.entries]
        key = tuple([(a["weight"].data_ptr(), a["weight"]._version, a["bias"].data_ptr(), a["bias"]._version,
                      w["weight"].data_ptr(), w["weight"]._version) for a, w in pd]) + (dev,)
        if key != self._key or not memo.enabled():
            self._build(dev)
            self._key = key
            self._memo = None
        last = getattr(self, "_memo", None)
        if not (STYLE_MEMO and memo.enabled()):
            memo_of = None
        if memo_of is not None and last is not None and last[0] is memo_of and last[1] == memo_of._version and last[2] == tuple(ws.shape):
            return last[3]
        out = self._compute(ws, dev)
        self._memo = (memo_of, memo_of._version, tuple(ws.shape), out) if memo_of is not None else None
        return out

    def _compute(self, ws, dev):
        N, self.num_ws = ws.shape[0], ws.shape[1]
        idx, bias, gain, table, total_waves = self._for_n(N, dev)
        Y = torch.matmul(ws.reshape(N * self.num_ws, -1).to(torch.float32), self.W_all_t)  # [N * num_ws, sum I]
        S = Y.reshape(-1)[idx].add_(bias).mul_(gain)
        d = None
        if self.w2_all is not None:
            d = ops.demod_coefs(self.w2_all, S, table, len(self.table), N, total_waves,
                                torch.empty((N * self.sumO,), dtype=torch.float32, device=dev))
        out = {}
        for (s_off, I, o_off, O), (bname, lname, _, _) in zip(self.slices, self.entries):
            st = S[s_off * N:(s_off + I) * N].view(N, I)
            dc = d[o_off * N:(o_off + O) * N].view(N, O) if o_off is not None else None
            out.setdefault(bname, {})[lname] = (st, dc)
        return out


def plan_entries(named_blocks, block_w0):
    """[(block name, layer name, layer, w index)] for blocks [(name, SynthesisBlock)] whose first w index is block_w0[i]."""
    ent = []
    for (name, blk), w0 in zip(named_blocks, block_w0):
        k = 0
        for lname in (("conv1",) if blk.in_channels == 0 else ("conv0", "conv1")) + ("torgb",):
            ent.append((name, lname, getattr(blk, lname), w0 + k))
            k += 1
    return ent


class SynthesisBlock(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture="skip",
                 resample_filter=(1, 3, 3, 1), conv_clamp=256, use_fp16=False, fp16_channels_last=False,
                 fused_modconv_default=True, **layer_kwargs):
        super().__init__()
        if architecture != "skip":
            raise NotImplementedError("the triplane backbone uses the 'skip' architecture (networks_stylegan2.py:396)")
        self.in_channels, self.w_dim, self.resolution, self.img_channels = in_channels, w_dim, resolution, img_channels
        self.is_last, self.architecture = is_last, architecture
        self.register_buffer("resample_filter", ops.setup_filter(resample_filter))
        self.num_conv = self.num_torgb = 0
        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2,
                                        resample_filter=resample_filter, conv_clamp=conv_clamp, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp,
                                    **layer_kwargs)
        self.num_conv += 1
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp)
        self.num_torgb += 1

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, update_emas=False, pre=None, x_image=None, next_styles=None,
                need_x=True, **layer_kwargs):
        """pre: {layer name: (styles, demod coefficients)} from a StylePlan (all affine layers of the network in one GEMM),
        or None: every layer runs its own affine like the reference (networks_stylegan2.py:342,377).
        x_image: x as the ops.ActImage the previous block's conv1 wrote for this block's conv0 (then x itself is not read by conv0);
        next_styles: the styles of the NEXT block's conv0 -> conv1 also writes its result as that layer's image and the block returns
        (x, img, image) instead of (x, img).
        need_x=False: the caller does not read the returned fp32 x (the next block takes the image, or this is the last block) — where
        the ToRGB layer rides on conv1's launch (_torgb_rides) x is then not written and None is returned in its place."""
        w_iter = iter(ws.unbind(dim=1))
        pre = pre or {}
        if self.in_channels == 0:
            x = self._parameters["const"].to(torch.float32).unsqueeze(0)
            if ws.shape[0] != 1:  # (batch 1: the convolution only reads x — no copy of the constant, networks_stylegan2.py:456-457)
                x = x.repeat([ws.shape[0], 1, 1, 1])
            if next_styles is not None and _hands_image(self.conv1):
                x, x_next = self.conv1(x, next(w_iter), pre=pre.get("conv1"), next_styles=next_styles, **layer_kwargs)
            else:
                x, x_next = self.conv1(x, next(w_iter), pre=pre.get("conv1"), **layer_kwargs), None
        else:
            # conv0 hands conv1 its operand (activation image) when conv1 can stage from one and its styles / demodulation
            # coefficients are known up front (a StylePlan)
            p1 = pre.get("conv1")
            img_ok = p1 is not None and p1[1] is not None and _takes_image(self.conv1, self.resolution) and self.conv1.in_channels % 8 == 0
            x0 = x_image if (x_image is not None and pre.get("conv0") is not None and pre["conv0"][1] is not None) else x.to(torch.float32)
            x = self.conv0(x0, next(w_iter), pre=pre.get("conv0"), next_styles=p1[0] if img_ok else None, **layer_kwargs)
            rides = _torgb_rides(self, x, pre)
            if rides is not None:  # ToRGB's channel sums from conv1's epilogue, finished by one small launch
                hand = next_styles if isinstance(x, ops.ActImage) else None
                x, x_next, part = self.conv1(x, next(w_iter), pre=p1, next_styles=hand, rgb=rides, want_y=need_x or (next_styles is not None and hand is None),
                                             **layer_kwargs)
                next(w_iter)
                t = self._modules["torgb"]
                img = ops.torgb_combine(part, bias=t._parameters["bias"], clamp=t.conv_clamp, skip=None if img is None else img.to(torch.float32),
                                        skip_filter=self._buffers["resample_filter"])
                if next_styles is not None:
                    return x, img, x_next
                return x, img
            if next_styles is not None and (isinstance(x, ops.ActImage) or _hands_image(self.conv1)):
                # conv1 hands the next block's conv0 its operand: from the pipelined kernel's epilogue, or (split-K layers of the small
                # maps) from the launch that sums the slices — no conversion pass in front of conv0 either way
                x, x_next = self.conv1(x, next(w_iter), pre=p1, next_styles=next_styles, **layer_kwargs)
            else:
                x, x_next = self.conv1(x, next(w_iter), pre=p1, **layer_kwargs), None
        # y = torgb(x); img = upsample2d(img, resample_filter); img = img.add_(y)  (networks_stylegan2.py:476-478): one launch
        img = self.torgb(x, next(w_iter), pre=pre.get("torgb"), skip=None if img is None else img.to(torch.float32),
                         skip_filter=self.resample_filter)
        if next_styles is not None:
            return x, img, x_next
        return x, img


def _pixel_unshuffle(t, f):
    """einops 'bs ch (h i) (w j) -> bs (i j ch) h w' (networks_stylegan2.py:632)."""
    b, ch, H, W = t.shape
    t = t.reshape(b, ch, H // f, f, W // f, f).permute(0, 3, 5, 1, 2, 4)
    return t.reshape(b, f * f * ch, H // f, W // f)


class SynthesisNetwork(_CacheFree):
    def __init__(self, w_dim, img_resolution, img_channels, cond_mode, channel_base=32768, channel_max=512,
                 num_fp16_res=4, **block_kwargs):
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        super().__init__()
        if num_fp16_res != 0:
            raise NotImplementedError("the triplane backbone is fp32 (num_fp16_res=0, trainers/train_eclustrousC.py:253)")
        self.cond_mode, self.w_dim, self.img_resolution, self.img_channels = cond_mode, w_dim, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.num_fp16_res = num_fp16_res
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        ch = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        self.num_ws = 0
        for res in self.block_resolutions:
            block = SynthesisBlock(ch[res // 2] if res > 4 else 0, ch[res], w_dim=w_dim, resolution=res,
                                   img_channels=img_channels, is_last=(res == img_resolution), **block_kwargs)
            self.num_ws += block.num_conv + (block.num_torgb if res == img_resolution else 0)
            setattr(self, f"b{res}", block)

    # ---- PAniC-3D conditioning between blocks (networks_stylegan2.py:551-694): element-wise glue on x / img.  What is added /
    # multiplied / concatenated depends only on the conditioning images and the level — not on x — so it is prepared once per
    # set of conditioning tensors (the 16 views of a subject share them: _util/eg3d_metrics3d.py / generate.py) and applied IN PLACE on
    # the block's fresh output: one launch per level instead of seven (flip, scale, resize, repeat, add, two-part cat), which at
    # batch 1 were ~40 launches the host could not issue as fast as the GPU ran them.  Same values: x[:, -k:] + t either way.
    def _cond_prepared(self, key, tensors, make):
        """make() cached under `key` for exactly these tensor OBJECTS at their current versions (strong references are kept, so
        a hit can never be another tensor that reused an address)."""
        if not memo.enabled():
            return make()
        cache = self.__dict__.setdefault("_cond_cache", {})
        hit = cache.get(key)
        if hit is not None and len(hit[0]) == len(tensors) and all(a is b and v == b._version for (a, v), b in zip(hit[0], tensors)):
            return hit[1]
        val = make()
        # Round 6: a term prepared for OTHER conditioning tensors of the same shape (the next subject) is written INTO the tensor the
        # previous subject's term lived in: the prepared terms keep their addresses, which a captured view (TriPlaneGenerator's
        # launch replay) has baked into its launches.  A term that cannot be updated in place is replaced, and `_cond_gen` says so.
        if hit is not None and hit[1].shape == val.shape and hit[1].dtype == val.dtype and hit[1].device == val.device \
                and not torch.is_grad_enabled():
            hit[1].copy_(val)
            val = hit[1]
        elif hit is not None:
            self.__dict__["_cond_gen"] = self.__dict__.get("_cond_gen", 0) + 1
        cache[key] = ([(t, t._version) for t in tensors], val)
        return val

    def clear_cond_cache(self):
        """Drop the prepared conditioning terms (they hold strong references to the last subject's conditioning images and their
        resized copies: a long-running server calls this between subjects, or when it is done; G.to(device) does it too)."""
        if self.__dict__.pop("_cond_cache", None) is not None:
            self.__dict__["_cond_gen"] = self.__dict__.get("_cond_gen", 0) + 1

    def _apply(self, fn):  # .to() / .cuda() / .float(): prepared terms live on the old device
        self.clear_cond_cache()
        self.__dict__.pop("_style_plan", None)
        self.__dict__.pop("_noise_pool", None)
        return super()._apply(fn)

    def _condition(self, lvl, res, x, img, cond, cm, chonkadd, image_styles=None, flag=None):
        """x, img: the block's FRESH outputs — owned by this call, written in place below.  (Inference only: under autograd, or if a
        caller ever aliased a block output, the in-place adds would be visible through the alias; the blocks return new tensors.)
        image_styles: the styles of the next block's conv0 when that layer stages from an activation image — where the level's only
        edit of x is ONE in-place add (the resnet chonk, `add_4`, `add_shuffle2_4`: the released model's modes) the add and the image
        of the edited x come from one launch (ops.act_to_image_add) and the image is returned as the third value (else None)."""
        x, img = self._condition_impl(lvl, res, x, img, cond, cm, chonkadd, image_styles, flag)
        ximg = self.__dict__.pop("_cond_image", None)
        return x, img, ximg

    def _add_to(self, x, t, c0, image_styles, flag, fuse):
        """x[:, c0 : c0 + C_t] += t, with the next conv0's image from the same launch where asked for and possible."""
        if fuse and image_styles is not None and x.is_contiguous() and x.dtype == torch.float32 and t.dtype == torch.float32 \
                and c0 % 8 == 0 and t.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and t.shape[0] in (1, x.shape[0]) and tuple(t.shape[2:]) == tuple(x.shape[2:]):
            self.__dict__["_cond_image"] = ops.act_to_image_add(x, image_styles, t, c0, saturated=flag)
        else:
            x[:, c0:c0 + t.shape[1]].add_(t)

    def _condition_impl(self, lvl, res, x, img, cond, cm, chonkadd, image_styles, flag):
        self.__dict__.pop("_cond_image", None)
        if self.cond_mode == "none":
            return x, img
        assert not (torch.is_grad_enabled() and (x.requires_grad or img.requires_grad)), "inference only: the conditioning is applied in place"
        if res == 8 and chonkadd > 0:  # resnet "chonk" added to the first channels of the 8x8 activations (:554-560)
            k = chonkadd
            chonk = cond["resnet_chonk"]
            t = self._cond_prepared(("chonk", k), [chonk], lambda: chonk[:, :k].to(x.dtype).clone())  # (a copy: an address that outlives the subject)
            self._add_to(x, t, 0, image_styles, flag, True)
            return x, img
        interp = torch.nn.functional.interpolate
        if self.cond_mode.startswith("ortho_front."):
            names = ["image_ortho_front"]
            if "gt_sides" in cm:
                names += ["image_ortho_left", "image_ortho_right"]
            if "dorthoA" in cm:
                names += ["image_dorthoA_left", "image_dorthoA_right"]
            srcs = [cond[n] for n in names]

            def make_cimg():
                cimg = cond["image_ortho_front"].flip(dims=(-2,))
                if "gt_sides" in cm:
                    cimg = torch.cat([cimg, cond["image_ortho_left"].permute(0, 1, 3, 2).flip(dims=(-1, -2)),
                                      cond["image_ortho_right"].permute(0, 1, 3, 2).flip(dims=(-1,))], dim=1)
                if "dorthoA" in cm:
                    cimg = torch.cat([cimg, cond["image_dorthoA_left"].permute(0, 1, 3, 2).flip(dims=(-1, -2)),
                                      cond["image_dorthoA_right"].permute(0, 1, 3, 2).flip(dims=(-1,))], dim=1)
                cimg = cimg * 2 - 1
                if "cond_img_norm_4" in cm:
                    cimg = 4 * cimg
                return cimg

            def resized(tag, unshuffle=False):
                """cimg at x's size, repeated to x.shape[1] / 4 channels ('add_4' / 'add_shuffle2_4'), prepared once"""
                def make():
                    cimg = self._cond_prepared("cimg", srcs, make_cimg)
                    t = _pixel_unshuffle(cimg, cimg.shape[-1] // x.shape[-1]) if unshuffle else interp(cimg, size=x.shape[-2:], mode="bilinear")
                    return t.repeat(1, int((x.shape[1] / 4) // t.shape[1]), 1, 1).contiguous()
                return self._cond_prepared((tag, lvl, tuple(x.shape[1:]), unshuffle), srcs, make)

            # (one add and nothing else that edits x at this level: the add may carry the next conv0's image along)
            only_add = ("add_4" in cm) != ("add_shuffle2_4" in cm) and not ({"concatfront", "mult_shuffle2_4", "crossavg_4", "crossavgt_38"} & cm)
            if "add_4" in cm:
                t = resized("add_4")
                self._add_to(x, t, x.shape[1] - t.shape[1], image_styles, flag, only_add)
            if "concatfront" in cm:
                t = self._cond_prepared(("concatfront", lvl, tuple(x.shape[2:])), srcs,
                                        lambda: interp(self._cond_prepared("cimg", srcs, make_cimg), size=x.shape[-2:], mode="bilinear"))
                x[:, -t.shape[1]:].copy_(t)
            if "add_shuffle2_4" in cm or "mult_shuffle2_4" in cm:
                t = resized("shuffle2_4", unshuffle=not (lvl < len(self.block_resolutions) - 2))
                if "add_shuffle2_4" in cm:
                    self._add_to(x, t, x.shape[1] - t.shape[1], image_styles, flag, only_add)
                else:
                    x[:, -t.shape[1]:].mul_(t)
            if "inj_6b_4" in cm and res == self.block_resolutions[-1]:
                front = cond["image_ortho_front"]
                t = self._cond_prepared(("inj_6b_4", tuple(img.shape[2:])), [front],
                                        lambda: interp((front.flip(dims=(-2,)) * 2 - 1) * 4, size=img.shape[-2:], mode="bilinear"))
                img[:, :t.shape[1]].add_(t)
        if "crossavg_4" in cm:
            k = int(x.shape[1] // 8)
            h, v = x[:, 0:k], x[:, k:2 * k]
            x = torch.cat([h.mean(dim=-1, keepdim=True).expand(h.shape), v.mean(dim=-2, keepdim=True).expand(v.shape),
                           x[:, 2 * k:]], dim=1)
        elif "crossavgt_38" in cm:
            k = int(x.shape[1] // 8)
            h, v, t = x[:, 0:k], x[:, k:2 * k], x[:, 2 * k:3 * k]
            x = torch.cat([h.mean(dim=-1, keepdim=True).expand(h.shape), v.mean(dim=-2, keepdim=True).expand(v.shape),
                           t.permute(0, 1, 3, 2), x[:, 3 * k:]], dim=1)
        return x, img

    def forward(self, ws, cond, latent_injection=None, stop_level=None, return_more=False, **block_kwargs):
        ws = ws.to(torch.float32)
        block_ws, w_idx = [], 0
        for res in self.block_resolutions:  # a block's ToRGB shares the next block's first w (:534-537)
            block = getattr(self, f"b{res}")
            block_ws.append(ws.narrow(1, w_idx, block.num_conv + block.num_torgb))
            w_idx += block.num_conv
        cm = set(self.cond_mode.split("."))
        chonk = [int(c.split("_")[-1]) for c in cm if c.startswith("reschonk_add_")]
        chonk = chonk[0] if chonk else 0
        x = img = None
        ximgs = []
        plan = self.__dict__.get("_style_plan")
        if plan is None:
            starts, w0 = [], 0
            for res in self.block_resolutions:
                starts.append(w0)
                w0 += getattr(self, f"b{res}").num_conv
            plan = StylePlan(plan_entries([(f"b{res}", getattr(self, f"b{res}")) for res in self.block_resolutions], starts))
            self.__dict__["_style_plan"] = plan
        pre = plan(ws, memo_of=ws)  # every layer's styles + demodulation coefficients: one GEMM + three small launches
        use_pool = self.__dict__.get("noise_pool")  # per-network override (None: the process default)
        if block_kwargs.get("noise_mode", "random") == "random" and (NOISE_POOL if use_pool is None else use_pool):  # all layers' random noise of this pass: two launches
            pool = self.__dict__.get("_noise_pool")
            if pool is None:
                blocks = [getattr(self, f"b{res}") for res in self.block_resolutions]
                pool = self.__dict__["_noise_pool"] = NoisePool([l for b in blocks for l in ([b.conv1] if b.in_channels == 0 else [b.conv0, b.conv1])])
            block_kwargs["noise_pool"] = pool.draw(ws.shape[0], ws.device)
        x_image = None
        for lvl, (res, cur_ws) in enumerate(zip(self.block_resolutions, block_ws)):
            # conv1 of this block writes its result also as the image the next block's conv0 stages from (no conversion pass in
            # front of that layer) — unless something between the blocks edits x: the conditioning of this level, a latent injection
            nxt = getattr(self, f"b{self.block_resolutions[lvl + 1]}") if lvl + 1 < len(self.block_resolutions) else None
            injected = latent_injection is not None and f"da_{lvl}" in latent_injection
            x_untouched = self.cond_mode == "none" and not injected
            ns_any = _next_conv0_styles(nxt, pre.get(f"b{self.block_resolutions[lvl + 1]}") if nxt is not None else None, res) if not injected else None
            ns = ns_any if x_untouched else None
            out = getattr(self, f"b{res}")(x, img, cur_ws, pre=pre[f"b{res}"], x_image=x_image, next_styles=ns, **block_kwargs)
            x, img, x_image = out if ns is not None else (out[0], out[1], None)
            # a conditioned level edits x between the blocks: the edit (where it is one in-place add) writes the next conv0's image itself
            x, img, cimg = self._condition(lvl, res, x, img, cond, cm, chonk, image_styles=None if x_untouched else ns_any,
                                           flag=_domain_flag(nxt._modules["conv0"], x.device) if (nxt is not None and not x_untouched) else None)
            if cimg is not None:
                x_image = cimg
            x, img = x.contiguous(), img.contiguous()
            ximgs.append((x, img))
            if latent_injection is not None:
                if f"da_{lvl}" in latent_injection:
                    x = x + latent_injection[f"da_{lvl}"]
                if f"db_{lvl}" in latent_injection:
                    img = img + latent_injection[f"db_{lvl}"]
        if stop_level is None:
            ret = img
        else:
            ret = ximgs[stop_level][1]
            for i in range(stop_level + 1, len(self.block_resolutions)):
                ret = ops.upsample2d(ret, getattr(self, f"b{self.block_resolutions[i]}").resample_filter)
        return (ret, {"ximgs": ximgs, "block_ws": block_ws}) if return_more else ret


class Generator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, cond_mode, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution, self.img_channels, self.cond_mode = img_resolution, img_channels, cond_mode
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels,
                                          cond_mode=cond_mode, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, cond_mode=cond_mode,
                                      **mapping_kwargs)

    def forward(self, z, c, cond, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, cond, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return self.synthesis(ws, cond, **synthesis_kwargs)
