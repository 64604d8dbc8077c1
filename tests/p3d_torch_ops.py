"""Plain PyTorch fp32 (CPU) restatements of the synthesis operators of panic3d_amd.ops, with the SAME signatures — test
infrastructure: `install(monkeypatch, ops)` swaps them in, so that the HOST logic of stylegan2.py / generator.py (block wiring,
StylePlan, the conditioning between blocks, activation-image hand-overs, noise) runs on CPU tensors and can be compared with the
reference's golden outputs in the `-m "not gpu"` suite.  Each function follows the reference lines it cites; none of this is
imported by the product.

An activation IMAGE (ops.ActImage: the consumer's modulated operand in the kernels' f16 hi / lo layout) is represented here by an
ops.ActImage whose `.data` is simply the modulated fp32 activation `styles * x` [N,C,H,W]."""
import numpy as np
import torch
import torch.nn.functional as F


def _fir(f, gain):
    return (f.to(torch.float32) * gain).flip([0, 1])  # upfirdn2d.py:193-196 (true convolution; gain = up^2)


def upfirdn_up2(x, f, pad):
    """upfirdn2d with up = 2: zero-insert, pad [x0, x1, y0, y1], FIR with f * 4 (upfirdn2d.py:169-213)."""
    N, C, H, W = x.shape
    xu = torch.zeros((N, C, 2 * H, 2 * W), dtype=x.dtype)
    xu[:, :, ::2, ::2] = x
    xu = F.pad(xu, [pad[0], pad[1], pad[2], pad[3]])
    k = _fir(f, 4.0)[None, None].repeat(C, 1, 1, 1)
    return F.conv2d(xu, k, groups=C)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1):
    assert up == 2 and padding == 0 and gain == 1 and tuple(f.shape) == (4, 4)
    return upfirdn_up2(x, f, [2, 1, 2, 1])  # upfirdn2d.py:341-350: ((fw + up - 1) // 2, (fw - up) // 2)


def upsample2d_add(x, f, add=None):
    y = upsample2d(x, f)
    return y if add is None else y + add


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None):
    """bias_act.py:93-122 (_bias_act_ref): x + b -> act -> * gain -> clamp."""
    if b is not None:
        x = x + b.reshape([-1 if i == dim else 1 for i in range(x.ndim)])
    if act == "lrelu":
        x = F.leaky_relu(x, 0.2 if alpha is None else alpha)
        g = np.sqrt(2) if gain is None else gain
    else:
        assert act == "linear"
        g = 1.0 if gain is None else gain
    if g != 1:
        x = x * float(g)
    if clamp is not None and clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def install(monkeypatch, ops):
    """Swap the restatements into panic3d_amd.ops (undone by pytest's monkeypatch at the end of the test)."""
    ActImage = ops.ActImage

    def act_to_image(x, styles=None, saturated=None):
        return ActImage(x if styles is None else x * styles[:, :, None, None], x.shape)

    def act_to_image_add(x, styles, add, c0, saturated=None):
        x[:, c0:c0 + add.shape[1]].add_(add)
        return ActImage(x * styles[:, :, None, None], x.shape)

    def modulated_conv2d(x, weight, styles, noise=None, up=1, padding=0, resample_filter=None, demodulate=True, bias=None,
                         act="linear", gain=None, clamp=None, weight_f16=None, dcoef=None, saturated=None, next_styles=None):
        """networks_stylegan2.py:40-97 in its non-fused form (:76-85: shared weights, x * styles in, * dcoefs out) + the bias_act of
        SynthesisLayer.forward (:350-352)."""
        if isinstance(x, ActImage):
            xm = x.data
            assert not demodulate or dcoef is not None
        else:
            xm = x * styles[:, :, None, None]
        w = weight.to(torch.float32)
        if up == 1:
            y = F.conv2d(xm, w, padding=padding)
        else:  # conv2d_resample.py:114-128: transposed conv (stride 2), then the FIR with pad [1,1,1,1], gain 4
            y = F.conv_transpose2d(xm, w.transpose(0, 1), stride=2)
            k = _fir(resample_filter, 4.0)[None, None].repeat(y.shape[1], 1, 1, 1)
            y = F.conv2d(F.pad(y, [1, 1, 1, 1]), k, groups=y.shape[1])
        if demodulate:
            if dcoef is None:
                dcoef = ((w.square().sum(dim=(2, 3))[None] * styles.square()[:, None, :]).sum(dim=2) + 1e-8).rsqrt()
            y = y * dcoef.reshape(y.shape[0], -1, 1, 1)
        if noise is not None:
            y = y + noise
        y = bias_act(y, bias, act=act, gain=gain, clamp=clamp)
        if next_styles is None:
            return y
        img = ActImage(y * next_styles[:, :, None, None], y.shape)
        return (y, img) if up == 1 else img

    def torgb(x, weight_t, out_channels, styles, bias=None, clamp=None, skip=None, skip_filter=None):
        """ToRGBLayer.forward (:366-380; styles already carry weight_gain) + `img = upsample2d(img) + y` (:476-478)."""
        y = bias_act(F.conv2d(x * styles[:, :, None, None], weight_t), bias, clamp=clamp)
        return y if skip is None else upsample2d(skip, skip_filter) + y

    def demod_coefs(w2_all, styles_all, table, L, N, total_waves, out):
        """d = rsqrt(sum_i W2[o,i] * s[n,i]^2 + 1e-8) for every layer of a StylePlan (stylegan2.StylePlan._for_n's table rows:
        [w2 offset, styles offset, out offset, O, I, first wave])."""
        for w2_off, s_off, o_off, O, I, _ in table[:L].tolist():
            w2 = w2_all[w2_off:w2_off + O * I].view(O, I)
            s = styles_all[s_off:s_off + N * I].view(N, I)
            out[o_off:o_off + N * O] = (s.square() @ w2.t() + 1e-8).rsqrt().reshape(-1)
        return out

    for name, fn in dict(act_to_image=act_to_image, act_to_image_add=act_to_image_add, modulated_conv2d=modulated_conv2d, torgb=torgb, demod_coefs=demod_coefs,
                         torgb_weights=lambda w: w.to(torch.float32), bias_act=bias_act, upsample2d=upsample2d, upsample2d_add=upsample2d_add,
                         conv_weight_layout=lambda I, O, W, up: 0,
                         conv_weights_to_f16=lambda w, split=False, layout=0: torch.zeros((2,) + (w.shape[0], w.shape[2] * w.shape[3], w.shape[1]) if split
                                                                                            else (w.shape[0], w.shape[2] * w.shape[3], w.shape[1]), dtype=torch.float16)).items():
        monkeypatch.setattr(ops, name, fn)
