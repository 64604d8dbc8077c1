#!/usr/bin/env python3
"""The activation-image path (GPU): an up-sampling layer hands the plain 3x3 layer behind it its operand.  Bit-identity against
the fp32-tensor path and the time of both, at the layers where the time is.  python tools/experiments/img_conv_check.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import panic3d_amd as P
ops = P.ops


def timeit(fn, n=30, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    torch.manual_seed(0)
    dev = torch.device("cuda")
    f = ops.setup_filter((1, 3, 3, 1)).to(dev)
    for (N, I, O, H) in [(1, 256, 128, 256), (1, 32, 256, 128), (1, 256, 128, 128), (2, 512, 512, 32), (4, 512, 256, 64), (1, 512, 512, 16)]:
        x = torch.randn(N, I, H, H, device=dev)
        w0, w1 = torch.randn(O, I, 3, 3, device=dev), torch.randn(O, O, 3, 3, device=dev)
        s0, s1 = torch.randn(N, I, device=dev) * 0.5 + 1.0, torch.randn(N, O, device=dev) * 0.5 + 1.0
        b0, b1 = torch.randn(O, device=dev), torch.randn(O, device=dev)
        d0 = ((w0[None] * s0[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt().contiguous()
        d1 = ((w1[None] * s1[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt().contiguous()
        wf0, wf1 = ops.conv_weights_to_f16(w0, split=True) if I % 16 == 0 else None, ops.conv_weights_to_f16(w1, split=True)
        nz0, nz1 = torch.randn(2 * H, 2 * H, device=dev) * 0.1, torch.randn(2 * H, 2 * H, device=dev) * 0.1
        k0 = dict(up=2, padding=1, resample_filter=f, demodulate=True, bias=b0, act="lrelu", dcoef=d0, noise=nz0, weight_f16=wf0)
        k1 = dict(up=1, padding=1, demodulate=True, bias=b1, act="lrelu", dcoef=d1, noise=nz1, weight_f16=wf1)
        ya = ops.modulated_conv2d(ops.modulated_conv2d(x, w0, s0, **k0), w1, s1, **k1)
        img = ops.modulated_conv2d(x, w0, s0, next_styles=s1, **k0)
        yb = ops.modulated_conv2d(img, w1, None, **k1)
        mid = ops.modulated_conv2d(x, w0, s0, **k0)
        row = dict(N=N, I=I, O=O, H=H, identical=bool(torch.equal(ya, yb)),
                   image_is_s_times_x=float((img.float() - mid * s1[:, :, None, None]).abs().max() / mid.abs().max()),
                   act_to_image_identical=bool(torch.equal(ops.act_to_image(mid, s1).data, img.data)))
        row["us_fp32_pair"] = timeit(lambda: ops.modulated_conv2d(ops.modulated_conv2d(x, w0, s0, **k0), w1, s1, **k1))
        row["us_image_pair"] = timeit(lambda: ops.modulated_conv2d(ops.modulated_conv2d(x, w0, s0, next_styles=s1, **k0), w1, None, **k1))
        print(json.dumps(row))


if __name__ == "__main__":
    main()
