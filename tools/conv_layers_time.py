#!/usr/bin/env python3
"""Per-layer time of the two-term convolution kernels at the shapes where the backbone / super-resolution time is (round 4).
    python tools/conv_layers_time.py [--n 30]            HIP-event time per operator call (us), batch 1
    rocprofv3 --kernel-trace --stats -d out -- python tools/conv_layers_time.py --n 10      per-kernel averages
conv1 layers are fed an activation image (what the blocks do); up layers write one."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import panic3d_amd as P
ops = P.ops


def timeit(fn, n, w=5):
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
    n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 30
    torch.manual_seed(0)
    dev = torch.device("cuda")
    f = ops.setup_filter((1, 3, 3, 1)).to(dev)
    rows = []
    # (kind, I, O, H_in): plain = 3x3 up 1 image-fed; up = transposed + FIR writing an image for a consumer
    shapes = [("plain", 256, 256, 256), ("plain", 128, 128, 512), ("up", 256, 128, 256), ("up", 32, 256, 128),
              ("plain", 512, 512, 64), ("up", 512, 256, 64), ("plain", 256, 256, 128), ("up", 256, 128, 128), ("plain", 128, 128, 256),
              ("up", 512, 512, 32), ("plain", 512, 512, 32)]
    for kind, I, O, H in shapes:
        N = 1
        x = torch.randn(N, I, H, H, device=dev)
        w = torch.randn(O, I, 3, 3, device=dev)
        s = torch.randn(N, I, device=dev) * 0.5 + 1.0
        s2 = torch.randn(N, O, device=dev) * 0.5 + 1.0
        b = torch.randn(O, device=dev)
        d = ((w[None] * s[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt().contiguous()
        wf = ops.conv_weights_to_f16(w, split=True) if I % 16 == 0 else None
        up = 2 if kind == "up" else 1
        nz = torch.randn(up * H, up * H, device=dev) * 0.1
        kw = dict(up=up, padding=1, resample_filter=f, demodulate=True, bias=b, act="lrelu", dcoef=d, noise=nz, weight_f16=wf)
        if kind == "plain":
            img = ops.act_to_image(x, s) if (wf is not None and H >= 32) else None
            ref = ops.modulated_conv2d(x, w, s, **kw)
            if img is not None:
                got = ops.modulated_conv2d(img, w, None, **kw)
                same = bool(torch.equal(got, ref))
                us = timeit(lambda: ops.modulated_conv2d(img, w, None, **kw), n)
            else:
                same, us = None, timeit(lambda: ops.modulated_conv2d(x, w, s, **kw), n)
            us32 = timeit(lambda: ops.modulated_conv2d(x, w, s, **kw), n)
            flops = 2.0 * I * O * 9 * H * H
            rows.append(dict(kind=kind, I=I, O=O, H=H, us_image_in=us, us_fp32_in=us32, tflops_fp32_equiv=flops / (us * 1e-6) / 1e12, image_equals_fp32=same))
        else:
            us = timeit(lambda: ops.modulated_conv2d(x, w, s, next_styles=s2, **kw), n) if O % 8 == 0 and wf is not None else None
            us32 = timeit(lambda: ops.modulated_conv2d(x, w, s, **kw), n)
            flops = 2.0 * I * O * 9 * H * H
            rows.append(dict(kind=kind, I=I, O=O, H=H, us_image_out=us, us_fp32_out=us32, tflops_fp32_equiv=flops / ((us or us32) * 1e-6) / 1e12))
        # accuracy against a float64 evaluation of the plain conv (small sample): the two-term kernel stays fp32-class
        if kind == "plain" and H <= 64:
            xs = (x * s[:, :, None, None]).double()
            r64 = torch.nn.functional.conv2d(xs, w.double(), padding=1) * d[:, :, None, None].double() + nz[None, None].double() + b[None, :, None, None].double()
            r64 = torch.nn.functional.leaky_relu(r64, 0.2) * (2 ** 0.5)
            rows[-1]["max_rel_err_vs_f64"] = float(((ref.double() - r64).abs().max() / r64.abs().max()).item())
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
