#!/usr/bin/env python3
"""A/B of the up-sampling layer's launch forms at the shapes of the backbone / super-resolution (batch 1, image in, image out):
round-5 form (k_modconv_up3 + FIR pass; P3D_UP4=0) against k_modconv_up4 with 8 / 16 grid rows per tile (P3D_UP4_RPW=0 / 2).
HIP-event time per call (us) and bit-equality of the image where the round-5 launch is unsplit.
    python tools/up4_ab.py [--n 30]"""
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
    shapes = [("sr.block1.conv0", 256, 128, 256), ("sr.block0.conv0", 32, 256, 128), ("b256.conv0", 256, 128, 128), ("b128.conv0", 512, 256, 64),
              ("b64.conv0", 512, 512, 32), ("b32.conv0", 512, 512, 16)]
    os.environ["P3D_UP4_MIN_WGS"] = "0"
    os.environ["P3D_UP4_MIN_I"] = "0"
    for name, I, O, H in shapes:
        x = torch.randn(1, I, H, H, device=dev)
        w = torch.randn(O, I, 3, 3, device=dev)
        s = torch.randn(1, I, device=dev) * 0.5 + 1.0
        s2 = torch.randn(1, O, device=dev) * 0.5 + 1.0
        b = torch.randn(O, device=dev)
        d = ((w[None] * s[:, None, :, None, None]).square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt().contiguous()
        wf = ops.conv_weights_to_f16(w, split=True)
        nz = torch.randn(2 * H, 2 * H, device=dev) * 0.1
        kw = dict(up=2, padding=1, resample_filter=f, demodulate=True, bias=b, act="lrelu", dcoef=d, noise=nz, weight_f16=wf)
        img = ops.act_to_image(x, s)
        row = dict(layer=name, I=I, O=O, H=H)
        outs = {}
        variants = [("r05", {"P3D_UP4": "0"}), ("up4_rpw2", {"P3D_UP4": "1", "P3D_UP4_RPW": "2"}), ("up4_rpw0", {"P3D_UP4": "1", "P3D_UP4_RPW": "0"})]
        for tag, env in variants:
            os.environ.update(env)
            call = lambda: ops.modulated_conv2d(img, w, None, next_styles=s2, **kw)
            outs[tag] = call().data.clone()
            row["us_" + tag] = round(timeit(call, n), 2)
        row["rpw2_equals_r05"] = bool(torch.equal(outs["r05"], outs["up4_rpw2"]))
        row["rpw0_equals_rpw2"] = bool(torch.equal(outs["up4_rpw0"], outs["up4_rpw2"]))
        row["max_abs_diff_vs_r05"] = float((outs["r05"].float() - outs["up4_rpw2"].float()).abs().max())
        flops = 2.0 * I * O * 9 * H * H
        row["TFLOP_s_fp32_equiv_best"] = round(flops / (min(row["us_up4_rpw2"], row["us_up4_rpw0"]) * 1e-6) / 1e12, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
