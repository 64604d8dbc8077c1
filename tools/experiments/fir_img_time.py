#!/usr/bin/env python3
"""Time of the up-sampling layer with an image output (k_modconv_up_h + k_fir4x4_img) vs an fp32 output (k_fir4x4_tiled), two shapes;
P3D_LIB selects a library variant."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import panic3d_amd as P
ops = P.ops
dev = torch.device("cuda")
torch.manual_seed(0)
f = ops.setup_filter((1, 3, 3, 1)).to(dev)


def t(fn, n=40):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return round(sorted(ts)[2], 1)


out = {"lib": os.path.basename(os.environ.get("P3D_LIB", "shipped"))}
for (I, O, H) in ((256, 128, 256), (32, 256, 128), (512, 256, 64)):
    x = torch.randn(1, I, H, H, device=dev); w = torch.randn(O, I, 3, 3, device=dev); s = torch.randn(1, I, device=dev) * 0.5 + 1
    s1 = torch.randn(1, O, device=dev); b = torch.randn(O, device=dev); d = torch.ones(1, O, device=dev)
    wf = ops.conv_weights_to_f16(w, split=True)
    kw = dict(up=2, padding=1, resample_filter=f, demodulate=True, bias=b, act="lrelu", dcoef=d, weight_f16=wf)
    out[f"{I}->{O}@{H}"] = {"fp32_out_us": t(lambda: ops.modulated_conv2d(x, w, s, **kw)), "image_out_us": t(lambda: ops.modulated_conv2d(x, w, s, next_styles=s1, **kw))}
print(json.dumps(out))
