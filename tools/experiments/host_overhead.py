#!/usr/bin/env python3
"""Host-side cost per operator call (issue rate with the GPU far behind or far ahead): tiny shapes, many calls, one sync."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import panic3d_amd as P
ops = P.ops
dev = torch.device("cuda")
torch.manual_seed(0)


def rate(fn, n=2000):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    return round(t_issue / n * 1e6, 1), round((time.perf_counter() - t) / n * 1e6, 1)


x = torch.randn(1, 64, 4, 4, device=dev); w = torch.randn(64, 64, 3, 3, device=dev); s = torch.randn(1, 64, device=dev)
b = torch.randn(64, device=dev); wf = ops.conv_weights_to_f16(w, split=True)
d = torch.ones(1, 64, device=dev); nz = torch.randn(4, 4, device=dev)
f = ops.setup_filter((1, 3, 3, 1)).to(dev)
wr = torch.randn(96, 64, 1, 1, device=dev); wt = ops.torgb_weights(wr)
out = {}
out["torch_add (us issue, us total)"] = rate(lambda: x + 1)
out["torch_empty"] = rate(lambda: torch.empty((1, 64, 4, 4), device=dev))
out["modulated_conv2d 3x3"] = rate(lambda: ops.modulated_conv2d(x, w, s, noise=nz, padding=1, bias=b, act="lrelu", weight_f16=wf, dcoef=d))
out["modulated_conv2d up"] = rate(lambda: ops.modulated_conv2d(x, w, s, up=2, padding=1, resample_filter=f, bias=b, act="lrelu", weight_f16=wf, dcoef=d))
out["torgb"] = rate(lambda: ops.torgb(x, wt, 96, s, bias=torch.zeros(96, device=dev)))
print(json.dumps(out, indent=1))
