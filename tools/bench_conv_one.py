#!/usr/bin/env python3
"""One modulated-conv shape, a few launches (for rocprofv3 --pmc passes):  bench_conv_one.py I O H up ks [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panic3d_amd import ops
MODE = next((a[2:] for a in sys.argv if a in ("--f16", "--f16x2")), None)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
I, O, H, up, ks = (int(a) for a in args[:5]); N = int(args[5]) if len(args) > 5 else 1
dev = "cuda"
f = ops.setup_filter([1, 3, 3, 1]).to(dev)
x = torch.randn(N, I, H, H, device=dev); w = torch.randn(O, I, ks, ks, device=dev); s = torch.randn(N, I, device=dev); b = torch.randn(O, device=dev)
wh = ops.conv_weights_to_f16(w, split=MODE == "f16x2") if MODE else None
for _ in range(5):
    ops.modulated_conv2d(x, w, s, up=up, padding=ks // 2, resample_filter=f, demodulate=ks == 3, bias=b, act="lrelu", weight_f16=wh)
torch.cuda.synchronize()
