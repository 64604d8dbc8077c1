#!/usr/bin/env python3
"""Per-layer timing of the HIP modulated convolution at the generator's shapes (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import panic3d_amd as P
from panic3d_amd import ops
dev = "cuda"
f = ops.setup_filter([1, 3, 3, 1]).to(dev)
F16 = "--f16" in sys.argv  # f16 MFMA operands (fp32 accumulate)
X2 = "--f16x2" in sys.argv  # two-term f16 operands (fp32-class results)
def run(I, O, H, up, ks, N=1):
    x = torch.randn(N, I, H, H, device=dev); w = torch.randn(O, I, ks, ks, device=dev); s = torch.randn(N, I, device=dev)
    b = torch.randn(O, device=dev)
    wh = ops.conv_weights_to_f16(w, split=X2) if (F16 or X2) and I % 16 == 0 else None
    fn = lambda: ops.modulated_conv2d(x, w, s, up=up, padding=ks // 2, resample_filter=f, demodulate=ks == 3, bias=b, act="lrelu" if ks == 3 else "linear", weight_f16=wh)
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    fl = 2 * ks * ks * I * O * H * H * N
    print(f"I={I:4d} O={O:4d} H={H:4d} up={up} ks={ks} N={N}: {dt*1e3:7.3f} ms  {fl/dt/1e12:6.1f} TF  ({fl/1e9:.1f} GFLOP)")
for cfg in [(256, 3, 256, 1, 1), (128, 3, 512, 1, 1), (32, 256, 128, 2, 3), (512, 512, 4, 1, 3), (512, 512, 16, 1, 3), (512, 512, 32, 2, 3), (512, 512, 64, 1, 3), (512, 256, 64, 2, 3), (256, 256, 128, 1, 3),
            (256, 128, 128, 2, 3), (128, 128, 256, 1, 3), (128, 96, 256, 1, 1), (256, 256, 256, 1, 3), (256, 128, 256, 2, 3), (128, 128, 512, 1, 3)]:
    run(*cfg)
