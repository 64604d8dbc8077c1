import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from panic3d_amd import ops
dev = "cuda"
f = ops.setup_filter([1, 3, 3, 1]).to(dev)
out = []
for (I, O, H, up) in [(256, 256, 256, 1), (128, 128, 512, 1), (256, 128, 256, 2), (32, 256, 128, 2), (512, 512, 64, 1), (512, 256, 64, 2)]:
    x = torch.randn(1, I, H, H, device=dev); w = torch.randn(O, I, 3, 3, device=dev); s = torch.randn(1, I, device=dev); b = torch.randn(O, device=dev)
    wh = ops.conv_weights_to_f16(w, split=True)
    fn = lambda: ops.modulated_conv2d(x, w, s, up=up, padding=1, resample_filter=f, demodulate=True, bias=b, act="lrelu", weight_f16=wh)
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, c in ev:
        a.record(); fn(); c.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(c) for a, c in ev)[len(ev) // 2]
    out.append(f"{I}->{O}@{H}{'up' if up == 2 else ''}: {ms * 1e3:.0f}us")
print(os.environ.get("P3D_LIB", "shipped").split("/")[-1], " | ".join(out))
