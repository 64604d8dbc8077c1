#!/usr/bin/env python3
"""One-configuration backbone timing for rocprofv3 --kernel-trace --stats (development aid): N synthesis passes at batch B.
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bb -o r -- python tools/profile_backbone.py [--batch 1] [--passes 10]"""
import argparse, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import panic3d_amd as P
P.stylegan2.STYLE_MEMO = False  # a pass includes its style computation (a new subject per pass)
from panic3d_amd import stylegan2 as sg
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--passes", type=int, default=10)
ap.add_argument("--sr", action="store_true", help="profile the 128^2 -> 512^2 super-resolution blocks (SuperresolutionHybrid8XDC) instead of the backbone")
a = ap.parse_args()
torch.manual_seed(0)
dev = "cuda"
G = sg.Generator(z_dim=512, c_dim=25, w_dim=512, img_resolution=256, img_channels=96, cond_mode="none",
                 mapping_kwargs={"num_layers": 2}, channel_base=32768, channel_max=512, num_fp16_res=0, conv_clamp=None).to(dev).eval()
if a.sr:
    from panic3d_amd import generator as gen
    sr = gen.SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True, channels_hidden=256).to(dev).eval()
    x = torch.randn(a.batch, 32, 128, 128, device=dev); rgb = x[:, :3].contiguous(); wsr = torch.randn(a.batch, 14, 512, device=dev)
    with torch.no_grad():
        for _ in range(3):
            sr(rgb, x, wsr, noise_mode="none")
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(a.passes):
            out = sr(rgb, x, wsr, noise_mode="none")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / a.passes
    print(json.dumps({"what": "superresolution", "batch": a.batch, "ms_per_pass": dt * 1e3, "passes_incl_warmup": a.passes + 3, "checksum": float(out.double().sum())}))
    sys.exit(0)
with torch.no_grad():
    ws = G.mapping(torch.randn(a.batch, 512, device=dev), torch.zeros(a.batch, 25, device=dev), {})
    for _ in range(3):
        G.synthesis(ws, {}, noise_mode="const")
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.passes):
        out = G.synthesis(ws, {}, noise_mode="const")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / a.passes
print(json.dumps({"batch": a.batch, "ms_per_pass": dt * 1e3, "passes_incl_warmup": a.passes + 3, "checksum": float(out.double().sum())}))
