#!/usr/bin/env python3
"""Phase timing of one TriPlaneGenerator.f call at the released model's sizes (the generator of tools/generate_subject.py):
mapping, backbone, ray generation, fused renderer, super-resolution.  Development aid."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
sys.argv = ["x"]
src = open(os.path.join(ROOT, "tools", "generate_subject.py")).read()
head = src[:src.index("def sync():")]
exec(head)
import panic3d_amd as P
from panic3d_amd import cameras
def sync(): torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    x0 = {"elevations": torch.zeros(1, device=dev), "azimuths": torch.zeros(1, device=dev), "cond": cond, "seeds": [0], "noise_mode": "const", "triplane_crop": 0.1, "cull_clouds": 0.5}
    for _ in range(3): G.f(dict(x0))
    T = {}
    def timed(name, fn, n=10):
        fn(); t = sync()
        for _ in range(n): r = fn()
        T[name] = (sync() - t) / n * 1e3
        return r
    xw = dict(x0); G.f(xw); ws = xw["ws"]
    timed("f_total", lambda: G.f(dict(x0)))
    timed("f_with_ws", lambda: G.f(dict(x0, ws=ws)))
    zs = xw["zs"]; cp = xw["camera_params"]
    timed("mapping_zplus", lambda: G.mapping_zplus(zs, cp, cond))
    planes = timed("planes", lambda: G._planes(ws, cond, noise_mode="const"))
    timed("camera_label+rays", lambda: cameras.perspective_rays(torch.stack([cameras.camera_label(0.0, 0.0, 1.0, 30.0)]).to(dev)[:, :16].view(-1, 4, 4), torch.stack([cameras.camera_label(0.0, 0.0, 1.0, 30.0)]).to(dev)[:, 16:25].view(-1, 3, 3), 128))
    fr = xw["force_rays"]
    ro = fr["ray_origins"].permute(0, 2, 3, 1).reshape(1, -1, 3).contiguous(); rd = fr["ray_directions"].permute(0, 2, 3, 1).reshape(1, -1, 3).contiguous()
    out = timed("renderer", lambda: G.renderer(planes, G.decoder, ro, rd, G.rendering_kwargs, triplane_crop=0.1, cull_clouds=0.5))
    feat = out[0].permute(0, 2, 1).reshape(1, 32, 128, 128).contiguous()
    timed("superres", lambda: G.superresolution(feat[:, :3].contiguous(), feat, ws, noise_mode="none"))
print(json.dumps({k: round(v, 3) for k, v in T.items()}))
