#!/usr/bin/env python3
"""ImportanceRenderer.forward_staged (one stand-alone kernel per stage + torch glue; the path of density_noise > 0) next to the fused
kernel: 128^2 rays x (48+48) and (96+96), bench.py's surface scene.  One JSON line."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import panic3d_amd as P
from panic3d_amd import cameras
import bench
dev = "cuda"
planes, raw, _, _ = bench.make_scene(dev, 0, 64, 20.0)


class FC:
    def __init__(self, w, b, i):
        self.weight, self.bias, self.weight_gain, self.bias_gain = w.to(dev), b.to(dev), 1 / np.sqrt(i), 1.0


class Dec:
    force_sigmoid = True


dec = Dec()
dec.net = [FC(raw[0], raw[1], 32), None, FC(raw[2], raw[3], 64)]
rend = P.ImportanceRenderer(use_triplane=True)
res = 128
o, d = cameras.rays_from_label(cameras.camera_label(0.0, 20.0, 1.0, 30.0)[None].to(dev), res)
pl = planes.to(dev)
out = {}
for S in (48, 96):
    ro = dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=S, depth_resolution_importance=S, white_back=True, use_triplane=1)
    for name, fn in (("fused_exact", lambda: rend(pl, dec, o, d, ro, triplane_crop=0.1, cull_clouds=0.5, exact=True)),
                     ("fused_tolerance", lambda: rend(pl, dec, o, d, ro, triplane_crop=0.1, cull_clouds=0.5)),
                     ("staged", lambda: rend.forward_staged(pl, dec, o, d, ro, triplane_crop=0.1, cull_clouds=0.5)),
                     ("staged_density_noise", lambda: rend(pl, dec, o, d, dict(ro, density_noise=0.5), triplane_crop=0.1, cull_clouds=0.5))):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        out[f"{name}_{S}p{S}_ms"] = (time.perf_counter() - t) / 10 * 1e3
print(json.dumps(dict(res=res, **out)))
