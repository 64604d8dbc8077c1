#!/usr/bin/env python3
"""Kernel time of the fused renderer for a few launch shapes under the current P3D_TILE_ORDER (read once by the library: run the script
once per order).  Surface scene of bench.py; exact and tolerance; HIP events, median of 15."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import panic3d_amd as P
from panic3d_amd import ops, cameras
import bench
dev = "cuda"
planes, raw, _, _ = bench.make_scene(dev, 0, 64, 20.0)
mlp = ops.prescale_mlp(*(x.to(dev) for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
nhwc = ops.planes_to_nhwc(planes.to(dev))
out = {"P3D_TILE_ORDER": os.environ.get("P3D_TILE_ORDER", "1 (default)")}
for res, N, S in ((256, 1, 48), (256, 4, 48), (384, 1, 48), (512, 1, 48), (512, 1, 96), (256, 1, 96), (1024, 1, 48)):
    ro = dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=S, depth_resolution_importance=S, white_back=True, use_triplane=1)
    lab = torch.stack([cameras.camera_label(0.0, 20.0 + 25.0 * i, 1.0, 30.0) for i in range(N)]).to(dev)
    o, d = cameras.rays_from_label(lab, res)
    R = res * res
    jit = torch.rand((N, R, S, 1), device=dev); u = torch.rand((N * R, S), device=dev)
    for fast in (False, True):
        opts = ops.make_opts(ro, triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True, fast_color=fast)
        for _ in range(3):
            ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res, per_view_clamp=N > 1)
        ts = []
        for _ in range(15):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res, per_view_clamp=N > 1); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        out[f"{res}^2 x{N} {S}+{S} {'tol' if fast else 'exact'}"] = round(float(np.median(ts)), 4)
print(json.dumps(out))
