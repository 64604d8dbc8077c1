#!/usr/bin/env python3
"""Small-launch kernel (k_render_pair) at 128^2 rays: time and wave-level decode steps with the exact early-outs on / off, on the
two bench scenes -> per-step cost and the fixed (non-decode) part of a wave."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import panic3d_amd as P
from panic3d_amd import ops, cameras
import bench
res, dev = 128, "cuda"
for scene in ("canonical", "surface"):
    for S in (48, 96):
        w = bench.Workload(scene, torch.device(dev), res, 20.0, 7)
        ro = dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=S, depth_resolution_importance=S, white_back=True, use_triplane=1)
        nhwc = ops.planes_to_nhwc(w.planes)
        R = res * res
        jit = torch.rand((1, R, S, 1), device=dev); u = torch.rand((R, S), device=dev)
        row = dict(scene=scene, S=S)
        for name, kw in (("pair", {}), ("pair_every_sample", dict(early_out=False)), ("rows32", dict(small_launch_kernel=False)),
                         ("rows32_tolerance", dict(small_launch_kernel=False, fast_color=True))):
            opts = ops.make_opts(ro, triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True, **kw)
            st = {}
            for _ in range(3):
                ops.render(nhwc, w.o, w.d, jit, u, w.mlp, opts, ray_tile_w=res, stats=st)
            ts = []
            for _ in range(10):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); ops.render(nhwc, w.o, w.d, jit, u, w.mlp, opts, ray_tile_w=res); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            row[name + "_ms"] = round(float(np.median(ts)), 4)
            tiles = R // (16 if st.get("small_launch_kernel") else 32)
            row[name + "_steps_per_wave"] = round(st.get("decode_steps", 0) / tiles, 1)
        print(json.dumps(row))
