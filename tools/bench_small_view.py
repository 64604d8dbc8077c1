#!/usr/bin/env python3
"""One small view (default 128^2 rays, 96+96 samples: what generate.py renders per view) on the three render kernels:
the small-launch kernels (8 rays x 4 samples and 16 rays x 2 samples per wave) and the 32-rays-per-wave kernel."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import panic3d_amd as P
from panic3d_amd import ops, cameras
import bench
res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = int(sys.argv[2]) if len(sys.argv) > 2 else 96
dev = "cuda"
planes, raw, _, _ = bench.make_scene(dev, 0, 64, 20.0)
mlp = ops.prescale_mlp(*(x.to(dev) for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
ro = dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=S, depth_resolution_importance=S, white_back=True, use_triplane=1)
nhwc = ops.planes_to_nhwc(planes.to(dev))
o, d = cameras.rays_from_label(cameras.camera_label(0.0, 20.0, 1.0, 30.0)[None].to(dev), res)
R = res * res
jit = torch.rand((1, R, S, 1), device=dev); u = torch.rand((R, S), device=dev)
out = {}
ref = None
for name, pair, fast in (("quad_8rays_x4samples", "quad", False), ("pair_16rays_x2samples", "pair", False), ("classic_32rays", False, False),
                         ("quad_tolerance", "quad", True), ("pair_tolerance", "pair", True), ("classic_32rays_tolerance", False, True), ("default_choice", True, False), ("default_choice_tolerance", True, True)):
    opts = ops.make_opts(ro, triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True, small_launch_kernel=pair, fast_color=fast)
    for _ in range(3):
        r = ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res)
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    out[name + "_ms"] = float(np.median(ts))
    if ref is None:
        ref = r
    elif not fast:
        out["identical"] = all(torch.equal(x, y) for x, y in zip(ref, r))
    else:
        out[name + "_max_abs_vs_exact"] = max(float((x - y).abs().max()) for x, y in zip(ref, r))
print(json.dumps(dict(res=res, samples=[S, S], **out)))
