#!/usr/bin/env python3
"""BASELINE config c5 on one GPU: density-only query of an N^3 grid (get_eg3d_volume's hot loop) on synthetic planes."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import panic3d_amd as P
from panic3d_amd import ops, volume
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = "cuda"
planes, raw, o, d = bench.make_scene(dev, 0, 64, 20.0)
mlp = ops.prescale_mlp(*(x.to(dev) for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
opts = ops.make_opts(dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=48, use_triplane=1), force_sigmoid=True)
nhwc = ops.planes_to_nhwc(planes.to(dev))
def run():
    out = torch.empty((1, N ** 3, 1), device=dev)
    step = 1 << 25
    for a in range(0, N ** 3, step):
        b = min(a + step, N ** 3)
        pts, _, _ = volume.create_samples(N, cube_length=0.7, device=dev, lo=a, hi=b)
        out[:, a:b], _ = ops.triplane_decode(nhwc, pts.contiguous(), mlp, opts, density_only=True)
    return out
run(); torch.cuda.synchronize(); t = time.perf_counter(); out = run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(json.dumps({"config": "c5", "grid": N, "points": N ** 3, "seconds": dt, "Gpoints_per_s": N ** 3 / dt / 1e9, "sigma_mean": float(out.mean())}))
