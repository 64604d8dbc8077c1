#!/usr/bin/env python3
"""generate.py:97-103 on one GPU, volume resident in HBM: N^3 density grid (BASELINE c5) -> iso-surface on the device.
    python tools/bench_mesh.py [N=512]
Prints one JSON line: ms per stage, mesh size, and the extractor's HBM rate against its algorithmic bytes (20 B / grid point)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
vs, org = 0.7 / (N - 1), -0.35


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return r, float(np.median(ts))


sig, t_grid = timed(lambda: ops.grid_density(nhwc, N, 0, N ** 3, vs, (org, org, org), mlp, opts))
(_, msk), t_grid_crop = timed(lambda: ops.grid_density(nhwc, N, 0, N ** 3, vs, (org, org, org), mlp, opts, crop_limit=0.25, skip_cropped=True))
(_, _), t_grid_crop_noskip = timed(lambda: ops.grid_density(nhwc, N, 0, N ** 3, vs, (org, org, org), mlp, opts, crop_limit=0.25))
dens, t_act = timed(lambda: ops.sigma2density(sig))
vol = dens.reshape(N, N, N)
level = 0.5
(v, f, nr, va), t_mc = timed(lambda: ops.marching_cubes(vol, level, flip0=True))
t0 = time.perf_counter(); host = {k: x.cpu().numpy() for k, x in dict(verts=v, faces=f, normals=nr, values=va).items()}; t_d2h = (time.perf_counter() - t0) * 1e3
print(json.dumps({"grid": N, "points": N ** 3, "grid_density_ms": t_grid, "grid_density_crop0.1_skip_ms": t_grid_crop, "grid_density_crop0.1_noskip_ms": t_grid_crop_noskip,
                  "cropped_fraction": float(msk.float().mean()), "sigma2density_ms": t_act, "marching_cubes_ms": t_mc,
                  "mesh_d2h_ms": t_d2h, "verts": len(v), "faces": len(f), "mesh_MB": sum(x.nbytes for x in host.values()) / 1e6,
                  "mc_GBps_algorithmic": 20.0 * N ** 3 / (t_mc * 1e-3) / 1e9, "volume_MB_kept_on_device": 4 * N ** 3 / 1e6}))
