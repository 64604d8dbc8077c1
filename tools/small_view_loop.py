#!/usr/bin/env python3
"""The pipeline's own renderer launch in a loop — 128^2 rays x (96+96), surface scene, tolerance mode of the final pass unless
--exact, the kernel the host picks (k_render_quad) unless --kernel pair|classic — for rocprofv3 passes (tools/pmc_small_view.sh)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import panic3d_amd as P
from panic3d_amd import ops, cameras
import bench
a = sys.argv[1:]
res = 128
S = 96
n = int(a[a.index("--n") + 1]) if "--n" in a else 20
kern = {"quad": "quad", "pair": "pair", "classic": False}[a[a.index("--kernel") + 1]] if "--kernel" in a else True
fast = "--exact" not in a
dev = "cuda"
planes, raw, _, _ = bench.make_scene(dev, 0, 64, 20.0)
mlp = ops.prescale_mlp(*(x.to(dev) for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
ro = dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=S, depth_resolution_importance=S, white_back=True, use_triplane=1)
nhwc = ops.planes_to_nhwc(planes.to(dev))
o, d = cameras.rays_from_label(cameras.camera_label(0.0, 20.0, 1.0, 30.0)[None].to(dev), res)
R = res * res
jit = torch.rand((1, R, S, 1), device=dev); u = torch.rand((R, S), device=dev)
opts = ops.make_opts(ro, triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True, small_launch_kernel=kern, fast_color=fast)
st = {}
r = ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res, stats=st)
for _ in range(n):
    r = ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res)
torch.cuda.synchronize()
print(json.dumps({"res": res, "samples": [S, S], "fast": fast, "kernel": st.get("small_launch_kind"), "decode_steps": st["decode_steps"], "decode_steps_full": st["decode_steps_full"],
                  "hit_fraction": float((r[2] > 0.5).float().mean()), "n": n}))
