import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import panic3d_amd as P, p3d_testing as T
from panic3d_amd import ops
N = 512
planes_np, raw = T.make_bench_scene("surface")
dev = "cuda"
mlp = ops.prescale_mlp(*(torch.from_numpy(x).to(dev) for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
opts = ops.make_opts(dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=48, use_triplane=1), force_sigmoid=True)
nhwc = ops.planes_to_nhwc(torch.from_numpy(planes_np).to(dev))
vs, org = 0.7 / (N - 1), -0.35
for staged, fast in ((False, False), (True, False), (False, True), (True, True)):
    f = lambda: ops.grid_density(nhwc, N, 0, N ** 3, vs, (org, org, org), mlp, opts, staged=staged, fast=fast)
    for _ in range(2): s = f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): s = f()
    torch.cuda.synchronize(); print("staged" if staged else "direct", "tolerance" if fast else "exact", (time.perf_counter() - t) / 5 * 1e3, "ms", float(s.double().mean()))
