import os, sys, ctypes, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import panic3d_amd as P
from panic3d_amd import ops
import bench
L = P._lib.lib()
L.p3d_pair_phase_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
res, dev = 128, "cuda"
names = ["setup+weights", "stratified", "coarse loop", "pdf/cdf", "draws+invcdf+sort", "merge pre-pass", "final loop", "outputs"]
for scene in ("canonical", "surface"):
    for S in (48, 96):
        w = bench.Workload(scene, torch.device(dev), res, 20.0, 7)
        ro = dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=S, depth_resolution_importance=S, white_back=True, use_triplane=1)
        nhwc = ops.planes_to_nhwc(w.planes)
        R = res * res
        jit = torch.rand((1, R, S, 1), device=dev); u = torch.rand((R, S), device=dev)
        opts = ops.make_opts(ro, triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True)
        for _ in range(3):
            ops.render(nhwc, w.o, w.d, jit, u, w.mlp, opts, ray_tile_w=res)
        buf = (ctypes.c_ulonglong * 16)(); L.p3d_pair_phase_read(buf, 1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.render(nhwc, w.o, w.d, jit, u, w.mlp, opts, ray_tile_w=res); b.record(); torch.cuda.synchronize()
        L.p3d_pair_phase_read(buf, 1)
        v = list(buf); waves = max(v[8], 1)
        print(scene, S, f"{a.elapsed_time(b):.3f} ms; per wave (us at 100 MHz ticks):", {n: round(v[i] / waves / 100.0, 1) for i, n in enumerate(names)}, "waves", v[8])
