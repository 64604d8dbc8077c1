import os, sys, ctypes
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import panic3d_amd as P, p3d_testing as T
from panic3d_amd import ops
L = P._lib.lib()
L.p3d_phase_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
dev = torch.device("cuda"); res, Sc, Sf = 512, 48, 48; R = res * res
ro = T.bench_rendering_kwargs(Sc, Sf)
for scene in ("canonical", "surface"):
    planes_np, raw = T.make_bench_scene(scene)
    nhwc = ops.planes_to_nhwc(torch.from_numpy(planes_np).to(dev))
    mlp = ops.prescale_mlp(*(torch.from_numpy(x).to(dev) for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
    o, d = P.cameras.rays_from_label(P.cameras.camera_label(0.0, 20.0, 1.0, 30.0)[None], res); o, d = o.to(dev), d.to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    jit = torch.rand((1, R, Sc, 1), device=dev, generator=g); u = torch.rand((R, Sf), device=dev, generator=g)
    for fast in (False, True):
        for early in (True, False):
            opts = ops.make_opts(ro, early_out=early, fast_color=fast, **T.BENCH_KW)
            for _ in range(2): ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res)
            buf = (ctypes.c_ulonglong * 17)(); L.p3d_phase_read(buf, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res); e1.record(); torch.cuda.synchronize()
            L.p3d_phase_read(buf, 1)
            b = list(buf); ms = e0.elapsed_time(e1)
            tot = b[16]
            print(f"{scene} fast={fast} early={early}: {ms:.2f} ms | wave lifetime ticks {tot/ max(b[7],1):.0f} per wave ({b[7]} waves) | "
                  f"coarse: gather {b[0]/tot:.3f} mlp {b[1]/tot:.3f} ({b[4]} steps, {b[0]/max(b[4],1):.0f}+{b[1]/max(b[4],1):.0f} ticks/step) | "
                  f"final: gather {b[2]/tot:.3f} mlp {b[3]/tot:.3f} ({b[5]} steps, {b[2]/max(b[5],1):.0f}+{b[3]/max(b[5],1):.0f} ticks/step) | other {1-(b[0]+b[1]+b[2]+b[3])/tot:.3f}"
                  f" || sections: weights->LDS+sync {b[8]/tot:.3f} stratified {b[9]/tot:.3f} coarse loop {b[10]/tot:.3f} cdf {b[11]/tot:.3f} draws+sort {b[12]/tot:.3f} final loop {b[13]/tot:.3f} [merge pre-pass {b[6]/tot:.3f} select/skip {b[14]/tot:.3f} march+composite {b[15]/tot:.3f}]")
