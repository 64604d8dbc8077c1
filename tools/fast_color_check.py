#!/usr/bin/env python3
"""P3D_FLAG_FAST_COLOR vs the exact contract on the GPU: error statistics and kernel time.
    python tools/fast_color_check.py [--res 512]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import panic3d_amd as P
import p3d_testing as T
from panic3d_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--sc", type=int, default=48)
ap.add_argument("--sf", type=int, default=48)
ap.add_argument("--seed", type=int, default=5)
ap.add_argument("--no-timing", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda")
res, Sc, Sf = a.res, a.sc, a.sf
R = res * res
ro = T.bench_rendering_kwargs(Sc, Sf)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = {}
for scene in ("canonical", "surface"):
    planes_np, raw = T.make_bench_scene(scene)
    nhwc = ops.planes_to_nhwc(torch.from_numpy(planes_np).to(dev))
    mlp = ops.prescale_mlp(*(torch.from_numpy(x).to(dev) for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
    o, d = P.cameras.rays_from_label(P.cameras.camera_label(0.0, 20.0, 1.0, 30.0)[None], res)
    o, d = o.to(dev), d.to(dev)
    g = torch.Generator(device=dev).manual_seed(a.seed)
    jit = torch.rand((1, R, Sc, 1), device=dev, generator=g)
    u = torch.rand((R, Sf), device=dev, generator=g)
    rec = {}
    for early in (True, False):
        ex = ops.make_opts(ro, early_out=early, **T.BENCH_KW)
        fa = ops.make_opts(ro, early_out=early, fast_color=True, **T.BENCH_KW)
        if a.no_timing:
            continue
        t_ex = timeit(lambda: ops.render(nhwc, o, d, jit, u, mlp, ex, ray_tile_w=res))
        t_fa = timeit(lambda: ops.render(nhwc, o, d, jit, u, mlp, fa, ray_tile_w=res))
        rec["ms_exact" + ("" if early else "_no_early_out")] = t_ex
        rec["ms_fast" + ("" if early else "_no_early_out")] = t_fa
    A = ops.render(nhwc, o, d, jit, u, mlp, ex, ray_tile_w=res)
    B = ops.render(nhwc, o, d, jit, u, mlp, fa, ray_tile_w=res)
    for name, x, y in zip(("feat", "depth", "wsum", "xyz"), A, B):
        df = (x - y).abs()
        df = torch.where(torch.isfinite(df), df, torch.zeros_like(df))
        per_ray = df.reshape(R, -1).max(dim=1).values
        rec[name] = {"max_abs": float(df.max()), "mean_abs": float(df.mean()),
                     "rays_over_2e-5": int((per_ray > 2e-5).sum()), "rays_over_1e-4": int((per_ray > 1e-4).sum())}
    img_a, img_b = A[0][..., :3] * 0.5 + 0.5, B[0][..., :3] * 0.5 + 0.5
    mse = float(((img_a - img_b) ** 2).mean())
    rec["psnr_fast_vs_exact_db"] = float("inf") if mse == 0 else 10 * np.log10(1 / mse)
    out[scene] = rec
    print(scene, json.dumps(rec))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"fast_color_check_{Sc}p{Sf}_seed{a.seed}.json"), "w"), indent=1)
