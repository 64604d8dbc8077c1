#!/usr/bin/env python3
"""bench.py — rendered rays/sec of the fused triplane renderer on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: launched by torch.distributed.run)

Workload (config.workload): BASELINE config c3 with synthetic assets — one 512x512-ray perspective view per rank per
step, 48 coarse + 48 importance samples per ray (= 96 decoded samples/ray on the reference's path), synthetic
spatially-coherent triplanes [1,3,32,256,256] (fp32), random OSGDecoder, triplane_crop=0.1, cull_clouds=0.5, white_back.
A step = ImportanceRenderer.forward end to end on the HIP path: NCHW->NHWC plane transpose, the two random draws
(torch.rand on device, renderer.py:324,371), the fused render kernel, the global depth clamp.  Multi-GPU: every rank
renders its own views of the sweep (no data-path collective); when launched by torch.distributed.run the K frames of all
ranks are gathered to rank 0 with ONE RCCL collective at the end of the sweep, inside the timed region.  Inputs (planes, rays, decoder) are resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  `roofline` prices the fused kernel against the HBM roofline using ALGORITHMIC bytes
((Sc+Sf)*1536 + 172 bytes per ray — what the reference's algorithm touches per ray; the kernel's exact
early-outs skip decodes whose result provably cannot change any output bit, `roofline.decode_steps_executed_frac` says
how many it executed; --no-early-out measures without them); `cpu_baseline` times the CPU oracle (a port of the reference
algorithm, OpenMP over rays) on a bounded sample of the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def make_scene(dev, seed, res, azim):
    """Synthetic subject: smooth blobs (16x16 noise upsampled) + 10% white noise, scale 4; decoder with a strong sigma
    row and a negative sigma bias so that about half of the rays hit an opaque surface and the rest stay empty — the
    coverage of a character in front of a white background, like a trained model's planes."""
    import panic3d_amd as P
    g = torch.Generator().manual_seed(seed)
    low = torch.randn(3, 32, 16, 16, generator=g)
    planes = torch.nn.functional.interpolate(low, size=(256, 256), mode="bilinear", align_corners=False)
    planes = ((planes + 0.1 * torch.randn(3, 32, 256, 256, generator=g)) * 4.0).reshape(1, 3, 32, 256, 256).contiguous()
    w0 = torch.randn(64, 32, generator=g)
    b0 = torch.randn(64, generator=g) * 0.5
    w1 = torch.randn(33, 64, generator=g)
    b1 = torch.randn(33, generator=g) * 0.5
    w1[0] *= 30.0
    b1[0] = -45.0  # with the x30 sigma row: ~55 % of the rays hit a surface, the rest see empty space (character-like coverage)
    label = P.cameras.camera_label(0.0, azim, 1.0, 30.0)
    o, d = P.cameras.rays_from_label(label[None], res)
    return planes, (w0, b0, w1, b1), o, d


def cpu_baseline(planes, raw, o, d, ro, kw, res, budget_s=15.0):
    """Time the CPU oracle on a centred crop of the same view, sized for ~budget_s seconds."""
    from oracle import oracle
    oracle.build()
    opts = oracle.make_opts(ro, **kw)
    mlp = oracle.prescale_mlp(*[x.numpy() for x in raw])
    pl = planes.numpy()
    Sc, Sf = opts.Sc, opts.Sf
    o2 = o.reshape(res, res, 3).numpy()
    d2 = d.reshape(res, res, 3).numpy()

    def run(side):
        a = (res - side) // 2
        oo = np.ascontiguousarray(o2[a:a + side, a:a + side].reshape(1, -1, 3))
        dd = np.ascontiguousarray(d2[a:a + side, a:a + side].reshape(1, -1, 3))
        rng = np.random.default_rng(0)
        jit = rng.random((1, side * side, Sc), dtype=np.float32)
        u = rng.random((side * side, max(Sf, 1)), dtype=np.float32)
        t = time.perf_counter()
        oracle.render(pl, oo, dd, jit, u, mlp, opts)
        return time.perf_counter() - t

    t0 = run(32)
    rate = 32 * 32 / t0
    side = int(min(res, max(32, (rate * budget_s) ** 0.5))) // 8 * 8
    t1 = run(side)
    return dict(value=side * side / t1, unit="rays/s", cores=os.cpu_count(), kind="port",
                sample=f"centre {side}x{side} rays of the same view, {Sc}+{Sf} samples/ray, {t1:.1f} s, "
                       f"oracle/p3d_oracle.c with OpenMP on {os.cpu_count()} host threads")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--sc", type=int, default=48)
    ap.add_argument("--sf", type=int, default=48)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-early-out", action="store_true", help="decode every sample (disable the exact early-outs)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # torch.distributed.run (also with one rank)
    if launched:
        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
    import panic3d_amd as P
    from panic3d_amd import ops, sharding
    P._lib.lib()

    res, Sc, Sf = a.res, a.sc, a.sf
    R = res * res
    ro = dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=Sc, depth_resolution_importance=Sf,
              disparity_space_sampling=False, clamp_mode="softplus", white_back=True, use_triplane=1)
    kw = dict(triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True)
    # each rank renders its own view of the sweep (weak scaling: one 512^2 view per rank per step)
    planes_c, raw_c, o_c, d_c = make_scene(dev, 0, res, azim=20.0 + 360.0 * rank / max(world, 1))
    planes, o, d = planes_c.to(dev), o_c.to(dev), d_c.to(dev)
    mlp = ops.prescale_mlp(*(x.to(dev) for x in raw_c), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
    opts = ops.make_opts(ro, early_out=not a.no_early_out, **kw)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    frames = torch.empty((a.steps, res, res, 4), dtype=torch.float32, device=dev) if launched else None

    def step(i=None):
        nhwc = ops.planes_to_nhwc(planes)
        jit = torch.rand((1, R, Sc, 1), dtype=torch.float32, device=dev)
        u = torch.rand((R, Sf), dtype=torch.float32, device=dev) if Sf > 0 else None
        if i is not None:
            ev[i][0].record()
        feat, depth, wsum, xyz = ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res)
        if i is not None:
            ev[i][1].record()
        if launched and i is not None:  # keep this rank's final RGBA frame (channels-last) for the sweep's single gather
            sharding.frames_rgba(feat, wsum, res, out=frames[i:i + 1], channels_last=True)
        return wsum

    for _ in range(a.warmup):
        ws = step()
    if launched:  # warm the collective too (communicator + buffer registration happen on first use)
        sharding.gather_frames(frames, counts=[a.steps] * world, dst=0, force=True)
    torch.cuda.synchronize()
    if launched:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        ws = step(i)
    if launched:  # the path's only collective: ONE gather of the sweep's final RGBA frames to rank 0 (inside the timed region)
        gathered = sharding.gather_frames(frames, counts=[a.steps] * world, dst=0, force=True)
        if rank == 0:
            assert gathered.shape == (world * a.steps, res, res, 4)
    torch.cuda.synchronize()
    if launched:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if launched:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
    st = {}
    ops.render(ops.planes_to_nhwc(planes), o, d, torch.rand((1, R, Sc, 1), device=dev), torch.rand((R, max(Sf, 1)), device=dev) if Sf > 0 else None,
               mlp, opts, ray_tile_w=res, stats=st)  # untimed: decode-step statistics of one launch
    if rank == 0:
        rays = world * R * a.steps
        bytes_per_ray = (Sc + Sf) * 1536 + 172
        achieved = R * bytes_per_ray / (kern_ms * 1e-3) / 1e9
        traffic = None
        pj = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pj):
            try:
                traffic = json.load(open(pj)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "rendered rays/sec at 512^2 img x 96 samples/ray", "value": rays / dt, "unit": "rays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"c3: {res}x{res} rays/view, {Sc}+{Sf} samples/ray, one view per GPU per step, synthetic "
                                   "triplanes [1,3,32,256,256], random OSGDecoder, crop=0.1 cull=0.5 white_back, "
                                   "step = transpose + rand draws + fused render + depth clamp" +
                                   ("; after the K steps ONE RCCL gather of all ranks' RGBA frames to rank 0, inside the timed region" if launched else ""),
                       "rays_per_step_per_gpu": R, "samples_per_ray": Sc + Sf, "parallelism": f"views x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "kernel": "k_render (p3d_render_f32)", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": R * bytes_per_ray,
                         "decode_steps_executed_frac": st["decode_steps"] / st["decode_steps_full"],
                         "early_out": not a.no_early_out},
            "wsum_mean": float(ws.mean().item()), "hit_fraction": float((ws > 0.5).float().mean().item()),
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(planes_c, raw_c, o_c, d_c, ro, kw, res)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
