#!/usr/bin/env python3
"""BASELINE config c5: marching-cubes density grid N^3 (default 512^3 = 134 M points), the fused density-only query with the
grid's slowest axis split into one contiguous slab per GPU, ONE RCCL gather of the sigma slabs to rank 0, iso-surface on rank 0.

    python tools/bench_c5.py [--grid 512]                                                        # one GPU
    python tools/bench_c5.py --gpus 8                                                            # 8 GPUs: starts its own 8 ranks (or run it under torch.distributed.run)
Prints one JSON line on rank 0."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--crop", type=float, default=None, help="triplane_crop (generate.py uses 0.1): masked points are not decoded")
ap.add_argument("--fast", action="store_true", help="tolerance-mode density decoder + LDS-staged texel boxes (opt-in)")
ap.add_argument("--check", action="store_true", help="print a sha256 of the gathered sigma grid (must not depend on the world size)")
ap.add_argument("--gpus", type=int, default=None, help="N > 1 without a launcher: start the N ranks (one per GPU) under torch.distributed.run")
a = ap.parse_args()
from panic3d_amd import sharding as _sh
_sh.ensure_ranks(a.gpus)  # re-executes under torch.distributed.run when needed; under a launcher WORLD_SIZE must match --gpus
rank, world, lrank = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
if a.gpus is not None and torch.cuda.device_count() < a.gpus:
    raise SystemExit(f"--gpus {a.gpus} but only {torch.cuda.device_count()} device(s) visible")
torch.cuda.set_device(lrank)
dev = torch.device("cuda", lrank)
if "RANK" in os.environ:
    dist.init_process_group("nccl", device_id=dev)
import panic3d_amd as P
from panic3d_amd import ops, sharding, volume
import bench

N = a.grid
planes, raw, _, _ = bench.make_scene(dev, 0, 64, 20.0)  # same seed on every rank -> identical planes
mlp = ops.prescale_mlp(*(x.to(dev) for x in raw), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
opts = ops.make_opts(dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=48, use_triplane=1), force_sigmoid=True)
nhwc = ops.planes_to_nhwc(planes.to(dev))
vs, org = 0.7 / (N - 1), -0.35
lo, hi = sharding.partition(N, world, rank)  # slab of the slowest grid axis
counts = [sharding.partition(N, world, r)[1] - sharding.partition(N, world, r)[0] for r in range(world)]


LAST = {}


def run():
    if a.crop is None:
        sig, msk = ops.grid_density(nhwc, N, lo * N * N, hi * N * N, vs, (org, org, org), mlp, opts, fast=a.fast), None
    else:
        sig, msk = ops.grid_density(nhwc, N, lo * N * N, hi * N * N, vs, (org, org, org), mlp, opts, crop_limit=0.35 - a.crop, skip_cropped=True, fast=a.fast)
    full = sharding.gather_frames(sig.reshape(hi - lo, N * N), counts, 0)
    fmsk = None if msk is None else sharding.gather_frames(msk.view(torch.uint8).reshape(hi - lo, N * N), counts, 0)
    if rank != 0:
        return None
    LAST["sigma"] = full
    dens = ops.sigma2density(full.reshape(N, N, N), None if fmsk is None else fmsk.reshape(N, N, N), None)
    return ops.marching_cubes(dens, 0.5, flip0=True)


def sync():
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    return time.perf_counter()


run()
t0 = sync()
K = 3
for _ in range(K):
    out = run()
t1 = sync()
if rank == 0:
    dt = (t1 - t0) / K
    res = {"config": "c5", "grid": N, "points": N ** 3, "n_gpus": world, "triplane_crop": a.crop, "tolerance_mode": a.fast, "seconds": dt, "Gpoints_per_s": N ** 3 / dt / 1e9,
           "verts": int(out[0].shape[0]), "faces": int(out[1].shape[0])}
    if a.check:
        import hashlib
        res["sha256"] = hashlib.sha256(LAST["sigma"].cpu().numpy().tobytes()).hexdigest()
    print(json.dumps(res))
if dist.is_initialized():
    dist.destroy_process_group()
