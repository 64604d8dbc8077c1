#!/usr/bin/env python3
"""BASELINE config c4: 360-degree sweep, 120 views x 512x512 rays of one subject, views sharded over the GPUs of a node
(one process per GPU), one RCCL gather of the final RGBA frames to rank 0.

    python tools/sweep360.py [--views 120] [--res 512]                       # one GPU
    python tools/sweep360.py --gpus 8                                        # 8 GPUs: starts its own 8 ranks (or run it under torch.distributed.run)

The subject's planes come from a random-init StyleGAN2-256 backbone run ONCE per rank on the HIP synthesis path (same
seed on every rank -> identical planes; cheaper than broadcasting 25 MB); every view is one fused-renderer launch.
Prints one JSON line on rank 0."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=120)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--exact", action="store_true", help="exact-contract final pass (default: the renderer's default, tolerance mode)")
ap.add_argument("--batch", type=int, default=1, help="views per launch (planes shared by the batch: P3D_FLAG_SHARED_PLANES)")
ap.add_argument("--check", action="store_true", help="print a sha256 of the gathered [views,4,res,res] tensor: every view draws from "
                "its own seeded generator (the reference seeds each view: _train/eg3dc/util/eg3dc_v0.py:72), so the hash must be the "
                "same for every world size and batch size")
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
from panic3d_amd import ops, sharding, stylegan2 as sg, cameras

torch.manual_seed(0)
G = sg.Generator(z_dim=512, c_dim=25, w_dim=512, img_resolution=256, img_channels=96, cond_mode="none",
                 mapping_kwargs={"num_layers": 2}, channel_base=32768, channel_max=512, num_fp16_res=0, conv_clamp=None).to(dev).eval()
g = torch.Generator().manual_seed(1)
w0, b0, w1, b1 = torch.randn(64, 32, generator=g), torch.randn(64, generator=g) * 0.5, torch.randn(33, 64, generator=g), torch.randn(33, generator=g) * 0.5
w1[0] *= 30.0; b1[0] = -45.0
mlp = ops.prescale_mlp(*(t.to(dev) for t in (w0, b0, w1, b1)), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
ro = dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=48, depth_resolution_importance=48, white_back=True, use_triplane=1)
opts = ops.make_opts(ro, triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True, fast_color=not a.exact)
res, R = a.res, a.res * a.res
azims = np.linspace(0, 360, a.views + 1)[:-1]
labels = torch.stack([cameras.camera_label(0.0, float(az), 1.0, 30.0) for az in azims]).to(dev)  # host camera maths, once

with torch.no_grad():
    def synth():
        ws = G.mapping(torch.randn(1, 512, generator=torch.Generator().manual_seed(2)).to(dev), torch.zeros(1, 25, device=dev), {})
        planes = G.synthesis(ws, {}, noise_mode="const").view(1, 3, 32, 256, 256) * 4.0
        return ops.planes_to_nhwc(planes.contiguous())

    def view_draws(v):  # per-view seeded draws (the reference's order: rand_like [1,R,Sc,1] then rand [R,Sf]): independent of
        g = torch.Generator(device=dev).manual_seed(1000 + v)  # which rank renders the view and of how views are batched
        return torch.rand((1, R, 48, 1), device=dev, generator=g), torch.rand((R, 48), device=dev, generator=g)

    def render_one(v, n=1):
        o, d = cameras.rays_from_label(labels[v:v + n], res)
        dr = [view_draws(v + k) for k in range(n)]
        jit, u = torch.cat([x[0] for x in dr]), torch.cat([x[1] for x in dr])
        # nhwc is [1,...]: shared by the n views; each view keeps its own depth-clamp range, like n calls of the reference
        feat, depth, wsum, xyz = ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res, per_view_clamp=True)
        return feat, wsum

    def render_local_batched():
        lo, hi = sharding.partition(a.views, world, rank)
        fr = [sharding.frames_rgba(*render_one(v, min(a.batch, hi - v)), res) for v in range(lo, hi, a.batch)]
        counts = [sharding.partition(a.views, world, r)[1] - sharding.partition(a.views, world, r)[0] for r in range(world)]
        return sharding.gather_frames(torch.cat(fr), counts, 0)

    nhwc = synth()
    render_one(0)  # warm-up
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    nhwc = synth()
    frames = sharding.render_views_sharded(render_one, a.views, res, dst=0) if a.batch == 1 else render_local_batched()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    dt = time.perf_counter() - t0
if rank == 0:
    assert frames.shape == (a.views, 4, res, res)
    out = {"config": "c4", "views": a.views, "res": res, "batch": a.batch, "final_pass": "exact" if a.exact else "tolerance", "n_gpus": world, "seconds": dt, "views_per_s": a.views / dt,
           "rays_per_s": a.views * R / dt, "alpha_mean": float(frames[:, 3].mean())}
    if a.check:
        import hashlib
        out["sha256"] = hashlib.sha256(frames.cpu().numpy().tobytes()).hexdigest()
    print(json.dumps(out))
if dist.is_initialized():
    dist.destroy_process_group()
