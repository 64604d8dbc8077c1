#!/usr/bin/env python3
"""BASELINE config c2 on one GPU: batch of 4 subjects, random-init StyleGAN2-256 backbone -> 4 x [3,32,256,256] planes ->
fused renderer at 256x256 rays, 48+48 samples (the planes of the four images are distinct: 100 MB of plane data per step)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import panic3d_amd as P
P.stylegan2.STYLE_MEMO = False  # every timed pass computes its styles (a new batch of subjects per pass)
from panic3d_amd import ops, cameras, stylegan2 as sg

torch.manual_seed(0)
dev = "cuda"
N, res, Sc, Sf = 4, 256, 48, 48
G = sg.Generator(z_dim=512, c_dim=25, w_dim=512, img_resolution=256, img_channels=96, cond_mode="none",
                 mapping_kwargs={"num_layers": 2}, channel_base=32768, channel_max=512, num_fp16_res=0, conv_clamp=None).to(dev).eval()
g = torch.Generator().manual_seed(1)
w0, b0, w1, b1 = torch.randn(64, 32, generator=g), torch.randn(64, generator=g) * 0.5, torch.randn(33, 64, generator=g), torch.randn(33, generator=g) * 0.5
w1[0] *= 30.0; b1[0] = -45.0
mlp = ops.prescale_mlp(*(t.to(dev) for t in (w0, b0, w1, b1)), 1 / np.sqrt(32), 1.0, 1 / np.sqrt(64), 1.0)
ro = dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=Sc, depth_resolution_importance=Sf, white_back=True, use_triplane=1)
EXACT = "--exact" in sys.argv  # exact-contract final pass (default: the renderer's default, tolerance mode)
opts = ops.make_opts(ro, triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True, fast_color=not EXACT)
labels = torch.stack([cameras.camera_label(0.0, 30.0 * i, 1.0, 30.0) for i in range(N)]).to(dev)
o, d = cameras.rays_from_label(labels, res)
R = res * res


def sync():
    torch.cuda.synchronize(); return time.perf_counter()


with torch.no_grad():
    ws = G.mapping(torch.randn(N, 512, device=dev), torch.zeros(N, 25, device=dev), {})

    def step():
        planes = G.synthesis(ws, {}, noise_mode="const").view(N, 3, 32, 256, 256) * 4.0
        nhwc = ops.planes_to_nhwc(planes.contiguous())
        jit = torch.rand((N, R, Sc, 1), device=dev); u = torch.rand((N * R, Sf), device=dev)
        return planes, ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res)

    for _ in range(2):
        step()
    K = 10
    t0 = sync()
    for _ in range(K):
        planes, out = step()
    t1 = sync()
    nhwc = ops.planes_to_nhwc(planes.contiguous())
    jit = torch.rand((N, R, Sc, 1), device=dev); u = torch.rand((N * R, Sf), device=dev)
    t2 = sync()
    for _ in range(K):
        ops.render(nhwc, o, d, jit, u, mlp, opts, ray_tile_w=res)
    t3 = sync()
print(json.dumps({"config": "c2", "final_pass": "exact" if EXACT else "tolerance", "batch": N, "rays_per_image": R, "samples": [Sc, Sf], "ms_backbone_plus_render": (t1 - t0) / K * 1e3,
                  "ms_render_only": (t3 - t2) / K * 1e3, "rays_per_s_render_only": N * R / ((t3 - t2) / K),
                  "rays_per_s_incl_backbone": N * R / ((t1 - t0) / K), "wsum_mean": float(out[2].mean())}))
