#!/usr/bin/env python3
"""hipGraph capture of one whole view (backbone -> fused renderer -> super-resolution) of TriPlaneGenerator.synthesis with
static shapes, replayed per view with new rays: removes the host/launch gaps between the ~100 small kernels of a view."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import panic3d_amd as P
from panic3d_amd.generator import TriPlaneGenerator
from panic3d_amd import cameras
dev = torch.device("cuda")
RK = {"image_resolution": 512, "disparity_space_sampling": False, "clamp_mode": "softplus",
      "superresolution_module": "training.superresolution.SuperresolutionHybrid8XDC", "c_gen_conditioning_zero": False,
      "c_scale": 1.0, "superresolution_noise_mode": "none", "decoder_lr_mul": 1.0, "sr_antialias": True, "white_back": True,
      "triplane_depth": 1, "use_triplane": 1, "tanh_rgb_output": False, "box_warp": 0.7, "ray_start": 0.5, "ray_end": 1.5,
      "depth_resolution": 96, "depth_resolution_importance": 96}
torch.manual_seed(0)
G = TriPlaneGenerator(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, sr_num_fp16_res=0,
                      mapping_kwargs={"num_layers": 2}, rendering_kwargs=RK, sr_kwargs={"channel_base": 32768, "channel_max": 512},
                      cond_mode="none", triplane_width=32, sr_channels_hidden=256, backbone_resolution=256, channel_base=32768,
                      channel_max=512, num_fp16_res=0, conv_clamp=None).to(dev).eval()
G.set_force_sigmoid(True)
res = 128
with torch.no_grad():
    ws = G.mapping(torch.randn(1, 512, device=dev), torch.zeros(1, 25, device=dev), {})
    c = cameras.camera_label(0, 0, 1.0, 30)[None].to(dev)
    o, d = cameras.rays_from_label(c, res)
    fr = {"ray_origins": o.reshape(1, res, res, 3).permute(0, 3, 1, 2).contiguous(),
          "ray_directions": d.reshape(1, res, res, 3).permute(0, 3, 1, 2).contiguous()}
    call = lambda: G.synthesis(ws, c, {}, neural_rendering_resolution=res, force_rays=fr, triplane_crop=0.1, cull_clouds=0.5, noise_mode="const")
    for _ in range(3):
        out = call()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        out = call()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t) / 10
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        gout = call()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for az in range(10):
        o2, d2 = cameras.rays_from_label(cameras.camera_label(0, 36.0 * az, 1.0, 30)[None].to(dev), res)
        fr["ray_origins"].copy_(o2.reshape(1, res, res, 3).permute(0, 3, 1, 2))
        fr["ray_directions"].copy_(d2.reshape(1, res, res, 3).permute(0, 3, 1, 2))
        g.replay()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t) / 10
    # same view eagerly vs replayed (different random draws -> compare loosely on the SR image)
    ref = call()["image"]
    fr0 = cameras.rays_from_label(c, res)
    fr["ray_origins"].copy_(fr0[0].reshape(1, res, res, 3).permute(0, 3, 1, 2)); fr["ray_directions"].copy_(fr0[1].reshape(1, res, res, 3).permute(0, 3, 1, 2))
    g.replay(); torch.cuda.synchronize()
print(json.dumps({"eager_ms_per_view": eager * 1e3, "graph_ms_per_view": graphed * 1e3, "mean_abs_diff_vs_eager": float((gout["image"] - ref).abs().mean())}))
