#!/usr/bin/env python3
"""The per-subject flow of _scripts/eval/generate.py:80-151 on the MI355X path, at the released model's sizes, with
random-init weights and a synthetic illustration (the checkpoint and the AnimeRecon data are not in this environment):
density grid 256^3 (get_eg3d_volume), then 4 orthographic + 12 perspective views through G.f with the front-view paste.
Prints a timing breakdown (one JSON line)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import panic3d_amd as P
from panic3d_amd.generator import TriPlaneGenerator
from panic3d_amd import volume

dev = torch.device("cuda")
RK = {"image_resolution": 512, "disparity_space_sampling": False, "clamp_mode": "softplus",
      "superresolution_module": "training.superresolution.SuperresolutionHybrid8XDC", "c_gen_conditioning_zero": True,  # train_eclustrousC.py:414 with the default --gen_pose_cond=False
     
      "gpc_reg_prob": 0.5, "c_scale": 1.0, "superresolution_noise_mode": "none", "density_reg": 0.25, "density_reg_p_dist": 0.004,
      "reg_type": "l1", "decoder_lr_mul": 1.0, "sr_antialias": True, "white_back": True, "triplane_depth": 1, "use_triplane": 1,
      "tanh_rgb_output": False, "box_warp": 0.7, "ray_start": 0.5, "ray_end": 1.5, "depth_resolution": 96,
      "depth_resolution_importance": 96, "avg_camera_radius": 1.0, "avg_camera_pivot": [0, 0, 0]}  # 96/96: eg3dc_v0.py:55-56
torch.manual_seed(0)
G = TriPlaneGenerator(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, sr_num_fp16_res=0,
                      mapping_kwargs={"num_layers": 2}, rendering_kwargs=RK,
                      sr_kwargs={"channel_base": 32768, "channel_max": 512, "fused_modconv_default": "inference_only"},
                      cond_mode="ortho_front.add_shuffle2_4.inj_6b_4.reschonk_add_64", triplane_width=32, sr_channels_hidden=256,
                      backbone_resolution=256, channel_base=32768, channel_max=512, fused_modconv_default="inference_only",
                      num_fp16_res=0, conv_clamp=None).to(dev).eval()
with torch.no_grad():
    for n, p in G.backbone.synthesis.named_parameters():
        if n.endswith("torgb.weight"):
            p.mul_(30.0)  # random-init ToRGB gives planes ~0: scale to O(1) features so that there is a surface to render
    G.decoder.net[2].weight[0] *= 20.0
G.set_force_sigmoid(True)
G.neural_rendering_resolution = 128
cond = {"image_ortho_front": torch.rand(1, 3, 512, 512, device=dev), "resnet_chonk": torch.randn(1, 64, 8, 8, device=dev)}
opts = {"triplane_crop": 0.1, "cull_clouds": 0.5,
        "paste_params": {"mode": "default", "thresh_weight": 0.95, "thresh_edges": 0.02, "thresh_occ": 0.05, "offset_occ": 0.01,
                         "thresh_dxyz": 0.000005}}
cam60 = torch.tensor(np.stack(np.meshgrid(np.linspace(60, -20, 5), np.linspace(-180, 150, 12))).T.reshape(60, -1)).float()
spin12 = [*range(42, 48), *range(36, 42)]
views = [(0, 0, -1), (0, 90, -1), (0, -90, -1), (0, 180, -1)] + [(float(cam60[v][0]), float(cam60[v][1]), 30) for v in spin12]

def sync():
    torch.cuda.synchronize(); return time.perf_counter()

with torch.no_grad():
    x0 = {"elevations": torch.zeros(1, device=dev), "azimuths": torch.zeros(1, device=dev), "cond": cond, "seeds": [0], "noise_mode": "const",
          "triplane_crop": 0.1, "cull_clouds": 0.5}
    xw = dict(x0)
    G.f(xw)  # warm-up
    for fov in (-1.0, 30.0):  # ... including the paste path, orthographic and perspective (first use of a torch kernel loads its code object)
        G.f({"elevations": torch.zeros(1, device=dev), "azimuths": 10 * torch.ones(1, device=dev), "fovs": fov * torch.ones(1, device=dev),
             "cond": cond, "seeds": [0], **opts})
    volume.mesh(G, xw["ws"], cond, resolution=32, level=0.5, triplane_crop=0.1, cull_clouds=0.5)  # warm-up (module load, allocator)
    t0 = sync()
    out = G.f(x0)
    t1 = sync()
    vol = volume.density_grid(G, x0["ws"], cond, resolution=256, triplane_crop=0.1, cull_clouds=0.5)
    dens = volume.to_volume(vol["densities"], 256)
    t2 = sync()
    level = 0.5  # generate.py:102
    tm0 = sync()
    mc = volume.mesh(G, x0["ws"], cond, resolution=256, level=level, triplane_crop=0.1, cull_clouds=0.5)  # generate.py:97-103 -> pkl dict
    tm1 = sync()
    imgs = []
    for elev, azim, fov in views:
        xin = {"elevations": elev * torch.ones(1, device=dev), "azimuths": azim * torch.ones(1, device=dev),
               "fovs": fov * torch.ones(1, device=dev), "cond": cond, "seeds": [0], **opts}
        o = G.f(xin)
        imgs.append(o["image"])
    t3 = sync()
    for elev, azim, fov in views:  # the same 16 views for the NEXT subject: labels and rays of a view are computed once per process (cameras.cached_view)
        xin = {"elevations": elev * torch.ones(1, device=dev), "azimuths": azim * torch.ones(1, device=dev),
               "fovs": fov * torch.ones(1, device=dev), "cond": cond, "seeds": [0], **opts}
        G.f(xin)
    t3b = sync()
    imgs2 = []
    for k, (elev, azim, fov) in enumerate(views):  # same flow, planes synthesised once per subject (x['use_cached_backbone'])
        xin = {"elevations": elev * torch.ones(1, device=dev), "azimuths": azim * torch.ones(1, device=dev),
               "fovs": fov * torch.ones(1, device=dev), "cond": cond, "seeds": [0], "noise_mode": "const",
               "cache_backbone": k == 0, "use_cached_backbone": k > 0, **opts}
        imgs2.append(G.f(xin)["image"])
    t4 = sync()
    t4 -= t3b - t3
    # all 16 views of the subject in ONE f() call: one backbone pass, one renderer launch per pass (shared planes), batched SR
    xin = {"elevations": torch.tensor([v[0] for v in views], device=dev, dtype=torch.float32),
           "azimuths": torch.tensor([v[1] for v in views], device=dev, dtype=torch.float32),
           "fovs": torch.tensor([v[2] for v in views], device=dev, dtype=torch.float32), "cond": cond, "seeds": [0], "noise_mode": "const", **opts}
    G.f(dict(xin))
    t5 = sync()
    ob = G.f(xin)
    t6 = sync()
    G.set_sr_mma_f16(True)  # opt-in: super-resolution convolutions on f16 MFMA operands (the reference's SR blocks are fp16 on GPU)
    G.f(dict(xin))
    t7 = sync()
    oh = G.f(dict(xin))
    t8 = sync()
    G.set_sr_mma_f16(False)
    sr_f16_psnr = float(10 * torch.log10(1.0 / ((oh["image_prepaste"] - ob["image_prepaste"]).double() ** 2).mean()))
if "--out" in sys.argv:  # the reference's per-subject files (generate.py:104-105,132-148)
    from panic3d_amd import outputs
    odn = sys.argv[sys.argv.index("--out") + 1]
    outputs.dump_mesh(mc, f"{odn}/marching_cubes.pkl")
    outputs.save_view(o, f"{odn}/rgb60_last.png", f"{odn}/xyza60_last.png", RK["box_warp"])
assert all(i.shape == (1, 3, 512, 512) and torch.isfinite(i).all() for i in imgs) and dens.shape == (1, 1, 256, 256, 256)
print(json.dumps({"one_view_f_ms": (t1 - t0) * 1e3, "density_grid_256_ms": (t2 - t1) * 1e3, "mesh_256_ms_incl_grid_and_d2h": (tm1 - tm0) * 1e3,
                  "mesh_verts": len(mc["verts"]), "mesh_faces": len(mc["faces"]), "views": len(views),
                  "views_with_paste_ms": (t3 - tm1) * 1e3, "ms_per_view": (t3 - tm1) * 1e3 / len(views), "ms_per_view_next_subjects": (t3b - t3) * 1e3 / len(views),
                  "subject_total_ms": (t3 - t0 - (tm0 - t2)) * 1e3, "ms_per_view_planes_cached": (t4 - t3) * 1e3 / len(views), "ms_per_view_all_views_one_call": (t6 - t5) * 1e3 / len(views),
                  "ms_per_view_all_views_one_call_sr_f16_operands": (t8 - t7) * 1e3 / len(views), "sr_f16_vs_fp32_psnr_db": sr_f16_psnr, "mean_alpha_last_view": float(o["image_weights"].mean()),
                  "paste_mask_mean_last_view": float(o["paste"]["mask"].mean())}))
