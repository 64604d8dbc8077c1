#!/usr/bin/env python3
"""tests/golden/fullsize_generator.npz — the REFERENCE's full-width TriPlaneGenerator (30 M parameters: StyleGAN2-256 backbone,
96-channel planes, SuperresolutionHybrid8XDC, the released model's constructor kwargs of SURVEY.md §8c) run once on CPU.

BUILD CONTAINER ONLY (imports /root/reference).  The checkpoint is absent, so the parameters come from a seeded recipe that
any implementation with the same parameter names can replay (tests/p3d_testing.fill_generator_params): the fixture holds only
outputs — the 512^2 image subsampled 4x, image_raw / image_depth / image_weights at 64^2, a subsampled slice of the planes, ws,
camera parameters and the seed of the two renderer draws.  VERDICT r02 item 7 ("G.f parity only at toy widths").

    python tests/golden/make_golden_fullsize.py
"""
import os
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("PROJECT_DN", "/root/reference")
os.environ.setdefault("PROJECT_NAME", "x")
sys.path[:0] = ["/root/reference"]
sys.path.append("/root/reference/_train/eg3dc/src")
sys.modules.setdefault("kornia", types.ModuleType("kornia"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import p3d_testing as T  # noqa: E402
from training.triplane import TriPlaneGenerator  # noqa: E402
import _databacks.lustrous_renders_v1 as dklustr  # noqa: E402

torch.set_grad_enabled(False)
SEED, NRR, DRAW_SEED = 31, 64, 77


def main():
    torch.set_num_threads(os.cpu_count())
    G = TriPlaneGenerator(**T.FULL_KW).eval()
    T.fill_generator_params(G, SEED)
    G.set_force_sigmoid(True)
    print("parameters:", sum(p.numel() for p in G.parameters()))
    z = torch.randn(1, 512, generator=torch.Generator().manual_seed(SEED + 1))
    cp = dklustr.camera_params_to_matrix("eg3d_lustrousB", elev=5.0, azim=25.0, dist=1.0, fov=30.0)
    c = cp["camera_label"][None].float()
    ws = G.mapping(z, c, {})
    R = NRR * NRR
    jit, u = T.make_random_draws(DRAW_SEED, 1, R, 48, 48)
    torch.manual_seed(DRAW_SEED)  # the renderer's rand_like [1,R,48,1] then rand [R,48]: make_random_draws(DRAW_SEED) replays them
    t = time.perf_counter()
    out = G.synthesis(ws, c, {}, neural_rendering_resolution=NRR, noise_mode="const", triplane_crop=0.1, cull_clouds=0.5)
    print("reference synthesis: %.1f s" % (time.perf_counter() - t))
    planes = out["triplane"]
    print("planes: mean |x| %.3f max %.2f; weights mean %.3f hit %.3f; image range [%.2f, %.2f]" % (
        float(planes.abs().mean()), float(planes.abs().max()), float(out["image_weights"].mean()),
        float((out["image_weights"] > 0.5).float().mean()), float(out["image"].min()), float(out["image"].max())))
    arrs = dict(image_sub4=out["image"][..., ::4, ::4].contiguous().numpy(), image_raw=out["image_raw"].numpy(),
                image_depth=out["image_depth"].numpy(), image_weights=out["image_weights"].numpy(), image_xyz=out["image_xyz"].numpy(),
                planes_sub8=planes[..., ::8, ::8].contiguous().numpy(), planes_abs_mean=np.float32(planes.abs().mean()),
                ws=ws.numpy(), z=z.numpy(), camera_params=c.numpy(), seed=np.int32(SEED), draw_seed=np.int32(DRAW_SEED),
                nrr=np.int32(NRR), torch_version=np.str_(torch.__version__))
    path = os.path.join(HERE, "fullsize_generator.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
