#!/usr/bin/env python3
"""tests/golden/bench_reference_block.npz — the REFERENCE renderer's output on a block of bench.py's frame.

BUILD CONTAINER ONLY (imports the unmodified reference from /root/reference on CPU; the GPU box has no reference tree).
bench.py's `verify.reference_block` re-renders exactly these rays with exactly these draws on the HIP path and prints
max-abs and PSNR of `image_raw` against this file: the "PSNR vs ref" half of BASELINE.json's metric, measured inside the
driver's own run against the real reference (not the port).

What is stored, per bench scene with a non-empty volume ('surface'): the centre SIDE x SIDE rays of bench.py's 512^2 view
(azimuth 20, fov 30), 48+48 samples, the reference's four outputs, seed of the two draws (p3d_testing.make_random_draws
reproduces the reference's rand_like / rand bit for bit — asserted here).

    python tests/golden/make_golden_bench.py
"""
import os
import sys
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
from training.volumetric_rendering.renderer import ImportanceRenderer  # noqa: E402
from training.volumetric_rendering.ray_sampler import RaySampler  # noqa: E402
from training.triplane import OSGDecoder  # noqa: E402
import _databacks.lustrous_renders_v1 as dklustr  # noqa: E402

torch.set_grad_enabled(False)
RES, SIDE, SEED = 512, 64, 4242


def block_rays(res=RES, side=SIDE, azim=20.0):
    cp = dklustr.camera_params_to_matrix("eg3d_lustrousB", elev=0.0, azim=azim, dist=1.0, fov=30.0)
    label = cp["camera_label"][None]
    o, d = RaySampler()(label[:, :16].view(-1, 4, 4), label[:, 16:25].view(-1, 3, 3), res)
    a = (res - side) // 2
    idx = (torch.arange(a, a + side)[:, None] * res + torch.arange(a, a + side)[None, :]).reshape(-1)
    return o[:, idx].contiguous(), d[:, idx].contiguous()


def main():
    out = {}
    for scene, (Sc, Sf) in (("surface", (48, 48)), ("surface96", (96, 96))):
        planes_np, raw = T.make_bench_scene("surface")
        dec = OSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32})
        for prm, want in zip((dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias), raw):
            prm.copy_(torch.from_numpy(want))
        dec.set_force_sigmoid(True)
        o, d = block_rays()
        ro = T.bench_rendering_kwargs(Sc, Sf)
        # the draws the reference is about to make, captured, must be the ones make_random_draws regenerates
        jit, u = T.make_random_draws(SEED, 1, SIDE * SIDE, Sc, Sf)
        torch.manual_seed(SEED)
        j2 = torch.rand(1, SIDE * SIDE, Sc, 1)
        u2 = torch.rand(SIDE * SIDE, Sf)
        assert np.array_equal(j2.numpy(), jit) and np.array_equal(u2.numpy(), u)
        torch.manual_seed(SEED)
        feat, depth, wsum, xyz = ImportanceRenderer(use_triplane=True)(torch.from_numpy(planes_np), dec, o, d, ro, **{k: v for k, v in T.BENCH_KW.items() if k != "force_sigmoid"})
        print(scene, "wsum mean", float(wsum.mean()), "hit", float((wsum > 0.5).float().mean()))
        p = scene + "_"
        out.update({p + "feat": feat.numpy(), p + "depth": depth.numpy(), p + "wsum": wsum.numpy(), p + "xyz": xyz.numpy(),
                    p + "rays_o": o.numpy(), p + "rays_d": d.numpy(), p + "Sc": np.int32(Sc), p + "Sf": np.int32(Sf)})
    out.update(seed=np.int32(SEED), res=np.int32(RES), side=np.int32(SIDE),
               planes_checksum=np.str_(T.checksum(T.make_bench_scene("surface")[0])), torch_version=np.str_(torch.__version__))
    path = os.path.join(HERE, "bench_reference_block.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
