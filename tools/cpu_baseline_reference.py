#!/usr/bin/env python3
"""Time the UNMODIFIED reference renderer (training.volumetric_rendering.renderer.ImportanceRenderer.forward + OSGDecoder,
imported from /root/reference, CPU torch) on the bench scenes, and check the CPU oracle against it on the same inputs.

BUILD CONTAINER ONLY (the GPU box has no /root/reference) — like tests/golden/make_golden.py.  Writes
profiles/cpu_baseline_reference.json, which bench.py prints as `cpu_baseline.reference_recorded` beside the live timing of
the port (oracle/p3d_oracle.c).  SURVEY.md §8(d) "CPU baseline beside it", BASELINE.md §3.

    python tools/cpu_baseline_reference.py [--quick]
"""
import argparse
import json
import os
import platform
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def reference_decoder(scene):
    """The reference's OSGDecoder carrying the scene's parameters; for 'canonical' the constructor under
    torch.manual_seed(0) must already give exactly those (this pins tests/p3d_testing.make_bench_scene)."""
    _, raw = T.make_bench_scene(scene)
    torch.manual_seed(0)
    dec = OSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32})
    if scene == "canonical":
        for got, want in zip((dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias), raw):
            assert np.array_equal(got.numpy(), want), "make_bench_scene('canonical') != OSGDecoder under torch.manual_seed(0)"
    else:
        for prm, want in zip((dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias), raw):
            prm.copy_(torch.from_numpy(want))
    dec.set_force_sigmoid(True)
    return dec


def rays(res, azim=20.0):
    cp = dklustr.camera_params_to_matrix("eg3d_lustrousB", elev=0.0, azim=azim, dist=1.0, fov=30.0)
    label = cp["camera_label"][None]
    return RaySampler()(label[:, :16].view(-1, 4, 4), label[:, 16:25].view(-1, 3, 3), res)


def run(scene, res, Sc, Sf, reps, check):
    planes_np, raw = T.make_bench_scene(scene)
    planes = torch.from_numpy(planes_np)
    dec = reference_decoder(scene)
    o, d = rays(res)
    ro = T.bench_rendering_kwargs(Sc, Sf)
    rend = ImportanceRenderer(use_triplane=True)
    times = []
    out = None
    for i in range(reps + 1):  # first call = warm-up
        torch.manual_seed(1234)  # the reference draws rand_like[N,R,Sc,1] then rand[N*R,Sf]: p3d_testing.make_random_draws(1234)
        t = time.perf_counter()
        out = rend(planes, dec, o, d, ro, triplane_crop=0.1, cull_clouds=0.5)
        dt = time.perf_counter() - t
        if i > 0:
            times.append(dt)
    med = float(np.median(times))
    rec = dict(scene=scene, res=res, Sc=Sc, Sf=Sf, rays=res * res, seconds_median=med, runs=reps,
               rays_per_s=res * res / med, samples_per_s=res * res * (Sc + Sf) / med)
    if check:  # the port (oracle) on the same planes / rays / draws: how far is it from the reference at bench scale?
        from oracle import oracle
        oracle.build()
        jit, u = T.make_random_draws(1234, 1, res * res, Sc, Sf)
        ref = oracle.render(planes_np, o.numpy(), d.numpy(), jit, u, oracle.prescale_mlp(*raw),
                            oracle.make_opts(ro, **T.BENCH_KW))
        names = ("feat", "depth", "wsum", "xyz")
        rec["oracle_vs_reference_max_abs"] = {n: float(np.max(np.abs(a.numpy().reshape(b.shape) - b)))
                                              for n, a, b in zip(names, out, ref)}
        img_ref = out[0][..., :3].numpy() * 0.5 + 0.5
        img_or = ref[0][..., :3] * 0.5 + 0.5
        mse = float(np.mean((img_ref - img_or.reshape(img_ref.shape)) ** 2))
        rec["oracle_vs_reference_psnr_db"] = float("inf") if mse == 0 else float(10 * np.log10(1.0 / mse))
        rec["wsum_mean"] = float(out[2].mean())
    print(json.dumps(rec))
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="c1 + 128^2 only")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    results = []
    for scene in ("canonical", "surface"):
        results.append(run(scene, 64, 32, 0, reps=3, check=True))              # BASELINE config c1
        results.append(run(scene, 128, 48, 48, reps=3, check=True))
        if not a.quick:
            results.append(run(scene, 256, 48, 48, reps=3, check=True))        # 256^2 x (48+48)
            # the headline configuration itself (BASELINE c3: 512^2 x (48+48)), un-chunked like every call of the reference: ~20 GB peak
            # (sampled_features [1,3,R*S,32] twice per pass), ~30 s per frame on 8 vCPU (VERDICT r04 "weak" 6: it was only extrapolated)
            results.append(run(scene, 512, 48, 48, reps=2, check=True))
    out = dict(what="unmodified reference ImportanceRenderer.forward (+OSGDecoder) on CPU torch, un-chunked calls, "
                    "torch.no_grad, median of 3 (2 at 512^2) after 1 warm-up", host="build container", cores=os.cpu_count(),
               cpu_model=cpu_model(), torch=torch.__version__, threads=torch.get_num_threads(), results=results)
    path = os.path.join(ROOT, "profiles", "cpu_baseline_reference.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
