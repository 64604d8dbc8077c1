"""The C ABI from a host with no PyTorch in it: examples/render_c_abi.cpp is compiled against include/panic3d_hip.h and
libpanic3d_hip.so only.  CPU: it builds (hipcc cross-compiles).  GPU: it renders the same bytes as panic3d_amd.ops.render."""
import os, subprocess, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "panic3d-anime-reconstruction_amd")


def build_example(out):
    import panic3d_amd
    panic3d_amd.build()
    cmd = [panic3d_amd._build._hipcc(), "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "examples", "render_c_abi.cpp"),
           "-I", os.path.join(ROOT, "include"), "-L", PKG, "-lpanic3d_hip", f"-Wl,-rpath,{PKG}", "-o", out]
    subprocess.check_call(cmd)
    return out


def test_example_builds_against_the_header_only(tmp_path):
    exe = build_example(str(tmp_path / "render_c_abi"))
    assert os.path.getsize(exe) > 0
    src = open(os.path.join(ROOT, "examples", "render_c_abi.cpp")).read()
    assert "torch" not in src.replace("PyTorch", "") and '#include "panic3d_hip.h"' in src


@pytest.mark.gpu
def test_cpp_host_renders_the_same_bytes(tmp_path, oracle):
    import torch
    import panic3d_amd as P
    import p3d_testing as T
    g = T.load_golden("render_32x32_16p16.npz")
    inp = T.golden_render_inputs(g)
    opts = P.ops.make_opts(inp["ro"], **inp["kw"])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    raw = [dev(x) for x in inp["raw_mlp"]]
    mlp = P.ops.prescale_mlp(*raw, inp["lr_mul"] / np.sqrt(32), inp["lr_mul"], inp["lr_mul"] / np.sqrt(64), inp["lr_mul"])
    N, _, _, H, W = inp["planes"].shape
    R = inp["rays_o"].shape[1]
    d = str(tmp_path)
    for name, a in (("planes", inp["planes"]), ("rays_o", inp["rays_o"]), ("rays_d", inp["rays_d"]), ("jitter", inp["jitter"]),
                    ("u", inp["u"]), ("w0", mlp[0].cpu().numpy()), ("b0", mlp[1].cpu().numpy()), ("w1", mlp[2].cpu().numpy()),
                    ("b1", mlp[3].cpu().numpy())):
        np.ascontiguousarray(a, dtype=np.float32).tofile(os.path.join(d, name + ".bin"))
    with open(os.path.join(d, "meta.txt"), "w") as f:
        f.write(f"{N} {H} {W} {R} 32 {opts.Sc} {opts.Sf} {opts.plane_mode} {opts.flags} " + " ".join(
            repr(float(np.float32(x))) for x in (opts.coord_scale, opts.ray_start, opts.ray_end, opts.depth_delta, opts.crop_limit, opts.cull_thresh)))
    exe = build_example(os.path.join(d, "render_c_abi"))
    out = subprocess.run([exe, d], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "gfx950" in out.stdout
    ref = P.ops.render(P.ops.planes_to_nhwc(dev(inp["planes"])), dev(inp["rays_o"]), dev(inp["rays_d"]), dev(inp["jitter"]), dev(inp["u"]),
                       mlp, opts, ray_tile_w=32)
    orc = oracle.render(inp["planes"], inp["rays_o"], inp["rays_d"], inp["jitter"], inp["u"],
                        oracle.prescale_mlp(*inp["raw_mlp"], lr_mul=inp["lr_mul"]), oracle.make_opts(inp["ro"], **inp["kw"]))
    for name, a, b in zip(("feat", "depth", "wsum", "xyz"), ref, orc):
        got = np.fromfile(os.path.join(d, name + ".bin"), dtype=np.float32).reshape(b.shape)
        assert np.array_equal(got, a.cpu().numpy()) and np.array_equal(got, b), name
