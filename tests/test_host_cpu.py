"""CPU (-m "not gpu"): host logic of the product — the C-ABI library loads and exports every declared symbol, option
conversion matches the oracle's, camera/ray generation matches the reference's golden vectors, view sharding +
frame gather work over gloo with world_size 2, and the product refuses CPU tensors (no fallback)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import p3d_testing as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def P():
    import panic3d_amd
    panic3d_amd.build()  # hipcc cross-compiles gfx950 without a GPU
    return panic3d_amd


def test_cabi_exports_every_declared_symbol(P):
    hdr = open(os.path.join(ROOT, "include", "panic3d_hip.h")).read()
    declared = set(re.findall(r"\b(p3d_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(P._lib.SIGNATURES), "include/panic3d_hip.h and _lib.SIGNATURES disagree"
    L = P._lib.lib()
    for name in declared:
        assert hasattr(L, name)
    assert b"gfx950" in L.p3d_build_info()
    assert L.p3d_render_workspace_bytes(1, 4096, 48, 48) >= 8


def test_cabi_argument_errors_without_gpu(P):
    import ctypes as C
    L = P._lib.lib()
    o = P.ops.make_opts(T.RENDERING_KWARGS)
    # null pointers / bad sizes are rejected before any launch
    assert L.p3d_planes_to_nhwc_f32(None, 3, 32, 8, 8, None, None) == -1
    assert L.p3d_triplane_decode_f32(None, 1, 8, 8, None, 1, None, None, None, None, C.byref(o), None, None, None) == -1
    assert L.p3d_sample_stratified_f32(0.5, 1.5, 0.02, 48, None, 1, None, None) == -1
    assert L.p3d_importance_f32(None, None, 0, 48, 48, None, None, None, None) == -1
    assert L.p3d_sigma2density_f32(None, None, 8, -1.0, None, None) == -1
    assert L.p3d_mc_workspace_bytes(1) == 0 and L.p3d_mc_workspace_bytes(1025) == 0 and L.p3d_mc_workspace_bytes(64) > 4 * 64 ** 3
    assert L.p3d_mc_count_f32(None, 8, 0, 0.5, None, 0, None, None) == -1
    assert L.p3d_conv_weights_to_f16(None, 4, 16, 3, None, None) == -1
    fake = C.c_void_p(16)  # never dereferenced: the range checks come first
    assert L.p3d_conv_weights_to_f16(fake, 4, 16, 2, fake, None) == -2  # ks must be 1 or 3
    assert L.p3d_modconv2d_f16mma_f32(fake, 1, 24, 8, 8, fake, fake, 8, 3, fake, 1, None, None, 0, None, 1, 1, 0.2, 1.0, -1.0, None, fake,
                                      fake, 1 << 20, None) == -2  # I % 16 != 0: use the fp32 entry point
    assert L.p3d_modconv2d_f16mma_f32(fake, 1, 32, 8, 8, fake, None, 8, 3, fake, 1, None, None, 0, None, 1, 1, 0.2, 1.0, -1.0, None, fake,
                                      fake, 1 << 20, None) == -1  # no f16 weights
    assert L.p3d_modconv2d_f32(None, 1, 32, 8, 8, fake, 8, 3, fake, 1, None, None, 0, None, 1, 1, 0.2, 1.0, -1.0, None, fake, fake, 1 << 20, None) == -1
    assert L.p3d_demod_coefs_f32(None, fake, fake, 3, 1, 24, fake, None) == -1 and L.p3d_demod_coefs_f32(fake, fake, fake, 65, 1, 24, fake, None) == -2
    assert L.p3d_depth_minmax_f32(None, 8, fake, fake, 16, None) == -1 and L.p3d_depth_minmax_f32(fake, 8, fake, fake, 8, None) == -3
    assert L.p3d_composite_workspace_bytes(1024, 96, 35) >= 8 and L.p3d_abi_version() == P._lib.P3D_ABI_VERSION
    assert L.p3d_render_workspace_bytes(4, 4096, 48, 48) >= 16 + 4 * 8  # one clamp range per view under P3D_FLAG_SHARED_PLANES


def test_opts_match_oracle(P, oracle):
    for kw in (dict(), dict(triplane_crop=0.1, cull_clouds=0.5), dict(binarize_clouds=0.4, triplane_crop=0.05),
               dict(force_sigmoid=True)):
        for ro in (T.RENDERING_KWARGS, dict(T.RENDERING_KWARGS, box_warp=1.0, depth_resolution=96, ray_start=2.25,
                                            ray_end=3.3, use_triplane=0, white_back=False)):
            a, b = P.ops.make_opts(ro, **kw), oracle.make_opts(ro, **{"force_sigmoid": False, **kw})
            for (name, _) in a._fields_:
                assert getattr(a, name) == getattr(b, name), name
    with pytest.raises(AssertionError):
        P.ops.make_opts(dict(T.RENDERING_KWARGS, clamp_mode="relu"))


def test_prescale_matches_oracle(P, oracle):
    raw = T.make_decoder_params(3, lr_mul=0.5)
    got = P.ops.prescale_mlp(*(torch.from_numpy(x) for x in raw), 0.5 / np.sqrt(32), 0.5, 0.5 / np.sqrt(64), 0.5)
    for a, b in zip(got, oracle.prescale_mlp(*raw, lr_mul=0.5)):
        assert np.array_equal(a.numpy(), b)


def test_cameras_match_reference_golden(P):
    z = T.load_golden("rays.npz")
    for i in range(3):
        e, a, f = z[f"persp{i}_cam"]
        lab = P.cameras.camera_label(e, a, 1.0, f)
        assert np.array_equal(lab.numpy(), z[f"persp{i}_label"][0])
        o, d = P.cameras.rays_from_label(lab[None], 16)
        assert np.array_equal(o.numpy(), z[f"persp{i}_o"]) and np.array_equal(d.numpy(), z[f"persp{i}_d"])
        fr = P.cameras.ortho_rays(e, a, 1.0, 0.7, 16)
        assert np.array_equal(fr["ray_origins"].numpy(), z[f"ortho{i}_o"])
        assert np.array_equal(fr["ray_directions"].numpy(), z[f"ortho{i}_d"])


def test_product_has_no_cpu_fallback(P):
    with pytest.raises(RuntimeError):
        P.ops.planes_to_nhwc(torch.zeros(1, 3, 32, 8, 8))
    r = P.ImportanceRenderer(use_triplane=True)
    with pytest.raises(RuntimeError):
        r.run_model(torch.zeros(1, 3, 32, 8, 8), _FakeDecoder(), torch.zeros(1, 4, 3), None, T.RENDERING_KWARGS)
    # the product never touches the oracle
    for fn in os.listdir(os.path.join(ROOT, "panic3d-anime-reconstruction_amd")):
        if fn.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "panic3d-anime-reconstruction_amd", fn)).read()


class _FC:
    def __init__(self, o, i):
        self.weight, self.bias = torch.zeros(o, i), torch.zeros(o)
        self.weight_gain, self.bias_gain = 1 / np.sqrt(i), 1


class _FakeDecoder:
    force_sigmoid = True

    def __init__(self):
        self.net = [_FC(64, 32), None, _FC(33, 64)]


def test_partition():
    from panic3d_amd import sharding
    for n, w in ((120, 8), (16, 3), (7, 7), (512, 8)):
        spans = [sharding.partition(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert sharding.partition(120, 8, 3) == (45, 60)  # BASELINE config c4: 15 views per GPU


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import panic3d_amd
from panic3d_amd import sharding
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
res, n_views = 8, {n_views}
def render_one(v):  # stands in for the HIP render: a frame whose content identifies the view
    feat = torch.full((1, res * res, 32), float(v)); wsum = torch.full((1, res * res, 1), 0.5 + v)
    return feat, wsum
out = sharding.render_views_sharded(render_one, n_views, res, dst=0)
if rank == 0:
    assert out.shape == (n_views, 4, res, res), out.shape
    for v in range(n_views):
        assert torch.all(out[v, :3] == v) and torch.all(out[v, 3] == 0.5 + v)
    print("GATHER_OK")
else:
    assert out is None
dist.destroy_process_group()
"""


def _run_gather_worker(tmp_path, world, n_views, port):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, n_views=n_views))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GATHER_OK" in r.stdout


def test_view_sharding_gloo_world2(tmp_path):
    _run_gather_worker(tmp_path, 2, 5, 29541)


def test_view_sharding_more_ranks_than_views(tmp_path):
    """A rank without views must not raise before the collective (the others would hang in it): it joins the gather with an
    empty stack (sharding.py:80)."""
    _run_gather_worker(tmp_path, 3, 2, 29543)


_STREAM_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import panic3d_amd
from panic3d_amd import sharding
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
K, res = 7, 4
frames = torch.zeros((K, res, res, 4))
g = sharding.FrameGather(frames, K, dst=0)
g.p2p = {p2p}  # True: the batched point-to-point fallback instead of gather()
chunk = 3
for i in range(K):  # "render" frame i, hand finished slices over while the loop goes on
    frames[i] = 100.0 * rank + i
    if (i + 1) % chunk == 0:
        g.push(i + 1 - chunk, i + 1)
if K % chunk:
    g.push(K - K % chunk, K)
out = g.finish()
if rank == 0:
    assert out.shape == (world * K, res, res, 4)
    for r in range(world):
        for i in range(K):
            assert torch.all(out[r * K + i] == 100.0 * r + i), (r, i)
    print("GATHER_OK")
else:
    assert out is None
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("p2p", [False, True])
def test_streamed_frame_gather_gloo_world3(tmp_path, p2p):
    """sharding.FrameGather (bench.py's collective: slices of the sweep sent while the next ones render) at world size 3:
    rank order, slice order and the ragged last slice; with gather() and with the batched point-to-point fallback."""
    script = tmp_path / "worker.py"
    script.write_text(_STREAM_WORKER.format(root=ROOT, p2p=p2p))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                        "--master-addr", "127.0.0.1", "--master-port", str(29547 + int(p2p)), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GATHER_OK" in r.stdout


def test_bench_gpus_flag_starts_its_own_ranks_dry_launch():
    """`python bench.py --gpus 2` — the driver's plain form, no torch.distributed.run around it — must start two ranks itself
    (the flag used to be parsed and never read).  --dry-launch stops after the rendezvous: gloo on CPU, no kernels."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_ranks_seen"] == 2 and line["world_size"] == 2 and "itself" in line["launcher"]
    # ... and the dry launch runs the SAME timed-sweep code as the GPU run (bench.timed_sweep) over stand-in frames: the multi-GPU
    # fields of the contract line are there — ranks seen, per-rank render ms, per-rank exposed gather ms, what moved into rank 0 —
    # and every rank's frames arrived in rank order
    assert len(line["per_rank"]["ms_per_step_render"]) == 2 and len(line["per_rank"]["gather_ms"]) == 2
    g = line["gather"]
    assert g["mode"].startswith("streamed") and g["content_ok"] is True and g["frames_per_rank"] == line["steps"] * line["views_per_step"]
    assert g["bytes_into_rank0"] == g["frames_per_rank"] * g["frame_bytes"] and g["exposed_ms_max"] >= g["exposed_ms_rank0"] >= 0
    # V views per launch, one gather at the end, three ranks with an odd step count
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-launch", "--steps", "5", "--warmup", "1",
                        "--views-per-step", "4", "--gather", "end"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_ranks_seen"] == 3 and line["views_per_step"] == 4 and line["gather"]["mode"].startswith("end")
    assert line["gather"]["content_ok"] is True and line["gather"]["frames_per_rank"] == 20 and len(line["per_rank"]["gather_ms"]) == 3
    # under a launcher the world size must match --gpus: a 3-rank launch of `--gpus 2` refuses to run
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                        "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stdout + r.stderr)


def test_stylegan2_state_dict_matches_reference_names(P):
    """Drop-in requirement (eg3dc_v0.py:49 copy_params_and_buffers(require_all=True)): identical parameter / buffer names
    and shapes as the reference's Generator — checked against the state_dict the reference itself produced."""
    from panic3d_amd import stylegan2 as sg
    kw = dict(z_dim=64, c_dim=25, w_dim=64, img_resolution=32, img_channels=96, mapping_kwargs={"num_layers": 2},
              channel_base=2048, channel_max=64, num_fp16_res=0, conv_clamp=None, fused_modconv_default="inference_only")
    for tag in ("none", "cond"):
        g = T.load_golden(f"syn_generator_{tag}.npz")
        G = sg.Generator(cond_mode=str(g["cond_mode"]), **kw)
        ref = {k[3:].replace("__", "."): v.shape for k, v in g.items() if k.startswith("sd_")}
        mine = {k: tuple(v.shape) for k, v in G.state_dict().items()}
        assert set(mine) == set(ref)
        import copy, pickle  # the reference pickles G for snapshots and deep-copies it: derived caches must not get in the way
        G.synthesis.b4.conv1.affine._scaled(torch.float32)
        G2 = pickle.loads(pickle.dumps(G))
        assert set(G2.state_dict()) == set(ref) and not hasattr(G2.synthesis.b4.conv1.affine, "_scaled_wb")
        assert set(copy.deepcopy(G).state_dict()) == set(ref)
        assert all(tuple(ref[k]) == mine[k] for k in ref)
    # full-size backbone of the released configuration: 14 ws, 96-channel 256^2 output
    G = sg.Generator(z_dim=512, c_dim=25, w_dim=512, img_resolution=256, img_channels=96, cond_mode="none",
                     mapping_kwargs={"num_layers": 2}, channel_base=32768, channel_max=512, num_fp16_res=0, conv_clamp=None)
    assert G.num_ws == 14 and G.synthesis.b256.conv1.weight.shape == (128, 128, 3, 3)


def test_create_samples_matches_reference_formula(P):
    """_util/eg3d_metrics3d.py:70-92 restated with the same float quirks; slab ranges concatenate to the full grid."""
    N, L = 12, 0.7
    full, origin, vs = P.volume.create_samples(N, cube_length=L)
    idx = torch.arange(0, N ** 3, 1, out=torch.LongTensor())
    ref = torch.zeros(N ** 3, 3)
    ref[:, 2] = idx % N
    ref[:, 1] = (idx.float() / N) % N
    ref[:, 0] = ((idx.float() / N) / N) % N
    o = np.array([0, 0, 0]) - L / 2
    for k, j in ((0, 2), (1, 1), (2, 0)):
        ref[:, k] = ref[:, k] * (L / (N - 1)) + o[j]
    assert torch.equal(full[0], ref)
    parts = [P.volume.create_samples(N, cube_length=L, lo=a, hi=b)[0] for a, b in ((0, 500), (500, 1000), (1000, N ** 3))]
    assert torch.equal(torch.cat(parts, dim=1), full)


def test_outputs_png_and_pkl_formats(tmp_path):
    """generate.py's on-disk outputs: PNG quantisation = clamp, x255, truncate (twodee_v1.py:184-185 via torchvision's
    to_pil_image); xyza composition (generate.py:143-146); mesh pickle keys (eg3d_metrics3d.py:203-209)."""
    import pickle
    from PIL import Image
    from panic3d_amd import outputs
    g = torch.Generator().manual_seed(0)
    img = torch.rand(1, 3, 8, 6, generator=g) * 1.4 - 0.2
    q = outputs.to_uint8_hwc(img)
    ref = (img[0].clamp(0, 1) * 255).numpy().astype(np.uint8).transpose(1, 2, 0)  # truncation, not rounding
    assert q.shape == (8, 6, 3) and np.array_equal(q, ref) and q.min() == 0 and q.max() == 255
    out = {"image": img, "image_xyz": torch.rand(1, 3, 8, 6, generator=g) * 0.7 - 0.35, "image_weights": torch.rand(1, 1, 8, 6, generator=g)}
    a, b = outputs.save_view(out, str(tmp_path / "v" / "front.png"), str(tmp_path / "v" / "front_xyza.png"), 0.7)
    assert np.array_equal(np.asarray(Image.open(a)), ref)
    x = np.asarray(Image.open(b))
    assert x.shape == (8, 6, 4) and Image.open(b).mode == "RGBA"
    exp = torch.cat([(out["image_xyz"] + 0.35) / 0.7, out["image_weights"]], 1)[0].clamp(0, 1).mul(255).to(torch.uint8).permute(1, 2, 0).numpy()
    assert np.array_equal(x, exp)
    mc = {"verts": np.zeros((3, 3), np.float32), "faces": np.array([[0, 1, 2]], np.int32), "normals": np.zeros((3, 3), np.float32),
          "values": np.ones(3, np.float32), "colors": np.zeros((3, 3), np.float32)}
    fn = outputs.dump_mesh(mc, str(tmp_path / "m" / "marching_cubes.pkl"))
    back = pickle.load(open(fn, "rb"))
    assert set(back) == {"verts", "faces", "normals", "values", "colors"} and back["faces"].dtype == np.int32
    with pytest.raises(RuntimeError):
        outputs.to_uint8_hwc(torch.zeros(2, 3, 4, 4))


def test_fully_connected_scaled_weight_cache(P):
    """FullyConnectedLayer caches weight * gain per parameter version: same values as the per-call multiplication, refreshed
    when the parameters change (load_state_dict / in-place updates), absent from the state_dict."""
    fc = P.stylegan2.FullyConnectedLayer(12, 7, lr_multiplier=0.5, bias_init=1.0)
    x = torch.randn(3, 12, generator=torch.Generator().manual_seed(0))
    ref = lambda: torch.addmm((fc.bias * fc.bias_gain).unsqueeze(0), x, (fc.weight * fc.weight_gain).t())
    with torch.no_grad():
        assert torch.equal(fc(x), ref())
        k0 = fc._scaled_key
        assert torch.equal(fc(x), ref()) and fc._scaled_key == k0
        fc.weight.mul_(2.0)
        assert torch.equal(fc(x), ref()) and fc._scaled_key != k0
        fc.load_state_dict({"weight": torch.ones(7, 12), "bias": torch.zeros(7)})
        assert torch.equal(fc(x), ref())
    assert set(fc.state_dict()) == {"weight", "bias"}


def test_ray_limits_box_matches_the_reference(P):
    """cameras.ray_limits_box + patch_ray_limits (math_utils.get_ray_limits_box, renderer.py:167-170) against the per-ray limits the
    reference itself produced for rendering_options ray_start = ray_end = 'auto' (tests/golden/render_auto_limits.npz)."""
    from panic3d_amd import cameras
    g = T.load_golden("render_auto_limits.npz")
    o, d = torch.from_numpy(g["rays_o"]), torch.from_numpy(g["rays_d"])
    a, b = cameras.ray_limits_box(o, d, float(g["meta_box_warp"]))
    assert int((b <= a).sum()) == 21 and torch.all(a[b <= a] == -1) and torch.all(b[b <= a] == -2)  # rays that miss the box
    a, b = cameras.patch_ray_limits(a, b)
    assert np.array_equal(a.reshape(1, -1).numpy(), g["ray_start"]) and np.array_equal(b.reshape(1, -1).numpy(), g["ray_end"])
    # no valid ray at all: nothing is patched (the reference's `if torch.any(is_ray_valid)`)
    a2, b2 = cameras.patch_ray_limits(torch.full((1, 4, 1), -1.0), torch.full((1, 4, 1), -2.0))
    assert torch.all(a2 == -1) and torch.all(b2 == -2)


def test_no_memo_environment_variable_is_read_at_import():
    import subprocess, sys
    code = "import os, sys; sys.path.insert(0, %r); import panic3d_amd as P; print(int(P.memo.enabled()))" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for val, want in (("1", "0"), ("0", "1"), (None, "1")):
        env = {k: v for k, v in os.environ.items() if k != "P3D_NO_MEMO"}
        if val is not None:
            env["P3D_NO_MEMO"] = val
        assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300).stdout.strip() == want


def test_sobel_restatement_against_a_dense_convolution_with_the_published_kernels():
    """VERDICT r03 item 7 (f4).  kornia is not installable here, so `kornia.filters.sobel` (triplane.py:632,652) is restated — and until
    round 4 the restatement (paste.sobel_magnitude, shifted slices; the HIP kernel is tested against it) was only ever compared with
    itself.  This is an INDEPENDENT formulation of kornia 0.6.5's published definition: `spatial_gradient(mode='sobel', order=1,
    normalized=True)` = a dense F.conv2d of the replicate-padded image with the 3x3 kernels
        gx = [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]] / 8,   gy = gx^T      (kornia.filters.kernels.get_sobel_kernel_3x3, normalize_kernel2d: / sum |k|)
    per channel (groups = C), then sqrt(gx^2 + gy^2 + eps), eps = 1e-6.  Also checked: it is a CORRELATION (conv2d does not flip;
    kornia flips the kernels once and F.conv3d un-flips them: the published gradient of a ramp rising to the right is positive)."""
    import torch.nn.functional as F
    from panic3d_amd import paste
    g = torch.Generator().manual_seed(12)
    x = torch.rand(2, 3, 37, 53, generator=g)
    x[:, :, 10:20, 15:30] += 2.0  # edges
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]]) / 8.0
    k = torch.stack([kx, kx.t()])[:, None]  # [2,1,3,3]
    C = x.shape[1]
    xp = F.pad(x, [1, 1, 1, 1], mode="replicate")
    gxy = F.conv2d(xp, k.repeat(C, 1, 1, 1), groups=C).reshape(2, C, 2, 37, 53)
    dense = torch.sqrt(gxy[:, :, 0] ** 2 + gxy[:, :, 1] ** 2 + 1e-6)
    mine = paste.sobel_magnitude(x)
    assert mine.shape == dense.shape and float((mine - dense).abs().max()) < 1e-6
    ramp = torch.arange(8.0)[None, None, None, :].repeat(1, 1, 8, 1)
    r = F.conv2d(F.pad(ramp, [1, 1, 1, 1], mode="replicate"), k)
    assert float(r[0, 0, 4, 4]) == 1.0 and float(r[0, 1, 4, 4]) == 0.0  # d/dx of a unit ramp = 1 after the /8 normalisation
    assert abs(float(paste.sobel_magnitude(ramp)[0, 0, 4, 4]) - float(np.sqrt(1 + 1e-6))) < 1e-6


def test_noise_pool_is_one_draw_sliced_per_layer(P):
    """noise_mode='random' for a whole pass: NoisePool draws every layer's noise in ONE randn call (the layers' execution order),
    applies every layer's strength in one multiply and hands each layer its slice — the values are randn(total) under the same
    seed, each layer's slice scaled by ITS strength, and a changed strength is noticed (host logic, CPU tensors)."""
    sg = P.stylegan2
    torch.manual_seed(0)
    net = sg.SynthesisNetwork(w_dim=32, img_resolution=32, img_channels=3, cond_mode="none", channel_base=256, channel_max=16, num_fp16_res=0)
    blocks = [getattr(net, f"b{r}") for r in net.block_resolutions]
    layers = [l for b in blocks for l in ([b.conv1] if b.in_channels == 0 else [b.conv0, b.conv1])]
    assert [l.resolution for l in layers] == [4, 8, 8, 16, 16, 32, 32]
    with torch.no_grad():
        for k, l in enumerate(layers):
            l.noise_strength.fill_(0.5 + k)
    pool = sg.NoisePool(layers)
    for N in (1, 2):
        total = sum(N * l.resolution ** 2 for l in layers)
        torch.manual_seed(7)
        raw = torch.randn(total)
        torch.manual_seed(7)
        pool.draw(N, torch.device("cpu"))
        o = 0
        for k, l in enumerate(layers):
            got = pool.take(l, N)
            n = N * l.resolution ** 2
            assert got.shape == (N, 1, l.resolution, l.resolution)
            assert torch.equal(got.reshape(-1), raw[o:o + n] * (0.5 + k))
            o += n
    with torch.no_grad():
        layers[2].noise_strength.fill_(100.0)  # (in-place: the version counter moves, the cached strengths are rebuilt)
    torch.manual_seed(7)
    pool.draw(1, torch.device("cpu"))
    assert float(pool.take(layers[2], 1).std()) > 50 and float(pool.take(layers[1], 1).std()) < 5


def test_cached_view_holds_the_rays_in_both_layouts(P):
    """make_view returns the rays of a view as [3,res,res] (the reference's force_rays layout) AND as [res^2,3] (what the renderer
    consumes, training/triplane.py:181-182 'b c h w -> b (h w) c'): the second must be exactly the rearranged first, for perspective
    and orthographic views."""
    cams = P.cameras
    for fov in (30.0, -1.0):
        label, o, d, of, df = cams.make_view(10.0, 35.0, 1.0, fov, 16, 0.7, torch.device("cpu"))
        assert label.shape == (25,) and o.shape == d.shape == (3, 16, 16) and of.shape == df.shape == (256, 3)
        assert torch.equal(o.permute(1, 2, 0).reshape(256, 3), of) and torch.equal(d.permute(1, 2, 0).reshape(256, 3), df)
        assert of.is_contiguous() and df.is_contiguous() and o.is_contiguous()


def test_one_launch_forms_of_the_view_glue_round_like_the_reference():
    """TriPlaneGenerator.synthesis writes `0.5 * (xyz + 1) * (-1, 1, -1)` (triplane.py:229) as addcmul(h, xyz, h), h = (-0.5, 0.5, -0.5),
    and `0.5 * image + 0.5` as add(0.5, image, alpha=0.5): one launch each.  x * 0.5 is exact, so each form rounds once, where the
    reference's rounds — the same bits for every finite float (checked here on values around the rounding-sensitive points)."""
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(200000, generator=g), torch.randn(100000, generator=g) * 1e-4 - 1.0, torch.randn(100000, generator=g) * 1e3,
                   torch.tensor([0.0, -1.0, 1.0, -1.0 + 2 ** -24, 1.0 - 2 ** -24, 3.4e38, -3.4e38, 1e-45, -1e-45, 1e-38])]).float()
    x = x[: (len(x) // 3) * 3].reshape(1, 3, -1, 1)
    sign = torch.tensor([-1.0, 1.0, -1.0])[None, :, None, None]
    ref = 0.5 * (x + 1) * sign
    h = sign * 0.5
    assert torch.equal(torch.addcmul(h, x, h), ref)
    assert torch.equal(torch.add(torch.tensor(0.5), x, alpha=0.5), 0.5 * x + 0.5)


def _oracle_stage_ops(monkeypatch, P, oracle, box_warp):
    """Replace the stand-alone stage operators of panic3d_amd.ops (each one HIP kernel) by the CPU oracle's restatement of the same
    stage, on CPU tensors: what is left of ImportanceRenderer.forward_staged is its HOST logic — the order of the stages, the sample
    points, where the noise goes, the masks, the merge — which can then run without a GPU."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    n = lambda x: x.detach().cpu().numpy()
    ops = P.ops

    def triplane_decode(planes, coords, mlp, opts, density_only=False):
        sigma, rgb = oracle.decode(n(planes), n(coords), tuple(n(m) for m in mlp), box_warp, plane_mode=int(opts.plane_mode),
                                   flags=int(opts.flags) & oracle.FLAG_FORCE_SIGMOID)
        return t(sigma), t(rgb)

    def composite(colors, densities, depths, white_back=True):
        rgb, depth, w = oracle.composite(n(colors), n(densities), n(depths), white_back=white_back)
        lead = tuple(colors.shape[:-2])
        return t(rgb).reshape(lead + (colors.shape[-1],)), t(depth).reshape(lead + (1,)), t(w).reshape(lead + (colors.shape[-2] - 1, 1))

    monkeypatch.setattr(ops, "planes_to_nhwc", lambda planes: planes)
    monkeypatch.setattr(ops, "triplane_decode", triplane_decode)
    monkeypatch.setattr(ops, "sample_stratified", lambda a, b, S, jit: t(oracle.sample_stratified(a, b, S, n(jit))))
    monkeypatch.setattr(ops, "composite", composite)
    monkeypatch.setattr(ops, "importance", lambda d, w, u: t(oracle.importance(n(d), n(w), n(u))[0]).reshape(tuple(d.shape[:2]) + (u.shape[-1], 1)))
    monkeypatch.setattr(ops, "unify_perm", lambda dc, df: t(oracle.unify_perm(n(dc), n(df))))


class _StageFC:
    def __init__(self, w, b, i, lr_mul=1.0):
        self.weight, self.bias, self.weight_gain, self.bias_gain = torch.from_numpy(w), torch.from_numpy(b), lr_mul / np.sqrt(i), lr_mul


def _cpu_decoder(raw, force_sigmoid=True, lr_mul=1.0):
    class Dec:
        pass
    d = Dec()
    d.force_sigmoid = force_sigmoid
    d.net = [_StageFC(raw[0], raw[1], 32, lr_mul), None, _StageFC(raw[2], raw[3], 64, lr_mul)]
    return d


def test_staged_path_host_logic_with_density_noise_vs_reference(P, oracle, monkeypatch):
    """The HOST side of ImportanceRenderer.forward_staged — which ImportanceRenderer.forward takes for rendering_options['density_noise']
    > 0 (renderer.py:276-277) — with every stage kernel replaced by the CPU oracle's restatement of that stage: against the REFERENCE's
    own render with its four draws captured (tests/golden/render_density_noise.npz), and without noise against the oracle's fused
    render of a fixture.  (The same path on the HIP kernels: tests/test_hip_parity.py.)"""
    g = T.load_golden("render_density_noise.npz")
    seed, res, Sc, Sf, dn = int(g["meta_seed"]), int(g["meta_res"]), int(g["meta_Sc"]), int(g["meta_Sf"]), float(g["meta_density_noise"])
    planes = T.make_planes(seed, 1, 256, 256, scale=4.0, smooth=8)
    assert T.checksum(planes) == str(g["planes_checksum"])
    raw = T.make_decoder_params(seed + 1, 1.0, float(g["meta_sigma_gain"]))
    ro = dict(T.RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf, density_noise=dn)
    _oracle_stage_ops(monkeypatch, P, oracle, ro["box_warp"])
    rend = P.ImportanceRenderer(use_triplane=True)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    out = rend(tt(planes), _cpu_decoder(raw), tt(g["rays_o"]), tt(g["rays_d"]), ro, triplane_crop=0.1, cull_clouds=0.5,
               jitter=tt(g["jitter"]), u=tt(g["u"]), density_noise_draws=(tt(g["noise_coarse"]), tt(g["noise_fine"])))
    for key, a, tol in zip(("feat", "depth", "wsum", "xyz"), out, (1e-4, 2e-5, 3e-5, 1e-4)):
        diff = np.abs(a.numpy().astype(np.float64) - g[key]).reshape(-1, a.shape[-1]).max(axis=1)
        assert (diff <= tol).mean() >= 0.995, (key, float(diff.max()))
    quiet = rend.forward_staged(tt(planes), _cpu_decoder(raw), tt(g["rays_o"]), tt(g["rays_d"]), dict(ro, density_noise=0),
                                triplane_crop=0.1, cull_clouds=0.5, jitter=tt(g["jitter"]), u=tt(g["u"]))
    assert float((out[0] - quiet[0]).abs().max()) > 1e-2  # the noise matters in this fixture
    # ... and without noise, a fixture with two views, binarised clouds and non-default limits: the oracle's fused render
    g2 = T.load_golden("render_variant_b.npz")
    inp = T.golden_render_inputs(g2)
    monkeypatch.undo()
    _oracle_stage_ops(monkeypatch, P, oracle, inp["ro"]["box_warp"])
    rend2 = P.ImportanceRenderer(use_triplane=bool(inp["ro"]["use_triplane"]))
    kw = {k: v for k, v in inp["kw"].items() if k != "force_sigmoid"}
    st = rend2.forward_staged(tt(inp["planes"]), _cpu_decoder(inp["raw_mlp"], inp["kw"]["force_sigmoid"], inp["lr_mul"]), tt(inp["rays_o"]),
                              tt(inp["rays_d"]), inp["ro"], jitter=tt(inp["jitter"]), u=tt(inp["u"]), **kw)
    ref = oracle.render(inp["planes"], inp["rays_o"], inp["rays_d"], inp["jitter"], inp["u"],
                        oracle.prescale_mlp(*inp["raw_mlp"], lr_mul=inp["lr_mul"]), oracle.make_opts(inp["ro"], **inp["kw"]))
    for key, a, b, tol in zip(("feat", "depth", "wsum", "xyz"), st, ref, (1e-4, 2e-5, 3e-5, 1e-4)):
        diff = np.abs(a.numpy().astype(np.float64) - b).reshape(-1, a.shape[-1]).max(axis=1)
        assert (diff <= tol).mean() >= 0.995, (key, float(diff.max()))


@pytest.mark.parametrize("tag", ["none", "cond", "cond2", "cond3", "cond4"])  # cond*: every branch of the conditioning glue
def test_generator_host_logic_vs_reference(P, monkeypatch, tag):
    """The HOST side of the StyleGAN2 backbone (stylegan2.py: mapping network, StylePlan, block wiring with the activation-image
    hand-overs, the PAniC-3D conditioning between blocks, constant noise) with every operator of panic3d_amd.ops replaced by its
    plain-PyTorch restatement (tests/p3d_torch_ops.py), on CPU: against the REFERENCE's own outputs for the five generator fixtures
    (tests/golden/make_golden_synthesis.py).  The same modules on the HIP kernels: tests/test_hip_synthesis.py."""
    import p3d_torch_ops
    sg = P.stylegan2
    p3d_torch_ops.install(monkeypatch, P.ops)
    g = T.load_golden(f"syn_generator_{tag}.npz")
    kw = dict(z_dim=64, c_dim=25, w_dim=64, img_resolution=32, img_channels=96, mapping_kwargs={"num_layers": 2},
              channel_base=2048, channel_max=64, num_fp16_res=0, conv_clamp=None, fused_modconv_default="inference_only")
    G = sg.Generator(cond_mode=str(g["cond_mode"]), **kw)
    G.load_state_dict({k[3:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("sd_")}, strict=True)
    G.eval()
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    cond = {k[5:]: tt(v) for k, v in g.items() if k.startswith("cond_") and k != "cond_mode"}
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    with torch.no_grad():
        ws = G.mapping(tt(g["z"]), tt(g["c"]), cond, truncation_psi=0.7, truncation_cutoff=4)
        assert np.abs(ws.numpy() - g["ws"]).max() < 1e-5
        assert np.abs(G.mapping(tt(g["z"]), tt(g["c"]), cond).numpy() - g["ws_psi1"]).max() < 1e-5
        img = G.synthesis(tt(g["ws"]), cond, noise_mode="const").numpy()
        assert img.shape == g["img"].shape and rel(img, g["img"]) < 1e-4, rel(img, g["img"])
        # the activation-image hand-overs are part of the wiring under test: without them the planes must not change
        monkeypatch.setattr(sg, "CONV_IMG", False)
        img2 = G.synthesis(tt(g["ws"]), cond, noise_mode="const").numpy()
        assert rel(img2, g["img"]) < 1e-4 and rel(img2, img) < 1e-5
        # random noise through the NoisePool: zero strengths in these fixtures? then the planes equal the constant-noise ones
        strengths = [float(m.noise_strength) for m in G.synthesis.modules() if isinstance(m, sg.SynthesisLayer)]
        img3 = G.synthesis(tt(g["ws"]), cond, noise_mode="random").numpy()
        assert np.isfinite(img3).all() and (any(strengths) or rel(img3, img2) < 1e-6)


def _oracle_render_op(monkeypatch, P, oracle):
    """ops.render (the fused renderer launch) replaced by the CPU oracle's render on CPU tensors — with shared planes (V views of one
    subject) and the per-view depth clamp handled the way the kernel's flags define them."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    n = lambda x: x.detach().cpu().numpy()

    def render(planes, rays_o, rays_d, jitter, u, mlp, opts, ray_tile_w=0, dumps=False, stats=None, per_view_clamp=False, ray_limits=None,
               rng_seed=None, weights_only=False):
        assert not dumps and rng_seed is None and ray_limits is None
        if weights_only:  # (ops.render: feat / xyz are not returned; wsum / depth are the full render's)
            full = render(planes, rays_o, rays_d, jitter, u, mlp, opts, ray_tile_w, per_view_clamp=per_view_clamp)
            return None, full[1], full[2], None
        oo = oracle.Opts(opts.coord_scale, opts.ray_start, opts.ray_end, opts.depth_delta, opts.crop_limit, opts.cull_thresh, opts.Sc, opts.Sf,
                         opts.plane_mode, int(opts.flags) & 31)
        N, R = rays_o.shape[0], rays_o.shape[1]
        pl, m = n(planes), tuple(n(x) for x in mlp)
        jit, uu = n(jitter).reshape(N, R, opts.Sc), (n(u).reshape(N, R, opts.Sf) if opts.Sf > 0 else None)
        if N > 1 and (per_view_clamp or pl.shape[0] == 1):  # every view its own call (= its own depth-clamp range), planes shared or not
            outs = [oracle.render(pl[min(i, pl.shape[0] - 1)][None], n(rays_o)[i][None], n(rays_d)[i][None], jit[i][None],
                                  None if uu is None else uu[i], m, oo) for i in range(N)]
            return tuple(t(np.concatenate([o[k] for o in outs])) for k in range(4))
        return tuple(t(a) for a in oracle.render(pl, n(rays_o), n(rays_d), jit, None if uu is None else uu.reshape(N * R, -1), m, oo))

    monkeypatch.setattr(P.ops, "render", render)


_TRI_RK = {"image_resolution": 512, "disparity_space_sampling": False, "clamp_mode": "softplus",
           "superresolution_module": "training.superresolution.SuperresolutionHybrid8XDC", "c_gen_conditioning_zero": False,
           "gpc_reg_prob": 0.5, "c_scale": 1.0, "superresolution_noise_mode": "none", "density_reg": 0.25,
           "density_reg_p_dist": 0.004, "reg_type": "l1", "decoder_lr_mul": 1.0, "sr_antialias": True, "white_back": True,
           "triplane_depth": 1, "use_triplane": 1, "tanh_rgb_output": False, "box_warp": 0.7, "ray_start": 0.5, "ray_end": 1.5,
           "depth_resolution": 12, "depth_resolution_importance": 12, "avg_camera_radius": 1.0, "avg_camera_pivot": [0, 0, 0]}
_TRI_KW = dict(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, sr_num_fp16_res=0,
               mapping_kwargs={"num_layers": 2}, rendering_kwargs=_TRI_RK,
               sr_kwargs={"channel_base": 32768, "channel_max": 512, "fused_modconv_default": "inference_only"},
               cond_mode="none", triplane_width=32, sr_channels_hidden=16, backbone_resolution=32, channel_base=1024,
               channel_max=32, fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None)


@pytest.mark.parametrize("fixture", ["syn_triplane_f", "syn_triplane_f_cond"])
def test_triplane_generator_f_host_logic_vs_reference(P, oracle, monkeypatch, fixture):
    """TriPlaneGenerator.f END TO END on CPU — seeds -> z -> ws -> planes -> renderer -> super-resolution, the dict-in / dict-out API of
    PAniC-3D (triplane.py:313-508) — with the synthesis operators replaced by their PyTorch restatements and the renderer launch by the
    CPU oracle: everything that is host logic in generator.py / stylegan2.py / cameras.py / renderer.py runs as shipped, against the
    REFERENCE's own G.f outputs (perspective + orthographic view; the conditioned generator that generate.py feeds).  The same call
    on the HIP kernels: tests/test_hip_synthesis.py."""
    import p3d_torch_ops
    from panic3d_amd.generator import TriPlaneGenerator
    p3d_torch_ops.install(monkeypatch, P.ops)
    _oracle_render_op(monkeypatch, P, oracle)
    monkeypatch.setattr(P.ops, "planes_to_nhwc", lambda planes: planes)
    g = T.load_golden(fixture + ".npz")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    cond_case = fixture.endswith("_cond")
    kw = dict(_TRI_KW, cond_mode=str(g["cond_mode"]), rendering_kwargs=dict(_TRI_RK, c_gen_conditioning_zero=True)) if cond_case else dict(_TRI_KW)
    G = TriPlaneGenerator(**kw)
    G.load_state_dict({k[3:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("sd_")}, strict=True)
    G.eval()
    G.set_force_sigmoid(True)
    G.set_render_exact(True)
    G._inject_draws = (tt(g["jitter"]), tt(g["u"]))
    if cond_case:
        cond = {k[5:]: tt(v) for k, v in g.items() if k.startswith("cond_") and k != "cond_mode"}
        x = dict(elevations=torch.tensor([5.0]), azimuths=torch.tensor([-30.0]), fovs=torch.tensor([30.0]), seeds=[7], cond=cond,
                 triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=16, noise_mode="const")
    else:
        x = dict(elevations=torch.tensor([0.0, 10.0]), azimuths=torch.tensor([20.0, 200.0]), fovs=torch.tensor([30.0, -1.0]), seeds=[3, 4],
                 cond={}, triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=16)
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))
    try:
        with torch.no_grad():
            out = G.f(x)
    finally:
        P.cameras.cached_view_clear()  # (CPU views must not stay in the process-wide cache)
    if not cond_case:
        assert np.abs(x["camera_params"].numpy() - g["camera_params"]).max() < 1e-6
    assert np.abs(x["ws"].numpy() - g["ws"]).max() < 1e-5
    assert rel(out["triplane"].numpy(), g["triplane"]) < 1e-4
    for k, tol in (("image_raw", 2e-3), ("image_weights", 2e-3), ("image_xyz", 2e-3)):
        d = np.abs(out[k].numpy() - g[k])
        assert d.max() < 20 * tol and d.mean() < tol, (k, d.max(), d.mean())
    d = np.abs(out["image"][..., ::4, ::4].numpy() - g["image_sub4"])
    assert out["image"].shape[1:] == (3, 512, 512) and d.mean() < 2e-3 and d.max() < 0.1, (d.mean(), d.max())


def _oracle_grid_ops(monkeypatch, P, oracle):
    """The two kernels of the density-grid query replaced by create_samples + the oracle's decoder and the oracle's activation pass;
    the iso-surface extractor by its C specification (oracle/p3d_oracle_mc.c)."""
    vol = P.volume
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))

    def grid_density(planes, grid_n, lo, hi, voxel_size, offsets, mlp, opts, crop_limit=None, skip_cropped=False, staged=None, fast=False):
        pts = vol.create_samples(grid_n, cube_length=_TRI_RK["box_warp"], lo=lo, hi=hi)[0]
        sigma, _ = oracle.decode(planes.numpy(), pts.numpy(), tuple(m.numpy() for m in mlp), _TRI_RK["box_warp"], plane_mode=int(opts.plane_mode),
                                 flags=int(opts.flags) & oracle.FLAG_FORCE_SIGMOID, density_only=True)
        if crop_limit is None:
            return t(sigma)
        lim = np.float32(crop_limit)
        return t(sigma), ((pts[..., 0].abs() > float(lim)) | (pts[..., 2].abs() > float(lim))).reshape(1, -1, 1)

    def marching_cubes(v, level, flip0=False, allow_degenerate=True):
        out = tuple(t(a) for a in oracle.marching_cubes(v.numpy(), level, flip0=flip0))
        return out if allow_degenerate else P.ops.drop_degenerate_faces(*out)

    monkeypatch.setattr(P.ops, "grid_density", grid_density)
    monkeypatch.setattr(P.ops, "sigma2density", lambda s, cropmask=None, cull=None: t(oracle.sigma2density(
        s.numpy(), None if cropmask is None else cropmask.numpy(), cull)))
    monkeypatch.setattr(P.ops, "marching_cubes", marching_cubes)


def test_density_grid_host_logic_vs_the_references_get_eg3d_volume(P, oracle, monkeypatch):
    """volume.density_grid / to_volume / create_samples — the host side of `_util/eg3d_metrics3d.py:94-183 get_eg3d_volume` — on CPU
    against the output of the reference's OWN get_eg3d_volume (tests/golden/volume_reference.npz): the grid query kernel is replaced
    by create_samples + the oracle's decoder, the activation pass by the oracle's, the backbone operators by their PyTorch
    restatements.  (On the HIP kernels: tests/test_hip_synthesis.py.)"""
    import p3d_torch_ops
    from panic3d_amd.generator import TriPlaneGenerator
    p3d_torch_ops.install(monkeypatch, P.ops)
    _oracle_render_op(monkeypatch, P, oracle)
    _oracle_stage_ops(monkeypatch, P, oracle, _TRI_RK["box_warp"])
    vol = P.volume
    _oracle_grid_ops(monkeypatch, P, oracle)
    gv, gt = T.load_golden("volume_reference.npz"), T.load_golden("syn_triplane_f.npz")
    G = TriPlaneGenerator(**_TRI_KW)
    G.load_state_dict({k[3:].replace("__", "."): torch.from_numpy(v) for k, v in gt.items() if k.startswith("sd_")}, strict=True)
    G.eval()
    G.set_force_sigmoid(True)
    N = int(gv["resolution"])
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))
    try:
        with torch.no_grad():
            x = dict(elevations=torch.zeros(1), azimuths=torch.zeros(1), seeds=[3], cond={}, neural_rendering_resolution=8)
            G.f(x)  # the reference obtains ws the same way (eg3d_metrics3d.py:101-109)
            ws = x["ws"]
            plain = vol.density_grid(G, ws, {}, resolution=N)
            masked = vol.density_grid(G, ws, {}, resolution=N, triplane_crop=0.1, cull_clouds=0.5)
            pts = vol.create_samples(N, cube_length=0.7)[0]
            rgb = G.sample_mixed(pts.contiguous(), None, ws, {}, noise_mode="const")["rgb"]
    finally:
        P.cameras.cached_view_clear()
    assert np.array_equal(vol.to_volume(pts, N).numpy(), gv["plain_coordinates"])
    sig = vol.to_volume(plain["sigmas"], N).numpy()
    assert sig.shape == gv["plain_sigmas"].shape == (1, 1, N, N, N) and rel(sig, gv["plain_sigmas"]) < 1e-3
    assert np.abs(vol.to_volume(plain["densities"], N).numpy() - gv["plain_densities"]).max() < 2e-4
    assert np.abs(vol.to_volume(rgb, N)[:, :3].numpy() - gv["plain_rgb3"]).max() < 2e-3
    dm, ref = vol.to_volume(masked["densities"], N).numpy(), gv["masked_densities"]
    same = (dm == -1e3) == (ref == -1e3)
    assert same.mean() > 0.995 and (ref == -1e3).mean() > 0.5  # thresholded masks: only boundary voxels may flip
    both = same & (ref != -1e3)
    assert both.sum() > 10 and np.abs(dm[both] - ref[both]).max() < 2e-4


def _torch_paste_front_op():
    """ops.paste_front (one fused kernel) as the torch formulation of the same post-process (triplane.py:607-691 with the
    restated Sobel of paste.py): resizes, masks, sampling of the illustration, lerp."""
    import torch.nn.functional as F
    from panic3d_amd import paste

    def paste_front_op(weights, xyz, occ, rays_o, rays_d, front, image, tw, te, to, td, bw, normalize_images):
        S = front.shape[-1]
        if len(front) == 1 and len(xyz) > 1:
            front = front.expand(len(xyz), -1, -1, -1)
        up = lambda t, mode="bilinear": F.interpolate(t, S, mode=mode)
        wmask = (up(weights) > tw).float()
        smask = (paste.sobel_magnitude(up(xyz)).norm(2, dim=1, keepdim=True) < te).float()
        fmask = up((occ < to).float())
        dmask = (up(paste.xyz_discrepancy(xyz, {"ray_origins": rays_o, "ray_directions": rays_d}), "nearest") < td).float()
        mask = wmask * smask * fmask * dmask
        pst = paste.sample_orthofront(front * 2 - 1 if normalize_images else front, up(xyz), bw)
        return dict(image=torch.lerp(image, pst, mask), paste=pst, mask=mask, mask_weights=wmask, mask_edges=smask, mask_occ=fmask, mask_dxyz=dmask)

    return paste_front_op


def _cpu_generator_env(monkeypatch, P, oracle):
    """Every device operator TriPlaneGenerator.f reaches, replaced by a CPU stand-in (PyTorch restatements of the synthesis operators,
    the CPU oracle for the renderer launch and the point decoder, the torch formulation of the paste): what then runs is the host
    logic of the package, as shipped."""
    import p3d_torch_ops
    p3d_torch_ops.install(monkeypatch, P.ops)
    _oracle_stage_ops(monkeypatch, P, oracle, _TRI_RK["box_warp"])
    _oracle_render_op(monkeypatch, P, oracle)
    monkeypatch.setattr(P.ops, "planes_to_nhwc", lambda planes: planes)
    monkeypatch.setattr(P.ops, "paste_front", _torch_paste_front_op())


def _cpu_fixture_generator(g, **over):
    from panic3d_amd.generator import TriPlaneGenerator
    G = TriPlaneGenerator(**dict(_TRI_KW, **over))
    G.load_state_dict({k[3:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("sd_")}, strict=True)
    G.eval()
    G.set_force_sigmoid(True)
    G.set_render_exact(True)
    return G


def test_f_many_views_of_one_subject_in_one_call_host_logic(P, oracle, monkeypatch):
    """Extension of the dict API, host side: ws / cond of batch 1 with V cameras = ONE backbone pass, ONE renderer call on shared
    planes with a depth clamp per view, batched super-resolution and paste.  Equal to V separate f() calls on the same ws and draws
    (CPU stand-ins for every device operator; the same on the HIP kernels: tests/test_hip_synthesis.py)."""
    _cpu_generator_env(monkeypatch, P, oracle)
    g = T.load_golden("syn_triplane_f.npz")
    G = _cpu_fixture_generator(g)
    V, res, S = 3, 16, 12
    R = res * res
    gen = torch.Generator().manual_seed(5)
    draws = [(torch.rand(V, R, S, 1, generator=gen), torch.rand(V * R, S, generator=gen)) for _ in range(2)]
    front = torch.rand(1, 3, 512, 512, generator=gen)
    el, az, fv = torch.tensor([0.0, 10.0, -5.0]), torch.tensor([0.0, 40.0, 200.0]), torch.tensor([-1.0, 30.0, 30.0])
    common = dict(seeds=[3], cond={"image_ortho_front": front}, triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=res,
                  noise_mode="const", paste_params={"mode": "default", "thresh_weight": 0.5, "thresh_edges": 0.2, "thresh_occ": 0.5,
                                                    "offset_occ": 0.01, "thresh_dxyz": 0.05})
    try:
        with torch.no_grad():
            with pytest.raises(RuntimeError):  # this fixture's generator is pose-conditioned: ws would differ per view
                G.f(dict(common, elevations=el, azimuths=az, fovs=fv))
            x0 = dict(common, elevations=el[:1], azimuths=az[:1], fovs=fv[:1], paste_params=None)
            G.f(x0)
            common["ws"] = x0["ws"]  # one subject: the same ws for every view
            G._inject_draws = [tuple(d) for d in draws]
            both = G.f(dict(common, elevations=el, azimuths=az, fovs=fv))
            assert G._inject_draws == [] and both["image"].shape == (V, 3, 512, 512) and both["triplane"].shape[0] == V
            for v in range(V):
                G._inject_draws = [(j[v:v + 1].contiguous(), u[v * R:(v + 1) * R].contiguous()) for j, u in draws]
                one = G.f(dict(common, elevations=el[v:v + 1], azimuths=az[v:v + 1], fovs=fv[v:v + 1]))
                for k in ("image_raw", "image_weights", "image_xyz", "image_depth"):  # depth too: one clamp range per view
                    assert torch.equal(both[k][v:v + 1], one[k]), (k, v)
                assert (both["image_prepaste"][v:v + 1] - one["image_prepaste"]).abs().max() < 1e-4
                assert ((both["paste"]["mask"][v:v + 1] - one["paste"]["mask"]).abs() > 1e-3).float().mean() < 1e-3
                assert (both["image"][v:v + 1] - one["image"]).abs().mean() < 1e-4
    finally:
        G._inject_draws = None
        P.cameras.cached_view_clear()


def test_paste_front_host_logic_vs_reference(P, oracle, monkeypatch):
    """f() with paste_params (generate.py:55-66) on CPU against the reference's paste_front (fixture syn_triplane_f.npz): the rays of
    the front-occlusion pass (built in one fma from the cached sign / shift tensors), the second renderer pass, the wiring of the
    masks — with the paste kernel replaced by the torch formulation of the same post-process, the renderer launch by the oracle and
    the synthesis operators by their PyTorch restatements.  (On the HIP kernels: tests/test_hip_synthesis.py.)"""
    import torch.nn.functional as F
    import p3d_torch_ops
    from panic3d_amd.generator import TriPlaneGenerator
    from panic3d_amd import paste
    p3d_torch_ops.install(monkeypatch, P.ops)
    _oracle_render_op(monkeypatch, P, oracle)
    monkeypatch.setattr(P.ops, "planes_to_nhwc", lambda planes: planes)
    monkeypatch.setattr(P.ops, "paste_front", _torch_paste_front_op())
    rays = {}
    real_render = P.ops.render

    def spy_render(planes, rays_o, rays_d, *a, **k):
        rays.setdefault("calls", []).append((rays_o.clone(), rays_d.clone()))
        return real_render(planes, rays_o, rays_d, *a, **k)

    monkeypatch.setattr(P.ops, "render", spy_render)
    g = T.load_golden("syn_triplane_f.npz")
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    G = TriPlaneGenerator(**_TRI_KW)
    G.load_state_dict({k[3:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("sd_")}, strict=True)
    G.eval()
    G.set_force_sigmoid(True)
    G.set_render_exact(True)
    G._inject_draws = [(tt(g["paste_draw0"]), tt(g["paste_draw1"])), (tt(g["paste_draw2"]), tt(g["paste_draw3"]))]
    front = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(11))
    xp = dict(elevations=torch.tensor([0.0]), azimuths=torch.tensor([0.0]), fovs=torch.tensor([-1.0]), seeds=[3],
              cond={"image_ortho_front": front}, triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=16,
              paste_params={"mode": "default", "thresh_weight": 0.5, "thresh_edges": 0.2, "thresh_occ": 0.5, "offset_occ": 0.01, "thresh_dxyz": 0.05})
    try:
        with torch.no_grad():
            out = G.f(xp)
    finally:
        P.cameras.cached_view_clear()
    assert G._inject_draws == [] and len(rays["calls"]) == 2  # both renderer passes ran
    # the occlusion pass's rays, bit for bit the reference's construction (triplane.py:565-571)
    ro = out["image_xyz"] * torch.tensor([-1, 1, -1])[None, :, None, None]
    ro[:, 2, :, :] -= _TRI_RK["ray_start"] - 0.01
    rd = torch.zeros_like(out["image_xyz"])
    rd[:, 2, :, :] = 1
    flat = lambda t_: t_.permute(0, 2, 3, 1).reshape(1, -1, 3)
    assert torch.equal(rays["calls"][1][0], flat(ro)) and torch.equal(rays["calls"][1][1], flat(rd))
    for k in ("mask", "mask_weights", "mask_edges", "mask_occ", "mask_dxyz"):
        a, b = out["paste"][k].numpy(), g["paste_" + k]
        assert a.shape == b.shape and (np.abs(a - b) > 1e-3).mean() < 0.01, k  # thresholded masks: <1 % boundary pixels
    sub = lambda t_: t_[..., ::4, ::4].numpy()
    assert np.abs(sub(out["paste"]["paste"]) - g["paste_paste_sub4"]).mean() < 2e-3
    assert np.abs(sub(out["image_prepaste"]) - g["paste_prepaste_sub4"]).mean() < 2e-3
    assert np.abs(sub(out["image"]) - g["paste_image_sub4"]).mean() < 5e-3


# ---- the memo layers in front of a G.f call (tests/p3d_shared_cases.py; the same cases run on the HIP kernels in tests/test_hip_synthesis.py) ----
def test_prepared_conditioning_follows_the_conditioning_tensors_cpu(P, oracle, monkeypatch):
    import p3d_shared_cases as MC
    _cpu_generator_env(monkeypatch, P, oracle)
    MC.prepared_conditioning_follows_the_conditioning_tensors(P, "cpu")


def test_style_plan_memo_follows_ws_and_parameters_cpu(P, oracle, monkeypatch):
    import p3d_shared_cases as MC
    _cpu_generator_env(monkeypatch, P, oracle)
    MC.style_plan_memo_follows_ws_and_parameters(P, "cpu")


def test_f_memoises_ws_only_while_nothing_the_mapping_reads_has_changed_cpu(P, oracle, monkeypatch):
    import p3d_shared_cases as MC
    _cpu_generator_env(monkeypatch, P, oracle)
    try:
        MC.f_memoises_ws_only_while_nothing_the_mapping_reads_has_changed(P, "cpu")
    finally:
        P.cameras.cached_view_clear()


@pytest.mark.parametrize("how", ["data", "dlpack"])
def test_one_switch_turns_every_memo_layer_off_and_hidden_writes_are_then_seen_cpu(P, oracle, monkeypatch, how):
    import p3d_shared_cases as MC
    _cpu_generator_env(monkeypatch, P, oracle)
    try:
        MC.one_switch_turns_every_memo_layer_off_and_hidden_writes_are_then_seen(P, how, "cpu")
    finally:
        P.cameras.cached_view_clear()


_GRID_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import panic3d_amd as P
from panic3d_amd import volume, ops
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N = {res}
# stand-ins for the two kernels of the grid query: sigma of grid point i is i (so the gathered grid shows every slab in its place)
ops.planes_to_nhwc = lambda planes: planes
def grid_density(planes, grid_n, lo, hi, vs, offsets, mlp, opts, crop_limit=None, skip_cropped=False, staged=None, fast=False):
    assert grid_n == N and 0 <= lo <= hi <= N ** 3 and lo % (N * N) == 0 and hi % (N * N) == 0  # whole slices of the slowest axis
    return torch.arange(lo, hi, dtype=torch.float32).reshape(1, -1, 1)
ops.grid_density = grid_density
ops.sigma2density = lambda s, cropmask=None, cull=None: s * 2
class FC:
    def __init__(self, o, i):
        self.weight, self.bias, self.weight_gain, self.bias_gain = torch.zeros(o, i), torch.zeros(o), 1.0, 1.0
class Dec:
    force_sigmoid = True
    net = [FC(64, 32), None, FC(33, 64)]
class G:
    rendering_kwargs = dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=12, depth_resolution_importance=12, white_back=True, use_triplane=1)
    decoder = Dec()
    renderer = P.ImportanceRenderer(use_triplane=True)
out = volume.density_grid_sharded(G, torch.zeros(1, 14, 512), {{}}, resolution=N, planes=torch.zeros(1, 3, 32, 8, 8))
if rank == 0:
    want = torch.arange(N ** 3, dtype=torch.float32).reshape(1, -1, 1)
    assert torch.equal(out["sigmas"], want) and torch.equal(out["densities"], want * 2), (out["sigmas"].shape,)
    print("GRID_OK")
else:
    assert out["sigmas"] is None and out["densities"] is None
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,res,port", [(2, 8, 29561), (3, 2, 29563)])
def test_density_grid_sharded_gloo(tmp_path, world, res, port):
    """BASELINE config c5 on N GPUs, host side (volume.density_grid_sharded): contiguous slabs of the slowest grid axis per rank, ONE
    gather of the sigma / density slabs to rank 0 — at world size 2, and with more ranks than grid slices (an empty slab still joins
    the gather).  The two kernels of the query are replaced by stand-ins whose output identifies the grid point."""
    script = tmp_path / "worker.py"
    script.write_text(_GRID_WORKER.format(root=ROOT, res=res))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GRID_OK" in r.stdout


def test_mesh_host_logic_equals_the_references_two_steps(P, oracle, monkeypatch):
    """volume.mesh (generate.py:97-103 without the volume leaving the device: the UN-flipped flat grid read reversed by the extractor,
    the colours decoded at exactly the vertices' voxels through create_samples' flat-index arithmetic) against the reference's two
    steps on the same numbers — get_eg3d_volume's flipped [N,N,N] volume and its full colour grid, then marching_cubes(vol, rgbs)
    indexing `rgbs[:3, a, b, c]` at `verts.astype(int)` (eg3d_metrics3d.py:166-177,186-210): same vertices, faces and colours.  CPU
    stand-ins: the oracle's decoder, activation pass and iso-surface specification."""
    _cpu_generator_env(monkeypatch, P, oracle)
    _oracle_grid_ops(monkeypatch, P, oracle)
    vol = P.volume
    g = T.load_golden("syn_triplane_f.npz")
    G = _cpu_fixture_generator(g)
    N = 20
    try:
        with torch.no_grad():
            x = dict(elevations=torch.zeros(1), azimuths=torch.zeros(1), seeds=[3], cond={}, neural_rendering_resolution=8)
            G.f(x)
            ws = x["ws"]
            planes = G._planes(ws, {}, noise_mode="const")
            dens = vol.density_grid(G, ws, {}, resolution=N, planes=planes)["densities"]
            level = float(dens.median())  # (a level this random-init volume crosses)
            fast = vol.mesh(G, ws, {}, resolution=N, level=level, planes=planes)
            pts = vol.create_samples(N, cube_length=_TRI_RK["box_warp"])[0]
            rgb = G.renderer.run_model(planes, G.decoder, pts.contiguous(), None, G.rendering_kwargs)["rgb"]
            two = vol.marching_cubes(vol.to_volume(dens, N)[0, 0].contiguous(), vol.to_volume(rgb, N)[0], _TRI_RK["box_warp"], level=level)
    finally:
        P.cameras.cached_view_clear()
    assert len(fast["faces"]) > 100 and set(fast) == {"verts", "faces", "normals", "values", "colors"}
    for k in ("verts", "faces", "normals", "values", "colors"):
        assert fast[k].shape == two[k].shape and np.array_equal(fast[k], two[k]), k
    assert np.abs(fast["verts"]).max() <= 0.35 + 1e-6 and fast["colors"].shape == (len(fast["verts"]), 3)


def test_mapping_zplus_takes_slot_i_of_the_ith_latent(P, oracle, monkeypatch):
    """TriPlaneGenerator.mapping_zplus (triplane.py:123-143): with one z PER w slot, slot i of the result is slot i of mapping(z_i) —
    also with resnet features in the conditioning — and the expanded-z shortcut of f() (one z for every slot) equals the general path."""
    _cpu_generator_env(monkeypatch, P, oracle)
    from panic3d_amd.generator import TriPlaneGenerator
    torch.manual_seed(9)
    G = TriPlaneGenerator(**dict(_TRI_KW, cond_mode="resnetcond_8")).eval()
    n = G.backbone.num_ws
    zs, c = torch.randn(2, n, 512), torch.randn(2, 25)
    cond = {"resnet_feats": torch.randn(2, 16)}
    with torch.no_grad():
        got = G.mapping_zplus(zs, c, cond, truncation_psi=0.8)
        for i in range(n):
            want = G.mapping(zs[:, i], c, cond, truncation_psi=0.8)[:, i]
            assert float((got[:, i] - want).abs().max()) < 1e-5, i
        one = torch.randn(2, 512)
        fast = G.mapping_zplus(one[:, None, :].expand(-1, n, -1), c, cond)
        slow = G.mapping_zplus(one[:, None, :].repeat(1, n, 1), c, cond)
        # (another batch size goes through another blocking of the same fp32 GEMM: round-off, not bits)
        assert float((fast - slow).abs().max()) < 1e-5 and fast.shape == (2, n, 512), float((fast - slow).abs().max())


@pytest.mark.parametrize("tag", ["none", "cond"])
def test_latent_injection_and_stop_level_vs_reference_cpu(P, oracle, monkeypatch, tag):
    import p3d_shared_cases as MC
    _cpu_generator_env(monkeypatch, P, oracle)
    MC.latent_injection_and_stop_level_vs_reference(P, tag, "cpu")


def test_f_options_vs_reference_cpu(P, oracle, monkeypatch):
    import p3d_shared_cases as MC
    _cpu_generator_env(monkeypatch, P, oracle)
    MC.f_options_vs_reference(P, "cpu")


@pytest.mark.skipif(not os.path.isdir("/root/reference/_train/eg3dc/src"), reason="needs the reference tree")
def test_noise_pool_off_reproduces_the_references_random_draws(P, monkeypatch):
    """with the pooled noise switched off (P3D_NOISE_POOL=0 / stylegan2.set_noise_pool(False) / net.noise_pool = False) every
    layer calls torch.randn itself in the reference's order (networks_stylegan2.py:342), so under the same torch seed the backbone with
    noise_mode='random' — what generate.py's G.f calls run — reproduces the REFERENCE's planes; with the pool on (the default) the
    same seed gives other noise values (one randn call per pass): same distribution, not the same stream."""
    import types
    import p3d_torch_ops
    sg = P.stylegan2
    os.environ.setdefault("PROJECT_DN", "/root/reference")
    os.environ.setdefault("PROJECT_NAME", "x")
    added = ["/root/reference", "/root/reference/_train/eg3dc/src"]
    sys.path[:0] = [added[0]]
    sys.path.append(added[1])
    sys.modules.setdefault("kornia", types.ModuleType("kornia"))
    try:
        from training.networks_stylegan2 import Generator as RefGenerator
    finally:
        for p_ in added:
            sys.path.remove(p_)
    p3d_torch_ops.install(monkeypatch, P.ops)
    g = T.load_golden("syn_generator_none.npz")
    kw = dict(z_dim=64, c_dim=25, w_dim=64, img_resolution=32, img_channels=96, mapping_kwargs={"num_layers": 2},
              channel_base=2048, channel_max=64, num_fp16_res=0, conv_clamp=None, fused_modconv_default="inference_only")
    sd = {k[3:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("sd_")}
    ours, ref = sg.Generator(cond_mode="none", **kw).eval(), RefGenerator(cond_mode="none", **kw).eval()
    ours.load_state_dict(sd, strict=True)
    ref.load_state_dict(sd, strict=True)
    with torch.no_grad():
        for G in (ours, ref):  # the fixture's noise strengths may be zero: make the random noise matter
            for n, p_ in G.named_parameters():
                if n.endswith("noise_strength"):
                    p_.fill_(0.3)
        ws = torch.from_numpy(g["ws_psi1"])
        torch.manual_seed(11)
        want = ref.synthesis(ws, {}, noise_mode="random")
        torch.manual_seed(12)
        other = ref.synthesis(ws, {}, noise_mode="random")
        assert float((want - other).abs().max()) > 1e-2  # the noise is visible in the planes
        prev = sg.set_noise_pool(False)  # the process-wide switch
        try:
            torch.manual_seed(11)
            got = ours.synthesis(ws, {}, noise_mode="random")
        finally:
            sg.set_noise_pool(prev)
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) < 2e-5 * scale  # the reference's draws, call for call
        ours.synthesis.noise_pool = False  # the per-network switch, with the process default back on
        torch.manual_seed(11)
        assert torch.equal(ours.synthesis(ws, {}, noise_mode="random"), got)
        ours.synthesis.noise_pool = None
        # (with the pool ON the CPU generator happens to give the same values too — its normal sampler works in blocks of 16 and every
        #  layer's count is a multiple of 16, so one randn(total) equals the layers' consecutive calls; the device generator (Philox,
        #  offset per call) does not have that property: tests/test_hip_synthesis.py checks there that the two modes differ)


def test_sobel_against_kornia(P):
    """f4: kornia.filters.sobel (training/triplane.py:632,652; kornia 0.6.5) vs paste.sobel_magnitude on the inputs of
    tests/golden/make_golden_sobel.py — compared when the fixture exists; FAILS with the generating command where kornia is importable
    and the fixture is not committed; skips (row stays UNPINNED) where neither is there."""
    from test_mcubes_cpu import _third_party_fixture
    (z,) = _third_party_fixture(["sobel_kornia.npz"], "kornia", "make_golden_sobel.py")
    from panic3d_amd import paste
    got = paste.sobel_magnitude(torch.from_numpy(z["x"]))
    assert got.shape == z["sobel"].shape and float((got - torch.from_numpy(z["sobel"])).abs().max()) < 1e-6, str(z["kornia_version"])


def test_paste_front_options_generate_py_does_not_use(P, oracle, monkeypatch):
    """`force_image` and `front_weight_erosion >= 1` (training/triplane.py:644-667; VERDICT r04 "missing" 7: they used to raise): both run on
    the torch formulation of the paste.  force_image: same masks, the pasted colours come from that image.  front_weight_erosion: one
    more render (the orthographic front view's `image_weights`, get_front_weights), its silhouette eroded by an e x e element and
    sampled (bilinearly) at the rendered xyz — a mask in [0, 1] that can only shrink the paste, and shrinks with e.  The erosion is kornia's definition for a
    flat element (geodesic border), checked against a brute-force minimum filter (PARITY UNPINNED against kornia itself, absent here)."""
    import torch.nn.functional as F
    from panic3d_amd import paste
    # erosion == brute force: window rows i - e//2 .. i - e//2 + e - 1, outside the image = +max (never the minimum)
    m = (torch.rand(1, 1, 9, 11, generator=torch.Generator().manual_seed(4)) > 0.3).float()
    for e in (1, 2, 3, 4):
        got, o = paste.erosion(m, e), e // 2
        want = torch.ones_like(m)
        for i in range(9):
            for j in range(11):
                win = [m[0, 0, a, b] for a in range(i - o, i - o + e) for b in range(j - o, j - o + e) if 0 <= a < 9 and 0 <= b < 11]
                want[0, 0, i, j] = min(win)
        assert torch.equal(got, want), e
    assert torch.equal(paste.erosion(m, 1), m)
    _cpu_generator_env(monkeypatch, P, oracle)
    g = T.load_golden("syn_triplane_f.npz")
    G = _cpu_fixture_generator(g)
    front = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(11))
    other = torch.rand(3, 512, 512, generator=torch.Generator().manual_seed(12))
    pp = {"mode": "default", "thresh_weight": 0.5, "thresh_edges": 0.2, "thresh_occ": 0.5, "offset_occ": 0.01, "thresh_dxyz": 0.05}
    mk = lambda **o: dict(elevations=torch.tensor([10.0]), azimuths=torch.tensor([25.0]), fovs=torch.tensor([-1.0]), seeds=[3],
                          cond={"image_ortho_front": front}, triplane_crop=0.1, cull_clouds=0.5, neural_rendering_resolution=16,
                          noise_mode="const", paste_params=dict(pp, **o))

    def run(**o):
        torch.manual_seed(5)  # the renderer's draws (torch.rand on the CPU stand-in) in the same order for every variant
        try:
            with torch.no_grad():
                return G.f(mk(**o))
        finally:
            P.cameras.cached_view_clear()
    base, forced, er1, er3 = run(), run(force_image=other), run(front_weight_erosion=1), run(front_weight_erosion=3)
    assert base["paste"]["mask"].mean() > 0.02  # something is pasted in this fixture
    # force_image: the masks are the view's, the paste is sampled from the other image
    assert torch.equal(forced["paste"]["mask"], base["paste"]["mask"]) and not torch.equal(forced["paste"]["paste"], base["paste"]["paste"])
    want = paste.sample_orthofront(other[None], F.interpolate(base["image_xyz"], 512, mode="bilinear"), G.rendering_kwargs["box_warp"])
    assert torch.allclose(forced["paste"]["paste"], want, atol=1e-6)
    assert torch.allclose(forced["image"], torch.lerp(base["image_prepaste"], want, base["paste"]["mask"]), atol=1e-6)
    # erosion: one more render, a front-weight mask in {0, 1}, monotone in e, only ever removing paste
    assert er1["paste"]["frontweight"].shape == base["image_weights"].shape and base["paste"]["frontweight"] is None
    fw1, fw3 = er1["paste"]["mask_frontweight"], er3["paste"]["mask_frontweight"]
    # (bilinear samples of a 0 / 1 silhouette: values in [0, 1]; the eroded silhouette is a subset, so the samples can only drop)
    assert float(fw1.min()) >= 0.0 and float(fw1.max()) <= 1.0 and bool((fw3 <= fw1 + 1e-6).all()) and fw3.sum() < fw1.sum()
    assert bool((er1["paste"]["mask"] <= base["paste"]["mask"]).all()) and bool((er3["paste"]["mask"] <= er1["paste"]["mask"] + 1e-6).all())
    assert torch.equal(er1["paste"]["mask"], base["paste"]["mask"] * fw1)
    # V views in ONE call with these options — the batch-1 front-weight mask and the forced image are expanded to the V views
    # (used to fail in grid_sample with a batch mismatch); every view equals its own single-view call
    try:
        with torch.no_grad():
            x0 = mk()
            x0["paste_params"] = None
            G.f(x0)
            ws = x0["ws"]
            el, az, fv = torch.tensor([10.0, 0.0]), torch.tensor([25.0, 60.0]), torch.tensor([-1.0, 30.0])
            for opt in (dict(force_image=other), dict(front_weight_erosion=2)):
                torch.manual_seed(5)
                both = G.f(dict(mk(**opt), ws=ws, elevations=el, azimuths=az, fovs=fv))
                assert both["paste"]["mask"].shape[0] == 2 and both["paste"]["paste"].shape == (2, 3, 512, 512)
                for v in range(2):
                    # (the renderer's draws differ between a V-view launch and a single one: the masks agree up to a few boundary pixels)
                    torch.manual_seed(5)
                    one = G.f(dict(mk(**opt), ws=ws, elevations=el[v:v + 1], azimuths=az[v:v + 1], fovs=fv[v:v + 1]))
                    assert ((both["paste"]["mask"][v:v + 1] - one["paste"]["mask"]).abs() > 1e-3).float().mean() < 8e-2, (opt.keys(), v)
                    if "force_image" in opt:
                        assert float((both["paste"]["paste"][v:v + 1] - one["paste"]["paste"]).abs().mean()) < 8e-2
    finally:
        P.cameras.cached_view_clear()
    # ... and with grad_sample the forced image receives a gradient through the sampling (the reference builds it outside no_grad, triplane.py:666-673)
    fi = other.clone().requires_grad_(True)
    xg = dict(mk(), paste_params=None)
    with torch.no_grad():
        out = G.f(xg)  # (fills xg's rays and camera parameters, as f() does for its caller)
    res = paste.paste_front_torch(G, xg, out, **dict(pp, force_image=fi, grad_sample=True))
    P.cameras.cached_view_clear()
    res["paste"].sum().backward()
    assert fi.grad is not None and float(fi.grad.abs().sum()) > 0


def test_run_model_with_density_noise(P, oracle, monkeypatch):
    """`run_model` with rendering_options['density_noise'] > 0 (renderer.py:276-277; used to raise): the decode, then
    `sigma += randn_like(sigma) * density_noise` with the caller's torch generator — the noise-free sigma plus exactly those draws."""
    _oracle_stage_ops(monkeypatch, P, oracle, 0.7)
    monkeypatch.setattr(P.ops, "planes_to_nhwc", lambda planes: planes)
    planes = torch.from_numpy(T.make_planes(5, 1, 32, 32))
    pts = torch.from_numpy(T.make_points(6, 1, 257, extent=0.4)).float()
    dec = _cpu_decoder(T.make_decoder_params(7), True, 1.0)
    rend = P.ImportanceRenderer(use_triplane=True)
    ro = dict(T.RENDERING_KWARGS)
    quiet = rend.run_model(planes, dec, pts, None, ro)
    torch.manual_seed(9)
    noisy = rend.run_model(planes, dec, pts, None, dict(ro, density_noise=0.5))
    torch.manual_seed(9)
    want = quiet["sigma"] + torch.randn_like(quiet["sigma"]) * 0.5
    assert torch.equal(noisy["sigma"], want) and torch.equal(noisy["rgb"], quiet["rgb"]) and float((noisy["sigma"] - quiet["sigma"]).abs().mean()) > 0.1


def test_prepare_rank_env_for_spawn_style_launchers(P, monkeypatch):
    """ranks made by torch.multiprocessing.spawn (or by setting RANK after the import) do not pass through the import-time
    hook; `sharding.prepare_rank_env()` gives them the dmabuf IPC setting RCCL needs on this driver — into os.environ (inherited by
    spawned children) or into an environment dict — and never overrides a value the user exported."""
    from panic3d_amd import sharding
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    env = {}
    assert sharding.prepare_rank_env(env) is True and env == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "1"}
    assert sharding.prepare_rank_env(env) is True and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"
    assert sharding.prepare_rank_env() is True and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"  # (no HIP runtime is up in the CPU suite)
