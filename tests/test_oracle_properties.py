"""Size-independent properties of the renderer, checked on the CPU oracle (the GPU path is bit-identical to it, so they
carry over): they hold for any input size and pin the oracle's semantics beyond the reference's golden vectors."""
import numpy as np
import pytest

import p3d_testing as T


def scene(seed, N=1, R=96, Sc=20, Sf=24, **ro_kw):
    rng = np.random.default_rng(seed)
    ro = dict(T.RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf, **ro_kw)
    planes = T.make_planes(seed, N, 40, 56, scale=3.0, smooth=8)
    raw = T.make_decoder_params(seed + 1, 1.0, 30.0)
    o = rng.standard_normal((N, R, 3)); o /= np.linalg.norm(o, axis=-1, keepdims=True)
    d = rng.uniform(-0.45, 0.45, (N, R, 3)) - o; d /= np.linalg.norm(d, axis=-1, keepdims=True)  # some rays miss the crop box
    jit, u = T.make_random_draws(seed + 2, N, R, Sc, Sf)
    return dict(ro=ro, planes=planes, raw=raw, o=o.astype(np.float32), d=d.astype(np.float32), jit=jit, u=u)


def render(oracle, s, ro=None, **kw):
    kw = {"force_sigmoid": True, **kw}
    return oracle.render(s["planes"], s["o"], s["d"], s["jit"], s["u"], oracle.prescale_mlp(*s["raw"]), oracle.make_opts(ro or s["ro"], **kw))


def test_rays_are_independent_up_to_the_depth_clamp(oracle):
    """Permuting the rays permutes feat / wsum / xyz exactly (the only cross-ray term is the global depth clamp)."""
    s = scene(1)
    R = s["o"].shape[1]
    perm = np.random.default_rng(0).permutation(R)
    a = render(oracle, s, triplane_crop=0.1, cull_clouds=0.5)
    Sf = s["ro"]["depth_resolution_importance"]
    s2 = dict(s, o=s["o"][:, perm], d=s["d"][:, perm], jit=s["jit"][:, perm], u=s["u"].reshape(1, R, Sf)[:, perm].reshape(R, Sf))
    b = render(oracle, s2, triplane_crop=0.1, cull_clouds=0.5)
    for k in (0, 2, 3):
        assert np.array_equal(a[k][:, perm], b[k])
    assert np.array_equal(a[1][:, perm], b[1])  # same ray set -> same global [min t, max t]
    half = dict(s, o=s["o"][:, :R // 2], d=s["d"][:, :R // 2], jit=s["jit"][:, :R // 2], u=s["u"][:R // 2])
    c = render(oracle, half, triplane_crop=0.1, cull_clouds=0.5)
    for k in (0, 2, 3):
        assert np.array_equal(a[k][:, :R // 2], c[k])


def test_weights_depth_and_background_relations(oracle):
    s = scene(2)
    wb = render(oracle, s, dict(s["ro"], white_back=True), triplane_crop=0.1, cull_clouds=0.5)
    nb = render(oracle, s, dict(s["ro"], white_back=False), triplane_crop=0.1, cull_clouds=0.5)
    W = wb[2]
    assert np.array_equal(W, nb[2]) and (W >= 0).all() and (W <= 1 + 1e-5).all() and 0.02 < W.mean() < 0.98
    # white background adds (1 - W) before the [-1,1] rescale: out_wb - out_nb = 2 (1 - W) (ray_marcher.py:52-55)
    assert np.abs((wb[0] - nb[0]) - 2 * (1 - W)).max() < 2e-6
    assert np.abs((wb[3] - nb[3]) - 2 * (1 - W)).max() < 2e-6
    ro = s["ro"]
    lo, hi = ro["ray_start"], ro["ray_end"] + (ro["ray_end"] - ro["ray_start"]) / (ro["depth_resolution"] - 1)
    assert (wb[1] >= lo - 1e-6).all() and (wb[1] <= hi + 1e-6).all()  # clamped into the range of the sampled depths


def test_masks_remove_density_exactly(oracle):
    s = scene(3)
    # a crop tighter than the scene: every sample is masked by position -> no weight at all, pure background
    out = render(oracle, s, triplane_crop=s["ro"]["box_warp"] / 2 - 1e-4)
    assert (out[2] == 0).all() and (out[0] == 1.0).all()  # white_back: 2 * (0 + 1 - 0) - 1
    # a cull threshold above 1 masks everything (the activated density 1 - exp(-softplus(sigma - 1)) never exceeds 1)
    out = render(oracle, s, cull_clouds=1.5)
    assert (out[2] == 0).all()
    # binarize: densities become +-1000 -> weights are 0 or saturate; rays either see nothing or are fully opaque
    out = render(oracle, s, binarize_clouds=0.5)
    W = out[2]
    assert (((W == 0) | (W > 0.999)).mean() > 0.9) and (W > 0.999).any() and (W == 0).any()


def test_single_pass_branch_is_the_coarse_composite(oracle):
    """N_importance == 0 (renderer.py:254-259, BASELINE config c1): one composite over the stratified samples."""
    s = scene(4, Sf=0)
    out = render(oracle, s)
    Sc = s["ro"]["depth_resolution"]
    t = oracle.sample_stratified(s["ro"]["ray_start"], s["ro"]["ray_end"], Sc, s["jit"])
    pts = s["o"][:, :, None, :] + t.reshape(1, -1, Sc, 1) * s["d"][:, :, None, :]
    sigma, rgb = oracle.decode(s["planes"], pts.reshape(1, -1, 3), oracle.prescale_mlp(*s["raw"]), s["ro"]["box_warp"], plane_mode=1)
    R = s["o"].shape[1]
    col = np.concatenate([rgb.reshape(1, R, Sc, 32), pts.reshape(1, R, Sc, 3).astype(np.float32)], -1)
    C, D, Wts = oracle.composite(col, sigma.reshape(1, R, Sc, 1), t.reshape(1, R, Sc, 1), white_back=True)
    C, Wts = np.asarray(C).reshape(1, R, 35), np.asarray(Wts).reshape(1, R, Sc - 1)
    assert np.array_equal(out[0], C[..., :32]) and np.array_equal(out[3], C[..., 32:])
    assert np.abs(out[2][..., 0] - Wts.sum(-1)).max() < 1e-6
