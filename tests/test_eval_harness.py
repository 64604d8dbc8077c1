"""CPU, build container only (skipped where /root/reference is absent): the checkpoint evaluation harness tools/eval_front.py,
end to end WITHOUT the released assets:

  * a small generator built by the REFERENCE's own class is pickled through the reference's own persistence
    (`torch_utils.persistence`, the format of `network-snapshot-*.pkl`: {'G', 'D', 'G_ema'}), and `eval_front.load_generator`
    loads it the way `_train/eg3dc/util/eg3dc_v0.py:25-62` does — `legacy.load_network_pkl`, this package's TriPlaneGenerator
    from the recorded init_args / init_kwargs, every tensor copied by name, 96+96, force_sigmoid;
  * a synthetic two-subject data root in the tool's prepared layout goes through `eval_front.run`: `G.f` per view with
    generate.py's inference options (device operators stood in on CPU: the machinery of tests/test_host_cpu.py), PNGs in
    generate.py's formats and paths, PSNR on the area-of-interest crop;
  * the PSNR function is measure.py:45's metric: equal to torchmetrics.PeakSignalNoiseRatio() where that package exists, and
    to the formula its source implements (range from the target of the call) everywhere.
"""
import json
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "_train", "eg3dc", "src")), reason="needs the reference tree")
sys.path.insert(0, os.path.join(ROOT, "tools"))

import eval_front  # noqa: E402
from test_dropin_reference import KWS, TRI_RK, ref_modules  # noqa: E402,F401  (the reference-import fixture)
from test_host_cpu import P, _cpu_generator_env  # noqa: E402,F401


def _snapshot(ref_triplane, path):
    """A network-snapshot pickle the way training_loop.py writes one: persistent classes, keys G / D / G_ema."""
    # conditioned like the released model is driven (generate.py:90-93 passes the front illustration and the ResNet "chonk" only)
    kw = dict(KWS[1], cond_mode="ortho_front.concatfront.inj_6b_4.crossavg_4.reschonk_add_8",
              rendering_kwargs=dict(TRI_RK, depth_resolution=8, depth_resolution_importance=8))
    torch.manual_seed(3)
    G = ref_triplane.TriPlaneGenerator(**kw).eval().requires_grad_(False)
    G.neural_rendering_resolution = 16
    with torch.no_grad():
        for n, p in G.backbone.synthesis.named_parameters():
            if n.endswith("torgb.weight"):
                p.mul_(30.0)
        G.decoder.net[2].weight[0] *= 20.0
    from training.networks_stylegan2 import FullyConnectedLayer  # any persistent nn.Module stands in for the discriminator
    with open(path, "wb") as fh:
        pickle.dump(dict(G=G, D=FullyConnectedLayer(4, 4), G_ema=G, training_set_kwargs=None, augment_pipe=None), fh)
    return G


def _data_root(root, names):
    from PIL import Image
    g = torch.Generator().manual_seed(0)
    for k, name in enumerate(names):
        d = os.path.join(root, name)
        os.makedirs(d)
        Image.fromarray((torch.rand(512, 512, 3, generator=g) * 255).to(torch.uint8).numpy()).save(os.path.join(d, "cond_image_ortho_front.png"))
        np.save(os.path.join(d, "resnet_chonk.npy"), torch.randn(64, 8, 8, generator=g).numpy())
        for view in ("front", "back"):
            rgba = (torch.rand(512, 512, 4, generator=g) * 255).to(torch.uint8).numpy()
            rgba[:64] = 0  # a transparent band: the white background of `.bg('w')` must show through
            Image.fromarray(rgba, "RGBA").save(os.path.join(d, f"gt_{view}.png"))
        json.dump([[40 + 8 * k, 100], [380, 300]], open(os.path.join(d, "roi.json"), "w"))
    open(os.path.join(root, "subjects.csv"), "w").write("\n".join(names) + "\n")


def test_psnr_is_measure_py_metric():
    g = torch.Generator().manual_seed(1)
    pred, gt = torch.rand(1, 3, 37, 29, generator=g), torch.rand(1, 3, 37, 29, generator=g) * 0.8
    want = 10 * np.log10(float(gt.max()) ** 2 / float(((pred.double() - gt.double()) ** 2).mean()))  # range = max(target) - min(0, ...)
    assert abs(eval_front.psnr_measure(pred, gt) - want) < 1e-9
    try:
        import torchmetrics
    except ImportError:
        torchmetrics = None
    if torchmetrics is not None:  # measure.py:45,119 literally
        assert abs(float(torchmetrics.PeakSignalNoiseRatio()(pred, gt)) - want) < 1e-4


def test_crop_on_white_is_the_reference_wrapper(ref_modules):
    """`I(img).crop(*roi).convert('RGBA').bg('w').convert('RGB').t()` (measure.py:116) through the reference's own image class."""
    try:
        from _util.twodee_v1 import I
    except Exception as e:  # noqa: BLE001 — its module imports half of the project's optional dependencies
        pytest.skip(f"_util.twodee_v1 not importable here: {type(e).__name__}: {e}")
    import _util.twodee_v1 as u2d
    if not hasattr(u2d, "TF"):  # its `crop` is torchvision's resized_crop; the module swallows the failed import
        pytest.skip("torchvision is not installed here: I.crop cannot run (the restatement is checked by the end-to-end test's geometry)")
    rgba = (torch.rand(512, 512, 4, generator=torch.Generator().manual_seed(2)) * 255).to(torch.uint8).numpy()
    roi = [[40, 100], [380, 300]]
    from PIL import Image
    want = I(Image.fromarray(rgba, "RGBA")).crop(*roi).convert("RGBA").bg("w").convert("RGB").t()
    got = eval_front.crop_on_white(rgba, roi)
    assert got.shape == want.shape and torch.equal(got, want.float())


def test_eval_harness_end_to_end_on_a_reference_pickle(ref_modules, tmp_path, P, oracle, monkeypatch):
    ref_triplane, misc = ref_modules
    pkl = str(tmp_path / "network-snapshot-000000.pkl")
    G_ref = _snapshot(ref_triplane, pkl)
    G = eval_front.load_generator(pkl, REF, device="cpu", depth_resolution=8, depth_resolution_importance=8)
    import panic3d_amd
    assert type(G) is panic3d_amd.generator.TriPlaneGenerator  # this package's class, not the unpickled one
    src = dict(misc.named_params_and_buffers(G_ref))
    for n, t in misc.named_params_and_buffers(G):
        assert torch.equal(t, src[n]), n
    assert G.neural_rendering_resolution == 16 and G.rendering_kwargs["depth_resolution"] == 8 and G.decoder.force_sigmoid is True
    # eg3dc_v0.py:55-56's default is what load_generator applies when not told otherwise
    assert eval_front.load_generator(pkl, REF, device="cpu").rendering_kwargs["depth_resolution_importance"] == 96

    data, out = str(tmp_path / "data"), str(tmp_path / "out")
    _data_root(data, ["subj_a", "subj_b"])
    _cpu_generator_env(monkeypatch, P, oracle)
    G.set_render_exact(True)  # (the CPU stand-in of the renderer launch is the oracle = the exact contract)
    rep = eval_front.run(G, data, out, subset="frontback", device="cpu")
    assert rep["subjects"] == 2 and set(rep["per_subject"]) == {"subj_a", "subj_b"} and rep["readme_front_psnr"] == 16.914
    from PIL import Image
    for name in ("subj_a", "subj_b"):
        for view in ("front", "back"):
            rgb, xyza = os.path.join(out, "ortho", name, view + ".png"), os.path.join(out, "ortho_xyza", name, view + ".png")
            assert Image.open(rgb).mode == "RGB" and Image.open(rgb).size == (512, 512) and Image.open(xyza).mode == "RGBA"
        # the printed number is measure.py's: PSNR of the written PNG against the ground truth on the ROI crop, on white
        roi = json.load(open(os.path.join(data, name, "roi.json")))
        pred = eval_front.crop_on_white(np.asarray(Image.open(os.path.join(out, "ortho", name, "front.png"))), roi)
        gt = eval_front.crop_on_white(np.asarray(Image.open(os.path.join(data, name, "gt_front.png"))), roi)
        assert pred.shape == (3, 380, 300)
        assert abs(rep["per_subject"][name]["front"] - eval_front.psnr_measure(pred, gt)) < 1e-9
        assert 0 < rep["per_subject"][name]["front"] < 20 and "back" in rep["per_subject"][name]  # random images: a few dB
    assert abs(rep["psnr_front"] - np.mean([rep["per_subject"][n]["front"] for n in rep["per_subject"]])) < 1e-9
    assert abs(rep["psnr_front_minus_readme"] - (rep["psnr_front"] - 16.914)) < 1e-9
    # the two subjects were conditioned on different illustrations: different renders
    a, b = (np.asarray(Image.open(os.path.join(out, "ortho", n, "front.png"))) for n in ("subj_a", "subj_b"))
    assert not np.array_equal(a, b)


def test_eval_views_are_generate_pys():
    v = eval_front.eval_views("all")
    assert [x[1] for x in v[:4]] == ["front", "left", "right", "back"] and len(v) == 16 and all(x[4] == -1 for x in v[:4])
    assert v[4][1] == "0042" and v[4][4] == 30.0 and v[-1][1] == "0041"
    assert eval_front.eval_views("front") == [("camO", "front", 0.0, 0.0, -1.0)]
