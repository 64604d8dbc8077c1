#!/usr/bin/env python3
"""Checkpoint-level evaluation on the MI355X path: load a released PAniC-3D pickle into THIS package's generator, render the
evaluation views of `_scripts/eval/generate.py` per subject, and score the front view the way `_scripts/eval/measure.py` does —
PSNR on the area-of-interest crop — printed beside the reference's published number (readme.md:83: front PSNR 16.914).

    # 1. once, where the reference tree + its assets live (the illustration -> render and the ResNet feature networks are the
    #    reference's own pre-processing, outside this package): write the per-subject inputs and ground truth as plain files
    python tools/eval_front.py prepare --reference /path/to/panic3d-anime-reconstruction --data eval_data [--limit N]
    # 2. on the MI355X box
    python tools/eval_front.py run --network <network-snapshot.pkl> --reference /path/to/panic3d-anime-reconstruction \\
        --data eval_data --out temp/eval/ecrutileE_eclustrousC_n120-00000-000200 [--subset front|all] [--mesh]

`--reference` is needed by `run` only to UNPICKLE: a StyleGAN-style snapshot carries its classes through
`torch_utils.persistence`, so `dnnlib`, `legacy` and `torch_utils` must be importable (`<reference>/_train/eg3dc/src`).  The
networks that then run are this package's: `legacy.load_network_pkl` -> `TriPlaneGenerator(*G.init_args, **G.init_kwargs)` ->
every parameter and buffer copied by name (require_all) -> `neural_rendering_resolution`, `rendering_kwargs`, force_sigmoid,
96 + 96 samples — `_train/eg3dc/util/eg3dc_v0.py:25-62` line by line.

Prepared data layout (`--data`): `subjects.csv` (one name per line) and per subject `<name>/`
    cond_image_ortho_front.png   RGB 512^2: the line-removed illustration on white      (generate.py:91 `image_ortho_front`)
    resnet_chonk.npy             float32: `x['resnet_features'][0]`                       (generate.py:92 `resnet_chonk`)
    gt_front.png [gt_back.png]   RGBA 512^2 ground-truth orthographic renders            (measure.py:116,122)
    roi.json                     [[top, left], [height, width]] = aligndata[bn]['area_of_interest'] (measure.py:108)

Status: the checkpoint `ecrutileE_eclustrousC_n120` and the AnimeRecon data are NOT in this environment, so the published
16.914 has not been reproduced here; what is tested (tests/test_eval_harness.py, CPU) is the machinery — a reference-built
generator pickled through the reference's own persistence, loaded by `load_generator`, two synthetic subjects rendered through
`G.f` with the device operators stood in, files written in generate.py's formats, PSNR equal to measure.py's metric.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

README_FRONT_PSNR = 16.914  # readme.md:83 (AnimeRecon, front view)
# generate.py:55-66
INFERENCE_OPTS = {"triplane_crop": 0.1, "cull_clouds": 0.5,
                  "paste_params": {"mode": "default", "thresh_weight": 0.95, "thresh_edges": 0.02, "thresh_occ": 0.05, "offset_occ": 0.01,
                                   "thresh_dxyz": 0.000005}}


def eval_views(subset="front"):
    """(camera kind, view name, elevation, azimuth, fov) of generate.py:108-117; 'front' = the view measure.py:116-119 scores."""
    views = [("camO", "front", 0.0, 0.0, -1.0)]
    if subset in ("frontback", "all"):
        views.append(("camO", "back", 0.0, 180.0, -1.0))
    if subset == "all":
        views[1:1] = [("camO", "left", 0.0, 90.0, -1.0), ("camO", "right", 0.0, -90.0, -1.0)]
        # dklustr.cam60 = meshgrid(elev linspace(60,-20,5), azim linspace(-180,150,12)); camsubs['spin12'] = 42..47, 36..41
        cam60 = np.stack(np.meshgrid(np.linspace(60, -20, 5), np.linspace(-180, 150, 12))).T.reshape(60, -1)
        views += [("camP", f"{v:04d}", float(cam60[v][0]), float(cam60[v][1]), 30.0) for v in [*range(42, 48), *range(36, 42)]]
    return views


# ---- the checkpoint ------------------------------------------------------------------------------------------------------------
def _unpickle_path(reference):
    src = os.path.join(reference, "_train", "eg3dc", "src")
    if not os.path.isdir(src):
        raise SystemExit(f"eval_front: {src} not found — --reference must be the root of the reference repository (needed to unpickle)")
    if src not in sys.path:
        sys.path.append(src)


def load_generator(network_pkl, reference, device="cuda", force_sigmoid=True, depth_resolution=96, depth_resolution_importance=96):
    """`load_eg3dc_model` (_train/eg3dc/util/eg3dc_v0.py:25-62) with the re-instantiated class taken from THIS package."""
    import panic3d_amd
    _unpickle_path(reference)
    import legacy  # the reference's (eg3dc_v0.py:14, 40-41)
    with open(network_pkl, "rb") as fp:
        data = legacy.load_network_pkl(fp)
    G_src = data["G_ema"].requires_grad_(False)
    G = panic3d_amd.generator.TriPlaneGenerator(*G_src.init_args, **G_src.init_kwargs).eval().requires_grad_(False)  # :47
    src = dict(list(G_src.named_parameters()) + list(G_src.named_buffers()))
    dst = dict(list(G.named_parameters()) + list(G.named_buffers()))
    missing, unused = sorted(set(dst) - set(src)), sorted(set(src) - set(dst))
    if missing or unused:  # misc.copy_params_and_buffers(..., require_all=True) (:49), both ways
        raise RuntimeError(f"checkpoint and generator disagree: missing in the pickle {missing[:5]}, not taken {unused[:5]}")
    with torch.no_grad():
        for name, t in dst.items():
            t.copy_(src[name].detach())
    G.neural_rendering_resolution = G_src.neural_rendering_resolution  # :50
    G.rendering_kwargs = dict(G_src.rendering_kwargs)                  # :51
    if force_sigmoid:
        G.set_force_sigmoid(True)                                      # :53-54 (generate.py:52)
    G.rendering_kwargs["depth_resolution"] = depth_resolution          # :55-56
    G.rendering_kwargs["depth_resolution_importance"] = depth_resolution_importance
    return G.to(device)


# ---- images the way the reference's wrapper handles them (_util/twodee_v1.py) ------------------------------------------------------
def _load_png(fn):
    from PIL import Image
    return np.asarray(Image.open(fn))


def crop_on_white(img_u8, roi):
    """`I(img).crop(*roi).convert('RGBA').bg('w').convert('RGB').t()` (measure.py:116-117; twodee_v1.py:272-294,533-534):
    rows roi[0][0] : +roi[1][0], columns roi[0][1] : +roi[1][1] (torchvision's crop: zero padding outside), alpha-composited over
    white with PIL's integer arithmetic, as float RGB [3,h,w] in [0,1]."""
    from PIL import Image
    (top, left), (h, w) = [[int(v) for v in p] for p in roi]
    im = Image.fromarray(np.asarray(img_u8)).convert("RGBA")
    im = im.crop((left, top, left + w, top + h))  # PIL pads with zeros (transparent) outside, like TF.crop
    white = Image.new("RGBA", im.size, (255, 255, 255, 255))
    rgb = np.asarray(Image.alpha_composite(white, im).convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(rgb).permute(2, 0, 1).contiguous()


def psnr_measure(pred, target):
    """measure.py:45 `torchmetrics.PeakSignalNoiseRatio()` called as `psnr(pred_rgb, gt_rgb)` (:119): default data_range=None, so
    the range comes from the TARGET of that call with the metric's zero-initialised trackers — max(target.max(), 0) -
    min(target.min(), 0) — base 10, mean squared error over all elements."""
    pred, target = pred.double(), target.double()
    mse = torch.mean((pred - target) ** 2)
    rng = torch.clamp(target.max(), min=0.0) - torch.clamp(target.min(), max=0.0)
    return float(10.0 * torch.log10(rng ** 2 / mse))


# ---- data ----------------------------------------------------------------------------------------------------------------------------
def read_subjects(data):
    with open(os.path.join(data, "subjects.csv")) as fh:
        return [l.strip() for l in fh if l.strip() and not l.startswith("#")]


def load_subject(data, name, device):
    d = os.path.join(data, name)
    front = _load_png(os.path.join(d, "cond_image_ortho_front.png"))
    if front.ndim != 3 or front.shape[2] < 3:
        raise RuntimeError(f"{name}: cond_image_ortho_front.png must be RGB")
    cond = {"image_ortho_front": torch.from_numpy(front[..., :3].astype(np.float32) / 255.0).permute(2, 0, 1)[None].contiguous().to(device),
            "resnet_chonk": torch.from_numpy(np.load(os.path.join(d, "resnet_chonk.npy")).astype(np.float32))[None].to(device)}
    with open(os.path.join(d, "roi.json")) as fh:
        roi = json.load(fh)
    return cond, roi


def view_files(out, name, cm, view):
    """generate.py:132-141: bn 'daredemoE/fandom_align/<id>/front' -> .../ortho/<id>/<view>.png etc.; here `name` plays <id>."""
    kind = {"camO": ("ortho", "ortho_xyza"), "camP": ("rgb60", "xyza60")}[cm]
    return os.path.join(out, kind[0], name, view + ".png"), os.path.join(out, kind[1], name, view + ".png")


# ---- the per-subject loop of generate.py:80-151 ----------------------------------------------------------------------------------------
def render_subject(G, cond, name, out, subset="front", seed=0, mesh=False, device="cuda"):
    from panic3d_amd import outputs, volume
    bw = G.rendering_kwargs["box_warp"]
    files = {}
    with torch.no_grad():
        if mesh:  # generate.py:86-105; the latent the way get_eg3d_volume takes it: one G.f call fills x['ws'] (eg3d_metrics3d.py:101-109)
            xin = {"elevations": torch.zeros(1, device=device), "azimuths": torch.zeros(1, device=device), "cond": cond, "seeds": [seed],
                   **INFERENCE_OPTS}
            G.f(xin)
            mc = volume.mesh(G, xin["ws"], cond, resolution=256, level=0.5, triplane_crop=INFERENCE_OPTS["triplane_crop"],
                             cull_clouds=INFERENCE_OPTS["cull_clouds"])
            files["mesh"] = outputs.dump_mesh(mc, os.path.join(out, "marching_cubes", name, "front.pkl"))
        for cm, view, elev, azim, fov in eval_views(subset):
            xin = {"elevations": elev * torch.ones(1, device=device), "azimuths": azim * torch.ones(1, device=device),
                   "fovs": fov * torch.ones(1, device=device), "cond": cond, "seeds": [seed], **INFERENCE_OPTS}
            o = G.f(xin, return_more=True)  # generate.py:130
            files[view] = outputs.save_view(o, *view_files(out, name, cm, view), bw)  # :143-148
    return files


def score_subject(data, name, out, roi):
    """measure.py:108-125, PSNR only (LPIPS / CLIP need networks that are not in this environment)."""
    res = {}
    (r0, c0), (h, w) = roi
    rois = {"front": roi, "back": [[r0, 512 - (c0 + w)], [h, w]]}  # measure.py:110 `roi_back`
    for view in ("front", "back"):
        gt_fn = os.path.join(data, name, f"gt_{view}.png")
        pred_fn = view_files(out, name, "camO", view)[0]
        if os.path.exists(gt_fn) and os.path.exists(pred_fn):
            res[view] = psnr_measure(crop_on_white(_load_png(pred_fn), rois[view]), crop_on_white(_load_png(gt_fn), rois[view]))
    return res


def run(G, data, out, subset="front", mesh=False, limit=None, device="cuda", seed=0):
    names = read_subjects(data)[:limit]
    per, fp32_subjects = {}, []
    for name in names:
        cond, roi = load_subject(data, name, device)
        # The default convolutions run on two-term f16 operands, exact to fp32 class for |s*x| <= 4094 and SATURATING beyond; nothing
        # bounds a released checkpoint's activations (conv_clamp=None, train_eclustrousC.py:554).  Every subject is therefore watched:
        # the flag word is reset before, read after, and a subject that tripped it is rendered AGAIN on fp32 operands and reported.
        G.watch_conv_domain(device)
        G.conv_domain_violated(reset=True)
        render_subject(G, cond, name, out, subset=subset, seed=seed, mesh=mesh, device=device)
        if G.conv_domain_violated(reset=True):
            G.set_conv_mma("f32")
            try:
                render_subject(G, cond, name, out, subset=subset, seed=seed, mesh=mesh, device=device)
            finally:
                G.set_conv_mma(None)
            fp32_subjects.append(name)
        per[name] = score_subject(data, name, out, roi)
    rep = {"subjects": len(names), "subset": subset, "out": out, "per_subject": per,
           "readme_front_psnr": README_FRONT_PSNR, "readme_source": "readme.md:83 (AnimeRecon, front)",
           "conv_operands": {"default": "two-term f16 (fp32-class inside |s*x| <= 4094)", "subjects_rerun_on_fp32_operands": fp32_subjects,
                             "note": "a subject listed here left the two-term domain; its files and scores come from the fp32-operand run"}}
    for view in ("front", "back"):
        vals = [p[view] for p in per.values() if view in p]
        if vals:
            rep[f"psnr_{view}"] = float(np.mean(vals))  # measure.py's table averages over subjects
    if "psnr_front" in rep:
        rep["psnr_front_minus_readme"] = rep["psnr_front"] - README_FRONT_PSNR
    return rep


# ---- preparation on a box that has the reference tree and its assets (generate.py:27-96; NOT runnable here) ---------------------------
def prepare(reference, data, limit=None, device="cuda"):
    """Write the prepared layout with the reference's own data backend and pre-processing networks.  Mirrors generate.py:27-47,68-96
    and measure.py:20-27,108-122; run it from the root of the reference repository's environment."""
    os.chdir(reference)
    sys.path[:0] = [reference]
    import _util.util_v1 as uutil
    from _databacks import lustrous_renders_v1 as dklustr
    from _train.img2img.util import rmline_wrapper
    from _train.danbooru_tagger.helpers.katepca import ResnetFeatureExtractorPCA
    dk = dklustr.DatabackendMinna()
    bns = [f"daredemoE/fandom_align/{bn}/front" for bn in uutil.read_bns("./_data/lustrous/subsets/daredemoE_test.csv")][:limit]
    aligndata = uutil.pload("./_data/lustrous/renders/daredemoE/fandom_align_alignment.pkl")
    rmline_model = rmline_wrapper.RMLineWrapper(("rmlineE_rmlineganA_n04", 199)).eval().to(device)
    resnet = ResnetFeatureExtractorPCA("./_data/lustrous/preprocessed/minna_resnet_feats_ortho/pca.pkl", 512).eval().to(device)
    names = []
    for bn in bns:
        x = dk[bn]
        with torch.no_grad():
            feats = resnet(x.image)
            kp = rmline_wrapper._apply_M_keypoints(aligndata[bn]["transformation"], aligndata[bn]["_alignment"]["source"]["keypoints"][
                aligndata[bn]["_alignment"]["source"]["_detection_used"]][None, ])[0, :, :2]
            img = rmline_model(x.image, kp)
        name = bn.split("/")[2]
        d = os.path.join(data, name)
        os.makedirs(d, exist_ok=True)
        img.bg("w").convert("RGB").save(os.path.join(d, "cond_image_ortho_front.png"))
        np.save(os.path.join(d, "resnet_chonk.npy"), feats[0].float().cpu().numpy())
        for view in ("front", "back"):
            dk[bn.replace("fandom_align", "ortho").replace("/front", "/" + view)].image.convert("RGBA").save(os.path.join(d, f"gt_{view}.png"))
        with open(os.path.join(d, "roi.json"), "w") as fh:
            json.dump([[int(v) for v in p] for p in aligndata[bn]["area_of_interest"]], fh)
        names.append(name)
    with open(os.path.join(data, "subjects.csv"), "w") as fh:
        fh.write("\n".join(names) + "\n")
    return names


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    pr = sub.add_parser("prepare")
    pr.add_argument("--reference", required=True)
    pr.add_argument("--data", required=True)
    pr.add_argument("--limit", type=int)
    rn = sub.add_parser("run")
    rn.add_argument("--network", required=True)
    rn.add_argument("--reference", required=True, help="root of the reference repository (unpickling needs dnnlib / legacy / torch_utils)")
    rn.add_argument("--data", required=True)
    rn.add_argument("--out", required=True)
    rn.add_argument("--subset", choices=("front", "frontback", "all"), default="front")
    rn.add_argument("--mesh", action="store_true", help="also the 256^3 density grid -> marching_cubes/<name>/front.pkl (generate.py:86-105)")
    rn.add_argument("--limit", type=int)
    rn.add_argument("--exact", action="store_true", help="exact-contract final pass (default: the renderer's tolerance mode)")
    a = ap.parse_args(argv)
    if a.cmd == "prepare":
        print(json.dumps({"prepared": len(prepare(os.path.abspath(a.reference), os.path.abspath(a.data), a.limit))}))
        return
    if not torch.cuda.is_available():
        raise SystemExit("eval_front run needs an MI355X: the HIP path has no CPU fallback")
    G = load_generator(a.network, os.path.abspath(a.reference))
    if a.exact:
        G.set_render_exact(True)
    print(json.dumps(run(G, a.data, a.out, subset=a.subset, mesh=a.mesh, limit=a.limit)))


if __name__ == "__main__":
    main()
