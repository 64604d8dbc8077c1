"""On-disk outputs of `_scripts/eval/generate.py` (SURVEY §8f-3 "on-disk PKL/PNG format compatibility"): the files a
subject produces are <view>.png (RGB), <view>_xyza.png (RGBA) and marching_cubes.pkl.  Host I/O only — no kernels here.

  * PNG quantisation follows the reference's image wrapper: `I(tensor).save(fn)` (generate.py:147-148) ->
    `TF.to_pil_image(data.float().clamp(0,1))` (_util/twodee_v1.py:184-185) -> torchvision's `pic.mul(255).byte()`:
    clamp to [0,1], scale by 255, TRUNCATE to uint8, CHW -> HWC, mode RGB / RGBA / L by channel count.
  * xyza = cat([(image_xyz + bw/2) / bw, image_weights], dim=1) (generate.py:143-146).
  * the mesh is pickled as a plain dict with the reference's keys (verts, faces, normals, values, colors;
    _util/eg3d_metrics3d.py:203-209 wraps them in its attribute-dict class, which readers index by key).
"""
import os
import pickle

import numpy as np
import torch


def to_uint8_hwc(img):
    """[1,C,H,W] or [C,H,W] float tensor / array in [0,1] -> uint8 [H,W,C] ([H,W] for C == 1), the reference's quantisation."""
    t = torch.as_tensor(img).detach().float().cpu()
    if t.dim() == 4:
        if t.shape[0] != 1:
            raise RuntimeError("one image at a time")
        t = t[0]
    if t.dim() != 3 or t.shape[0] not in (1, 3, 4):
        raise RuntimeError("image must be [C,H,W] with C in {1,3,4}")
    q = t.clamp(0, 1).mul(255).to(torch.uint8).permute(1, 2, 0).contiguous().numpy()
    return q[..., 0] if q.shape[-1] == 1 else q


def save_png(img, fn):
    from PIL import Image  # imported lazily: the render path does not need PIL
    os.makedirs(os.path.dirname(os.path.abspath(fn)), exist_ok=True)
    Image.fromarray(to_uint8_hwc(img)).save(fn)
    return fn


def xyza(out, box_warp):
    """generate.py:143-146."""
    return torch.cat([(out["image_xyz"] + box_warp / 2) / box_warp, out["image_weights"]], dim=1)


def save_view(out, fn_rgb, fn_xyza, box_warp):
    """The two PNGs generate.py writes per view (:147-148) from the dict G.f returns."""
    return save_png(out["image"], fn_rgb), save_png(xyza(out, box_warp), fn_xyza)


def dump_mesh(mc, fn):
    """generate.py:104-105 (`uutil.pdump(mc, fn_march)`)."""
    os.makedirs(os.path.dirname(os.path.abspath(fn)), exist_ok=True)
    keys = ("verts", "faces", "normals", "values", "colors")
    with open(fn, "wb") as fh:
        pickle.dump({k: np.asarray(mc[k]) for k in keys if k in mc}, fh)
    return fn
