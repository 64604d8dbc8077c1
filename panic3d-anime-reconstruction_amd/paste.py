"""`paste_front` — the front-view paste post-process of TriPlaneGenerator.f (training/triplane.py:553-691): where the
rendered surface is visible from the orthographic front view, the super-resolved colour is replaced by the input
illustration sampled at the rendered xyz.  Two launches: ONE extra pass through the fused renderer for the front-occlusion
test (rays from the rendered surface points towards the front plane, triplane.py:565-578), then ONE fused kernel
(`ops.paste_front` -> p3d_paste_front_f32, csrc/p3d_paste.hip) for the four masks, the sampling of the illustration and the
lerp — the reference runs three bilinear resizes, a Sobel, a nearest resize, a grid_sample and a lerp over 512^2 x N pixels.

kornia is not a dependency: the kernel restates kornia 0.6.5 `kornia.filters.sobel(x, normalized=True, eps=1e-6)` (3x3
Sobel kernels divided by 8, replicate padding, sqrt(gx^2 + gy^2 + eps)) — "parity unpinned": kornia cannot be installed here
(profiles/history/r02_notes.txt; tests/golden/make_golden_sobel.py generates the pin where it can).  `front_weight_erosion >= 1`
(an extra front-view render + kornia.morphology.erosion, restated below) and `force_image` — the two options
_scripts/eval/generate.py:55-66 never passes — run on the torch formulation (`paste_front_torch`), not on the fused kernel.  `sobel_magnitude`,
`sample_orthofront` and `xyz_discrepancy` below are the torch formulation of the same steps, kept as the kernel's reference
in tests/ (`paste_front_torch`).
"""
import contextlib

import torch
import torch.nn.functional as F


def sobel_magnitude(x, eps=1e-6):
    """3x3 Sobel gradient magnitude, kernels / 8, replicate padding — a restatement of kornia.filters.sobel's published definition,
    PARITY UNPINNED against kornia itself (absent here: tests/golden/make_golden_sobel.py + test_sobel_against_kornia wait for a box
    that has it).  Written with shifted slices: on ROCm a 1-channel
    F.conv2d of a 512^2 image goes through MIOpen and costs ~150 ms per call, the slices ~0.1 ms."""
    xp = F.pad(x, [1, 1, 1, 1], mode="replicate")
    tl, tc, tr = xp[..., :-2, :-2], xp[..., :-2, 1:-1], xp[..., :-2, 2:]
    ml, mr = xp[..., 1:-1, :-2], xp[..., 1:-1, 2:]
    bl, bc, br = xp[..., 2:, :-2], xp[..., 2:, 1:-1], xp[..., 2:, 2:]
    gx = ((tr - tl) + 2.0 * (mr - ml) + (br - bl)) * 0.125
    gy = ((bl - tl) + 2.0 * (bc - tc) + (br - tr)) * 0.125
    return torch.sqrt(gx * gx + gy * gy + eps)


def sample_orthofront(front_rgb, view_xyz, bw):
    """Colour of the front illustration at the (x, y) of every rendered surface point (triplane.py:555-564)."""
    vij = 1 - (view_xyz[:, [1, 0]] + bw / 2) / bw
    return F.grid_sample(front_rgb.permute(0, 1, 3, 2), vij.permute(0, 2, 3, 1) * 2 - 1, padding_mode="border",
                         mode="bilinear", align_corners=False)


def front_occlusion(G, x, out, offset=0.01):
    """Accumulated opacity between every rendered surface point and the front plane (triplane.py:565-578): rays from the
    surface points along +z.  The reference calls G.f again for this (backbone + renderer + super-resolution, and with its
    default noise_mode='random' on different planes); only `image_weights` is used, so this renders the SAME planes
    (`out['triplane']`) once more through the fused renderer and skips the rest."""
    # (G._sign: the (-1, 1, -1) tensor kept on the device — a torch.tensor(..., device=...) here is a pageable host->device copy
    # that waits for everything queued on the stream, i.e. for the whole view: ~1 ms of idle GPU per pasted view, tools/host_profile.py)
    # ro = xyz * (-1, 1, -1); ro.z -= ray_start - offset   (triplane.py:568-570) as ONE launch: fma(xyz, sign, shift) — xyz * sign is
    # exact, so the single rounding is the reference's subtraction; rd = (0, 0, 1) everywhere is a constant kept per shape
    xyz = out["image_xyz"]
    N, _, H, W = xyz.shape
    dev = xyz.device
    consts = G.__dict__.get("_occ_consts")
    key = (dev, N, H, W, float(G.rendering_kwargs["ray_start"]), float(offset))
    if consts is None or consts[0] != key:
        shift = torch.zeros((1, 3, 1, 1), dtype=torch.float32, device=dev)
        shift[0, 2] = -(G.rendering_kwargs["ray_start"] - offset)
        rd_flat = torch.zeros((N, H * W, 3), dtype=torch.float32, device=dev)
        rd_flat[..., 2] = 1
        consts = G.__dict__["_occ_consts"] = (key, shift, rd_flat)
    ro = torch.addcmul(consts[1], xyz, G._sign(dev))
    flat = lambda t: t.permute(0, 2, 3, 1).reshape(N, H * W, 3).contiguous()
    draws = G._inject_draws or (None, None)
    if isinstance(draws, list):
        draws = draws.pop(0)
    # only the accumulated opacity of this second render is read (triplane.py:578): a weights-only launch — no colour decode
    _, _, wsum, _ = G.renderer(out["triplane"], G.decoder, flat(ro), consts[2], G.rendering_kwargs,
                               triplane_crop=x.get("triplane_crop"), cull_clouds=x.get("cull_clouds"),
                               binarize_clouds=x.get("binarize_clouds"), jitter=draws[0], u=draws[1], weights_only=True)
    return wsum.permute(0, 2, 1).reshape(N, 1, H, W)


def xyz_discrepancy(xyz, rays):
    """Distance of the rendered xyz from its own ray (triplane.py:600-605)."""
    a, n = rays["ray_origins"], rays["ray_directions"]
    p = xyz * torch.tensor([-1, 1, -1], device=xyz.device)[None, :, None, None]
    return ((p - a) - ((p - a) * n).sum(dim=1, keepdim=True) * n).norm(2, dim=1, keepdim=True)


def paste_front(G, x, out, mode="default", thresh_weight=0.95, thresh_edges=0.02, thresh_occ=0.05, offset_occ=0.01,
                thresh_dxyz=0.01, front_weight_erosion=0, grad_sample=False, force_image=None, **kwargs):
    """training/triplane.py:607-691 on the fused kernel; same arguments and return keys."""
    from . import ops
    if front_weight_erosion >= 1 or force_image is not None:
        # the two options _scripts/eval/generate.py never passes (triplane.py:644-667): the torch formulation below carries them — the
        # extra front-view render + erosion, or another image to paste — on the same renderer launches; not on the fused paste kernel
        return paste_front_torch(G, x, out, mode=mode, thresh_weight=thresh_weight, thresh_edges=thresh_edges, thresh_occ=thresh_occ,
                                 offset_occ=offset_occ, thresh_dxyz=thresh_dxyz, front_weight_erosion=front_weight_erosion,
                                 grad_sample=grad_sample, force_image=force_image, **kwargs)
    with torch.no_grad():
        occ = front_occlusion(G, x, out, offset=offset_occ)
        res = ops.paste_front(out["image_weights"], out["image_xyz"], occ, x["force_rays"]["ray_origins"], x["force_rays"]["ray_directions"],
                              x["cond"]["image_ortho_front"], out["image"], thresh_weight, thresh_edges, thresh_occ, thresh_dxyz,
                              G.rendering_kwargs["box_warp"], x["normalize_images"])
    return {"image": res["image"], "paste": res["paste"], "mask": res["mask"], "mask_weights": res["mask_weights"],
            "mask_edges": res["mask_edges"], "mask_occ": res["mask_occ"], "mask_dxyz": res["mask_dxyz"],
            "mask_frontweight": torch.ones_like(res["mask_dxyz"]), "frontweight": None}


def paste_front_torch(G, x, out, mode="default", thresh_weight=0.95, thresh_edges=0.02, thresh_occ=0.05, offset_occ=0.01,
                      thresh_dxyz=0.01, front_weight_erosion=0, grad_sample=False, force_image=None, **kwargs):
    """The same post-process as a composition of torch ops (the reference's own formulation): the comparator of the fused kernel, and
    the path of the two options generate.py does not use — `front_weight_erosion >= 1` (triplane.py:644-658: the paste is also masked
    by the eroded silhouette of a front-view render) and `force_image` (:666-667: paste that image instead of the conditioning one)."""
    view_xyz = out["image_xyz"]
    front_rgb = x["cond"]["image_ortho_front"]
    if len(front_rgb) == 1 and len(view_xyz) > 1:  # V views of one subject in one call (generator.synthesis)
        front_rgb = front_rgb.expand(len(view_xyz), -1, -1, -1)
    S = front_rgb.shape[-1]
    with torch.no_grad():
        wmask = (F.interpolate(out["image_weights"], S, mode="bilinear") > thresh_weight).float()
        smask = sobel_magnitude(F.interpolate(view_xyz, S, mode="bilinear")).norm(2, dim=1, keepdim=True)
        smask = (smask < thresh_edges).float()
        fmask = (front_occlusion(G, x, out, offset=offset_occ) < thresh_occ).float()
        fmask = F.interpolate(fmask, S, mode="bilinear")
        dmask = F.interpolate(xyz_discrepancy(view_xyz, x["force_rays"]), S, mode="nearest")
        dmask = (dmask < thresh_dxyz).float()
        frontw = None
        fwmask = torch.ones_like(dmask)
        if front_weight_erosion >= 1:  # triplane.py:644-658
            frontw = front_weights(G, x)
            fwmask = erosion((frontw > 0.5).float(), int(front_weight_erosion))
            if len(fwmask) == 1 and len(view_xyz) > 1:  # one front-view render, V views in this call
                fwmask = fwmask.expand(len(view_xyz), -1, -1, -1)
            fwmask = sample_orthofront(fwmask, F.interpolate(view_xyz, S, mode="bilinear"), G.rendering_kwargs["box_warp"])
            fwmask = F.interpolate(fwmask, S, mode="nearest")
        mask = wmask * smask * fmask * dmask * fwmask
    # the image to paste is built OUTSIDE no_grad, as the reference builds it (triplane.py:666-673): with grad_sample a gradient
    # reaches force_image / the conditioning image through the sampling below
    if force_image is None:
        tocopy = front_rgb if not x["normalize_images"] else front_rgb * 2 - 1
    else:  # the reference takes its image wrapper (`force_image.t()[None,]`, triplane.py:667); a [C,H,W] / [1,C,H,W] tensor works too
        t = force_image.t() if not torch.is_tensor(force_image) else force_image
        tocopy = (t[None] if t.dim() == 3 else t).to(mask.device)
        if len(tocopy) == 1 and len(view_xyz) > 1:
            tocopy = tocopy.expand(len(view_xyz), -1, -1, -1)
    with (contextlib.nullcontext() if grad_sample else torch.no_grad()):
        paste = sample_orthofront(tocopy, F.interpolate(view_xyz, S, mode="bilinear"), G.rendering_kwargs["box_warp"])
    return {"image": torch.lerp(out["image"], paste, mask), "paste": paste, "mask": mask, "mask_weights": wmask,
            "mask_edges": smask, "mask_occ": fmask, "mask_dxyz": dmask, "mask_frontweight": fwmask, "frontweight": frontw}


def front_weights(G, x):
    """`get_front_weights` (triplane.py:579-599): `image_weights` of an orthographic front view of the same subject."""
    dev = x["cond"]["image_ortho_front"].device
    xin = {k: v for k, v in x.items() if k not in ("paste_params", "camera_params", "conditioning_params", "force_rays")}
    xin["elevations"], xin["azimuths"], xin["fovs"] = torch.zeros(1, device=dev), torch.zeros(1, device=dev), -torch.ones(1, device=dev)
    if "distances" in xin and len(xin["distances"]) != 1:  # V views in the caller's dict: ONE front view here, at the first view's distance
        xin["distances"] = xin["distances"][:1]
    for k in ("image", "image_raw", "image_depth", "image_weights", "triplane", "image_xyz", "normalize_images"):  # (the caller's outputs are not inputs)
        xin.pop(k, None)
    xin["normalize_images"] = x["normalize_images"]
    return G.f(xin, return_more=True)["image_weights"]


def erosion(mask, e, max_val=1e4):
    """`kornia.morphology.erosion(mask, torch.ones(e, e))` as kornia 0.6.5 defines it for a flat structuring element: origin (e // 2,
    e // 2), border_type 'geodesic' (the image is padded with max_val, so the border itself does not erode), output = minimum over the
    window.  PARITY UNPINNED against kornia (absent here), like the Sobel filter above."""
    o = e // 2
    padded = F.pad(mask, [o, e - o - 1, o, e - o - 1], value=max_val)
    return -F.max_pool2d(-padded, kernel_size=e, stride=1)
