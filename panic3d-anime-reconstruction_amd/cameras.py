"""Host-side camera / ray generation of the hot path's callers — negligible work, plain torch.

  camera_label(elev, azim, dist, fov)    <- _databacks/lustrous_renders_v1.py:33-75 camera_params_to_matrix('eg3d_lustrousB')
  perspective_rays(cam2world, intr, res) <- training/volumetric_rendering/ray_sampler.py:24-62 RaySampler.forward
  ortho_rays(elev, azim, dist, bw, res)  <- _databacks/lustrous_renders_v1.py:78-104 get_rays_ortho
"""
import numpy as np
import torch
from scipy.spatial.transform import Rotation

_FLIP_A = np.diag([-1.0, 1.0, -1.0, 1.0])
_FLIP_B = np.diag([1.0, -1.0, -1.0, 1.0])


def camera_matrices(elev, azim, dist, fov):
    """(extrinsic cam2world 4x4, intrinsic 3x3) float32 tensors; fov in degrees, focal = 0.5 / tan(fov/2)."""
    elev, azim, dist, fov = (float(v) for v in (elev, azim, dist, fov))
    focal = 0.5 / np.tan((fov / 2) * np.pi / 180)
    intr = np.asarray([[focal, 0, 0.5], [0, focal, 0.5], [0, 0, 1]], dtype=np.float32)
    Rm = np.eye(4)
    Rm[:3, :3] = Rotation.from_euler("xyz", [elev, azim, 0], degrees=True).as_matrix().T
    Rm[[0, 2]] *= -1
    Rm[2, -1] = -dist
    extr = _FLIP_A @ np.linalg.inv(Rm) @ _FLIP_B
    return torch.tensor(extr).float(), torch.tensor(intr).float()


def camera_label(elev, azim, dist, fov):
    """The 25-float conditioning label: flatten(extrinsic) ++ flatten(intrinsic)."""
    e, i = camera_matrices(elev, azim, dist, fov)
    return torch.cat([e.flatten(), i.flatten()])


def perspective_rays(cam2world, intrinsics, resolution):
    """cam2world [N,4,4], intrinsics [N,3,3] -> origins [N,res^2,3], unit directions [N,res^2,3].
    Pixel centres (i + 0.5)/res; x follows the column index (the reference's uv.flip(0))."""
    N, dev = cam2world.shape[0], cam2world.device
    M = resolution * resolution
    cam = cam2world[:, :3, 3]
    fx, fy = intrinsics[:, 0, 0:1], intrinsics[:, 1, 1:2]
    cx, cy, sk = intrinsics[:, 0, 2:3], intrinsics[:, 1, 2:3], intrinsics[:, 0, 1:2]
    ar = torch.arange(resolution, dtype=torch.float32, device=dev)
    uv = torch.stack(torch.meshgrid(ar, ar, indexing="ij")) * (1.0 / resolution) + (0.5 / resolution)
    uv = uv.flip(0).reshape(2, -1).transpose(1, 0)
    x_cam = uv[:, 0].unsqueeze(0).expand(N, M)
    y_cam = uv[:, 1].unsqueeze(0).expand(N, M)
    z_cam = torch.ones((N, M), device=dev)
    x_lift = (x_cam - cx + cy * sk / fy - sk * y_cam / fy) / fx * z_cam
    y_lift = (y_cam - cy) / fy * z_cam
    pts = torch.stack((x_lift, y_lift, z_cam, torch.ones_like(z_cam)), dim=-1)
    world = torch.bmm(cam2world, pts.permute(0, 2, 1)).permute(0, 2, 1)[:, :, :3]
    dirs = torch.nn.functional.normalize(world - cam[:, None, :], dim=2)
    return cam.unsqueeze(1).repeat(1, M, 1), dirs


def rays_from_label(label, resolution):
    """camera label(s) [N,25] -> rays, as training/triplane.py:393-396 splits them."""
    label = label.reshape(-1, 25)
    return perspective_rays(label[:, :16].reshape(-1, 4, 4), label[:, 16:25].reshape(-1, 3, 3), resolution)


def ortho_rays(elev, azim, dist, boxwarp, resolution, device=None):
    """Orthographic rays: origins on a boxwarp-wide plane at distance dist, direction (0,0,-1), rotated by
    Euler xyz(-elev, azim, 0).  Returns dict(ray_origins, ray_directions) of [1,3,res,res] like the reference."""
    e, a, r, bw = float(elev), float(azim), int(resolution), boxwarp
    mg = torch.arange(r, device=device)
    lin = (mg + 0.5) / r * bw - bw / 2
    gx, gy = torch.meshgrid(lin, -lin, indexing="xy")
    o = torch.stack([gx, gy, torch.zeros(r, r, device=device)])
    both = torch.stack([o, o + torch.tensor([0.0, 0.0, -1.0], device=device)[:, None, None]])
    both[:, 2] += dist
    rot = torch.tensor(Rotation.from_euler("xyz", [-e, a, 0.0], degrees=True).as_matrix(), device=device, dtype=both.dtype)
    t = (rot @ both.permute(0, 2, 3, 1)[..., None]).permute(-1, 0, 3, 1, 2)[0]
    return {"ray_origins": t[0][None], "ray_directions": (t[1] - t[0])[None]}


def ortho_rays_flat(elev, azim, dist, boxwarp, resolution, device=None):
    """The [1,res^2,3] form the renderer consumes (training/triplane.py:181-182 rearrange 'b c h w -> b (h w) c')."""
    fr = ortho_rays(elev, azim, dist, boxwarp, resolution, device)
    o = fr["ray_origins"].permute(0, 2, 3, 1).reshape(1, resolution * resolution, 3).contiguous()
    d = fr["ray_directions"].permute(0, 2, 3, 1).reshape(1, resolution * resolution, 3).contiguous()
    return o, d


def ray_limits_box(rays_o, rays_d, box_side_length):
    """math_utils.get_ray_limits_box (volumetric_rendering/math_utils.py:46-98): entry / exit distance of every ray through the
    axis-aligned box of side `box_side_length` centred at the origin; (-1, -2) for rays that miss it.  The same fp32
    operations in the same order (1 / d, (bound - o) * invdir, max / min), as tensor ops: rays [..., 3] -> two [..., 1]."""
    shape = rays_o.shape
    o, d = rays_o.detach().reshape(-1, 3), rays_d.detach().reshape(-1, 3)
    lo = torch.full((1,), -1 * (box_side_length / 2), dtype=o.dtype, device=o.device)
    hi = torch.full((1,), 1 * (box_side_length / 2), dtype=o.dtype, device=o.device)
    invdir = 1 / d
    neg = invdir < 0
    near, far = torch.where(neg, hi, lo), torch.where(neg, lo, hi)  # bounds[sign], bounds[1 - sign] per axis
    t0, t1 = (near - o) * invdir, (far - o) * invdir
    tmin, tmax = t0[:, 0], t1[:, 0]
    valid = ~((tmin > t1[:, 1]) | (t0[:, 1] > tmax))
    tmin, tmax = torch.max(tmin, t0[:, 1]), torch.min(tmax, t1[:, 1])
    valid = valid & ~((tmin > t1[:, 2]) | (t0[:, 2] > tmax))
    tmin, tmax = torch.max(tmin, t0[:, 2]), torch.min(tmax, t1[:, 2])
    tmin = torch.where(valid, tmin, torch.full_like(tmin, -1))
    tmax = torch.where(valid, tmax, torch.full_like(tmax, -2))
    return tmin.reshape(*shape[:-1], 1), tmax.reshape(*shape[:-1], 1)


def patch_ray_limits(ray_start, ray_end):
    """renderer.py:167-170: rays that miss the box take [min, max] of the valid rays' START distances (sic).  No host
    synchronisation: with no valid ray at all the limits stay as they are, like the reference's `if torch.any(...)`."""
    valid = ray_end > ray_start
    big = torch.finfo(ray_start.dtype).max
    lo = torch.where(valid, ray_start, torch.full_like(ray_start, big)).min()
    hi = torch.where(valid, ray_start, torch.full_like(ray_start, -big)).max()
    some = valid.any()
    fix = (~valid) & some
    return torch.where(fix, lo, ray_start), torch.where(fix, hi, ray_end)


import functools


# generate.py renders 16 fixed poses per subject (4 ortho + 12 perspective): 32 entries hold that set at two resolutions.  The
# entries are device tensors (4 x 3 x res^2 fp32 — the rays in image layout and in the renderer's layout: 12 MB per view at 512^2, 0.8 MB
# at the pipeline's 128^2), so the cache is kept SMALL — a 360-degree sweep or
# random evaluation poses would otherwise pin one entry per unique pose (3 GB at 512^2 with the old 512 entries) — and is
# dropped by cached_view_clear() (e.g. after moving a generator to another device).
def make_view(elev, azim, dist, fov, resolution, boxwarp, device, dtype=torch.float32):
    """(camera label [25], ray origins [3,res,res], ray directions [3,res,res], and the same rays as [res^2,3] — the layout the
    renderer consumes, training/triplane.py:181-182) of one view, computed now."""
    elev, azim, dist, fov, resolution, boxwarp = float(elev), float(azim), float(dist), float(fov), int(resolution), float(boxwarp)
    label = camera_label(elev, azim, dist, fov).to(dtype).to(device)
    if fov < 0:  # negative fov = orthographic view (training/triplane.py:402-414)
        r = ortho_rays(elev, azim, dist, boxwarp, resolution, device=device)
        o, d = r["ray_origins"][0].contiguous(), r["ray_directions"][0].contiguous()
        flat = lambda t: t.permute(1, 2, 0).reshape(resolution * resolution, 3).contiguous()
        return label, o, d, flat(o), flat(d)
    ro, rd = perspective_rays(label[:16].view(1, 4, 4), label[16:25].view(1, 3, 3), resolution)
    chw = lambda t: t.reshape(resolution, resolution, 3).permute(2, 0, 1).contiguous()
    return label, chw(ro), chw(rd), ro.reshape(resolution * resolution, 3).contiguous(), rd.reshape(resolution * resolution, 3).contiguous()


_cached_view = functools.lru_cache(maxsize=32)(make_view)


def cached_view(elev, azim, dist, fov, resolution, boxwarp, device, dtype=torch.float32):
    """(camera label [25], ray origins [3,res,res], ray directions [3,res,res], both again as [res^2,3]) of one view, memoised on the view's parameters:
    generate.py renders the same 16 poses for every subject, and the ray generation is a dozen small launches plus host maths.
    The tensors are shared between calls: callers stack / copy them, they never write into them."""
    return _cached_view(float(elev), float(azim), float(dist), float(fov), int(resolution), float(boxwarp), torch.device(device), dtype)


def cached_view_clear():
    """Drop the memoised views (device tensors)."""
    _cached_view.cache_clear()
