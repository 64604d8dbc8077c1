#!/usr/bin/env python3
"""Quick timing of the fused render kernel (development aid; bench.py is the contract benchmark)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import panic3d_amd as P
from panic3d_amd import ops

res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Sc = int(sys.argv[2]) if len(sys.argv) > 2 else 48
Sf = int(sys.argv[3]) if len(sys.argv) > 3 else 48
N = int(sys.argv[4]) if len(sys.argv) > 4 else 1
torch.manual_seed(0)
dev = "cuda"
low = torch.randn(N * 3, 32, 16, 16, device=dev)
planes = (torch.nn.functional.interpolate(low, size=(256, 256), mode="bilinear") + 0.1 * torch.randn(N * 3, 32, 256, 256, device=dev)).reshape(N, 3, 32, 256, 256).contiguous()
w0 = torch.randn(64, 32, device=dev); b0 = torch.randn(64, device=dev) * .5; w1 = torch.randn(33, 64, device=dev); b1 = torch.randn(33, device=dev) * .5
w1[0] *= 20
mlp = ops.prescale_mlp(w0, b0, w1, b1, 1 / np.sqrt(32), 1, 1 / np.sqrt(64), 1)
ro = dict(box_warp=0.7, ray_start=0.5, ray_end=1.5, depth_resolution=Sc, depth_resolution_importance=Sf, white_back=True, use_triplane=1)
opts = ops.make_opts(ro, triplane_crop=0.1, cull_clouds=0.5, force_sigmoid=True)
R = res * res
ys, xs = torch.meshgrid(torch.arange(res, device=dev), torch.arange(res, device=dev), indexing="ij")
o = torch.stack([(xs + .5) / res * .7 - .35, (ys + .5) / res * .7 - .35, torch.full_like(xs, 1.0, dtype=torch.float32)], -1).reshape(1, R, 3).float().repeat(N, 1, 1).contiguous()
d = torch.tensor([0, 0, -1.0], device=dev).expand(N, R, 3).contiguous()
jit = torch.rand(N, R, Sc, device=dev); u = torch.rand(N * R, max(Sf, 1), device=dev)
pl = ops.planes_to_nhwc(planes)
for tw in (res, 0):
    for _ in range(2):
        out = ops.render(pl, o, d, jit, u, mlp, opts, ray_tile_w=tw)
    torch.cuda.synchronize()
    t = time.time(); K = 5
    for _ in range(K):
        out = ops.render(pl, o, d, jit, u, mlp, opts, ray_tile_w=tw)
    torch.cuda.synchronize()
    dt = (time.time() - t) / K
    print(f"res={res} S={Sc}+{Sf} N={N} tile_w={tw}: {dt*1e3:.2f} ms  {N*R/dt/1e6:.2f} Mrays/s  {N*R*(Sc+Sf)/dt/1e9:.3f} Gsamples/s  wsum mean {out[2].mean().item():.3f}")
