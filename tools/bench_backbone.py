#!/usr/bin/env python3
"""Timing of the StyleGAN2-256 triplane backbone (and the 512^2 super-resolution blocks) on the HIP synthesis operators.
Secondary measurement (the contract benchmark is bench.py): ms per image and achieved fp32 MFMA TFLOP/s."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import panic3d_amd as P
P.stylegan2.STYLE_MEMO = False  # a pass includes its style computation (a new subject per pass)
from panic3d_amd import stylegan2 as sg, generator as gen

torch.manual_seed(0)
dev = "cuda"


def flops_backbone():
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128}
    f = 0
    for res, c in ch.items():
        cin = ch[res // 2] if res > 4 else 0
        if cin:
            f += 2 * 9 * cin * c * (res // 2) ** 2  # transposed conv (useful MACs)
        f += 2 * 9 * c * c * res * res
        f += 2 * c * 96 * res * res
    return f


def timeit(fn, n=10, groups=5):
    """Median over `groups` back-to-back groups of n calls (a mean over one group carried the occasional one-off — a first
    allocation of a 134 MB activation, a lazily set kernel attribute — into the number: 7 ms 'super-resolution' lines)."""
    for _ in range(3):
        fn()
    ts = []
    for _ in range(groups):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) / n)
    return float(np.median(ts))


G = sg.Generator(z_dim=512, c_dim=25, w_dim=512, img_resolution=256, img_channels=96, cond_mode="none",
                 mapping_kwargs={"num_layers": 2}, channel_base=32768, channel_max=512, num_fp16_res=0, conv_clamp=None).to(dev).eval()
out = {}
with torch.no_grad():
    for N in (1, 4):
        ws = G.mapping(torch.randn(N, 512, device=dev), torch.zeros(N, 25, device=dev), {})
        dt = timeit(lambda: G.synthesis(ws, {}, noise_mode="const"))
        fl = flops_backbone() * N
        out[f"backbone_N{N}"] = {"ms": dt * 1e3, "GFLOP": fl / 1e9, "TFLOP/s": fl / dt / 1e12, "frac_of_157.3": fl / dt / 157.3e12}
    sr = gen.SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True, channels_hidden=256).to(dev).eval()
    x = torch.randn(1, 32, 128, 128, device=dev); rgb = x[:, :3].contiguous(); ws = torch.randn(1, 14, 512, device=dev)
    dt = timeit(lambda: sr(rgb, x, ws, noise_mode="none"))
    fl = 2 * (9 * 32 * 256 * 128 ** 2 + 9 * 256 * 256 * 256 ** 2 + 256 * 3 * 256 ** 2 + 9 * 256 * 128 * 256 ** 2 + 9 * 128 * 128 * 512 ** 2 + 128 * 3 * 512 ** 2)
    out["superres_N1"] = {"ms": dt * 1e3, "GFLOP": fl / 1e9, "TFLOP/s": fl / dt / 1e12, "frac_of_157.3": fl / dt / 157.3e12}
    for m in sr.modules():  # opt-in f16 MFMA operands (TriPlaneGenerator.set_sr_mma_f16)
        if isinstance(m, (sg.SynthesisLayer, sg.ToRGBLayer)):
            m.mma_f16 = True
    dt = timeit(lambda: sr(rgb, x, ws, noise_mode="none"))
    out["superres_N1_f16_operands"] = {"ms": dt * 1e3, "TFLOP/s": fl / dt / 1e12, "frac_of_2500_f16": fl / dt / 2.5e15}
    x16 = x.expand(16, -1, -1, -1).contiguous(); rgb16 = x16[:, :3].contiguous(); ws16 = ws.expand(16, -1, -1).contiguous()
    dt = timeit(lambda: sr(rgb16, x16, ws16, noise_mode="none"), n=3, groups=3)
    out["superres_N16_f16_operands"] = {"ms_per_image": dt * 1e3 / 16, "TFLOP/s": 16 * fl / dt / 1e12}
print(json.dumps(out))
