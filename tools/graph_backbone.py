#!/usr/bin/env python3
"""How much of a batch-1 backbone / super-resolution pass is host launch overhead?  Eager timing vs hipGraph replay of the same
pass (torch.cuda.CUDAGraph captures the torch glue and the C-ABI launches alike: both are plain launches on the capturing stream)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import panic3d_amd as P
P.stylegan2.STYLE_MEMO = False  # a pass includes its style computation (a new subject per pass)
from panic3d_amd import stylegan2 as sg, generator as gen

dev = "cuda"
torch.manual_seed(0)


def measure(call, n=20):
    for _ in range(3):
        call()
    es = []
    for _ in range(5):  # median of five groups: one-off allocations / lazily set attributes stay out of the number
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            call()
        torch.cuda.synchronize()
        es.append((time.perf_counter() - t) / n)
    eager = sorted(es)[2]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = call()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return eager * 1e3, (time.perf_counter() - t) / n * 1e3


out = {}
with torch.no_grad():
    G = sg.Generator(z_dim=512, c_dim=25, w_dim=512, img_resolution=256, img_channels=96, cond_mode="none",
                     mapping_kwargs={"num_layers": 2}, channel_base=32768, channel_max=512, num_fp16_res=0, conv_clamp=None).to(dev).eval()
    for N in (1, 4):
        ws = G.mapping(torch.randn(N, 512, device=dev), torch.zeros(N, 25, device=dev), {})
        e, r = measure(lambda: G.synthesis(ws, {}, noise_mode="const"))
        out[f"backbone_N{N}"] = {"eager_ms": e, "graph_replay_ms": r}
    sr = gen.SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True, channels_hidden=256).to(dev).eval()
    x = torch.randn(1, 32, 128, 128, device=dev); rgb = x[:, :3].contiguous(); wsr = torch.randn(1, 14, 512, device=dev)
    e, r = measure(lambda: sr(rgb, x, wsr, noise_mode="none"))
    out["superres_N1"] = {"eager_ms": e, "graph_replay_ms": r}
print(json.dumps(out))
