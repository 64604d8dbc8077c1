import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import panic3d_amd as P
from panic3d_amd import stylegan2 as sg, generator as gen
dev = "cuda"
torch.manual_seed(0)
with torch.no_grad():
    G = sg.Generator(z_dim=512, c_dim=25, w_dim=512, img_resolution=256, img_channels=96, cond_mode="none",
                     mapping_kwargs={"num_layers": 2}, channel_base=32768, channel_max=512, num_fp16_res=0, conv_clamp=None).to(dev).eval()
    for N in (1, 4):
        ws = G.mapping(torch.randn(N, 512, device=dev), torch.zeros(N, 25, device=dev), {})
        for _ in range(5):
            G.synthesis(ws, {}, noise_mode="const")
    sr = gen.SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True, channels_hidden=256).to(dev).eval()
    x = torch.randn(1, 32, 128, 128, device=dev); rgb = x[:, :3].contiguous(); wsr = torch.randn(1, 14, 512, device=dev)
    ts = []
    st0 = torch.cuda.memory_stats()
    for i in range(30):
        torch.cuda.synchronize(); t = time.perf_counter()
        sr(rgb, x, wsr, noise_mode="none")
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    st1 = torch.cuda.memory_stats()
    keys = ["num_device_alloc", "num_device_free", "num_alloc_retries", "allocation.all.allocated", "segment.all.allocated", "segment.all.freed"]
    print(json.dumps({"ms": [round(t, 2) for t in ts], "stats_delta": {k: st1.get(k, 0) - st0.get(k, 0) for k in keys},
                      "reserved_GB": torch.cuda.memory_reserved() / 2**30}))
