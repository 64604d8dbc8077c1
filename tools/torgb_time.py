#!/usr/bin/env python3
"""HIP-event time of the stand-alone ToRGB launch (p3d_torgb_f32) at the backbone's shapes, with and without the skip image.
    python tools/torgb_time.py [--n 50]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import panic3d_amd as P
ops = P.ops


def timeit(fn, n, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 50
    dev = torch.device("cuda")
    f = ops.setup_filter((1, 3, 3, 1)).to(dev)
    for I, O, H in [(128, 96, 256), (256, 96, 128), (512, 96, 64), (512, 96, 32), (512, 96, 16), (512, 96, 8), (512, 96, 4), (128, 3, 512), (256, 3, 256)]:
        x = torch.randn(1, I, H, H, device=dev)
        w = torch.randn(O, I, 1, 1, device=dev)
        s = torch.randn(1, I, device=dev)
        b = torch.randn(O, device=dev)
        skip = torch.randn(1, O, H // 2, H // 2, device=dev)
        wt = ops.torgb_weights(w)
        row = dict(I=I, O=O, H=H)
        row["us_skip"] = round(timeit(lambda: ops.torgb(x, wt, O, s, bias=b, skip=skip, skip_filter=f), n), 2)
        row["us_noskip"] = round(timeit(lambda: ops.torgb(x, wt, O, s, bias=b), n), 2)
        mb = (x.numel() + 1.25 * O * H * H) * 4 / 1e6
        row["MB"] = round(mb, 1)
        row["TB_s_skip"] = round(mb / row["us_skip"], 2)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
