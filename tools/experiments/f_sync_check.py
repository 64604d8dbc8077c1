#!/usr/bin/env python3
"""G.f with the view parameters as device tensors (one device -> host copy per call) vs host tensors (none): how much of a call is
the pipeline drain + refill behind that copy."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
sys.argv = ["x"]
src = open(os.path.join(ROOT, "tools", "generate_subject.py")).read()
exec(src[:src.index("def sync():")])
def sync(): torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    base = {"cond": cond, "seeds": [0], "noise_mode": "const", "triplane_crop": 0.1, "cull_clouds": 0.5}
    xd = dict(base, elevations=torch.zeros(1, device=dev), azimuths=torch.zeros(1, device=dev))
    xh = dict(base, elevations=torch.zeros(1), azimuths=torch.zeros(1))
    xw = dict(xd); G.f(xw); ws = xw["ws"]
    out = {}
    for name, x0 in (("device_params", xd), ("host_params", xh)):
        for _ in range(3): G.f(dict(x0, ws=ws))
        t = sync()
        for _ in range(20): G.f(dict(x0, ws=ws))
        out[name + "_ms"] = round((sync() - t) / 20 * 1e3, 3)
        t = sync()
        for _ in range(20): G.f(dict(x0, ws=ws)); torch.cuda.synchronize()
        out[name + "_synced_each_call_ms"] = round((sync() - t) / 20 * 1e3, 3)
print(json.dumps(out))
