#!/usr/bin/env python3
"""Where the HOST time of one TriPlaneGenerator.f call goes (the generator of tools/generate_subject.py, eager, one view per call,
synchronised after every call like _scripts/eval/generate.py): cProfile over N calls, top functions by own time, next to the
wall time per call with and without a synchronisation between calls.  Development aid (GPU box)."""
import os, sys, time, json, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
argv = sys.argv[:]
sys.argv = ["x"]
src = open(os.path.join(ROOT, "tools", "generate_subject.py")).read()
exec(src[:src.index("def sync():")])
def sync(): torch.cuda.synchronize(); return time.perf_counter()
N = 40
paste = "--paste" in argv
extra = dict(opts) if paste else {"triplane_crop": 0.1, "cull_clouds": 0.5}
def xin(k):
    elev, azim, fov = views[k % len(views)]
    return {"elevations": elev * torch.ones(1, device=dev), "azimuths": azim * torch.ones(1, device=dev), "fovs": fov * torch.ones(1, device=dev),
            "cond": cond, "seeds": [0], "noise_mode": "const", **extra}
with torch.no_grad():
    for k in range(20):
        G.f(xin(k))
    t = sync()
    for k in range(N):
        G.f(xin(k)); torch.cuda.synchronize()
    synced = (sync() - t) / N * 1e3
    t = sync()
    for k in range(N):
        G.f(xin(k))
    free = (sync() - t) / N * 1e3
    # host time alone: how long python needs to ISSUE one call (no synchronisation inside the loop, timed before the final sync)
    t = sync()
    for k in range(N):
        G.f(xin(k))
    issue = (time.perf_counter() - t) / N * 1e3
    sync()
    pr = cProfile.Profile()
    pr.enable()
    for k in range(N):
        G.f(xin(k)); torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s)
ps.sort_stats("tottime").print_stats(45)
ps.sort_stats("cumulative").print_stats(45)
print(json.dumps({"paste": paste, "ms_per_call_synced": synced, "ms_per_call_free_running": free, "ms_host_issue_per_call": issue, "calls": N}))
print(s.getvalue().replace(ROOT + "/", ""))
