#!/usr/bin/env python3
"""One G.f call under rocprofv3 --kernel-trace: which launches are not ours, and how much of the call the GPU idles.
  rocprofv3 --kernel-trace --output-format csv -d OUT -o r -- python tools/experiments/f_trace.py ; python tools/experiments/f_trace.py --read OUT"""
import os, sys, time, json, csv
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = list(csv.DictReader(open(os.path.join(sys.argv[2], "r_kernel_trace.csv"))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "fill" in r["Kernel_Name"].lower() and int(r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", 0)) >= 0 and r["Kernel_Name"].startswith("MARK")]
    # calls are separated by a torch.zeros(7777) marker
    idx = [i for i, r in enumerate(rows) if r.get("Grid_Size_X", r.get("Grid_Size", "")) and "vectorized_elementwise_kernel" in r["Kernel_Name"] and r.get("Workgroup_Size_X", "") and False]
    # simpler: split on gaps > 1 ms
    calls, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 1_000_000:
            calls.append(cur); cur = []
        cur.append(b)
    calls.append(cur)
    c = calls[-2] if len(calls) > 2 else calls[-1]
    t0, t1 = int(c[0]["Start_Timestamp"]), int(c[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in c)
    print(f"launches {len(c)}, span {(t1 - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
    prev = None
    for r in c:
        gap = (int(r["Start_Timestamp"]) - prev) / 1e3 if prev is not None else 0.0
        prev = int(r["End_Timestamp"])
        print(f"  {r['Kernel_Name'][:70]:70s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us   gap before {gap:6.1f}")
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
sys.argv = ["x"]
src = open(os.path.join(ROOT, "tools", "generate_subject.py")).read()
exec(src[:src.index("def sync():")])
with torch.no_grad():
    x0 = {"elevations": torch.zeros(1, device=dev), "azimuths": torch.zeros(1, device=dev), "cond": cond, "seeds": [0], "noise_mode": "const", "triplane_crop": 0.1, "cull_clouds": 0.5}
    for _ in range(3):
        G.f(dict(x0))
    xw = dict(x0); G.f(xw); ws = xw["ws"]
    for _ in range(4):
        torch.cuda.synchronize(); time.sleep(0.01)
        G.f(dict(x0, ws=ws))
    torch.cuda.synchronize()
