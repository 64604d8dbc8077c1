#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc pass (counter_collection.csv):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d <dir> -- <cmd>
    python tools/mfma_util.py <dir> profiles/<tag>_mfma_util.json [name filter ...]
mfma_util = MFMA busy cycles / (GPU active cycles x 1024 SIMDs); GRBM_GUI_ACTIVE is summed over the 8 XCDs (divided by 8 here)."""
import collections, csv, glob, json, os, sys

d, out, filt = sys.argv[1], sys.argv[2], sys.argv[3:] or ["k_modconv", "k_render<"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if any(f in r["Kernel_Name"] for f in filt):
            key = f'{r["Kernel_Name"].replace("void ", "").split("(")[0]} grid={r["Grid_Size"]}'
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        act = m["GRBM_GUI_ACTIVE"] / 8
        res[k] = {"mfma_instructions": m.get("SQ_INSTS_MFMA"), "valu_instructions": m.get("SQ_INSTS_VALU"),
                  "mfma_busy_cycles": m["SQ_VALU_MFMA_BUSY_CYCLES"], "gpu_active_cycles": act,
                  "mfma_util": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * 1024), "launches": len(c["GRBM_GUI_ACTIVE"])}
json.dump({"note": __doc__.strip().splitlines()[-1], "kernels": res}, open(out, "w"), indent=1)
for k, v in res.items():
    print(f'{k:60s} util {v["mfma_util"]:.3f}  valu/mfma {(v["valu_instructions"] or 0) / max(v["mfma_instructions"] or 1, 1):.2f}')
