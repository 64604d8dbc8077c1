#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/<dir>) into small tracked files under profiles/.

    python tools/summarize_prof.py <tag> <kernel_stats_dir> [<pmc_dir> ...]

Writes profiles/<tag>_kernel_stats.csv (kernel names truncated to 80 chars), profiles/<tag>_pmc.json (per-launch
averages of every counter for the fused kernel k_render) and profiles/pmc_latest.json (HBM traffic per launch:
FETCH_SIZE and WRITE_SIZE are reported in KiB; FETCH_SIZE gets the gfx950 x2 correction of
/opt/skills/guides/MI355X_MICROARCH.md §HBM because the gathers are 16-B-per-lane loads).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    ks = glob.glob(os.path.join(stats_dir, "*kernel_stats.csv"))
    if ks:
        rows = list(csv.reader(open(ks[0])))
        with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            for r in rows:
                r[0] = r[0][:80]
                w.writerow(r)
    pmc = {}
    meta = {}
    for d in pmc_dirs:
        for fn in glob.glob(os.path.join(d, "*counter_collection.csv")):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(fn)):
                if ("k_render<" in r["Kernel_Name"] or r["Kernel_Name"].startswith("k_render(")) and "finish" not in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                              "LDS_Block_Size", "Scratch_Size")}
            for k, v in acc.items():
                pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    if pmc:
        out = {"kernel": "k_render", "dispatch": meta, "counters": pmc}
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            fetch = pmc["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
            write = pmc["WRITE_SIZE"]["mean_per_launch"] * 1024
            out["hbm_bytes_per_launch"] = fetch + write
            out["fetch_bytes_corrected"] = fetch
            out["write_bytes"] = write
            json.dump({"hbm_bytes_per_launch": fetch + write, "source": f"profiles/{tag}_pmc.json"},
                      open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"))
        json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
    if ks:
        print(open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")).read()[:1200])
    print(json.dumps(pmc)[:1200])


if __name__ == "__main__":
    main()
