#!/usr/bin/env python3
"""gpurun_out/<tag>_smallpmc (tools/pmc_small_view.sh) -> profiles/<tag>_small_view_pmc.json: per-launch counter means of the
128^2-ray renderer launch and the fractions derived from them (units as in tools/summarize_prof.py)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag + "_smallpmc")
out = {"tag": tag, "render_src_sha": open(os.path.join(src, "render_src_sha.txt")).read().strip(),
       "what": "ops.render, 128^2 rays x (96+96), surface scene of bench.py, the kernel the host picks (tools/small_view_loop.py)", "modes": {}}
is_render = lambda n: ("k_render" in n) and "finish" not in n
for mode in ("tol", "exact"):
    ent = {}
    js = os.path.join(src, f"stats_{mode}.json")
    if os.path.exists(js):
        lines = [l for l in open(js).read().splitlines() if l.startswith("{")]
        if lines:
            ent["run"] = json.loads(lines[-1])
    ks = glob.glob(os.path.join(src, f"stats_{mode}", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        for r in csv.DictReader(open(ks[0])):
            if is_render(r["Name"]):
                ent["kernel"] = r["Name"][:60]
                ent["rocprof_kernel_avg_us"] = float(r["AverageNs"]) / 1e3
                ent["rocprof_kernel_min_us"], ent["rocprof_kernel_max_us"], ent["calls"] = float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, int(r["Calls"])
    acc, meta = collections.defaultdict(list), {}
    for fn in glob.glob(os.path.join(src, f"pmc_{mode}_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if is_render(r["Kernel_Name"]):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size") if k in r}
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    ent["dispatch"], ent["counters_mean_per_launch"] = meta, c
    d = {}
    if "GRBM_GUI_ACTIVE" in c:
        clk = c["GRBM_GUI_ACTIVE"] / 8
        d["gpu_active_clocks"] = clk
        if "SQ_ACTIVE_INST_VALU" in c:
            d["valu_active_frac_of_simd_cycles"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (clk * 1024)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (clk * 1024)
        if "SQ_BUSY_CYCLES" in c:
            d["sq_busy_frac"] = c["SQ_BUSY_CYCLES"] / c["GRBM_GUI_ACTIVE"] if c["GRBM_GUI_ACTIVE"] else None
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        wc = c["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
            if k in c:
                d[k.lower() + "_frac_of_wave_cycles"] = c[k] / wc
        if "gpu_active_clocks" in d:
            d["mean_resident_waves_per_simd"] = wc * 4 / (d["gpu_active_clocks"] * 1024)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"]:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    ent["derived"] = d
    out["modes"][mode] = ent
p = os.path.join(ROOT, "profiles", f"{tag}_small_view_pmc.json")
json.dump(out, open(p, "w"), indent=1)
print(p)
for m, e in out["modes"].items():
    print(m, e.get("kernel"), e.get("rocprof_kernel_avg_us"), json.dumps(e.get("derived")))
