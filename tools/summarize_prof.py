#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/collect_profile.sh (gpurun_out/<tag>/) into small tracked files under profiles/.

    python tools/summarize_prof.py <tag>

Writes, per scene (canonical / surface) and mode (exact_early = the default launch since round 3: exact-contract final pass,
exact early-outs on; exact_noearly = every sample decoded; early / noearly = the same in the tolerance mode, bench.py --fast):
  profiles/<tag>_<scene>_<mode>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (kernel names cut to 80 chars)
  profiles/<tag>_pmc.json                          per-launch averages of every counter for k_render + dispatch info
  profiles/pmc_latest.json                         {"<scene>/<exact|tolerance>": {kernel_src_sha, render_src_sha, hbm_bytes_per_launch, bounds, source}} —
                                                   bench.py prints it only when render_src_sha (the renderer's translation unit) matches the sources it runs.
Units / corrections: FETCH_SIZE and WRITE_SIZE are KiB; FETCH_SIZE gets the gfx950 x2 correction of
/opt/skills/guides/MI355X_MICROARCH.md §HBM (the gathers are 16-B-per-lane loads); GRBM_GUI_ACTIVE is summed over the 8 XCDs
(÷8 = active GPU clocks of the launch); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4 = clocks);
SQ_VALU_MFMA_BUSY_CYCLES counts clocks.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_CU, N_SIMD = 256, 1024


def is_render(name):
    return ("k_render<" in name or name.startswith("k_render(") or "k_render_pair" in name or "k_render_quad" in name) and "finish" not in name


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    sha = open(os.path.join(src, "kernel_src_sha.txt")).read().strip()
    # the renderer's translation unit (what a k_render capture is a capture OF): recorded by the box (collect_profile.sh), or computed
    # here when the box's sources are the ones in this tree
    sys.path.insert(0, ROOT)
    import panic3d_amd as P
    rs = os.path.join(src, "render_src_sha.txt")
    if os.path.exists(rs):
        render_sha = open(rs).read().strip()
    else:
        assert P._build.source_hash() == sha, "the capture is from other kernel sources than this tree: no render_src_sha to stamp"
        render_sha = P._build.render_source_hash()
    allpmc, latest = {}, {}
    for scene in ("canonical", "surface"):
        for mode in ("early", "noearly", "exact_early", "exact_noearly"):
            ks = glob.glob(os.path.join(src, f"stats_{scene}_{mode}", "**", "*kernel_stats.csv"), recursive=True)
            kern_ns = None
            if ks:
                rows = list(csv.reader(open(ks[0])))
                with open(os.path.join(prof, f"{tag}_{scene}_{mode}_kernel_stats.csv"), "w", newline="") as f:
                    w = csv.writer(f)
                    for r in rows:
                        if is_render(r[0]):
                            kern_ns = float(r[3])  # AverageNs
                        r[0] = r[0][:80]
                        w.writerow(r)
            acc = collections.defaultdict(list)
            meta = {}
            for fn in glob.glob(os.path.join(src, f"pmc_{scene}_{mode}_*", "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(fn)):
                    if is_render(r["Kernel_Name"]):
                        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                        meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                                  "LDS_Block_Size", "Scratch_Size")}
            if not acc and kern_ns is None:
                continue
            c = {k: sum(v) / len(v) for k, v in acc.items()}
            ent = {"dispatch": meta, "counters": {k: {"mean_per_launch": c[k], "launches": len(acc[k])} for k in c},
                   "rocprof_kernel_avg_ms": None if kern_ns is None else kern_ns / 1e6}
            b = {}
            if "GRBM_GUI_ACTIVE" in c:
                clk = c["GRBM_GUI_ACTIVE"] / 8
                b["gpu_active_clocks"] = clk
                if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
                    b["tcp_lookups_per_clk_per_cu"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (clk * N_CU)
                    # what tools/ubench/l1_gather sustains at this kernel's occupancy (8 waves per CU): per-lane gathers of the
                    # kernel's layout 1.0 lane-requests per clk per CU, a quad on one texel 4.1, lane-random 16-byte pieces 2.7
                    b["tcp_lookups_ubench_per_clk_per_cu"] = {"per_lane_layout": 1.0, "quad_on_one_texel": 4.1, "random_pieces": 2.7}
                if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                    b["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (clk * N_SIMD)
                if "SQ_INSTS_VALU" in c:
                    b["valu_issue_frac_at_2clk_per_inst"] = (c["SQ_INSTS_VALU"] - c.get("SQ_INSTS_MFMA", 0.0)) * 2 / (clk * N_SIMD)
                if "SQ_ACTIVE_INST_VALU" in c:
                    b["valu_active_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (clk * N_SIMD)
            if "SQ_WAVE_CYCLES" in c:
                for k, nm in (("SQ_WAIT_INST_ANY", "wait_inst_frac_of_wave_cycles"), ("SQ_WAIT_ANY", "wait_any_frac_of_wave_cycles"),
                              ("SQ_ACTIVE_INST_ANY", "active_inst_frac_of_wave_cycles")):
                    if k in c:
                        b[nm] = c[k] / c["SQ_WAVE_CYCLES"]
            if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
                b["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            b["counter_source"] = f"profiles/{tag}_pmc.json [{scene}/{mode}] (rocprofv3 --pmc, one group per pass)"
            ent["bounds"] = b
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                ent["fetch_bytes_corrected"] = c["FETCH_SIZE"] * 1024 * 2
                ent["write_bytes"] = c["WRITE_SIZE"] * 1024
                ent["hbm_bytes_per_launch"] = ent["fetch_bytes_corrected"] + ent["write_bytes"]
            allpmc[f"{scene}/{mode}"] = ent
            if mode in ("early", "exact_early"):  # the two timed launches of bench.py: --fast and the default (exact)
                latest[f"{scene}/{'tolerance' if mode == 'early' else 'exact'}"] = {
                    "kernel_src_sha": sha, "render_src_sha": render_sha, "hbm_bytes_per_launch": ent.get("hbm_bytes_per_launch"), "bounds": b,
                    "rocprof_kernel_avg_ms": ent["rocprof_kernel_avg_ms"], "source": f"profiles/{tag}_pmc.json [{scene}/{mode}]"}
    json.dump({"tag": tag, "kernel_src_sha": sha, "render_src_sha": render_sha, "captures": allpmc}, open(os.path.join(prof, f"{tag}_pmc.json"), "w"), indent=1)
    if latest:
        json.dump(latest, open(os.path.join(prof, "pmc_latest.json"), "w"), indent=1)
    # the bench lines the runs themselves printed
    lines = {}
    for fn in sorted(glob.glob(os.path.join(src, "stats_*.json"))):
        txt = [l for l in open(fn).read().splitlines() if l.startswith("{")]
        if txt:
            lines[os.path.basename(fn)[:-5]] = json.loads(txt[-1])
    if lines:
        json.dump(lines, open(os.path.join(prof, f"{tag}_bench_under_rocprof.json"), "w"), indent=1)
    for k, v in allpmc.items():
        print(k, "kernel ms", v["rocprof_kernel_avg_ms"], json.dumps(v["bounds"]), "hbm", v.get("hbm_bytes_per_launch"))
    if os.path.exists(os.path.join(src, "failures.txt")):
        print(open(os.path.join(src, "failures.txt")).read())


if __name__ == "__main__":
    main()
