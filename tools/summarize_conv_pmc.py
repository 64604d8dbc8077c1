#!/usr/bin/env python3
"""Condense gpurun_out/<tag>_convpmc (tools/pmc_backbone.sh) into profiles/<tag>_mfma_util.json: one row per launch of the LAST
backbone pass and the LAST super-resolution pass, in execution order — kernel, grid, duration (kernel trace), MFMA-busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES / (active clocks x 1024 SIMDs)), SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, VALU-active fraction, LDS bank
conflict cycles / LDS active cycles.  GRBM_GUI_ACTIVE is summed over the 8 XCDs (/8 = active clocks of the launch).

    python tools/summarize_conv_pmc.py <tag>
"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag + "_convpmc")
N_SIMD = 1024


def short(name):
    import re
    m = re.match(r"_Z(\d+)", name)  # (rocprofv3 leaves some non-template kernels mangled: _Z19k_splitk_reduce_img12ReduceParams...)
    if m:
        n = int(m.group(1))
        return name[m.end():m.end() + n]
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def ours(name):
    return short(name).startswith("k_")


def last_pass(rows, key_start):
    """rows of the last pass: from the last launch whose name contains key_start (the first kernel of a pass) to the end of the
    p3d kernels."""
    idx = [i for i, r in enumerate(rows) if key_start in r["Kernel_Name"]]
    return rows[idx[-1]:] if idx else rows


_ss = os.path.join(src, "synthesis_src_sha.txt")
out = {"tag": tag, "kernel_src_sha": open(os.path.join(src, "kernel_src_sha.txt")).read().strip(),
       "synthesis_src_sha": open(_ss).read().strip() if os.path.exists(_ss) else None,  # the key bench.py's pipeline block matches on
       "note": __doc__.strip().splitlines()[1], "passes": {}}
for what, first in (("bb", "k_demod_plan"), ("sr", "k_demod_plan")):
    tr = glob.glob(os.path.join(src, f"trace_{what}", "**", "*kernel_trace.csv"), recursive=True)
    if not tr:
        continue
    rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
    seg = [r for r in last_pass(rows, first) if ours(r["Kernel_Name"])]
    launches = [{"kernel": short(r["Kernel_Name"]), "grid": f'{r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]}',
                 "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "lds": int(r["LDS_Block_Size"]),
                 "vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"])} for r in seg]
    # counters: per dispatch, same order of p3d launches in the pmc runs (one pass after 3 warm-up passes -> take the last len(seg))
    for grp in ("a", "b"):
        fn = glob.glob(os.path.join(src, f"pmc_{what}_{grp}", "**", "*counter_collection.csv"), recursive=True)
        if not fn:
            continue
        per = collections.OrderedDict()
        for r in csv.DictReader(open(fn[0])):
            if not ours(r["Kernel_Name"]):
                continue
            per.setdefault(int(r["Dispatch_Id"]), {"kernel": short(r["Kernel_Name"]), "grid": r["Grid_Size"]})[r["Counter_Name"]] = float(r["Counter_Value"])
        disp = [per[k] for k in sorted(per)][-len(launches):]
        for L, d in zip(launches, disp):
            if d["kernel"] != L["kernel"]:
                L.setdefault("pmc_mismatch", []).append(d["kernel"])
                continue
            clk = d.get("GRBM_GUI_ACTIVE", 0) / 8
            if grp == "a" and clk:
                L["active_clocks"] = clk
                L["mfma_busy_frac"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (clk * N_SIMD)
                if d.get("SQ_WAVE_CYCLES"):
                    L["wait_inst_frac_of_wave_cycles"] = d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"]
                    L["wait_any_frac_of_wave_cycles"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
                L["mfma_insts"], L["valu_insts"] = d.get("SQ_INSTS_MFMA"), d.get("SQ_INSTS_VALU")
            if grp == "b" and clk:
                L["valu_active_frac"] = d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (clk * N_SIMD)
                if d.get("SQ_ACTIVE_INST_LDS"):
                    L["lds_bank_conflict_frac_of_lds_cycles"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_ACTIVE_INST_LDS"]
    tot = sum(L["us"] for L in launches)
    out["passes"][what] = {"launches": len(launches), "p3d_kernel_us": tot, "rows": launches}
    print(what, "launches", len(launches), "p3d kernel time %.1f us" % tot)
    for L in launches:
        print("  %-28s %-16s %7.1f us  mfma %s  wait_inst %s  valu %s  ldsconf %s" % (
            L["kernel"], L["grid"], L["us"], *("%.2f" % L[k] if k in L else "  - " for k in
            ("mfma_busy_frac", "wait_inst_frac_of_wave_cycles", "valu_active_frac", "lds_bank_conflict_frac_of_lds_cycles"))))
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_mfma_util.json"), "w"), indent=1)
