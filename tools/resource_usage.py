#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of every shipped instantiation.

    python tools/resource_usage.py [-o profiles/history/r03_resource_usage.txt] [-D P3D_XYZ=1 ...]

Compiles each csrc/*.hip for gfx950 with the product's flags plus `-Rpass-analysis=kernel-resource-usage` (no GPU needed) and
prints one row per kernel: SGPRs, VGPRs, AGPRs, scratch bytes per lane, waves per SIMD, SGPR / VGPR spills, static LDS.
`--keep DIR` also leaves the ISA (`*.s`) of every source there, for reading the loops.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import panic3d_amd as P  # noqa: E402

B = P._build


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", x).replace("void ", "") for x in out[:len(names)]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", "--out")
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--keep")
    ap.add_argument("--sources", nargs="*", default=B.SOURCES)
    a = ap.parse_args()
    flags = [f for f in B.HIPCC_FLAGS if f != "-shared"] + ["-D" + d for d in a.D]
    lines = ["# kernel resource usage, hipcc " + " ".join(flags) + " -Rpass-analysis=kernel-resource-usage",
             "# kernel sources: " + B.source_hash(),
             "%-64s %5s %5s %5s %8s %4s %7s %7s %7s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratchB", "occ", "sSpill", "vSpill", "LDS")]
    tmp = a.keep or tempfile.mkdtemp()
    os.makedirs(tmp, exist_ok=True)
    for s in a.sources:
        obj = os.path.join(tmp, s + ".o")
        cmd = [B._hipcc()] + flags + ["-c", os.path.join(B.CSRC, s), "-o", obj, "-Rpass-analysis=kernel-resource-usage"]
        if a.keep:
            cmd.append("--save-temps=obj")
        txt = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
        blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
        names = demangle([b.split("\n")[0].strip() for b in blocks])
        lines.append("# " + s)
        for nm, b in zip(names, blocks):
            def g(k):
                m = re.search(k + r": (\d+)", b)
                return int(m.group(1)) if m else -1
            lines.append("%-64s %5d %5d %5d %8d %4d %7d %7d %7d" % (
                nm[:64], g("TotalSGPRs"), g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                g("SGPRs Spill"), g("VGPRs Spill"), g(r"LDS Size \[bytes/block\]")))
    text = "\n".join(lines) + "\n"
    sys.stdout.write(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
