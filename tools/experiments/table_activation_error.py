#!/usr/bin/env python3
"""What would the 'table activations' lever of DESIGN.md §9 do to the agreement with the reference?  CPU-only experiment.

Builds the oracle twice — as shipped (polynomial exp / log1p softplus and sigmoid in the decoder) and with
-DOR_EXPERIMENT_TABLE_ACT (cubic-Hermite tables with step 1/16 for the 64 hidden softplus and the 32 colour sigmoids; the
marcher's own softplus / exp are untouched) — and compares both against every golden the reference produced:
max |error| per output, inverse-CDF index mismatches, and how far the two variants are from each other.
    python tools/experiments/table_activation_error.py            (writes profiles/history/r01_table_activation_experiment.txt)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import p3d_testing as T
from oracle import oracle as O

exp_so = "/tmp/libp3d_oracle_tableact.so"
# a patched TEMPORARY copy of the oracle: the decoder's two activation call sites go to the table functions
src = open(os.path.join(ROOT, "oracle", "p3d_oracle.c")).read()
marker = "static void or_sample_plane("
head, tail = src.split(marker, 1)
tail = tail.replace("h[n] = or_softplus(a);", "h[n] = or_softplus_h(a);").replace("float sg = or_sigmoid(a);", "float sg = or_sigmoid_c(a);")
assert "or_softplus_h(a)" in tail and "or_sigmoid_c(a)" in tail
patched = head + '#include "%s"\n' % os.path.join(ROOT, "tools", "experiments", "table_activation_oracle.h") + marker + tail
open("/tmp/p3d_oracle_tableact.c", "w").write(patched)
subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-mavx2", "-fopenmp", "-fPIC", "-shared",
                       "-I" + os.path.join(ROOT, "oracle"), "-include", "math.h", "/tmp/p3d_oracle_tableact.c", os.path.join(ROOT, "oracle", "p3d_oracle_mc.c"),
                       "-o", exp_so, "-lm"])
base_lib = O.lib()
exp_lib = ctypes.CDLL(exp_so)


def with_lib(lib, fn):
    O._LIB = lib
    try:
        return fn()
    finally:
        O._LIB = base_lib


lines = []
def log(s=""):
    print(s); lines.append(s)

log("decoder activations: shipped polynomials vs cubic-Hermite tables (step 1/16), both against the reference's goldens")
log(f"{'fixture':28s} {'variant':6s} {'feat':>9s} {'depth':>9s} {'wsum':>9s} {'xyz':>9s} {'inds!=':>7s} {'sigma_c':>9s}")
worst = {}
for name in T.RENDER_GOLDENS:
    g = T.load_golden(name + ".npz")
    inp = T.golden_render_inputs(g)
    outs = {}
    for tag, lib in (("poly", base_lib), ("table", exp_lib)):
        f = lambda: O.render(inp["planes"], inp["rays_o"], inp["rays_d"], inp["jitter"], inp["u"],
                             O.prescale_mlp(*inp["raw_mlp"], lr_mul=inp["lr_mul"]), O.make_opts(inp["ro"], **inp["kw"]), dumps=True)
        feat, depth, wsum, xyz, d = with_lib(lib, f)
        outs[tag] = (feat, depth, wsum, xyz, d)
        e = [float(np.abs(a - g[k]).max()) for a, k in ((feat, "feat"), (depth, "depth"), (wsum, "wsum"), (xyz, "xyz"))]
        nidx = int((d["inds"] != g["inds"]).sum()) if "inds" in g and d.get("inds") is not None and g["inds"].size else 0
        es = float(np.abs(d["sigma_coarse"] - g["sigma_coarse"]).max()) if "sigma_coarse" in g else float("nan")
        log(f"{name:28s} {tag:6s} {e[0]:9.2e} {e[1]:9.2e} {e[2]:9.2e} {e[3]:9.2e} {nidx:7d} {es:9.2e}")
        for k, v in zip(("feat", "depth", "wsum", "xyz"), e):
            worst[(tag, k)] = max(worst.get((tag, k), 0.0), v)
    a, b = outs["poly"], outs["table"]
    log(f"{'':28s} {'p-t':6s} " + " ".join(f"{float(np.abs(x - y).max()):9.2e}" for x, y in zip(a[:4], b[:4])) +
        f" {int((a[4]['inds'] != b[4]['inds']).sum()) if a[4].get('inds') is not None else 0:7d}")
log()
log("worst case over the render fixtures (tolerances of tests/test_oracle_golden.py: feat 1e-4, depth 2e-5, wsum 3e-5, xyz 1e-4):")
for tag in ("poly", "table"):
    log(f"  {tag:6s} " + "  ".join(f"{k} {worst[(tag, k)]:.2e}" for k in ("feat", "depth", "wsum", "xyz")))
open(os.path.join(ROOT, "profiles", "r01_table_activation_experiment.txt"), "w").write("\n".join(lines) + "\n")
