#!/usr/bin/env python3
"""Ablation builds of the convolution kernels (development aid; timing only — results are wrong by construction): copies of
csrc/p3d_synthesis.hip with one stage of k_modconv_w2 / k_modconv_up_h<true> removed, built into tools/experiments/ablate/lib_<tag>.so
(load with P3D_LIB=...).  The shipped sources are not touched."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CS = os.path.join(ROOT, "panic3d-anime-reconstruction_amd", "csrc")
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(CS, "p3d_synthesis.hip")).read()

def variant(tag, edits):
    s = src
    for old, new in edits:
        assert old in s, (tag, old[:60])
        s = s.replace(old, new)
    s = s.replace('#include "../../include/panic3d_hip.h"', '#include "%s/include/panic3d_hip.h"' % ROOT)
    fn = os.path.join(HERE, f"p3d_synthesis_{tag}.hip")
    open(fn, "w").write(s)
    return fn

V = {
    "base": [],
    # w2: no activation re-staging after the first chunk
    "w2_nox": [("        if (more) conv_gload_w(p, pl, xn, sn, ic0 + 16, ic_end, rg);\n        const char* xh = xs[0][buf] + xlane;", "        const char* xh = xs[0][buf] + xlane;"),
               ("        if (more) conv_lstore_w(xs[0][buf ^ 1], pl, rg, p.sat);   // (waits", "        // (waits")],
    # w2: no weight re-staging after the first chunk
    "w2_now": [("        if (more) conv_glds_wh(p, pl, ws, tid, ic0 + 16, ic_end, 0);\n        // phase 2", "        // phase 2"),
               ("        if (more) conv_glds_wh(p, pl, ws, tid, ic0 + 16, ic_end, 1);\n        buf ^= 1;", "        buf ^= 1;")],
    # w2 + up_h: no epilogue stores (kept alive by an impossible condition)
    "nostore": [("                if (ch >= p.O) continue;\n                float v = acc[a][b][r] * HX_SPLIT_UNSCALE;", "                if (ch >= p.O || p.N > 0) continue;\n                float v = acc[a][b][r] * HX_SPLIT_UNSCALE;"),
                ("                if (ch < p.O) yout[(((size_t)n * p.O + ch) * p.OH + oy) * p.OW + ox] = SPLIT ?", "                if (ch < p.O && p.N < 0) yout[(((size_t)n * p.O + ch) * p.OH + oy) * p.OW + ox] = SPLIT ?")],
    # up_h<true>: no activation re-staging / no weight re-staging
    "up_nox": [("        if (more) conv_gload_h<SPLIT ? 0 : NT, false>(p, pl, xn, sn, ic0 + 16, ic_end, rg);\n        const char* xb = xs[buf] + xlane;\n        const char* wb = ws[SPLIT ? 0 : buf] + wlane;\n        auto run_pass", "        const char* xb = xs[buf] + xlane;\n        const char* wb = ws[SPLIT ? 0 : buf] + wlane;\n        auto run_pass"),
               ("            if (more) conv_lstore_hx<0, true>(xs[buf ^ 1], pl, rg, p.sat);\n            __builtin_amdgcn_s_waitcnt(0);\n            __syncthreads();         // a_hi is free", "            __builtin_amdgcn_s_waitcnt(0);\n            __syncthreads();         // a_hi is free")],
    "up_now": [("            if (more) conv_glds_w2<NT>(p, pl, ws[0], tid, ic0 + 16, ic_end, 0);\n            run_pass(WBYTES, 0);", "            run_pass(WBYTES, 0);"),
               ("            if (more) conv_glds_w2<NT>(p, pl, ws[0], tid, ic0 + 16, ic_end, 1);\n        } else {", "        } else {")],
}
procs = []
for tag, edits in V.items():
    fn = variant(tag, edits)
    out = os.path.join(HERE, f"lib_{tag}.so")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC", "-shared",
           f'-DP3D_SRC_HASH="abl_{tag}"', os.path.join(CS, "p3d_kernels.hip"), fn, os.path.join(CS, "p3d_mcubes.hip"), os.path.join(CS, "p3d_paste.hip"), "-o", out]
    procs.append((tag, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
for tag, p in procs:
    o = p.communicate()[0].decode()
    print(tag, "rc", p.returncode, o[-300:] if p.returncode else "")
