"""Developer aid: per-phase clock accounting inside k_render (s_memtime, per-wave LDS accumulators, one atomic per wave).

Instruments a TEMPORARY copy of the kernel sources (the product stays free of measurement hooks), builds it as
build/lib_phase.so and leaves the tree untouched:

    python tools/experiments/phase_timing/build_instrumented.py
    gpurun -- 'P3D_LIB=$PWD/build/lib_phase.so python tools/phase_timing.py'

The patches are anchored on source text; when an anchor no longer matches the script stops with the anchor it missed.
Slots: 0/1 coarse gather / MLP, 2/3 final gather / MLP, 4/5 coarse / final decode steps, 6 merge pre-pass, 7 waves,
8 weights -> LDS, 9 stratified, 10 coarse loop, 11 cdf, 12 draws + sort, 13 final loop, 14 select / skip, 15 march + composite,
16 wave lifetime.  Results of round 2: profiles/history/r02_notes.txt.
"""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CSRC = os.path.join(ROOT, "panic3d-anime-reconstruction_amd", "csrc")
tmp = tempfile.mkdtemp(prefix="p3d_phase_")
work = os.path.join(tmp, "pkg", "csrc")
shutil.copytree(CSRC, work)
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))


def patch(path, pairs):
    s = open(path).read()
    for a, b in pairs:
        if s.count(a) != 1:
            sys.exit(f"anchor not found exactly once in {os.path.basename(path)}: {a[:80]!r}")
        s = s.replace(a, b, 1)
    open(path, "w").write(s)

DECODE = [
    ('template <bool WANT_RGB, bool QUADG = false, bool LAZY = false, typename RSRC>\nP3D_DEV bool p3d_decode_wave(const float* lds, RSRC rs,',
     '__device__ unsigned long long g_phase[17];\n#define P3D_T() __builtin_amdgcn_s_memtime()\nP3D_DEV unsigned long long* p3d_phase_slots() {\n    __shared__ unsigned long long slots[8][16];\n    return slots[(threadIdx.x >> 6) & 7];\n}\ntemplate <bool WANT_RGB, bool QUADG = false, bool LAZY = false, typename RSRC>\nP3D_DEV bool p3d_decode_wave(const float* lds, RSRC rs,'),
    ('    const f32x16 X = p3d_gather_features<QUADG>(rs, g, cfg, px, py, pz, live);\n    return p3d_decode_features<WANT_RGB, LAZY>(lds, cfg, X, px, pz, sigma_out, rgb, live);\n',
     '    const unsigned long long t0 = P3D_T();\n    f32x16 X = p3d_gather_features<QUADG>(rs, g, cfg, px, py, pz, live);\n    p3d_pin16(X);\n    const unsigned long long t1 = P3D_T();\n    const bool have_ = p3d_decode_features<WANT_RGB, LAZY>(lds, cfg, X, px, pz, sigma_out, rgb, live);\n    asm volatile("" : "+v"(sigma_out));\n    const unsigned long long t2 = P3D_T();\n    if (__lane_id() == 0) { unsigned long long* sl = p3d_phase_slots(); sl[WANT_RGB ? 2 : 0] += t1 - t0; sl[WANT_RGB ? 3 : 1] += t2 - t1; sl[WANT_RGB ? 5 : 4] += 1ull; }\n    return have_;\n'),
    ('    const f32x16 X = p3d_gather_features<QUADG>(rs, g, cfg, px, py, pz, live);\n    return p3d_decode_features_fast<WANT_RGB, LAZY>(lds, cfg, X, px, pz, sigma_out, rgb, live);\n',
     '    const unsigned long long t0 = P3D_T();\n    f32x16 X = p3d_gather_features<QUADG>(rs, g, cfg, px, py, pz, live);\n    p3d_pin16(X);\n    const unsigned long long t1 = P3D_T();\n    const bool have_ = p3d_decode_features_fast<WANT_RGB, LAZY>(lds, cfg, X, px, pz, sigma_out, rgb, live);\n    asm volatile("" : "+v"(sigma_out));\n    const unsigned long long t2 = P3D_T();\n    if (__lane_id() == 0) { unsigned long long* sl = p3d_phase_slots(); sl[2] += t1 - t0; sl[3] += t2 - t1; sl[5] += 1ull; }\n    return have_;\n'),
]
KERNELS = [
    ('    extern __shared__ __attribute__((aligned(16))) float lds[];\n    p3d_load_mlp_to_lds(lds, p.w0, p.b0, p.w1, p.b1, !FAST);\n    if constexpr (FAST) p3d_load_mlp_f16_to_lds(lds, p.w0, p.w1);',
     '    extern __shared__ __attribute__((aligned(16))) float lds[];\n    const unsigned long long tw0 = P3D_T();\n    if ((threadIdx.x & 63) == 0) { unsigned long long* sl = p3d_phase_slots(); for (int q = 0; q < 16; ++q) sl[q] = 0; }\n    p3d_load_mlp_to_lds(lds, p.w0, p.b0, p.w1, p.b1, !FAST);\n    if constexpr (FAST) p3d_load_mlp_f16_to_lds(lds, p.w0, p.w1);'),
    ('    const int nwaves = blockDim.x >> 6;\n    long long tile = bs * nwaves + wave;\n    if (tile >= p.ntiles) return;  // no workgroup barrier below this line',
     '    const int nwaves = blockDim.x >> 6;\n    long long tile = bs * nwaves + wave;\n    if (tile >= p.ntiles) return;  // no workgroup barrier below this line\n    unsigned long long tm_ = P3D_T();\n    auto mark = [&](int slot) { const unsigned long long now = P3D_T(); if ((threadIdx.x & 63) == 0) p3d_phase_slots()[slot] += now - tm_; tm_ = now; };\n    mark(8);'),
    ('    float tmin = __builtin_inff(), tmax = -__builtin_inff();\n    if (Sf > 0) {\n        // ---- coarse pass, densities only -> ray-marcher weights: renderer.py:179-211',
     '    mark(9);\n    float tmin = __builtin_inff(), tmax = -__builtin_inff();\n    if (Sf > 0) {\n        // ---- coarse pass, densities only -> ray-marcher weights: renderer.py:179-211'),
    ('        // ---- sample_importance / sample_pdf: renderer.py:328-387 (per ray; both lanes of a pair compute the same)\n        const int Ns = Sc - 3;',
     '        mark(10);\n        // ---- sample_importance / sample_pdf: renderer.py:328-387 (per ray; both lanes of a pair compute the same)\n        const int Ns = Sc - 3;'),
    ('        const float* uu = p.u + ray * Sf;\n        constexpr int DB = 8;  // draws in flight',
     '        mark(11);\n        const float* uu = p.u + ray * Sf;\n        constexpr int DB = 8;  // draws in flight'),
    ('    // ---- final pass: merge on the fly (ties: coarse first = stable), decode, composite [rgb | xyz]:',
     '    mark(12);\n    // ---- final pass: merge on the fly (ties: coarse first = stable), decode, composite [rgb | xyz]:'),
    ('    // ---- outputs.  white_back and the [-1,1] rescale are per ray (ray_marcher.py:52-55); the depth clamp is global.',
     '    mark(13);\n    // ---- outputs.  white_back and the [-1,1] rescale are per ray (ray_marcher.py:52-55); the depth clamp is global.'),
    ('    if (lane == 0) {\n        atomicMin(p.gminmax, p3d_f2ord(tmin));\n        atomicMax(p.gminmax + 1, p3d_f2ord(tmax));\n        if (p.per_view_clamp) {  // a tile never straddles two views',
     '    if (lane == 0) {\n        unsigned long long* sl = p3d_phase_slots();\n        for (int q = 0; q < 16; ++q) if (q != 7) atomicAdd(&g_phase[q], sl[q]);\n        atomicAdd(&g_phase[16], P3D_T() - tw0); atomicAdd(&g_phase[7], 1ull);\n        atomicMin(p.gminmax, p3d_f2ord(tmin));\n        atomicMax(p.gminmax + 1, p3d_f2ord(tmax));\n        if (p.per_view_clamp) {  // a tile never straddles two views'),
    ('const char* p3d_build_info(void)',
     'int p3d_phase_read(unsigned long long* out, int reset) {\n    hipDeviceSynchronize();\n    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 17);\n    if (reset) { unsigned long long z[17] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)); }\n    return 0;\n}\nconst char* p3d_build_info(void)'),
    ('        if constexpr (EARLY) {\n            // the merge, once:',
     '        const unsigned long long tp_ = P3D_T();\n        if constexpr (EARLY) {\n            // the merge, once:'),
    ('            ci = 0;  // from here on: coarse samples among the first m merged ones (the fine index is m - ci)\n        }',
     '            ci = 0;  // from here on: coarse samples among the first m merged ones (the fine index is m - ci)\n        }\n        if ((threadIdx.x & 63) == 0) p3d_phase_slots()[6] += P3D_T() - tp_;'),
    ('            bool take_c, known = false;  // known: sigma = -1000 without a decode (reached here only behind a sigma > 602)\n            float t;',
     '            bool take_c, known = false;  // known: sigma = -1000 without a decode (reached here only behind a sigma > 602)\n            float t;\n            const unsigned long long ta_ = P3D_T();'),
    ('            const bool have = EARLY ? !done : true;\n            const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;',
     '            if ((threadIdx.x & 63) == 0) p3d_phase_slots()[14] += P3D_T() - ta_;\n            const bool have = EARLY ? !done : true;\n            const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;'),
    ('            const bool marching = have && !first;\n            float w = 0.0f, tm = 0.0f;',
     '            const unsigned long long tb_ = P3D_T();\n            const bool marching = have && !first;\n            float w = 0.0f, tm = 0.0f;'),
    ('                prev_skipped = skipped;\n                first = false;\n                if constexpr (!EARLY) ++m;\n            }\n        }\n    }',
     '                prev_skipped = skipped;\n                first = false;\n                if constexpr (!EARLY) ++m;\n            }\n            if ((threadIdx.x & 63) == 0) { p3d_phase_slots()[15] += P3D_T() - tb_; }\n        }\n    }'),
]
patch(os.path.join(work, "p3d_decode.hpp"), DECODE)
patch(os.path.join(work, "p3d_kernels.hip"), KERNELS)
os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
out = os.path.join(ROOT, "build", "lib_phase.so")
srcs = [os.path.join(work, f) for f in ("p3d_kernels.hip", "p3d_synthesis.hip", "p3d_mcubes.hip", "p3d_paste.hip")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
       "-fPIC", "-shared", *srcs, "-o", out]
subprocess.check_call(cmd)
shutil.rmtree(tmp)
print("built", out)
