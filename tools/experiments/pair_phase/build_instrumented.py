"""Developer aid: per-phase clock accounting inside k_render_pair (s_memtime at the phase boundaries, one atomic per wave and phase).
Instruments a TEMPORARY copy of the kernel sources and builds it as build/lib_pairphase.so; the tree stays untouched.

    python tools/experiments/pair_phase/build_instrumented.py
    gpurun -- 'P3D_LIB=$PWD/build/lib_pairphase.so python tools/experiments/pair_phase/read_phases.py'

Slots: 0 weights -> LDS + setup, 1 stratified, 2 coarse loop, 3 pdf / cdf, 4 draws + inverse CDF + sort, 5 merge pre-pass,
6 final loop, 7 outputs, 8 waves.
"""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import panic3d_amd as P
tmp = tempfile.mkdtemp(prefix="p3d_pairphase_")
work = os.path.join(tmp, "pkg", "csrc")
shutil.copytree(P._build.CSRC, work)
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
path = os.path.join(work, "p3d_kernels.hip")
s = open(path).read()
i0 = s.index("void k_render_pair(RenderParams p) {")
i1 = s.index("__global__ void k_minmax_init(")
head, seg, tail = s[:i0], s[i0:i1], s[i1:]


def rep(a, b):
    global seg
    if seg.count(a) != 1:
        sys.exit(f"anchor not found exactly once: {a[:90]!r} ({seg.count(a)})")
    seg = seg.replace(a, b, 1)


MARK = lambda k: f"    {{ const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&g_pair[{k}], t_ - tph); tph = t_; }}\n"
rep("    __syncthreads();\n    const int lane = threadIdx.x & 63,", "    unsigned long long tph = __builtin_amdgcn_s_memtime();\n    __syncthreads();\n    const int lane = threadIdx.x & 63,")
rep("    bool unsorted = false;\n    float tcmin", MARK(0) + "    bool unsorted = false;\n    float tcmin")
rep("    float tmin = __builtin_inff(), tmax = -__builtin_inff();\n    auto is_cropped", MARK(1) + "    float tmin = __builtin_inff(), tmax = -__builtin_inff();\n    auto is_cropped")
rep("        const int Ns = Sc - 3;\n", MARK(2) + "        const int Ns = Sc - 3;\n")
rep("        const float* uu = p.u + ray * Sf;\n", MARK(3) + "        const float* uu = p.u + ray * Sf;\n")
rep("    // ---- final pass: two merged samples per step\n", MARK(4) + "    // ---- final pass: two merged samples per step\n")
rep("        int m = 0, ci = 0;  // merged samples consumed so far", MARK(5) + "        int m = 0, ci = 0;  // merged samples consumed so far")
rep("    {\n        const float Wt = st.W;\n        float d = st.D / Wt;\n        if (d != d) d = __builtin_inff();\n", MARK(6) + "    {\n        const float Wt = st.W;\n        float d = st.D / Wt;\n        if (d != d) d = __builtin_inff();\n")
rep("    if (lane == 0) {\n        atomicMin(p.gminmax, p3d_f2ord(tmin));", MARK(7) + "    if (lane == 0) atomicAdd(&g_pair[8], 1ull);\n    if (lane == 0) {\n        atomicMin(p.gminmax, p3d_f2ord(tmin));")
head = head.replace("template <int NF>\n__global__ __launch_bounds__(64 * P3D_RENDER_WAVES, 1) ", "__device__ unsigned long long g_pair[16];\ntemplate <int NF>\n__global__ __launch_bounds__(64 * P3D_RENDER_WAVES, 1) ", 1)
assert "g_pair[16]" in head
tail = tail.replace('extern "C" {', 'extern "C" {\nvoid p3d_pair_phase_read(unsigned long long* out, int reset) {\n    hipDeviceSynchronize();\n    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair), 16 * 8);\n    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_pair), z, 16 * 8); }\n}\n', 1)
open(path, "w").write(head + seg + tail)
out = os.path.join(ROOT, "build", "lib_pairphase.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
flags = [f for f in P._build.HIPCC_FLAGS]
cmd = [P._build._hipcc()] + flags + ['-DP3D_SRC_HASH="pairphase"', "-I", os.path.join(tmp, "include")] + [os.path.join(work, x) for x in P._build.SOURCES] + ["-o", out]
subprocess.check_call(cmd)
print(out)
shutil.rmtree(tmp)
