// p3d_mcubes.hip — iso-surface extraction of the density grid on the device: replaces the host-side
// skimage.measure.marching_cubes call of _util/eg3d_metrics3d.py:186-210 (called from _scripts/eval/generate.py:98-103)
// so that the 512^3 density volume never leaves HBM; only the mesh (a few MB) does.  gfx950 only.
//
// HBM-bound integer / index work.  One workgroup owns a GROUP of MC_G consecutive grid rows (fixed a; b0 .. b0+MC_G-1; all c),
// one thread 4 consecutive points of each of them: the (a,b,c) arithmetic is scalar, every volume access is a coalesced 16-byte
// row read, the 2 x (MC_G+1) rows a group touches are all requested before the first is used (the kernels are latency-bound
// otherwise: measured 1.9 TB/s with one row per workgroup), and each row is fetched once per group instead of twice.
// Groups are walked in flat grid order, which fixes the vertex / face order (oracle/p3d_oracle_mc.c reproduces it -> bit-exact):
//   k_mc_classify    per grid point: #owned crossed edges (0..3) and #triangles of its cube (table) -> one packed sum per group
//   k_mc_scan_rows   one workgroup per plane a: exclusive scan of that plane's group sums, plane totals
//   k_mc_scan_planes one workgroup: exclusive scan of the n plane totals (64-bit), grand totals
//   k_mc_compact     same walk + in-group scans -> global vertex id per point, packed with its 3-bit crossed-edge mask into
//                    vert_info[n^3] (the only per-point array, 4 B), and one compact record per vertex / per triangle, parked in
//                    the output buffers themselves
//   k_mc_emit_verts  one thread per vertex: position / normal / value from its record
//   k_mc_emit_tris   one thread per triangle: the three vertex ids from vert_info of the owning grid points (corners of the
//                    same cube: L1/L2 hits)
// Algorithmic bytes per grid point: 3 volume reads (4 B each; the a+1 plane comes from L2) + vert_info write + read = 20 B
// -> 2.7 GB at 512^3 (+ the mesh itself).
// The case table is include/p3d_mc_table.h (derived by tools/gen_mc_table.py, not copied from anywhere).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define P3D_MC_QUAL static __device__ const
#include "../../include/p3d_mc_table.h"
#include "../../include/panic3d_hip.h"

#define MC_T 256      // max threads per workgroup; a thread owns MC_PT consecutive points of the row
#define MC_PT 4
#define MC_MAXN 1024  // = MC_T * MC_PT: one workgroup covers whole rows
#define MC_G 8        // rows per workgroup

struct McParams {
    const float* vol;
    int n;
    int flip0;
    float level;
    int gpp;                       // row groups per plane = ceil(n / MC_G)
    unsigned* row_sum;             // [n*gpp] packed (tris << 16 | verts) per row group (<= 40960 / 24576)
    uint2* row_off;                // [n*gpp] exclusive (verts, tris) offsets of the group within its plane
    unsigned long long* plane;     // [2][n] plane totals, then exclusive plane offsets (verts / tris)
    unsigned long long* totals;    // [2]
    unsigned* vert_info;           // [n^3] (first vertex id of the point << 3) | crossed-edge mask
    float* verts;
    float* normals;
    float* values;
    int* faces;
};

__device__ __forceinline__ const float* mc_row_ptr(const McParams& p, int a, int b) {
    const int aa = p.flip0 ? p.n - 1 - a : a;
    return p.vol + ((size_t)aa * p.n + b) * p.n;
}

// r[c .. c+4] of one volume row (the 5th value closes the last cube of the thread); VEC: n % 4 == 0 -> one 16-byte load
template <bool VEC>
__device__ __forceinline__ void mc_load5(const float* r, int c, int n, float v[5]) {
    if (VEC) {
        const float4 q = *reinterpret_cast<const float4*>(r + c);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = c + j < n ? r[c + j] : 0.0f;
    }
    v[4] = c + 4 < n ? r[c + 4] : 0.0f;
}

// The volume rows of one group: f[r][0] = row (a, b0+r), f[r][1] = row (a+1, b0+r), r = 0..MC_G (the extra row closes the last
// cubes); each holds the thread's points c .. c+4.  Rows past the border stay 0 and are masked by the callers' hb / ha.
struct McGroup { float f[MC_G + 1][2][5]; };

template <bool VEC>
__device__ __forceinline__ void mc_load_group(const McParams& p, int a, int b0, int c, McGroup& T) {
    const int n = p.n;
    const bool ha = a + 1 < n;
#pragma unroll
    for (int r = 0; r <= MC_G; ++r) {
#pragma unroll
        for (int i = 0; i < 5; ++i) T.f[r][0][i] = T.f[r][1][i] = 0.0f;
        if (b0 + r < n) {  // uniform
            mc_load5<VEC>(mc_row_ptr(p, a, b0 + r), c, n, T.f[r][0]);
            if (ha) mc_load5<VEC>(mc_row_ptr(p, a + 1, b0 + r), c, n, T.f[r][1]);
        }
    }
}

// point j (0..3) of row r of the group: crossed owned edges (bit 0/1/2: along c/b/a) and cube case (or -1)
__device__ __forceinline__ void mc_point(const McParams& p, const McGroup& T, int a, int b, int r, int c, int j, unsigned& cross,
                                         int& ccase) {
    const int n = p.n;
    const float lv = p.level;
    const bool valid = c + j < n, hc = c + j + 1 < n, hb = b + 1 < n, ha = a + 1 < n;
    const bool i0 = T.f[r][0][j] > lv, ic = T.f[r][0][j + 1] > lv, ib = T.f[r + 1][0][j] > lv, ia = T.f[r][1][j] > lv;
    cross = !valid ? 0u : (((hc && ic != i0) ? 1u : 0u) | ((hb && ib != i0) ? 2u : 0u) | ((ha && ia != i0) ? 4u : 0u));
    // corner v at (da,db,dc) = (v>>2&1, v>>1&1, v&1): 0 self, 1 +c, 2 +b, 3 +b+c, 4 +a, 5 +a+c, 6 +a+b, 7 +a+b+c
    ccase = (hc && hb && ha) ? ((i0 ? 1 : 0) | (ic ? 2 : 0) | (ib ? 4 : 0) | (T.f[r + 1][0][j + 1] > lv ? 8 : 0) | (ia ? 16 : 0) |
                                (T.f[r][1][j + 1] > lv ? 32 : 0) | (T.f[r + 1][1][j] > lv ? 64 : 0) |
                                (T.f[r + 1][1][j + 1] > lv ? 128 : 0))
                             : -1;
}

// exclusive scan of one unsigned per thread over the workgroup (<= 4 waves); *total = workgroup sum
__device__ __forceinline__ unsigned mc_wg_scan(unsigned x, unsigned* total, unsigned* lds /* [4] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    unsigned inc = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned y = __shfl_up(inc, d, 64);
        if (lane >= d) inc += y;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < MC_T / 64; ++w) {
        const unsigned s = w < nw ? lds[w] : 0u;
        before += w < wave ? s : 0u;
        all += s;
    }
    __syncthreads();
    *total = all;
    return before + inc - x;
}

// the case tables live in LDS: lanes of a wave look up different cases (divergent byte reads)
__device__ __forceinline__ void mc_stage_ntri(unsigned char* l_ntri) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) l_ntri[i] = P3D_MC_NTRI[i];
}

// exclusive scans of MC_G per-thread counts (each <= 20, workgroup sums <= 5120) packed two per word; tot[r] = row totals
__device__ __forceinline__ void mc_scan_rows8(const unsigned cnt[MC_G], unsigned excl[MC_G], unsigned tot[MC_G], unsigned* lds) {
#pragma unroll
    for (int k = 0; k < MC_G / 2; ++k) {
        unsigned t;
        const unsigned e = mc_wg_scan(cnt[2 * k] | (cnt[2 * k + 1] << 16), &t, lds);
        excl[2 * k] = e & 0xffffu; excl[2 * k + 1] = e >> 16;
        tot[2 * k] = t & 0xffffu; tot[2 * k + 1] = t >> 16;
    }
}

template <bool VEC>
__global__ __launch_bounds__(MC_T) void k_mc_classify(McParams p) {
    __shared__ unsigned lds[4];
    __shared__ unsigned char l_ntri[256];
    const int grp = blockIdx.x, a = grp / p.gpp, b0 = (grp - a * p.gpp) * MC_G;
    const int c = MC_PT * threadIdx.x;
    McGroup T;
    if (c < p.n) mc_load_group<VEC>(p, a, b0, c, T);  // all volume loads in flight while the table is staged
    mc_stage_ntri(l_ntri);
    __syncthreads();
    unsigned acc = 0;
    if (c < p.n) {
#pragma unroll
        for (int r = 0; r < MC_G; ++r) {
            if (b0 + r >= p.n) break;
#pragma unroll
            for (int j = 0; j < MC_PT; ++j) {
                unsigned cross; int cs;
                mc_point(p, T, a, b0 + r, r, c, j, cross, cs);
                acc += (unsigned)__popc(cross) | ((cs >= 0 ? (unsigned)l_ntri[cs] : 0u) << 16);
            }
        }
    }
    unsigned total;
    mc_wg_scan(acc, &total, lds);
    if (threadIdx.x == 0) p.row_sum[grp] = total;
}

// block = plane a; gpp <= 128 row groups, one per thread
__global__ __launch_bounds__(MC_T) void k_mc_scan_rows(McParams p) {
    __shared__ unsigned lds[MC_T / 64 + 1];
    const int a = blockIdx.x, i = threadIdx.x;
    const unsigned x = i < p.gpp ? p.row_sum[(size_t)a * p.gpp + i] : 0u;
    unsigned totv, tott;
    const unsigned ev = mc_wg_scan(x & 0xffffu, &totv, lds);
    const unsigned et = mc_wg_scan(x >> 16, &tott, lds);
    if (i < p.gpp) p.row_off[(size_t)a * p.gpp + i] = make_uint2(ev, et);
    if (i == 0) { p.plane[a] = totv; p.plane[p.n + a] = tott; }
}

__global__ __launch_bounds__(MC_T) void k_mc_scan_planes(McParams p) {
    __shared__ unsigned long long lds[MC_T];
    for (int which = 0; which < 2; ++which) {
        unsigned long long* arr = p.plane + (size_t)which * p.n;
        unsigned long long v[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 4 * threadIdx.x + k;
            v[k] = i < p.n ? arr[i] : 0ull;
            s += v[k];
        }
        lds[threadIdx.x] = s;
        __syncthreads();
        unsigned long long before = 0, all = 0;
        for (int j = 0; j < MC_T; ++j) {  // 256 LDS broadcasts; this kernel runs once per extraction
            const unsigned long long x = lds[j];
            before += j < (int)threadIdx.x ? x : 0ull;
            all += x;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 4 * threadIdx.x + k;
            if (i < p.n) arr[i] = before;
            before += v[k];
        }
        if (threadIdx.x == 0) p.totals[which] = all;
    }
}

// central difference along `axis` (0: c, 1: b, 2: a) at grid point (a,b,c), one-sided at the borders
__device__ __forceinline__ float mc_grad1(const McParams& p, int a, int b, int c, int axis) {
    const int i = axis == 0 ? c : (axis == 1 ? b : a);
    const int lo = i > 0 ? i - 1 : i, hi = i + 1 < p.n ? i + 1 : i;
    float flo, fhi;
    if (axis == 0) { const float* r = mc_row_ptr(p, a, b); flo = r[lo]; fhi = r[hi]; }
    else if (axis == 1) { flo = mc_row_ptr(p, a, lo)[c]; fhi = mc_row_ptr(p, a, hi)[c]; }
    else { flo = mc_row_ptr(p, lo, b)[c]; fhi = mc_row_ptr(p, hi, b)[c]; }
    const float d = fhi - flo;
    return (hi - lo == 2) ? 0.5f * d : d;
}

// Pass C: the same streaming walk as k_mc_classify, now with the global offsets known: writes vert_info for every point and ONE
// compact record per vertex and per triangle INTO THE OUTPUT BUFFERS THEMSELVES — values[vid] (as u32) = (point id << 2) | edge slot,
// faces[3t] = point id of the cube, faces[3t+1] = case | (index of the triangle in its cube << 8) — so that the two emit kernels
// run one thread per vertex / per triangle (dense, perfectly balanced; each thread reads its own record and overwrites it).
template <bool VEC>
__global__ __launch_bounds__(MC_T) void k_mc_compact(McParams p) {
    __shared__ unsigned lds[4];
    __shared__ unsigned char l_ntri[256];
    const int grp = blockIdx.x, a = grp / p.gpp, b0 = (grp - a * p.gpp) * MC_G;
    const int c = MC_PT * threadIdx.x;
    McGroup T;
    unsigned cv[MC_G], ct[MC_G], ev[MC_G], et[MC_G], tv[MC_G], tt[MC_G];
#pragma unroll
    for (int r = 0; r < MC_G; ++r) cv[r] = ct[r] = 0;
    if (c < p.n) mc_load_group<VEC>(p, a, b0, c, T);
    mc_stage_ntri(l_ntri);
    __syncthreads();
    if (c < p.n) {
#pragma unroll
        for (int r = 0; r < MC_G; ++r) {
            if (b0 + r >= p.n) break;
#pragma unroll
            for (int j = 0; j < MC_PT; ++j) {
                unsigned cross; int cs;
                mc_point(p, T, a, b0 + r, r, c, j, cross, cs);
                cv[r] += (unsigned)__popc(cross);
                ct[r] += cs >= 0 ? (unsigned)l_ntri[cs] : 0u;
            }
        }
    }
    mc_scan_rows8(cv, ev, tv, lds);
    mc_scan_rows8(ct, et, tt, lds);
    if (c >= p.n) return;
    const uint2 off = p.row_off[grp];
    unsigned long long vbase = p.plane[a] + off.x, tbase = p.plane[p.n + a] + off.y;
    unsigned* vrec = reinterpret_cast<unsigned*>(p.values);
#pragma unroll
    for (int r = 0; r < MC_G; ++r) {
        const int b = b0 + r;
        if (b >= p.n) break;
        unsigned vid = (unsigned)(vbase + ev[r]), tid = (unsigned)(tbase + et[r]);
        vbase += tv[r]; tbase += tt[r];
        const unsigned pid0 = ((unsigned)a * p.n + b) * p.n + c;
        unsigned info[MC_PT];
#pragma unroll
        for (int j = 0; j < MC_PT; ++j) {
            unsigned cross; int cs;
            mc_point(p, T, a, b, r, c, j, cross, cs);
            info[j] = (vid << 3) | cross;
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (cross & (1u << s)) vrec[vid++] = ((pid0 + j) << 2) | (unsigned)s;
            const unsigned nt = cs >= 0 ? (unsigned)l_ntri[cs] : 0u;
            for (unsigned i = 0; i < nt; ++i, ++tid) {
                p.faces[3 * (size_t)tid] = (int)(pid0 + j);
                p.faces[3 * (size_t)tid + 1] = cs | (int)(i << 8);
            }
        }
        unsigned* vi = p.vert_info + pid0;
        if (VEC) *reinterpret_cast<uint4*>(vi) = make_uint4(info[0], info[1], info[2], info[3]);
        else {
#pragma unroll
            for (int j = 0; j < MC_PT; ++j) if (c + j < p.n) vi[j] = info[j];
        }
    }
}

// one thread per vertex
__global__ __launch_bounds__(256) void k_mc_emit_verts(McParams p, unsigned nverts) {
    const unsigned vid = blockIdx.x * 256u + threadIdx.x;
    if (vid >= nverts) return;
    const unsigned rec = reinterpret_cast<const unsigned*>(p.values)[vid];
    const unsigned pid = rec >> 2;
    const int s = (int)(rec & 3u), n = p.n;
    const int c = (int)(pid % (unsigned)n), ab = (int)(pid / (unsigned)n), b = ab % n, a = ab / n;
    const int a1 = a + (s == 2), b1 = b + (s == 1), c1 = c + (s == 0);
    const float f0 = mc_row_ptr(p, a, b)[c], f1 = mc_row_ptr(p, a1, b1)[c1];
    const float t = (p.level - f0) / (f1 - f0);
    float pos[3] = {(float)a, (float)b, (float)c};  // (a, b, c) = skimage's (axis 0, 1, 2) vertex order
    pos[2 - s] += t;
    float g[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float g0 = mc_grad1(p, a, b, c, ax), g1 = mc_grad1(p, a1, b1, c1, ax);
        g[ax] = __builtin_fmaf(t, g1 - g0, g0);
    }
    // |g| through binary64: sqrt in double then rounding to float IS the correctly rounded float sqrt (53 >= 2*24+2); the f32
    // hardware sqrt is not.  normal = -g/|g| in (a,b,c) order: g[2] is d/da.
    const float len2 = __builtin_fmaf(g[0], g[0], __builtin_fmaf(g[1], g[1], g[2] * g[2]));
    const float len = (float)__dsqrt_rn((double)len2);
    const bool ok = len > 0.0f;
    float* V = p.verts + 3 * (size_t)vid;
    float* N = p.normals + 3 * (size_t)vid;
    V[0] = pos[0]; V[1] = pos[1]; V[2] = pos[2];
    N[0] = ok ? -g[2] / len : 0.0f;
    N[1] = ok ? -g[1] / len : 0.0f;
    N[2] = ok ? -g[0] / len : 0.0f;
    p.values[vid] = f0 > f1 ? f0 : f1;
}

// one thread per triangle
__global__ __launch_bounds__(256) void k_mc_emit_tris(McParams p, unsigned ntris) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= ntris) return;
    int* F = p.faces + 3 * (size_t)t;
    const unsigned pid = (unsigned)F[0];
    const int rec = F[1], cs = rec & 255, i = rec >> 8;
    const size_t n = p.n;
    int id[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int e = P3D_MC_TRI[cs][3 * i + k];
        // lower corner of edge e: P3D_MC_EDGE[e][0] = 0,2,4,6, 0,1,4,5, 0,1,2,3 packed 3 bits each; slot = e >> 2
        const int v0 = (int)((0x688b08d10ull >> (3 * e)) & 7), slot = e >> 2;
        const size_t owner = pid + ((v0 >> 2) & 1) * n * n + ((v0 >> 1) & 1) * n + (v0 & 1);
        const unsigned info = p.vert_info[owner];
        id[k] = (int)((info >> 3) + __popc(info & ((1u << slot) - 1u)));
    }
    F[0] = id[0]; F[1] = id[1]; F[2] = id[2];
}

extern "C" {

size_t p3d_mc_workspace_bytes(int n) {
    if (n < 2 || n > MC_MAXN) return 0;
    const size_t nn = (size_t)n * n, ng = (size_t)n * ((n + MC_G - 1) / MC_G);
    return 256 + 16 * (size_t)MC_MAXN + 12 * ((ng + 3) / 4 * 4) + 4 * nn * n;
}

static int mc_params(McParams& p, const float* vol, int n, int flip0, float level, void* workspace, size_t workspace_bytes) {
    if (!vol || !workspace) return P3D_E_ARG;
    if (n < 2 || n > MC_MAXN) return P3D_E_RANGE;
    if (workspace_bytes < p3d_mc_workspace_bytes(n) || ((uintptr_t)workspace & 15)) return P3D_E_WORKSPACE;
    p.vol = vol; p.n = n; p.flip0 = flip0 ? 1 : 0; p.level = level;
    p.gpp = (n + MC_G - 1) / MC_G;
    const size_t ng = ((size_t)n * p.gpp + 3) / 4 * 4;  // keeps vert_info 16-byte aligned
    char* w = (char*)workspace;
    p.totals = (unsigned long long*)w;            w += 256;
    p.plane = (unsigned long long*)w;             w += 16 * (size_t)MC_MAXN;
    p.row_off = (uint2*)w;                        w += 8 * ng;
    p.row_sum = (unsigned*)w;                     w += 4 * ng;
    p.vert_info = (unsigned*)w;
    p.verts = p.normals = p.values = nullptr; p.faces = nullptr;
    return P3D_OK;
}

int p3d_mc_count_f32(const float* vol, int n, int flip0, float level, void* workspace, size_t workspace_bytes,
                     uint64_t* out_counts, void* stream) {
    McParams p;
    int rc = mc_params(p, vol, n, flip0, level, workspace, workspace_bytes);
    if (rc) return rc;
    if (!out_counts) return P3D_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const unsigned threads = (unsigned)(((n + MC_PT - 1) / MC_PT + 63) / 64 * 64);
    if (n % 4 == 0 && ((uintptr_t)vol & 15) == 0) hipLaunchKernelGGL(k_mc_classify<true>, dim3((unsigned)(n * p.gpp)), dim3(threads), 0, s, p);
    else hipLaunchKernelGGL(k_mc_classify<false>, dim3((unsigned)(n * p.gpp)), dim3(threads), 0, s, p);
    hipLaunchKernelGGL(k_mc_scan_rows, dim3((unsigned)n), dim3(MC_T), 0, s, p);
    hipLaunchKernelGGL(k_mc_scan_planes, dim3(1), dim3(MC_T), 0, s, p);
    hipError_t e = hipMemcpyAsync(out_counts, p.totals, 16, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
    e = hipGetLastError();
    return e == hipSuccess ? P3D_OK : (int)e;
}

int p3d_mc_emit_f32(const float* vol, int n, int flip0, float level, void* workspace, size_t workspace_bytes, int64_t nverts,
                    int64_t ntris, float* out_verts, float* out_normals, float* out_values, int32_t* out_faces, void* stream) {
    McParams p;
    int rc = mc_params(p, vol, n, flip0, level, workspace, workspace_bytes);
    if (rc) return rc;
    if (nverts < 0 || ntris < 0) return P3D_E_ARG;
    if (nverts >= (1ll << 29) || ntris >= (1ll << 29)) return P3D_E_RANGE;  // vert_info packs the vertex id in 29 bits
    if (nverts == 0 || ntris == 0) return P3D_OK;
    if (!out_verts || !out_normals || !out_values || !out_faces) return P3D_E_ARG;
    p.verts = out_verts; p.normals = out_normals; p.values = out_values; p.faces = out_faces;
    hipStream_t s = (hipStream_t)stream;
    const unsigned threads = (unsigned)(((n + MC_PT - 1) / MC_PT + 63) / 64 * 64);
    if (n % 4 == 0 && ((uintptr_t)vol & 15) == 0) hipLaunchKernelGGL(k_mc_compact<true>, dim3((unsigned)(n * p.gpp)), dim3(threads), 0, s, p);
    else hipLaunchKernelGGL(k_mc_compact<false>, dim3((unsigned)(n * p.gpp)), dim3(threads), 0, s, p);
    hipLaunchKernelGGL(k_mc_emit_verts, dim3((unsigned)((nverts + 255) / 256)), dim3(256), 0, s, p, (unsigned)nverts);
    hipLaunchKernelGGL(k_mc_emit_tris, dim3((unsigned)((ntris + 255) / 256)), dim3(256), 0, s, p, (unsigned)ntris);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? P3D_OK : (int)e;
}

}  // extern "C"
