// p3d_mcubes.hip — iso-surface extraction of the density grid on the device: replaces the host-side
// skimage.measure.marching_cubes call of _util/eg3d_metrics3d.py:186-210 (called from _scripts/eval/generate.py:98-103)
// so that the 512^3 density volume never leaves HBM; only the mesh (a few MB) does.  gfx950 only.
//
// HBM-bound integer / index work.  One 256-thread workgroup owns one grid ROW (fixed a, b; all c), so the (a,b,c) index
// arithmetic is scalar and every volume access is a coalesced row read; rows are walked in flat grid order, which fixes the
// vertex / face order (oracle/p3d_oracle_mc.c reproduces it -> bit-exact parity):
//   k_mc_classify   per grid point: #owned crossed edges (0..3) and #triangles of its cube (table) -> one packed sum per row
//   k_mc_scan_rows  one workgroup per plane a: exclusive scan of that plane's n row sums, plane totals
//   k_mc_scan_planes one workgroup: exclusive scan of the n plane totals (64-bit), grand totals
//   k_mc_verts      in-row scan -> global vertex id per point, packed with its 3-bit crossed-edge mask into vert_info[n^3]
//                   (the only per-point array, 4 B); writes position / normal / value of every vertex
//   k_mc_tris       in-row scan -> triangle id; fetches the three vertex ids of every triangle from vert_info of the owning
//                   grid points (corners of the same cube: L1/L2 hits); writes faces
// Algorithmic bytes per grid point: 3 volume reads (4 B each, neighbour rows come from L2) + vert_info write + read = 20 B
// -> 2.7 GB at 512^3 (+ the mesh itself).
// The case table is include/p3d_mc_table.h (derived by tools/gen_mc_table.py, not copied from anywhere).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define P3D_MC_QUAL static __device__ const
#include "../../include/p3d_mc_table.h"
#include "../../include/panic3d_hip.h"

#define MC_T 256      // max threads per workgroup; a thread owns MC_PT consecutive points of the row
#define MC_PT 4
#define MC_MAXN 1024  // = MC_T * MC_PT: one workgroup covers a whole row

struct McParams {
    const float* vol;
    int n;
    int flip0;
    float level;
    unsigned* row_sum;             // [n*n] packed (tris << 16 | verts) per row (<= 5120 / 3072)
    uint2* row_off;                // [n*n] exclusive (verts, tris) offsets of the row within its plane
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

// The thread's MC_PT points c .. c+3 of row (a,b): crossed owned edges (bit 0/1/2: along c/b/a) and cube case (or -1),
// f[0][j] = value at the point, f[1..3][j] = the (a,b+1), (a+1,b), (a+1,b+1) rows (0 past the border; masked by hb / ha).
struct McTile {
    float f[4][5];
    unsigned cross[MC_PT];
    int ccase[MC_PT];
};

template <bool VEC>
__device__ __forceinline__ void mc_tile(const McParams& p, int a, int b, int c, McTile& T) {
    const int n = p.n;
    const bool hb = b + 1 < n, ha = a + 1 < n;
    const float lv = p.level;
    mc_load5<VEC>(mc_row_ptr(p, a, b), c, n, T.f[0]);
    if (hb) mc_load5<VEC>(mc_row_ptr(p, a, b + 1), c, n, T.f[1]);
    if (ha) mc_load5<VEC>(mc_row_ptr(p, a + 1, b), c, n, T.f[2]);
    if (ha && hb) mc_load5<VEC>(mc_row_ptr(p, a + 1, b + 1), c, n, T.f[3]);
#pragma unroll
    for (int j = 0; j < MC_PT; ++j) {
        const bool valid = c + j < n, hc = c + j + 1 < n;
        const bool i0 = T.f[0][j] > lv;
        const bool ic = T.f[0][j + 1] > lv, ib = T.f[1][j] > lv, ia = T.f[2][j] > lv;
        T.cross[j] = !valid ? 0u : (((hc && ic != i0) ? 1u : 0u) | ((hb && ib != i0) ? 2u : 0u) | ((ha && ia != i0) ? 4u : 0u));
        // corner v at (da,db,dc) = (v>>2&1, v>>1&1, v&1): 0 self, 1 +c, 2 +b, 3 +b+c, 4 +a, 5 +a+c, 6 +a+b, 7 +a+b+c
        T.ccase[j] = (hc && hb && ha)
                         ? ((i0 ? 1 : 0) | (ic ? 2 : 0) | (ib ? 4 : 0) | (T.f[1][j + 1] > lv ? 8 : 0) | (ia ? 16 : 0) |
                            (T.f[2][j + 1] > lv ? 32 : 0) | (T.f[3][j] > lv ? 64 : 0) | (T.f[3][j + 1] > lv ? 128 : 0))
                         : -1;
    }
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

template <bool VEC>
__global__ __launch_bounds__(MC_T) void k_mc_classify(McParams p) {
    __shared__ unsigned lds[4];
    __shared__ unsigned char l_ntri[256];
    const int row = blockIdx.x, a = row / p.n, b = row - a * p.n;
    const int c = MC_PT * threadIdx.x;
    unsigned acc = 0;
    McTile T;
    if (c < p.n) mc_tile<VEC>(p, a, b, c, T);  // volume loads in flight while the table is staged
    mc_stage_ntri(l_ntri);
    __syncthreads();
    if (c < p.n) {
#pragma unroll
        for (int j = 0; j < MC_PT; ++j)
            acc += (unsigned)__popc(T.cross[j]) | ((T.ccase[j] >= 0 ? (unsigned)l_ntri[T.ccase[j]] : 0u) << 16);
    }
    unsigned total;
    mc_wg_scan(acc, &total, lds);
    if (threadIdx.x == 0) p.row_sum[row] = total;
}

// block = plane a; n <= 1024 rows, 4 per thread
__global__ __launch_bounds__(MC_T) void k_mc_scan_rows(McParams p) {
    __shared__ unsigned lds[MC_T / 64 + 1];
    const int a = blockIdx.x;
    const unsigned* rs = p.row_sum + (size_t)a * p.n;
    unsigned v[4], t[4], sv = 0, st = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = 4 * threadIdx.x + k;
        const unsigned x = i < p.n ? rs[i] : 0u;
        v[k] = x & 0xffffu; t[k] = x >> 16;
        sv += v[k]; st += t[k];
    }
    unsigned totv, tott;
    unsigned ev = mc_wg_scan(sv, &totv, lds);
    unsigned et = mc_wg_scan(st, &tott, lds);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = 4 * threadIdx.x + k;
        if (i < p.n) p.row_off[(size_t)a * p.n + i] = make_uint2(ev, et);
        ev += v[k]; et += t[k];
    }
    if (threadIdx.x == 0) { p.plane[a] = totv; p.plane[p.n + a] = tott; }
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

template <bool VEC>
__global__ __launch_bounds__(MC_T) void k_mc_verts(McParams p) {
    __shared__ unsigned lds[4];
    const int row = blockIdx.x, a = row / p.n, b = row - a * p.n;
    const int c = MC_PT * threadIdx.x;
    McTile T;
    unsigned mine = 0;
    if (c < p.n) {
        mc_tile<VEC>(p, a, b, c, T);
#pragma unroll
        for (int j = 0; j < MC_PT; ++j) mine += (unsigned)__popc(T.cross[j]);
    }
    unsigned total;
    const unsigned excl = mc_wg_scan(mine, &total, lds);
    if (c >= p.n) return;
    unsigned long long vid = p.plane[a] + p.row_off[row].x + excl;
    unsigned info[MC_PT];
    {
        unsigned run = (unsigned)vid;
#pragma unroll
        for (int j = 0; j < MC_PT; ++j) { info[j] = (run << 3) | T.cross[j]; run += (unsigned)__popc(T.cross[j]); }
    }
    unsigned* vi = p.vert_info + (size_t)row * p.n + c;
    if (VEC) *reinterpret_cast<uint4*>(vi) = make_uint4(info[0], info[1], info[2], info[3]);
    else {
#pragma unroll
        for (int j = 0; j < MC_PT; ++j) if (c + j < p.n) vi[j] = info[j];
    }
    if (!mine) return;
#pragma unroll
    for (int j = 0; j < MC_PT; ++j) {
        const unsigned cross = T.cross[j];
        if (!cross) continue;
        const int cj = c + j;
        const float f0 = T.f[0][j];
        float g0[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) g0[ax] = mc_grad1(p, a, b, cj, ax);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (!(cross & (1u << s))) continue;
            const int a1 = a + (s == 2), b1 = b + (s == 1), c1 = cj + (s == 0);
            const float f1 = s == 0 ? T.f[0][j + 1] : (s == 1 ? T.f[1][j] : T.f[2][j]);
            const float t = (p.level - f0) / (f1 - f0);
            float pos[3] = {(float)a, (float)b, (float)cj};  // (a, b, c) = skimage's (axis 0, 1, 2) vertex order
            pos[2 - s] += t;
            float g[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const float g1 = mc_grad1(p, a1, b1, c1, ax);
                g[ax] = __builtin_fmaf(t, g1 - g0[ax], g0[ax]);
            }
            // |g| through binary64: sqrt in double then rounding to float IS the correctly rounded float sqrt (53 >= 2*24+2);
            // the f32 hardware sqrt is not.  normal = -g/|g| in (a,b,c) order: g[2] is d/da.
            const float len2 = __builtin_fmaf(g[0], g[0], __builtin_fmaf(g[1], g[1], g[2] * g[2]));
            const float len = (float)__dsqrt_rn((double)len2);
            const bool ok = len > 0.0f;
            float* V = p.verts + 3 * vid;
            float* N = p.normals + 3 * vid;
            V[0] = pos[0]; V[1] = pos[1]; V[2] = pos[2];
            N[0] = ok ? -g[2] / len : 0.0f;
            N[1] = ok ? -g[1] / len : 0.0f;
            N[2] = ok ? -g[0] / len : 0.0f;
            p.values[vid] = f0 > f1 ? f0 : f1;
            ++vid;
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(MC_T) void k_mc_tris(McParams p) {
    __shared__ unsigned lds[4];
    __shared__ unsigned char l_ntri[256];
    __shared__ signed char l_tri[256 * P3D_MC_ROW];
    const int row = blockIdx.x, a = row / p.n, b = row - a * p.n;
    if (a + 1 >= p.n || b + 1 >= p.n) return;  // uniform: no cubes start on the last plane / row
    const int c = MC_PT * threadIdx.x;
    McTile T;
    unsigned nt[MC_PT] = {0, 0, 0, 0}, mine = 0;
    if (c < p.n) mc_tile<VEC>(p, a, b, c, T);
    mc_stage_ntri(l_ntri);
    __syncthreads();
    if (c < p.n) {
#pragma unroll
        for (int j = 0; j < MC_PT; ++j) { nt[j] = T.ccase[j] >= 0 ? (unsigned)l_ntri[T.ccase[j]] : 0u; mine += nt[j]; }
    }
    unsigned total;
    const unsigned excl = mc_wg_scan(mine, &total, lds);
    if (!total) return;  // uniform: most rows of a real volume hold no surface -> the 4 KB triangle table is not staged
    for (int i = threadIdx.x; i < 256 * P3D_MC_ROW / 4; i += blockDim.x)
        reinterpret_cast<int*>(l_tri)[i] = reinterpret_cast<const int*>(&P3D_MC_TRI[0][0])[i];
    __syncthreads();
    if (!mine) return;
    int* F = p.faces + 3 * (p.plane[p.n + a] + p.row_off[row].y + excl);
    const size_t n = p.n;
#pragma unroll
    for (int j = 0; j < MC_PT; ++j) {
        if (!nt[j]) continue;
        const size_t pid = (size_t)row * n + c + j;
        const signed char* tri = l_tri + T.ccase[j] * P3D_MC_ROW;
        for (unsigned t = 0; t < nt[j]; ++t) {
            int id[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int e = tri[3 * t + k];
                // lower corner of edge e: P3D_MC_EDGE[e][0] = 0,2,4,6, 0,1,4,5, 0,1,2,3 packed 3 bits each; slot = e >> 2
                const int v0 = (int)((0x688b08d10ull >> (3 * e)) & 7), slot = e >> 2;
                const size_t owner = pid + ((v0 >> 2) & 1) * n * n + ((v0 >> 1) & 1) * n + (v0 & 1);
                const unsigned info = p.vert_info[owner];
                id[k] = (int)((info >> 3) + __popc(info & ((1u << slot) - 1u)));
            }
            F[0] = id[0]; F[1] = id[1]; F[2] = id[2];
            F += 3;
        }
    }
}

extern "C" {

size_t p3d_mc_workspace_bytes(int n) {
    if (n < 2 || n > MC_MAXN) return 0;
    const size_t nn = (size_t)n * n;
    return 256 + 16 * (size_t)MC_MAXN + 12 * nn + 4 * nn * n;
}

static int mc_params(McParams& p, const float* vol, int n, int flip0, float level, void* workspace, size_t workspace_bytes) {
    if (!vol || !workspace) return P3D_E_ARG;
    if (n < 2 || n > MC_MAXN) return P3D_E_RANGE;
    if (workspace_bytes < p3d_mc_workspace_bytes(n) || ((uintptr_t)workspace & 15)) return P3D_E_WORKSPACE;
    const size_t nn = (size_t)n * n;
    p.vol = vol; p.n = n; p.flip0 = flip0 ? 1 : 0; p.level = level;
    char* w = (char*)workspace;
    p.totals = (unsigned long long*)w;            w += 256;
    p.plane = (unsigned long long*)w;             w += 16 * (size_t)MC_MAXN;
    p.row_off = (uint2*)w;                        w += 8 * nn;
    p.row_sum = (unsigned*)w;                     w += 4 * nn;
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
    if (n % 4 == 0 && ((uintptr_t)vol & 15) == 0) hipLaunchKernelGGL(k_mc_classify<true>, dim3((unsigned)(n * n)), dim3(threads), 0, s, p);
    else hipLaunchKernelGGL(k_mc_classify<false>, dim3((unsigned)(n * n)), dim3(threads), 0, s, p);
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
    if (n % 4 == 0 && ((uintptr_t)vol & 15) == 0) {
        hipLaunchKernelGGL(k_mc_verts<true>, dim3((unsigned)(n * n)), dim3(threads), 0, s, p);
        hipLaunchKernelGGL(k_mc_tris<true>, dim3((unsigned)(n * n)), dim3(threads), 0, s, p);
    } else {
        hipLaunchKernelGGL(k_mc_verts<false>, dim3((unsigned)(n * n)), dim3(threads), 0, s, p);
        hipLaunchKernelGGL(k_mc_tris<false>, dim3((unsigned)(n * n)), dim3(threads), 0, s, p);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? P3D_OK : (int)e;
}

}  // extern "C"
