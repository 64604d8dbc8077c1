/* p3d_oracle_mc.c — CPU ORACLE of the iso-surface extractor.  TEST INFRASTRUCTURE ONLY (same rule as
 * p3d_oracle.c: only tests/, smoke() and bench.py's cpu_baseline leg may load it).
 *
 * Parity status: PARITY UNPINNED against the reference.  The reference calls skimage.measure.marching_cubes(vol, level,
 * spacing=(1,1,1), gradient_direction='descent', step_size=1, allow_degenerate=False, method='lewiner')
 * (_util/eg3d_metrics3d.py:186-210).  scikit-image (unpinned in _env/Dockerfile) is neither installed here nor vendored
 * under /root/reference, and its Lewiner case tables cannot be restated from memory, so this is a SPECIFICATION of our own
 * extractor, pinned by properties instead (tests/test_mcubes_cpu.py): closed 2-manifold on padded volumes, consistent outward
 * orientation, vertices on the level set of the trilinear edge interpolant, Euler characteristic 2 and area/volume
 * convergence on a sphere.  What matches skimage by construction: vertex coordinates in index space (axis 0, 1, 2 order,
 * spacing 1), linear interpolation along grid edges, one shared vertex per crossed edge, int32 faces, outward normals for
 * 'descent', values = local maximum of the data.  What differs: triangulation inside ambiguous cubes (Lewiner resolves
 * them by trilinear tests; here: include/p3d_mc_table.h, inside corners always isolated), vertex/face ORDER, and degenerate
 * triangles (a grid value exactly equal to `level`) are kept.
 *
 * Contract (binary32, the HIP kernels in csrc/p3d_mcubes.hip follow it op for op):
 *   V(a,b,c)   = vol[((flip0 ? n-1-a : a)*n + b)*n + c]      (flip0: read the un-flipped grid of eg3d_metrics3d.py:166-168)
 *   inside(x)  = x > level
 *   vertices   are owned by the grid point at the lower end of their edge; order: grid point (a,b,c) lexicographic, then
 *                edge along c, along b, along a.  t = (level - f0) / (f1 - f0); position = index + t on the edge's axis;
 *   gradient   per axis: 0.5*(V(i+1) - V(i-1)) inside, V(i+1) - V(i) / V(i) - V(i-1) at the borders;
 *                g = fmaf(t, g1 - g0, g0); len = (float)sqrt((double)fmaf(gc,gc, fmaf(gb,gb, ga*ga))) — the correctly
 *                rounded binary32 square root; normal = -g/len (0 if len == 0)
 *   value      = max(f0, f1)
 *   faces      cube (a,b,c) lexicographic, then table order; vertex id = first vertex of the owner + rank of the edge slot.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include "../include/p3d_mc_table.h"

typedef struct { const float* vol; int n; int flip0; float level; } mc_vol;

static inline float mcv(const mc_vol* m, int a, int b, int c) {
    int aa = m->flip0 ? m->n - 1 - a : a;
    return m->vol[((int64_t)aa * m->n + b) * m->n + c];
}

static inline unsigned mc_cross(const mc_vol* m, int a, int b, int c) {
    const int n = m->n;
    const int i0 = mcv(m, a, b, c) > m->level;
    unsigned x = 0;
    if (c + 1 < n && (mcv(m, a, b, c + 1) > m->level) != i0) x |= 1u;
    if (b + 1 < n && (mcv(m, a, b + 1, c) > m->level) != i0) x |= 2u;
    if (a + 1 < n && (mcv(m, a + 1, b, c) > m->level) != i0) x |= 4u;
    return x;
}

static inline int mc_case(const mc_vol* m, int a, int b, int c) {
    int cs = 0;
    for (int v = 0; v < 8; ++v)
        if (mcv(m, a + ((v >> 2) & 1), b + ((v >> 1) & 1), c + (v & 1)) > m->level) cs |= 1 << v;
    return cs;
}

static inline float mc_grad(const mc_vol* m, int a, int b, int c, int axis) {
    int i = axis == 0 ? c : (axis == 1 ? b : a);
    int lo = i > 0 ? i - 1 : i, hi = i + 1 < m->n ? i + 1 : i;
    float flo, fhi;
    if (axis == 0) { flo = mcv(m, a, b, lo); fhi = mcv(m, a, b, hi); }
    else if (axis == 1) { flo = mcv(m, a, lo, c); fhi = mcv(m, a, hi, c); }
    else { flo = mcv(m, lo, b, c); fhi = mcv(m, hi, b, c); }
    float d = fhi - flo;
    return (hi - lo == 2) ? 0.5f * d : d;
}

void or_mc_count(const float* vol, int n, int flip0, float level, int64_t* counts) {
    mc_vol m = {vol, n, flip0, level};
    int64_t nv = 0, nt = 0;
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b)
            for (int c = 0; c < n; ++c) {
                nv += __builtin_popcount(mc_cross(&m, a, b, c));
                if (a + 1 < n && b + 1 < n && c + 1 < n) nt += P3D_MC_NTRI[mc_case(&m, a, b, c)];
            }
    counts[0] = nv;
    counts[1] = nt;
}

/* verts/normals [nv][3], values [nv], faces [nt][3]; returns 0, or -1 when the scratch allocation fails */
int or_mc_emit(const float* vol, int n, int flip0, float level, float* verts, float* normals, float* values, int32_t* faces) {
    mc_vol m = {vol, n, flip0, level};
    const int64_t npts = (int64_t)n * n * n;
    uint32_t* info = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)npts);
    if (!info) return -1;
    int64_t vid = 0;
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b)
            for (int c = 0; c < n; ++c) {
                const unsigned x = mc_cross(&m, a, b, c);
                info[((int64_t)a * n + b) * n + c] = ((uint32_t)vid << 3) | x;
                if (!x) continue;
                const float f0 = mcv(&m, a, b, c);
                float g0[3];
                for (int ax = 0; ax < 3; ++ax) g0[ax] = mc_grad(&m, a, b, c, ax);
                for (int s = 0; s < 3; ++s) {
                    if (!(x & (1u << s))) continue;
                    const int a1 = a + (s == 2), b1 = b + (s == 1), c1 = c + (s == 0);
                    const float f1 = mcv(&m, a1, b1, c1);
                    const float t = (level - f0) / (f1 - f0);
                    float pos[3] = {(float)a, (float)b, (float)c};
                    pos[2 - s] += t;
                    float g[3];
                    for (int ax = 0; ax < 3; ++ax) {
                        const float g1 = mc_grad(&m, a1, b1, c1, ax);
                        g[ax] = fmaf(t, g1 - g0[ax], g0[ax]);
                    }
                    const float len = (float)sqrt((double)fmaf(g[0], g[0], fmaf(g[1], g[1], g[2] * g[2])));
                    for (int k = 0; k < 3; ++k) {
                        verts[3 * vid + k] = pos[k];
                        normals[3 * vid + k] = len > 0.0f ? -g[2 - k] / len : 0.0f;
                    }
                    values[vid] = f0 > f1 ? f0 : f1;
                    ++vid;
                }
            }
    int64_t fid = 0;
    for (int a = 0; a + 1 < n; ++a)
        for (int b = 0; b + 1 < n; ++b)
            for (int c = 0; c + 1 < n; ++c) {
                const int cs = mc_case(&m, a, b, c);
                for (int k = 0; k < 3 * P3D_MC_NTRI[cs]; ++k) {
                    const int e = P3D_MC_TRI[cs][k];
                    const int v0 = P3D_MC_EDGE[e][0], slot = e >> 2;
                    const uint32_t w = info[((int64_t)(a + ((v0 >> 2) & 1)) * n + (b + ((v0 >> 1) & 1))) * n + (c + (v0 & 1))];
                    faces[3 * fid + k] = (int32_t)((w >> 3) + __builtin_popcount(w & ((1u << slot) - 1u)));
                }
                fid += P3D_MC_NTRI[cs];
            }
    free(info);
    return 0;
}
