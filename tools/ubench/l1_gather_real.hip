// Microbenchmark: the vector L1's service time for k_render's REAL gather pattern (one wave-step of the 512^2 x (48+48) bench):
// 32 rays of an 8x4 pixel tile (Morton lane order), neighbouring rays 0.38 texel apart; plane A is screen-aligned (both
// coordinates follow the pixel), planes B / C carry the depth: one coordinate follows the pixel, the other is jittered over a
// depth bin of 7.6 texels.  Per sample and plane: 4 bilinear taps x 64 B per lane (4 x dwordx4).  Planes of 256x256 texels x
// 128 B (8 MB each, L2-resident), a fresh tile position every step.  Layout variants of the lane -> (ray, channel half, piece):
//   0  shipped: ray = l & 31, half = l >> 5, load k fetches piece k
//   1  piece rotated by the quad index
//   2  ray = l >> 1, half = l & 1 (a line's two halves in neighbouring lanes)
//   3  as 2, piece rotated by ray
//   4  shipped layout but NO jitter on the depth coordinate (what a deterministic-depth render would see)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(const float* tab, float* out, int iters) {
    const unsigned lane = threadIdx.x & 63, gid = blockIdx.x * blockDim.x + threadIdx.x, wid = gid >> 6;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, 3u * 256 * 256 * 128, 0x00020000);
    const unsigned j = (MODE == 2 || MODE == 3) ? (lane >> 1) : (lane & 31), h = (MODE == 2 || MODE == 3) ? (lane & 1) : (lane >> 5);
    const int lx = (j & 1) | ((j >> 1) & 6), ly = ((j >> 1) & 1) | ((j >> 3) & 2);  // 8x4 pixel tile, Morton order
    const unsigned rot = MODE == 1 ? (j >> 2) : MODE == 3 ? j : 0u;
    unsigned s = wid * 2654435761u + 12345u, sl = gid * 747796405u + 2891336453u;
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;   // wave-uniform: tile position
        sl = sl * 1664525u + 1013904223u; // per ray: depth jitter (same for both halves of a ray: derive from j)
        const unsigned sj = (wid * 64 + j) * 2246822519u + it * 3266489917u;
        const float jit = MODE == 4 ? 0.5f : (float)((sj * 1664525u + 1013904223u) >> 8) * (1.0f / 16777216.0f);
        const float bx = 8.0f + (float)((s >> 8) & 0xff) * 0.9f, by = 8.0f + (float)((s >> 16) & 0xff) * 0.9f;
        const float u = bx + 0.38f * (float)lx, v = by + 0.38f * (float)ly, w = by * 0.7f + 20.0f + 7.6f * jit;
        i32x4 vreg[12];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const float cx = p == 2 ? v : u, cy = p == 0 ? v : w;
            const int x0 = (int)cx, y0 = (int)cy;
            float a = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned off = (unsigned)p * (256u * 256u * 128u) + (unsigned)(((y0 + (t >> 1)) & 255) * 256 + ((x0 + (t & 1)) & 255)) * 128u + h * 64u;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    vreg[(t & 2 ? 8 : 0) / 2 * 0 + kk + 4 * (t & 1)] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (((unsigned)kk + rot) & 3u) * 16u, 0, 0);
                }
                if (t & 1) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) a += __builtin_bit_cast(float, vreg[q].x) + __builtin_bit_cast(float, vreg[q].w);
                }
            }
            acc += a;
        }
    }
    out[gid] = acc;
}
template <int MODE>
void run(const char* name, const float* tab) {
    float* d;
    const int nb = 256 * 4, iters = 150;  // 4 blocks x 8 waves per CU resident: 8 waves per CU x 4 ... (512 threads: 2 blocks/CU by regs)
    (void)hipMalloc(&d, (size_t)nb * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<nb, 512>>>(tab, d, 4);
    (void)hipEventRecord(e0);
    k<MODE><<<nb, 512>>>(tab, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)nb * 8 * iters * 48 / 256.0;  // wave gather instructions per CU
    printf("mode %d  %-52s %.3f ms  %.1f clk per gather instruction per CU (at 2.4 GHz)\n", MODE, name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
    (void)hipFree(d);
}
int main() {
    float* tab; (void)hipMalloc(&tab, (size_t)3 * 256 * 256 * 128); (void)hipMemset(tab, 0, (size_t)3 * 256 * 256 * 128);
    run<0>("shipped layout", tab);
    run<1>("piece rotated by quad", tab);
    run<2>("(ray, half) = (l >> 1, l & 1)", tab);
    run<3>("(l >> 1, l & 1) + piece rotated by ray", tab);
    run<4>("shipped layout, no depth jitter", tab);
    return 0;
}
