// Microbenchmark: what does the vector L1 (TCP) of a gfx950 CU sustain for 16-B-per-lane gathers that HIT in L1?
// One 1024-thread block per CU (16 waves, 4 per SIMD), all waves of a block gather inside ONE window of `win` bytes
// (8 KB: L1-resident for sure; 64 KB: L2-served), 16 loads in flight per lane.  Cycles by s_memtime (shader clock).
//   mode 0  every lane its own random 16-B chunk                                    (64 distinct 16-B pieces / instr)
//   mode 1  every lane its own random 64-B chunk, 4 instrs cover it                 (today's gather, random texels)
//   mode 2  quad-cooperative: 4 lanes of a quad read the 4 pieces of one 64-B chunk (16 chunks / instr)
//   mode 3  fully coalesced: lane i reads base + 16 i, random 1-KB base per instr
//   mode 4  k_render's lane layout: lane (j, h) reads texel[j] * 128 + 64 h + 16 p, texel[j] random per lane pair
//   mode 5  as 4, but the 4 lanes of a quad share one texel (adjacent rays on one texel)
//   mode 6  as 4, but all 32 samples on 4 texels
//   mode 7  every lane random 4-B dword (buffer_load_dword)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(const float* tab, unsigned win, float* out, long long* cyc, int iters) {
    const unsigned lane = threadIdx.x & 63, gid = blockIdx.x * blockDim.x + threadIdx.x;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(tab + (size_t)blockIdx.x * (win / 4)), 0, win, 0x00020000);
    unsigned s = gid * 2654435761u + 12345u;
    float acc = 0;
    const unsigned n16 = win / 16, n64 = win / 64, n128 = win / 128, n1k = win / 1024;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        i32x4 v[16];
        unsigned r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { s = s * 1664525u + 1013904223u; r[q] = s >> 8; }
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { unsigned x = (r[q & 3] * (2 * q + 1)) >> 4; v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (x % n16) * 16u, 0, 0); }
        } else if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int p = 0; p < 4; ++p) v[q * 4 + p] = __builtin_amdgcn_raw_buffer_load_b128(rs, (r[q] % n64) * 64u + p * 16u, 0, 0);
        } else if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#define BC(O) { unsigned c = __builtin_amdgcn_mov_dpp(r[q], (O) * 0x55, 0xf, 0xf, false); \
                v[q * 4 + (O)] = __builtin_amdgcn_raw_buffer_load_b128(rs, (c % n64) * 64u + (lane & 3) * 16u, 0, 0); }
                BC(0) BC(1) BC(2) BC(3)
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                unsigned b = __builtin_amdgcn_readfirstlane(r[q & 3] * (2 * q + 1) >> 4);
                v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (b % n1k) * 1024u + lane * 16u, 0, 0);
            }
        } else if (MODE == 4 || MODE == 5 || MODE == 6) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned src = MODE == 4 ? (lane & 31) : MODE == 5 ? (lane & 28) : (lane & 24);
                unsigned t = __shfl(r[q], src);
                unsigned base = (t % n128) * 128u + (lane >> 5) * 64u;
#pragma unroll
                for (int p = 0; p < 4; ++p) v[q * 4 + p] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + p * 16u, 0, 0);
            }
        } else if (MODE >= 8 && MODE <= 15) {
            // k_render-like layouts with variations.  ntex: distinct texels among the 32 samples (32, or 8 = a quad shares one)
            //   8: (j,h) = (l&31, l>>5), piece rotated by j          9: same, piece rotated by quad index (j>>2)
            //  10: as 9 with a quad sharing a texel                  11: (j,h) = (l>>1, l&1), no rotation
            //  12: as 11, piece rotated by j                         13: as 8 with a quad sharing a texel
            //  14: as 4 but half h reads the OTHER line (texel*256 + 128 h): are the two halves of a line the problem?
            //  15: (j,h) = (l&31, l>>5), piece rotated by j + 2h
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned j = (MODE == 11 || MODE == 12) ? (lane >> 1) : (lane & 31), hh = (MODE == 11 || MODE == 12) ? (lane & 1) : (lane >> 5);
                const unsigned srcj = (MODE == 10 || MODE == 13) ? (j & 28) : j;
                const unsigned srcl = (MODE == 11 || MODE == 12) ? srcj * 2 : srcj;
                unsigned t = __shfl(r[q], srcl);
                unsigned base = MODE == 14 ? (t % (win / 256)) * 256u + hh * 128u : (t % n128) * 128u + hh * 64u;
                const unsigned rot = (MODE == 8 || MODE == 12 || MODE == 13) ? j : (MODE == 9 || MODE == 10) ? (j >> 2) : MODE == 15 ? j + 2 * hh : 0u;
#pragma unroll
                for (int p = 0; p < 4; ++p) v[q * 4 + p] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + ((p + rot) & 3u) * 16u, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                unsigned x = (r[q & 3] * (2 * q + 1)) >> 4;
                int w = __builtin_amdgcn_raw_buffer_load_b32(rs, (x % (win / 4)) * 4u, 0, 0);
                v[q] = (i32x4){w, w, w, w};
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += __builtin_bit_cast(float, v[q].x) + __builtin_bit_cast(float, v[q].w);
    }
    long long t1 = __builtin_readcyclecounter();
    out[gid] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
static int g_threads = 1024;  // threads per block = per CU (argv[1]): 16 waves by default, 8 = k_render's occupancy
template <int MODE>
void run(const char* name, const float* tab, unsigned win) {
    float* d; long long* c;
    const int nb = 256, iters = 400;
    (void)hipMalloc(&d, (size_t)nb * 1024 * 4); (void)hipMalloc(&c, nb * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<nb, g_threads>>>(tab, win, d, c, 4);
    hipEventRecord(e0);
    k<MODE><<<nb, g_threads>>>(tab, win, d, c, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long hc[256]; (void)hipMemcpy(hc, c, nb * 8, hipMemcpyDeviceToHost);
    double mx = 0; for (int i = 0; i < nb; ++i) mx = hc[i] > mx ? hc[i] : mx;
    const double reqs_per_cu = (double)g_threads * iters * 16;  // lane requests issued by one CU's block
    const double bytes = (MODE == 7 ? 4.0 : 16.0);
    printf("mode %d win %6u  %-44s %.3f ms  %8.0f kcyc  %.2f lane-req/clk/CU  %.1f B/clk/CU  (%.2f TB/s chip at wall time)\n", MODE, win, name, ms,
           mx / 1e3, reqs_per_cu / mx, reqs_per_cu * bytes / mx, 256.0 * reqs_per_cu * bytes / ms / 1e9);
    (void)hipFree(d); (void)hipFree(c);
}
int main(int argc, char** argv) {
    if (argc > 1) g_threads = atoi(argv[1]);
    printf("threads per CU: %d\n", g_threads);
    const unsigned wins[2] = {8192, 131072};
    float* tab; (void)hipMalloc(&tab, (size_t)256 * 131072); (void)hipMemset(tab, 0, (size_t)256 * 131072);
    for (unsigned w : wins) {
        run<0>("lane-random 16-B chunks", tab, w);
        run<1>("lane-private 64-B chunk x4 instr", tab, w);
        run<2>("quad-cooperative 64-B", tab, w);
        run<3>("fully coalesced 1 KB / instr", tab, w);
        run<4>("k_render layout, 32 random texels", tab, w);
        run<5>("k_render layout, quad shares a texel (8)", tab, w);
        run<6>("k_render layout, 4 texels per instr", tab, w);
        run<7>("lane-random dword", tab, w);
        run<8>("render layout, piece rot by j", tab, w);
        run<9>("render layout, piece rot by quad", tab, w);
        run<10>("quad shares texel, rot by quad", tab, w);
        run<11>("(j,h)=(l>>1,l&1), no rot", tab, w);
        run<12>("(j,h)=(l>>1,l&1), rot by j", tab, w);
        run<13>("quad shares texel, rot by j", tab, w);
        run<14>("halves on different lines", tab, w);
        run<15>("render layout, rot by j+2h", tab, w);
    }
    return 0;
}
