// Microbenchmark: L1 (TCP) cost of divergent 16-B gathers vs quad-cooperative 64-B gathers on gfx950.
// A 24 MB table; every block gathers inside its own 16 KB window (L1-resident, like a ray tile's texels); 16 B per lane.
//   MODE 0: every lane reads its own random 64-B-aligned chunk, 16 B at a time (4 instructions cover its 64 B)  [today's gather]
//   MODE 1: the 4 lanes of a quad read the 4 consecutive 16-B pieces of ONE random chunk per instruction      [quad-cooperative]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* tab, unsigned nchunks, float* out, int iters) {
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, nchunks * 64u, 0x00020000);
    unsigned lane = threadIdx.x & 63, gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned s = gid * 2654435761u + 12345u;
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned ch[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { s = s * 1664525u + 1013904223u; ch[q] = (blockIdx.x * 997u % (nchunks - 256u)) + ((s >> 8) & 255u); }  // 16 KB window per block: L1 hits
        i32x4 v[16];
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int p = 0; p < 4; ++p) v[q * 4 + p] = __builtin_amdgcn_raw_buffer_load_b128(rs, ch[q] * 64u + p * 16u, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
            {
#define BC(O) { unsigned c = __builtin_amdgcn_mov_dpp(ch[q], (O) * 0x55, 0xf, 0xf, false); /* quad_perm broadcast of lane O */ \
                v[q * 4 + (O)] = __builtin_amdgcn_raw_buffer_load_b128(rs, c * 64u + (lane & 3) * 16u, 0, 0); }
                BC(0) BC(1) BC(2) BC(3)
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += __builtin_bit_cast(float, v[q].x) + __builtin_bit_cast(float, v[q].w);
    }
    out[gid] = acc;
}
template <int MODE>
void run(const char* name, const float* tab, unsigned nchunks) {
    float* d; (void)hipMalloc(&d, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<2048, 256>>>(tab, nchunks, d, 2);
    hipEventRecord(e0);
    k<MODE><<<2048, 256>>>(tab, nchunks, d, 200);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = 2048.0 * 256 * 200 * 16 * 16;
    printf("%-28s %.3f ms  %.2f TB/s of 16-B lane requests\n", name, ms, bytes / ms / 1e9);
    (void)hipFree(d);
}
int main() {
    unsigned nchunks = 24u * 1024 * 1024 / 64;
    float* tab; (void)hipMalloc(&tab, (size_t)nchunks * 64); (void)hipMemset(tab, 0, (size_t)nchunks * 64);
    run<0>("lane-private 16-B gathers", tab, nchunks);
    run<1>("quad-cooperative 64-B", tab, nchunks);
    return 0;
}
