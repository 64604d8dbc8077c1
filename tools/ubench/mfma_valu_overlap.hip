// Microbenchmark: do v_mfma_f32_32x32x2_f32 and VALU overlap (a) across the two waves of a SIMD, (b) inside one wave?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MODE 0: MFMA only; 1: VALU only; 2: both in every wave (independent streams); 3: waves 0-3 MFMA, waves 4-7 VALU
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
    const int wave = threadIdx.x >> 6;
    f32x16 c0 = {0}, c1 = {0};
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    if (do_mfma && do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name) {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 512>>>(d, 10, 0.999f, 0.001f);
    hipEventRecord(e0);
    k<MODE><<<256, 512>>>(d, iters, 0.999f, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms  (per iteration: 8 MFMA and/or 192 VALU per wave; 2 waves/SIMD)\n", name, ms);
    hipFree(d);
}
int main() {
    run<0>("MFMA only (all 8 waves)");
    run<1>("VALU only (all 8 waves)");
    run<2>("MFMA + VALU interleaved inside every wave");
    run<3>("waves 0-3 MFMA only, waves 4-7 VALU only");
    return 0;
}
