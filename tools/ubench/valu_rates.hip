// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_ldexp_f32 on gfx950 (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
    float x[16];
    f2 y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = (f2){x[i], x[i] + 1}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], (f2){a, a}, (f2){b, b});
            if (MODE == 2) x[i] = __builtin_ldexpf(x[i], 1) * a;
            if (MODE == 3) x[i] = x[i] / (a + x[i]);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int flops_per) {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256 * 8, 256>>>(d, 100, 0.999f, 0.001f);
    hipEventRecord(e0);
    k<MODE><<<256 * 8, 256>>>(d, iters, 0.999f, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = 256.0 * 8 * 256 * iters * 16;  // lane-instructions
    printf("%s: %.3f ms, %.2f T lane-instr/s, %.2f TFLOP/s-equivalent\n", name, ms, ops / ms / 1e9, ops * flops_per / ms / 1e9);
    hipFree(d);
}
int main() {
    run<0>("v_fma_f32", 2);
    run<1>("v_pk_fma_f32 (2 per lane)", 4);
    run<2>("v_ldexp_f32 + v_mul", 1);
    run<3>("f32 IEEE div + add", 1);
    return 0;
}
