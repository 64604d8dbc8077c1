// How does v_mfma_f32_32x32x16_bf16 accumulate?  Compare the hardware result of D = C + sum_{k<16} a_k*b_k (one output
// element per test, random bf16 inputs with varied exponents) against candidate CPU models.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// each wave computes a 32x32 tile; we use A row i = test's a-vector, B col j = b-vector; read back diagonal-ish elements
__global__ void k(const uint16_t* A, const uint16_t* B, const float* Cin, float* D) {
    // A [32][16], B [16][32] (as B^T [32][16]), Cin [32][32] row-major
    int lane = threadIdx.x;
    int i = lane & 31, kh = lane >> 5;
    s16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[i * 16 + kh * 8 + e]; b[e] = B[i * 16 + kh * 8 + e]; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * kh; c[r] = Cin[row * 32 + i]; }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * kh; D[row * 32 + i] = c[r]; }
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1, 1);
    std::uniform_int_distribution<int> E(-6, 6);
    int trials = 200, match_exact = 0, match_seq = 0, match_seqrev = 0, match_pair = 0, match_dbl_seq = 0, total = 0;
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, 32 * 16 * 2); hipMalloc(&dB, 32 * 16 * 2); hipMalloc(&dC, 32 * 32 * 4); hipMalloc(&dD, 32 * 32 * 4);
    for (int t = 0; t < trials; ++t) {
        std::vector<uint16_t> A(32 * 16), B(32 * 16);
        std::vector<float> C(32 * 32), D(32 * 32);
        auto rb = [&]() { float f = std::ldexp(U(rng), E(rng)); uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
        for (auto& x : A) x = rb();
        for (auto& x : B) x = rb();
        for (auto& x : C) x = std::ldexp(U(rng), E(rng));
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            float c = C[i * 32 + j];
            long double ex = c; double ds = c; float s = c, sr = c;
            float p[16];
            for (int kk = 0; kk < 16; ++kk) p[kk] = bf2f(A[i * 16 + kk]) * bf2f(B[j * 16 + kk]);  // exact in f32
            for (int kk = 0; kk < 16; ++kk) { ex += (long double)p[kk]; ds += (double)p[kk]; s = s + p[kk]; }
            for (int kk = 15; kk >= 0; --kk) sr = sr + p[kk];
            float pr[16]; for (int kk = 0; kk < 16; ++kk) pr[kk] = p[kk];
            for (int w = 8; w >= 1; w >>= 1) for (int kk = 0; kk < w; ++kk) pr[kk] = pr[kk] + pr[kk + w];
            float pair = c + pr[0];
            float hw = D[i * 32 + j];
            total++;
            match_exact += (hw == (float)ex); match_seq += (hw == s); match_seqrev += (hw == sr); match_pair += (hw == pair);
            match_dbl_seq += (hw == (float)ds);
        }
    }
    printf("total %d  exact-sum-round-once %d  double-seq-round-once %d  f32-seq %d  f32-seq-reversed %d  f32-tree+c %d\n", total,
           match_exact, match_dbl_seq, match_seq, match_seqrev, match_pair);
    return 0;
}
