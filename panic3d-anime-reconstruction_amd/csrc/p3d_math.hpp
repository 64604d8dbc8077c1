// p3d_math.hpp — device-side scalar math of the arithmetic contract (include/p3d_numerics.h) for gfx950.
// Every fused multiply-add is an explicit __builtin_fmaf; the translation unit is compiled with
// -ffp-contract=off and without fast-math, so the compiler neither fuses nor reassociates anything.
// f32 division is IEEE (hipcc default: correctly rounded divide), subnormals are kept (gfx9 default).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p3d_numerics.h"

#define P3D_DEV __device__ __forceinline__

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

P3D_DEV float p3d_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// exp(x): range reduction by ln2 (two-part constant) + degree-6 polynomial, scaled by v_ldexp_f32 (exact 2^n scaling,
// IEEE rounding into subnormals, saturating integer conversion) — no clamps needed for x <= 0: the result underflows to 0.
P3D_DEV float p3d_exp_core(float x) {
    float n = __builtin_rintf(x * P3D_LOG2E);
    float r = p3d_fma(n, -P3D_LN2_HI, x);
    r = p3d_fma(n, -P3D_LN2_LO, r);
    float p = P3D_EXP_C6;
    p = p3d_fma(p, r, P3D_EXP_C5);
    p = p3d_fma(p, r, P3D_EXP_C4);
    p = p3d_fma(p, r, P3D_EXP_C3);
    p = p3d_fma(p, r, P3D_EXP_C2);
    p = p3d_fma(p, r, P3D_EXP_C1);
    p = p3d_fma(p, r, P3D_EXP_C0);
    return __builtin_ldexpf(p, (int)n);
}

// general argument (the alpha path: -(rho*dl) can be positive when depths are not sorted)
P3D_DEV float p3d_exp(float x) {
    float y = p3d_exp_core(x);
    y = (x < P3D_EXP_LO) ? 0.0f : y;
    return (x > P3D_EXP_HI) ? __builtin_inff() : y;
}

// x <= 0: softplus / sigmoid / cull paths.  The clamp makes -inf safe (inf - inf in the reduction); below P3D_EXP_LO the
// ldexp underflows to exactly 0, so the value equals p3d_exp(x) on the whole domain.
P3D_DEV float p3d_exp_nonpos(float x) { return p3d_exp_core(__builtin_fmaxf(x, P3D_EXP_LO)); }

P3D_DEV float p3d_log1p01(float z) {
    float q = P3D_L1P_C8;
    q = p3d_fma(q, z, P3D_L1P_C7);
    q = p3d_fma(q, z, P3D_L1P_C6);
    q = p3d_fma(q, z, P3D_L1P_C5);
    q = p3d_fma(q, z, P3D_L1P_C4);
    q = p3d_fma(q, z, P3D_L1P_C3);
    q = p3d_fma(q, z, P3D_L1P_C2);
    q = p3d_fma(q, z, P3D_L1P_C1);
    q = p3d_fma(q, z, P3D_L1P_C0);
    return q * z;
}

// torch Softplus(beta=1, threshold=20).  No threshold select: for x > 20 the sum already rounds to x (p3d_numerics.h).
P3D_DEV float p3d_softplus(float x) {
    float z = p3d_exp_nonpos(-__builtin_fabsf(x));
    return __builtin_fmaxf(x, 0.0f) + p3d_log1p01(z);
}

// 1/d for d in [1,2]: linear seed + three Newton steps, all fma (bit-reproducible; an IEEE division costs ~15 VALU slots)
P3D_DEV float p3d_rcp12(float d) {
    float r = p3d_fma(d, -P3D_RCP_A, P3D_RCP_B);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float e = p3d_fma(-d, r, 1.0f);
        r = p3d_fma(r, e, r);
    }
    return r;
}

P3D_DEV float p3d_sigmoid(float x) {
    float z = p3d_exp_nonpos(-__builtin_fabsf(x));
    float r = p3d_rcp12(1.0f + z);
    return (x >= 0.0f) ? r : z * r;
}

// order-preserving float <-> uint32 map (for atomic min/max over arbitrary-sign floats)
P3D_DEV uint32_t p3d_f2ord(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
P3D_DEV float p3d_ord2f(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __builtin_bit_cast(float, u);
}
