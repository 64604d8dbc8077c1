// p3d_decode.hpp — wavefront-level fused triplane gather + OSGDecoder MLP for gfx950 (CDNA4).
//
// One wave64 decodes 32 sample points at a time.  Lane l serves sample j = l & 31 and channel half h = l >> 5:
//   * gather: for each of the 3 planes and 4 bilinear taps the lane reads 16 contiguous channels (64 B, 4 x
//     buffer_load_dwordx4) of the channels-last plane texel; the lane pair (j,0)/(j,1) covers the texel's 128-B line.
//     Out-of-range taps are given an out-of-bounds buffer offset, for which the hardware returns zeros
//     (grid_sample padding_mode='zeros', renderer.py:80).
//   * layer 1 (32 -> 64) and layer 2 (64 -> 32 rgb) run on the matrix cores as v_mfma_f32_32x32x2_f32 chains with the
//     weights as the A operand and the samples on the N axis.  An f32 MFMA is bitwise a k-ordered fmaf chain, which is
//     exactly what include/p3d_numerics.h specifies, so the CPU oracle can restate it.  The operand mapping is chosen so
//     that no cross-lane shuffle is ever needed: the B operand of layer 1 is the lane's own 16 gathered channels, and the
//     B operand of layer-2 instruction (t,s) is accumulator register s of layer-1 tile t (after softplus).
//   * the sigma row (output 0) is two half-chains on the VALU joined by one v_permlane32_swap.
//
// Reference: training/volumetric_rendering/renderer.py:52-81,138-153,266-280; training/triplane.py:516-544.
#pragma once
#include "p3d_math.hpp"
#include "../../include/panic3d_hip.h"

// ---- LDS image of the decoder parameters (per workgroup), in MFMA operand order -------------------------------
//   W0A [2][4][64][4] : tile t, s4, lane l, e  -> w0[32t + (l&31)][16(l>>5) + 4*s4 + e]
//   W1A [2][4][64][4] : tile t, s4, lane l, e  -> w1[1 + (l&31)][nlo(t, 4*s4+e) + 4(l>>5)]
//   B0P [2][2][16]    : half h, tile t, reg r  -> b0[32t + rowof(r) + 4h]
//   B1P [2][16]       : half h, reg r          -> b1[1 + rowof(r) + 4h]
//   W1S [2][32]       : half h, (t,s)          -> w1[0][nlo(t,s) + 4h]
//   B1S [4]           : b1[0], 0, 0, 0
// with rowof(r) = (r&3) + 8(r>>2) and nlo(t,s) = 32t + rowof(s).
#define P3D_LDS_W0A 0
#define P3D_LDS_W1A 2048
#define P3D_LDS_B0P 4096
#define P3D_LDS_B1P 4160
#define P3D_LDS_W1S 4192
#define P3D_LDS_B1S 4256
#define P3D_LDS_MLP_FLOATS 4260
#ifdef P3D_EXPERIMENT_TABLE_ACT  // timing experiment (never shipped): cubic-Hermite activation tables in LDS
#undef P3D_LDS_MLP_FLOATS
#define P3D_TAB_N 320
#define P3D_LDS_TAB_SP 4260                          // [321][4] softplus tail g(u) = log1p(exp(-u)), u = |x|, step 1/16
#define P3D_LDS_TAB_SG (4260 + 4 * (P3D_TAB_N + 1))  // [321][4] sigmoid(u)
#define P3D_LDS_MLP_FLOATS (4260 + 8 * (P3D_TAB_N + 1))
P3D_DEV float p3d_tab_eval(const float* tab, float u) {
    float t = __builtin_fminf(u, (float)P3D_TAB_N / 16.0f) * 16.0f;
    int i = (int)t;
    float f = t - (float)i;
    const f32x4 c = *(const f32x4*)(tab + 4 * i);
    return p3d_fma(p3d_fma(p3d_fma(c[3], f, c[2]), f, c[1]), f, c[0]);
}
#define P3D_ACT_SOFTPLUS(lds, x) (__builtin_fmaxf(x, 0.0f) + p3d_tab_eval((lds) + P3D_LDS_TAB_SP, __builtin_fabsf(x)))
P3D_DEV float p3d_sigmoid_tab(const float* lds, float x) {
    float s = p3d_tab_eval(lds + P3D_LDS_TAB_SG, __builtin_fabsf(x));
    return x >= 0.0f ? s : 1.0f - s;
}
#define P3D_ACT_SIGMOID(lds, x) p3d_sigmoid_tab(lds, x)
#else
#define P3D_ACT_SOFTPLUS(lds, x) p3d_softplus(x)
#define P3D_ACT_SIGMOID(lds, x) p3d_sigmoid(x)
#endif

#ifdef P3D_ABL_NOMFMA  // timing experiment: matrix-core work replaced by one VALU op
P3D_DEV f32x16 p3d_fake_mfma(float a, float b, f32x16 c) { c[0] = p3d_fma(a, b, c[0]); return c; }
#define P3D_MFMA(a, b, c) p3d_fake_mfma(a, b, c)
#else
#define P3D_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#endif

P3D_DEV int p3d_rowof(int r) { return (r & 3) + 8 * (r >> 2); }

// Cooperative load by the whole workgroup (blockDim.x threads).  Caller must __syncthreads() afterwards.
P3D_DEV void p3d_load_mlp_to_lds(float* lds, const float* w0, const float* b0, const float* w1, const float* b1) {
    for (int idx = threadIdx.x; idx < 2048; idx += blockDim.x) {
        int e = idx & 3, l = (idx >> 2) & 63, s4 = (idx >> 8) & 3, t = idx >> 10;
        int s = 4 * s4 + e, i = l & 31, h = l >> 5;
        lds[P3D_LDS_W0A + idx] = w0[(32 * t + i) * 32 + 16 * h + s];
        lds[P3D_LDS_W1A + idx] = w1[(1 + i) * 64 + 32 * t + p3d_rowof(s) + 4 * h];
    }
    for (int idx = threadIdx.x; idx < 64; idx += blockDim.x) {
        int r = idx & 15, t = (idx >> 4) & 1, h = idx >> 5;
        lds[P3D_LDS_B0P + idx] = b0[32 * t + p3d_rowof(r) + 4 * h];
        int s = idx & 15, tt = (idx >> 4) & 1;
        lds[P3D_LDS_W1S + idx] = w1[32 * tt + p3d_rowof(s) + 4 * h];
    }
    for (int idx = threadIdx.x; idx < 32; idx += blockDim.x) {
        int r = idx & 15, h = idx >> 4;
        lds[P3D_LDS_B1P + idx] = b1[1 + p3d_rowof(r) + 4 * h];
    }
    if (threadIdx.x < 4) lds[P3D_LDS_B1S + threadIdx.x] = threadIdx.x == 0 ? b1[0] : 0.0f;
#ifdef P3D_EXPERIMENT_TABLE_ACT
    for (int i = threadIdx.x; i <= P3D_TAB_N; i += blockDim.x) {
        const double hh = 1.0 / 16.0, u0 = i * hh, u1 = u0 + hh;
        double g0 = log1p(exp(-u0)), g1 = log1p(exp(-u1)), dg0 = -1.0 / (1.0 + exp(u0)), dg1 = -1.0 / (1.0 + exp(u1));
        double s0 = 1.0 / (1.0 + exp(-u0)), s1 = 1.0 / (1.0 + exp(-u1)), ds0 = s0 * (1 - s0), ds1 = s1 * (1 - s1);
        if (i == P3D_TAB_N) { g0 = g1 = dg0 = dg1 = 0.0; s0 = s1 = 1.0; ds0 = ds1 = 0.0; }
        float* a = lds + P3D_LDS_TAB_SP + 4 * i;
        a[0] = (float)g0; a[1] = (float)(hh * dg0); a[2] = (float)(3 * (g1 - g0) - hh * (2 * dg0 + dg1)); a[3] = (float)(2 * (g0 - g1) + hh * (dg0 + dg1));
        float* b = lds + P3D_LDS_TAB_SG + 4 * i;
        b[0] = (float)s0; b[1] = (float)(hh * ds0); b[2] = (float)(3 * (s1 - s0) - hh * (2 * ds0 + ds1)); b[3] = (float)(2 * (s0 - s1) + hh * (ds0 + ds1));
    }
#endif
}

struct P3dPlaneGeom {
    float halfW, halfH, fW, fH;  // 0.5*W, 0.5*H, (float)W, (float)H
    int W;
    uint32_t plane_bytes;        // H*W*32*4
};

#define P3D_OOB_OFFSET 0x7ffffff0u

// One plane, this lane's 16 channels: F.grid_sample(bilinear, zeros, align_corners=False) — renderer.py:80.
P3D_DEV void p3d_tap_offsets(const P3dPlaneGeom& g, uint32_t plane_off, uint32_t chan_off, float gx, float gy,
                             uint32_t off[4], float wgt[4], bool live = true) {
    float ix = (gx + 1.0f) * g.halfW - 0.5f;
    float iy = (gy + 1.0f) * g.halfH - 0.5f;
    // !live: this lane's result is known not to matter (cropped / dead ray): give it out-of-bounds offsets so that it issues
    // no L1 lookups (the gather rate is the kernel's binding limit) — it then decodes an all-zero feature vector
    bool inr = live && (ix > -1.0f) && (ix < g.fW) && (iy > -1.0f) && (iy < g.fH);
    float fx0 = __builtin_floorf(ix), fy0 = __builtin_floorf(iy);
    float wx1 = ix - fx0, wy1 = iy - fy0;
    float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    wgt[0] = inr ? wy0 * wx0 : 0.0f;
    wgt[1] = inr ? wy0 * wx1 : 0.0f;
    wgt[2] = inr ? wy1 * wx0 : 0.0f;
    wgt[3] = inr ? wy1 * wx1 : 0.0f;
    int x0 = inr ? (int)fx0 : 0, y0 = inr ? (int)fy0 : 0;
    int H = (int)g.fH;
    bool vx0 = inr && x0 >= 0, vx1 = inr && (x0 + 1 < g.W), vy0 = y0 >= 0, vy1 = (y0 + 1 < H);
    uint32_t base = plane_off + chan_off + (uint32_t)((y0 * g.W + x0) * 128);
    uint32_t row = (uint32_t)g.W * 128u;
    off[0] = (vx0 && vy0) ? base : P3D_OOB_OFFSET;
    off[1] = (vx1 && vy0) ? base + 128u : P3D_OOB_OFFSET;
    off[2] = (vx0 && vy1) ? base + row : P3D_OOB_OFFSET;
    off[3] = (vx1 && vy1) ? base + row + 128u : P3D_OOB_OFFSET;
}

template <typename RSRC>
P3D_DEV f32x16 p3d_load16(RSRC rs, uint32_t off) {
    f32x16 v;
#ifdef P3D_ABL_NOLOAD  // timing experiment: no gather traffic
    float f = __builtin_bit_cast(float, off);
    v = (f32x16){f, f, f, f, f, f, f, f, f, f, f, f, f, f, f, f};
    return v;
#endif
    i32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    i32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off + 16, 0, 0);
    i32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off + 32, 0, 0);
    i32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off + 48, 0, 0);
    f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b), fc = __builtin_bit_cast(f32x4, c),
          fd = __builtin_bit_cast(f32x4, d);
    v.s0 = fa.x; v.s1 = fa.y; v.s2 = fa.z; v.s3 = fa.w;
    v.s4 = fb.x; v.s5 = fb.y; v.s6 = fb.z; v.s7 = fb.w;
    v.s8 = fc.x; v.s9 = fc.y; v.sa = fc.z; v.sb = fc.w;
    v.sc = fd.x; v.sd = fd.y; v.se = fd.z; v.sf = fd.w;
    return v;
}

P3D_DEV f32x16 p3d_bilerp(const float wgt[4], const f32x16& v00, const f32x16& v01, const f32x16& v10, const f32x16& v11) {
    f32x16 f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float a = wgt[0] * v00[c];
        a = p3d_fma(wgt[1], v01[c], a);
        a = p3d_fma(wgt[2], v10[c], a);
        a = p3d_fma(wgt[3], v11[c], a);
        f[c] = a;
    }
    return f;
}

template <typename RSRC>
P3D_DEV f32x16 p3d_sample_plane(RSRC rs, const P3dPlaneGeom& g, uint32_t plane_off, uint32_t chan_off, float gx,
                                float gy) {
    uint32_t off[4];
    float wgt[4];
    p3d_tap_offsets(g, plane_off, chan_off, gx, gy, off, wgt);
    f32x16 v00 = p3d_load16(rs, off[0]);
    f32x16 v01 = p3d_load16(rs, off[1]);
    f32x16 v10 = p3d_load16(rs, off[2]);
    f32x16 v11 = p3d_load16(rs, off[3]);
    f32x16 f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float a = wgt[0] * v00[c];
        a = p3d_fma(wgt[1], v01[c], a);
        a = p3d_fma(wgt[2], v10[c], a);
        a = p3d_fma(wgt[3], v11[c], a);
        f[c] = a;
    }
    return f;
}

// exchange with the partner lane (l ^ 32)
P3D_DEV float p3d_partner(float x) {
    uint32_t xi = __builtin_bit_cast(uint32_t, x);
    auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    uint32_t y = (__lane_id() < 32) ? r[1] : r[0];
    return __builtin_bit_cast(float, y);
}

struct P3dDecodeCfg {
    float coord_scale, crop_limit, cull_thresh;
    int plane_mode, flags;
};

// Decode one sample per lane pair.  All 64 lanes must be active.  lds = workgroup MLP image (p3d_load_mlp_to_lds).
// Returns sigma (after masks) and, if WANT_RGB, this lane's 16 colour channels: register r holds channel
// rowof(r) + 4h  (channels {0-3,8-11,16-19,24-27} for h = 0, {4-7,12-15,20-23,28-31} for h = 1).
// live = false on a lane: its gathers are suppressed (see p3d_tap_offsets) and its outputs are garbage except that the
// position-only crop mask still applies.
template <bool WANT_RGB, typename RSRC>
P3D_DEV void p3d_decode_wave(const float* lds, RSRC rs, const P3dPlaneGeom& g, const P3dDecodeCfg& cfg, float px,
                             float py, float pz, float& sigma_out, f32x16& rgb, bool live = true) {
    const int lane = __lane_id();
    const int h = lane >> 5;
    const uint32_t chan_off = (uint32_t)h * 64u;
    float qx = px * cfg.coord_scale, qy = py * cfg.coord_scale, qz = pz * cfg.coord_scale;  // renderer.py:77
    // generate_planes / project_onto_planes: renderer.py:26-66
    // Plane-at-a-time software pipeline: the taps of plane p+1 are in flight while plane p is interpolated, which bounds
    // the live tap registers to 2 x 64 (the compiler would otherwise hoist all 48 loads = 192 VGPRs).
    float g2x = cfg.plane_mode ? qy : qz, g2y = cfg.plane_mode ? qz : qx;
    uint32_t of0[4], of1[4], of2[4];
    float wg0[4], wg1[4], wg2[4];
    p3d_tap_offsets(g, 0u, chan_off, qx, qy, of0, wg0, live);
    p3d_tap_offsets(g, g.plane_bytes, chan_off, qx, qz, of1, wg1, live);
    p3d_tap_offsets(g, 2u * g.plane_bytes, chan_off, g2x, g2y, of2, wg2, live);
    f32x16 a00 = p3d_load16(rs, of0[0]), a01 = p3d_load16(rs, of0[1]), a10 = p3d_load16(rs, of0[2]), a11 = p3d_load16(rs, of0[3]);
    f32x16 b00 = p3d_load16(rs, of1[0]), b01 = p3d_load16(rs, of1[1]), b10 = p3d_load16(rs, of1[2]), b11 = p3d_load16(rs, of1[3]);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 X = p3d_bilerp(wg0, a00, a01, a10, a11);
    __builtin_amdgcn_sched_barrier(0);
    a00 = p3d_load16(rs, of2[0]); a01 = p3d_load16(rs, of2[1]); a10 = p3d_load16(rs, of2[2]); a11 = p3d_load16(rs, of2[3]);
    __builtin_amdgcn_sched_barrier(0);
    {
        f32x16 f1 = p3d_bilerp(wg1, b00, b01, b10, b11);
#pragma unroll
        for (int c = 0; c < 16; ++c) X[c] = X[c] + f1[c];
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        f32x16 f2 = p3d_bilerp(wg2, a00, a01, a10, a11);
#pragma unroll
        for (int c = 0; c < 16; ++c) X[c] = (X[c] + f2[c]) * P3D_THIRD;  // triplane.py:530 mean(1)
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- layer 1 on the matrix cores: acc[t][r] = b0[n] + sum_k w0[n][k] X[k], n = 32t + rowof(r) + 4h
    const f32x4* b0p = (const f32x4*)(lds + P3D_LDS_B0P + h * 32);
    f32x16 acc0, acc1;
    {
        f32x4 q0 = b0p[0], q1 = b0p[1], q2 = b0p[2], q3 = b0p[3];
        f32x4 r0 = b0p[4], r1 = b0p[5], r2 = b0p[6], r3 = b0p[7];
        acc0 = (f32x16){q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        acc1 = (f32x16){r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
    }
    const f32x4* w0a = (const f32x4*)(lds + P3D_LDS_W0A) + lane;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        f32x4 a0 = w0a[(0 * 4 + s4) * 64];
        f32x4 a1 = w0a[(1 * 4 + s4) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = P3D_MFMA(a0[e], X[4 * s4 + e], acc0);
            acc1 = P3D_MFMA(a1[e], X[4 * s4 + e], acc1);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // Softplus: triplane.py:524
        acc0[r] = P3D_ACT_SOFTPLUS(lds, acc0[r]);
        acc1[r] = P3D_ACT_SOFTPLUS(lds, acc1[r]);
    }
    // ---- sigma row on the VALU: two half chains joined across the lane pair (triplane.py:543)
    const f32x4* w1s = (const f32x4*)(lds + P3D_LDS_W1S + h * 32);
    float sa = (h == 0) ? lds[P3D_LDS_B1S] : 0.0f;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        f32x4 w = w1s[s4];
#pragma unroll
        for (int e = 0; e < 4; ++e) sa = p3d_fma(w[e], acc0[4 * s4 + e], sa);
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        f32x4 w = w1s[4 + s4];
#pragma unroll
        for (int e = 0; e < 4; ++e) sa = p3d_fma(w[e], acc1[4 * s4 + e], sa);
    }
    float sigma = sa + p3d_partner(sa);

    if (WANT_RGB) {  // ---- layer 2 rows 1..32 on the matrix cores + sigmoid (triplane.py:539-542)
        const f32x4* b1p = (const f32x4*)(lds + P3D_LDS_B1P + h * 16);
        f32x4 q0 = b1p[0], q1 = b1p[1], q2 = b1p[2], q3 = b1p[3];
        f32x16 o = (f32x16){q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        const f32x4* w1a = (const f32x4*)(lds + P3D_LDS_W1A) + lane;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            f32x4 a = w1a[(0 * 4 + s4) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) o = P3D_MFMA(a[e], acc0[4 * s4 + e], o);
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            f32x4 a = w1a[(1 * 4 + s4) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) o = P3D_MFMA(a[e], acc1[4 * s4 + e], o);
        }
        const bool fs = (cfg.flags & P3D_FLAG_FORCE_SIGMOID) != 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sg = P3D_ACT_SIGMOID(lds, o[r]);
            rgb[r] = fs ? sg : sg * 1.002f - 0.001f;
        }
    }
    // ---- masks on raw sigma: renderer.py:138-153,187-198
    if (cfg.flags & P3D_FLAG_CROP) {
        if (__builtin_fabsf(px) > cfg.crop_limit || __builtin_fabsf(pz) > cfg.crop_limit) sigma = P3D_SIGMA_MASKED;
    }
    if (cfg.flags & (P3D_FLAG_CULL | P3D_FLAG_BINARIZE)) {
        float a = 1.0f - p3d_exp_nonpos(-p3d_softplus(sigma - 1.0f));
        if (cfg.flags & P3D_FLAG_BINARIZE)
            sigma = (a < cfg.cull_thresh) ? P3D_SIGMA_MASKED : P3D_SIGMA_SOLID;
        else if (a < cfg.cull_thresh)
            sigma = P3D_SIGMA_MASKED;
    }
    sigma_out = sigma;
}
