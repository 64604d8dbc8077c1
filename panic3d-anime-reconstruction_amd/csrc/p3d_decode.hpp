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
#define P3D_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

// ---- tolerance mode of the FINAL pass (P3D_FLAG_FAST_COLOR) --------------------------------------------
// Both layers on v_mfma_f32_32x32x16_f16 (16x the f32 MFMA rate) with every operand split into two f16 terms
// (x = xh + xl, xh = f16(x) by truncation, xl = f16(x - xh)): x*w ~ xh*wh + xl*wh + xh*wl, fp32 accumulation — about 2^-21
// relative per product — and the activations on the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32).
// Extra LDS image (only in the FAST kernels), in MFMA operand order, 16 B per lane:
//   W0H [2 t][2 q][2 hi/lo][64 lanes][8 f16] : w0[32t + (l&31)][16(l>>5) + 8q + i]
//   W1H [2 t][2 pp][2 hi/lo][64 lanes][8 f16]: w1[1 + (l&31)][32t + rowof(8pp + i) + 4(l>>5)]   (overlays W1A: the f32
//                                               colour weights are never used by a FAST kernel)
#define P3D_LDS_W0H P3D_LDS_MLP_FLOATS
#define P3D_LDS_W1H P3D_LDS_W1A
#define P3D_LDS_FAST_FLOATS (P3D_LDS_MLP_FLOATS + 2048)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define P3D_MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

P3D_DEV int p3d_rowof(int r) { return (r & 3) + 8 * (r >> 2); }

// Cooperative load by the whole workgroup (blockDim.x threads).  Caller must __syncthreads() afterwards.
// with_w1a = false: a FAST kernel, whose f16 colour weights overlay W1A (p3d_load_mlp_f16_to_lds).
// with_w0a = false: a tolerance-mode kernel that never runs the exact layer 1 either (the density query): its `lds` base may
// then point P3D_LDS_B0P floats BEFORE the allocation (nothing below P3D_LDS_B0P is touched).
P3D_DEV void p3d_load_mlp_to_lds(float* lds, const float* w0, const float* b0, const float* w1, const float* b1,
                                 bool with_w1a = true, bool with_w0a = true) {
    for (int idx = threadIdx.x; idx < 2048; idx += blockDim.x) {
        int e = idx & 3, l = (idx >> 2) & 63, s4 = (idx >> 8) & 3, t = idx >> 10;
        int s = 4 * s4 + e, i = l & 31, h = l >> 5;
        if (with_w0a) lds[P3D_LDS_W0A + idx] = w0[(32 * t + i) * 32 + 16 * h + s];
        if (with_w1a) lds[P3D_LDS_W1A + idx] = w1[(1 + i) * 64 + 32 * t + p3d_rowof(s) + 4 * h];
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
}

// FAST kernels: the f16 hi/lo operand images.  Call AFTER p3d_load_mlp_to_lds (W1H overlays W1A, which such a kernel never
// reads) and before its __syncthreads().
P3D_DEV void p3d_load_mlp_f16_to_lds(float* lds, const float* w0, const float* w1, bool with_w1h = true) {
    _Float16* w0h = (_Float16*)(lds + P3D_LDS_W0H);
    _Float16* w1h = (_Float16*)(lds + P3D_LDS_W1H);
    for (int idx = threadIdx.x; idx < 2048; idx += blockDim.x) {  // idx = ((t*2 + q) * 64 + l) * 8 + i
        const int i = idx & 7, l = (idx >> 3) & 63, q = (idx >> 9) & 1, t = idx >> 10;
        const int m = l & 31, hh = l >> 5;
        const float a = w0[(32 * t + m) * 32 + 16 * hh + 8 * q + i];
        const float b = w1[(1 + m) * 64 + 32 * t + p3d_rowof(8 * q + i) + 4 * hh];
        const _Float16 ah = (_Float16)a, bh = (_Float16)b;
        const int o = (((t * 2 + q) * 2) * 64 + l) * 8 + i;  // hi image of chunk (t,q); the lo image follows 64 lanes later
        w0h[o] = ah; w0h[o + 512] = (_Float16)(a - (float)ah);
        if (with_w1h) { w1h[o] = bh; w1h[o + 512] = (_Float16)(b - (float)bh); }
    }
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
    // no L1 lookups — it then decodes an all-zero feature vector
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

// The lane's 64 B of one tap as four 16-B loads (per-lane gathers: one lane = one sample's half texel).  The vector L1 charges
// such an instruction by the distinct texels it touches (tools/ubench/l1_gather*.hip, in-kernel ablations in
// profiles/history/r02_notes.txt); rotating the 16-B slot by the quad index helps in the micro-benchmark and not in the kernels, sharing
// a texel inside a quad helps in both: p3d_load16_quad below.
template <typename RSRC>
P3D_DEV f32x16 p3d_load16(RSRC rs, uint32_t off) {
    f32x16 v;
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
// compiler barrier that makes 16 values "observed" at this point of the program (no instruction is emitted)
P3D_DEV void p3d_pin16(f32x16& v) {
    asm volatile("" : "+v"(v.s0), "+v"(v.s1), "+v"(v.s2), "+v"(v.s3), "+v"(v.s4), "+v"(v.s5), "+v"(v.s6), "+v"(v.s7), "+v"(v.s8),
                      "+v"(v.s9), "+v"(v.sa), "+v"(v.sb), "+v"(v.sc), "+v"(v.sd), "+v"(v.se), "+v"(v.sf) : : "memory");
}
#ifndef P3D_GATHER_DEPTH
#define P3D_GATHER_DEPTH 4  // taps in flight per lane (2 / 3 / 4 / 6 measured within 3 % of each other: profiles/history/r02_notes.txt)
#endif
// Tap-granular software pipeline over the 12 taps of a sample (3 planes x nw, ne, sw, se): P3D_GATHER_DEPTH taps (16 registers
// each) are in flight while the oldest one is folded into its plane's bilinear sum, in the contract's order (nw, ne, sw, se;
// plane 0 + plane 1, + plane 2, x 1/3).  The depth bounds the live tap registers (the compiler would otherwise hoist all 48
// loads = 192 VGPRs).  load(k) returns this lane's 16 channels of tap k.
template <int CTRL>  // quad_perm: lane i of every quad reads lane (CTRL >> 2i) & 3 of the same quad
P3D_DEV uint32_t p3d_quad_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
P3D_DEV float p3d_quad_f(float v) { return __builtin_bit_cast(float, p3d_quad_u<CTRL>(__builtin_bit_cast(uint32_t, v))); }
// QUAD: the registers of a tap are [ray of the quad][4 channels] (p3d_load16_quad), so register c takes the weight of ray c >> 2.
template <bool QUAD = false, typename LOAD>
P3D_DEV f32x16 p3d_fold_taps(const float wg[12], LOAD load) {
    f32x16 tap[P3D_GATHER_DEPTH];
#pragma unroll
    for (int k = 0; k < P3D_GATHER_DEPTH; ++k) tap[k] = load(k);
    f32x16 X, f;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        __builtin_amdgcn_sched_barrier(0);
        const f32x16 v = tap[k % P3D_GATHER_DEPTH];
        float w4[4];
        if constexpr (QUAD) {
            w4[0] = p3d_quad_f<0x00>(wg[k]); w4[1] = p3d_quad_f<0x55>(wg[k]); w4[2] = p3d_quad_f<0xaa>(wg[k]); w4[3] = p3d_quad_f<0xff>(wg[k]);
        } else {
            w4[0] = w4[1] = w4[2] = w4[3] = wg[k];
        }
        if ((k & 3) == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) f[c] = w4[c >> 2] * v[c];
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) f[c] = p3d_fma(w4[c >> 2], v[c], f[c]);
        }
        // pin the partial sums here: without it the compiler sinks all the arithmetic below the last load (sched_barrier only
        // orders machine instructions that already sit on either side of it) and every tap stays live.  At the end of a plane the
        // value that lives on is X, so X is what gets pinned and f simply dies there (pinning f AFTER `X = f` made f a modified
        // copy of X: 16 v_mov per plane, 48 of a decode step's ~1600 vector instructions in the round-4 ISA).
        if (k == 3) {
            p3d_pin16(f);
            X = f;
        } else if (k == 7) {
#pragma unroll
            for (int c = 0; c < 16; ++c) X[c] = X[c] + f[c];
            p3d_pin16(X);
        } else if (k == 11) {
#pragma unroll
            for (int c = 0; c < 16; ++c) X[c] = (X[c] + f[c]) * P3D_THIRD;  // triplane.py:530 mean(1)
            p3d_pin16(X);
        } else {
            p3d_pin16(f);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (k + P3D_GATHER_DEPTH < 12) tap[k % P3D_GATHER_DEPTH] = load(k + P3D_GATHER_DEPTH);
    }
    __builtin_amdgcn_sched_barrier(0);
    return X;
}

// ---- quad-cooperative gathers ----------------------------------------------------------------------------------------------
// The vector L1 charges a gather instruction by the distinct texels it touches (profiles/history/r02_notes.txt: "every ray its own
// texel" 63 clk, "a quad shares a texel" 16 clk per instruction), so the four lanes of a quad (4 consecutive samples, same
// channel half) fetch the half-texel of ONE of their samples per instruction — lane i piece i, 64 contiguous bytes — instead
// of four pieces of four different texels: instruction r serves sample r of the quad.  A lane then holds 4 channels of each of
// the quad's 4 samples; the bilinear fold is per channel, so it runs on that layout unchanged (the weight of sample r comes
// from lane r by DPP) — every (sample, channel) value is produced by the same multiply / fma chain as before, on another lane:
// bit-identical — and ONE 4 x 4 transpose inside the quad per sample (not per tap) restores "lane = sample".

// tap registers [4 r + d] = channels 4 i + d (of this lane's half) of sample r of the quad; off = THIS lane's tap offset
template <typename RSRC>
P3D_DEV f32x16 p3d_load16_quad(RSRC rs, uint32_t off) {
    const uint32_t piece = ((uint32_t)__lane_id() & 3u) * 16u;
    const i32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(p3d_quad_u<0x00>(off) + piece), 0, 0);
    const i32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(p3d_quad_u<0x55>(off) + piece), 0, 0);
    const i32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(p3d_quad_u<0xaa>(off) + piece), 0, 0);
    const i32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(p3d_quad_u<0xff>(off) + piece), 0, 0);
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b), fc = __builtin_bit_cast(f32x4, c),
                fd = __builtin_bit_cast(f32x4, d);
    f32x16 v;
    v.s0 = fa.x; v.s1 = fa.y; v.s2 = fa.z; v.s3 = fa.w;
    v.s4 = fb.x; v.s5 = fb.y; v.s6 = fb.z; v.s7 = fb.w;
    v.s8 = fc.x; v.s9 = fc.y; v.sa = fc.z; v.sb = fc.w;
    v.sc = fd.x; v.sd = fd.y; v.se = fd.z; v.sf = fd.w;
    return v;
}

// in: [4 r + d] = channels 4 i + d of sample r (lane i of the quad); out: [4 p + d] = channels 4 p + d of this lane's sample.
// rotate the rows by the lane index, move row q from lane (i - q) & 3 (one DPP each), rotate back.  The lane-dependent
// selects are written as bit-field inserts under per-lane masks ((m & a) | (~m & b) = one v_bfi / v_bitop3 each): as `?:` on
// lane predicates the optimiser re-derived them into compare-and-select chains on the lane index — 700 instructions per
// sample instead of 76 (seen in the ISA of the first version).
P3D_DEV uint32_t p3d_bsel(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }  // m ? a : b, m = 0 or ~0
P3D_DEV f32x16 p3d_quad_transpose(const f32x16& in) {
    const uint32_t i = (uint32_t)__lane_id() & 3u;
    uint32_t m0 = 0u - (i & 1u), m1 = 0u - ((i >> 1) & 1u);
    asm volatile("" : "+v"(m0), "+v"(m1));  // opaque: keep them masks
    uint32_t x[16], y[16], r[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float f = in[c];  // (through a scalar: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this clang)
        x[c] = __builtin_bit_cast(uint32_t, f);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {  // y[q] = x[(i + q) & 3]
        uint32_t b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = p3d_bsel(m0, x[4 * ((q + 1) & 3) + d], x[4 * q + d]);
#pragma unroll
        for (int q = 0; q < 4; ++q) y[4 * q + d] = p3d_bsel(m1, b[(q + 2) & 3], b[q]);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {  // r[q] (lane i) = y[q] of lane (i - q) & 3  =  channels piece (i - q) & 3 of sample i
        r[d] = y[d];
        r[4 + d] = p3d_quad_u<0x93>(y[4 + d]);   // lanes 0..3 read 3, 0, 1, 2
        r[8 + d] = p3d_quad_u<0x4e>(y[8 + d]);   // 2, 3, 0, 1
        r[12 + d] = p3d_quad_u<0x39>(y[12 + d]); // 1, 2, 3, 0
    }
    f32x16 out;
#pragma unroll
    for (int d = 0; d < 4; ++d) {  // out[p] = r[(i - p) & 3]
        uint32_t b[4], c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = p3d_bsel(m0, r[4 * ((q + 3) & 3) + d], r[4 * q + d]);   // b[q] = r[(q - i0) & 3]
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = p3d_bsel(m1, b[(q + 2) & 3], b[q]);                     // c[q] = r[(q - i) & 3]
        // out[p] = r[(i - p) & 3] = c[(2 i - p) & 3]: for even i c[(-p) & 3], for odd i c[(2 - p) & 3]
#pragma unroll
        for (int pz = 0; pz < 4; ++pz) out[4 * pz + d] = __builtin_bit_cast(float, p3d_bsel(m0, c[(2 - pz) & 3], c[(4 - pz) & 3]));
    }
    return out;
}

// This lane's 16 interpolated feature channels (16h .. 16h+15) of the sample at (px,py,pz): the three bilinear plane samples
// and their mean (renderer.py:68-81, triplane.py:530), gathered straight from the planes (per-lane L1 gathers).
template <bool QUADG = false, typename RSRC>
P3D_DEV f32x16 p3d_gather_features(RSRC rs, const P3dPlaneGeom& g, const P3dDecodeCfg& cfg, float px, float py, float pz,
                                   bool live) {
    const int h = __lane_id() >> 5;
    const uint32_t chan_off = (uint32_t)h * 64u;
    float qx = px * cfg.coord_scale, qy = py * cfg.coord_scale, qz = pz * cfg.coord_scale;  // renderer.py:77
    // generate_planes / project_onto_planes: renderer.py:26-66
    float g2x = cfg.plane_mode ? qy : qz, g2y = cfg.plane_mode ? qz : qx;
    uint32_t of[12];
    float wg[12];
    p3d_tap_offsets(g, 0u, chan_off, qx, qy, of, wg, live);
    p3d_tap_offsets(g, g.plane_bytes, chan_off, qx, qz, of + 4, wg + 4, live);
    p3d_tap_offsets(g, 2u * g.plane_bytes, chan_off, g2x, g2y, of + 8, wg + 8, live);
    if constexpr (QUADG) return p3d_quad_transpose(p3d_fold_taps<true>(wg, [&](int k) { return p3d_load16_quad(rs, of[k]); }));
    else return p3d_fold_taps(wg, [&](int k) { return p3d_load16(rs, of[k]); });
}

// ---- LDS-staged gather for the regular-grid query (north_star's "LDS-staged plane tiles") -------------------------------
// The 32 consecutive grid points of a wave-step (one grid row: x and y fixed, z advancing by about half a texel) read 384
// taps that fall on ~80 distinct texels: a 2 x 2..3 block of plane (x,y) and two 2 x ~18 strips of the planes that carry z.
// The vector L1's service time per gather instruction is what bounds the density query, so the wave loads each
// plane's texel box ONCE, coalesced (whole 128-byte lines, <= 15 loads per lane instead of 48), parks it in its own LDS region
// and gathers the taps from there.  Same values (out-of-plane texels are stored as zeros), same arithmetic: bit-identical.
#define P3D_BOX_TEXELS 40                                  // texels per plane box (xspan * yspan must fit)
#define P3D_BOX_STRIDE 144u                                // bytes per parked texel: 128 + 16 of padding, so that consecutive texels
                                                           // start 36 banks apart and the 16 lanes of a ds_read_b128 group (<= 8
                                                           // neighbouring texels x 4 banks) never collide
#define P3D_BOX_PLANE_FLOATS (P3D_BOX_TEXELS * 36)
#define P3D_BOX_FLOATS_PER_WAVE (3 * P3D_BOX_PLANE_FLOATS)  // 16.9 KB per wave

struct P3dBox { int xmin, ymin, xspan, yspan; };  // wave-uniform

// texel of the (nw) tap of one plane; inr as in p3d_tap_offsets
P3D_DEV void p3d_plane_texel(const P3dPlaneGeom& g, float gx, float gy, int& x0, int& y0, bool& inr) {
    const float ix = (gx + 1.0f) * g.halfW - 0.5f, iy = (gy + 1.0f) * g.halfH - 0.5f;
    inr = (ix > -1.0f) && (ix < g.fW) && (iy > -1.0f) && (iy < g.fH);
    x0 = (int)__builtin_floorf(ix);
    y0 = (int)__builtin_floorf(iy);
}

// Box of one plane from the first and the last sample of the tile (the coordinates are monotone along a grid row), verified
// against every lane.  Returns false (wave-uniform) when the taps do not fit: the caller then gathers directly.
P3D_DEV bool p3d_make_box(int x0, int y0, bool inr, P3dBox& b) {
    const int xa = __builtin_amdgcn_readlane(x0, 0), xb = __builtin_amdgcn_readlane(x0, 31);
    const int ya = __builtin_amdgcn_readlane(y0, 0), yb = __builtin_amdgcn_readlane(y0, 31);
    b.xmin = xa < xb ? xa : xb;
    b.ymin = ya < yb ? ya : yb;
    const int xmax = xa < xb ? xb : xa, ymax = ya < yb ? yb : ya;
    b.xspan = xmax - b.xmin + 2;
    b.yspan = ymax - b.ymin + 2;
    const bool inside = inr && x0 >= b.xmin && x0 <= xmax && y0 >= b.ymin && y0 <= ymax;
    return __builtin_amdgcn_ballot_w64(!inside) == 0 && b.xspan * b.yspan <= P3D_BOX_TEXELS;
}

// All 64 lanes: load the box of one plane (<= 40 texels = 320 16-byte pieces, 5 per lane) into registers.
template <typename RSRC>
P3D_DEV void p3d_box_load(RSRC rs, const P3dPlaneGeom& g, uint32_t plane_off, const P3dBox& b, i32x4 v[5]) {
    const int lane = __lane_id();
    const int npieces = b.xspan * b.yspan * 8, H = (int)g.fH;
    const float inv = 1.0f / (float)b.xspan;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int f = lane + 64 * r, tex = f >> 3, piece = f & 7;
        const int by = (int)(((float)tex + 0.5f) * inv);  // tex / xspan for the small integers involved
        const int bx = tex - by * b.xspan;
        const int x = b.xmin + bx, y = b.ymin + by;
        const bool ok = f < npieces && x >= 0 && x < g.W && y >= 0 && y < H;  // out-of-plane texels: zeros (padding_mode='zeros')
        const uint32_t off = ok ? plane_off + (uint32_t)((y * g.W + x) * 128 + piece * 16) : P3D_OOB_OFFSET;
        v[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    }
}
P3D_DEV void p3d_box_store(float* box_plane /* this wave's region of one plane */, const i32x4 v[5]) {
    const int lane = __lane_id();
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int f = lane + 64 * r;  // piece f = texel (f >> 3), 16-byte piece (f & 7)
        *(i32x4*)((char*)box_plane + (f >> 3) * P3D_BOX_STRIDE + (f & 7) * 16) = v[r];
    }
}

// tap byte offsets inside the wave's box region + bilinear weights of one plane (cf. p3d_tap_offsets)
P3D_DEV void p3d_tap_offsets_box(const P3dPlaneGeom& g, const P3dBox& b, uint32_t region_off, uint32_t chan_off, float gx, float gy,
                                 uint32_t off[4], float wgt[4], bool live) {
    float ix = (gx + 1.0f) * g.halfW - 0.5f;
    float iy = (gy + 1.0f) * g.halfH - 0.5f;
    bool inr = live && (ix > -1.0f) && (ix < g.fW) && (iy > -1.0f) && (iy < g.fH);
    float fx0 = __builtin_floorf(ix), fy0 = __builtin_floorf(iy);
    float wx1 = ix - fx0, wy1 = iy - fy0;
    float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    wgt[0] = inr ? wy0 * wx0 : 0.0f;
    wgt[1] = inr ? wy0 * wx1 : 0.0f;
    wgt[2] = inr ? wy1 * wx0 : 0.0f;
    wgt[3] = inr ? wy1 * wx1 : 0.0f;
    // the caller has verified that every lane's texel lies inside the box (p3d_make_box); a lane that is not live carries zero
    // weights and reads the box origin
    const int tx = inr ? (int)fx0 - b.xmin : 0, ty = inr ? (int)fy0 - b.ymin : 0;
    const uint32_t base = region_off + chan_off + (uint32_t)(ty * b.xspan + tx) * P3D_BOX_STRIDE;
    const uint32_t row = (uint32_t)b.xspan * P3D_BOX_STRIDE;
    off[0] = base; off[1] = base + P3D_BOX_STRIDE; off[2] = base + row; off[3] = base + row + P3D_BOX_STRIDE;
}
P3D_DEV f32x16 p3d_lds16(const float* box, uint32_t off) {
    const f32x4* q = (const f32x4*)((const char*)box + off);
    const f32x4 fa = q[0], fb = q[1], fc = q[2], fd = q[3];
    f32x16 v;
    v.s0 = fa.x; v.s1 = fa.y; v.s2 = fa.z; v.s3 = fa.w;
    v.s4 = fb.x; v.s5 = fb.y; v.s6 = fb.z; v.s7 = fb.w;
    v.s8 = fc.x; v.s9 = fc.y; v.sa = fc.z; v.sb = fc.w;
    v.sc = fd.x; v.sd = fd.y; v.se = fd.z; v.sf = fd.w;
    return v;
}

// The staged gather in two halves, so that the caller can keep the NEXT tile's box loads in flight (in registers) while it
// decodes the current tile:
//   p3d_stage_plan   boxes of a tile (wave-uniform) + the 15 box loads per lane issued into registers; false = does not fit
//   p3d_stage_commit registers -> this wave's LDS region (the previous tile's taps must have been consumed)
//   p3d_gather_features_boxed  the taps from LDS, folded like p3d_gather_features (same values, same arithmetic)
struct P3dStage {
    P3dBox b[3];
    i32x4 v[15];
};
template <typename RSRC>
P3D_DEV bool p3d_stage_plan(RSRC rs, const P3dPlaneGeom& g, const P3dDecodeCfg& cfg, float px, float py, float pz, P3dStage& st) {
    const float qx = px * cfg.coord_scale, qy = py * cfg.coord_scale, qz = pz * cfg.coord_scale;
    const float g2x = cfg.plane_mode ? qy : qz, g2y = cfg.plane_mode ? qz : qx;
    int x0, y0;
    bool inr, fits;
    p3d_plane_texel(g, qx, qy, x0, y0, inr);
    fits = p3d_make_box(x0, y0, inr, st.b[0]);
    p3d_plane_texel(g, qx, qz, x0, y0, inr);
    fits = p3d_make_box(x0, y0, inr, st.b[1]) && fits;
    p3d_plane_texel(g, g2x, g2y, x0, y0, inr);
    fits = p3d_make_box(x0, y0, inr, st.b[2]) && fits;
    if (!fits) return false;
    p3d_box_load(rs, g, 0u, st.b[0], st.v);
    p3d_box_load(rs, g, g.plane_bytes, st.b[1], st.v + 5);
    p3d_box_load(rs, g, 2u * g.plane_bytes, st.b[2], st.v + 10);
    return true;
}
P3D_DEV void p3d_stage_commit(float* box, const P3dStage& st) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the previous tile's tap reads come first
    __builtin_amdgcn_wave_barrier();
    p3d_box_store(box, st.v);
    p3d_box_store(box + P3D_BOX_PLANE_FLOATS, st.v + 5);
    p3d_box_store(box + 2 * P3D_BOX_PLANE_FLOATS, st.v + 10);
    // the box is written and read by different lanes of this wave only: LDS operations of a wave execute in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
P3D_DEV f32x16 p3d_gather_features_boxed(const float* box, const P3dBox b[3], const P3dPlaneGeom& g, const P3dDecodeCfg& cfg,
                                         float px, float py, float pz, bool live) {
    const int h = __lane_id() >> 5;
    const uint32_t chan_off = (uint32_t)h * 64u;
    const float qx = px * cfg.coord_scale, qy = py * cfg.coord_scale, qz = pz * cfg.coord_scale;
    const float g2x = cfg.plane_mode ? qy : qz, g2y = cfg.plane_mode ? qz : qx;
    uint32_t of[12];
    float wg[12];
    p3d_tap_offsets_box(g, b[0], 0u, chan_off, qx, qy, of, wg, live);
    p3d_tap_offsets_box(g, b[1], P3D_BOX_PLANE_FLOATS * 4u, chan_off, qx, qz, of + 4, wg + 4, live);
    p3d_tap_offsets_box(g, b[2], 2u * P3D_BOX_PLANE_FLOATS * 4u, chan_off, g2x, g2y, of + 8, wg + 8, live);
    return p3d_fold_taps(wg, [&](int k) { return p3d_lds16(box, of[k]); });
}

// masks on raw sigma: renderer.py:138-153,187-198
P3D_DEV float p3d_apply_masks(const P3dDecodeCfg& cfg, float px, float pz, float sigma) {
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
    return sigma;
}

// The decoder on already gathered features X (this lane's 16 channels), then the masks.
// LAZY: colour on demand.  A sample's colour enters the result only through the two interval weights next to it, and a masked
// sample (sigma = -1000 after crop / cull) has exactly-zero weights unless a neighbour's sigma is >= ~794 — which the marcher
// checks on the exact weights and answers with a colour decode after all (k_render's `skipped` guards).  So when NO live lane
// of the wave has an unmasked sigma, layer 2 and the 16 sigmoids are skipped: the function returns false and rgb is not
// written.  (Wave-uniform decision: in an empty or culled region every final-pass step takes this exit.)
template <bool WANT_RGB, bool LAZY = false>
P3D_DEV bool p3d_decode_features(const float* lds, const P3dDecodeCfg& cfg, const f32x16& X, float px, float pz, float& sigma_out,
                                 f32x16& rgb, bool live = true);

template <bool WANT_RGB, bool QUADG = false, bool LAZY = false, typename RSRC>
P3D_DEV bool p3d_decode_wave(const float* lds, RSRC rs, const P3dPlaneGeom& g, const P3dDecodeCfg& cfg, float px,
                             float py, float pz, float& sigma_out, f32x16& rgb, bool live = true) {
    const f32x16 X = p3d_gather_features<QUADG>(rs, g, cfg, px, py, pz, live);
    return p3d_decode_features<WANT_RGB, LAZY>(lds, cfg, X, px, pz, sigma_out, rgb, live);
}

template <bool WANT_RGB, bool LAZY>
P3D_DEV bool p3d_decode_features(const float* lds, const P3dDecodeCfg& cfg, const f32x16& X, float px, float pz, float& sigma_out,
                                 f32x16& rgb, bool live) {
    const int lane = __lane_id();
    const int h = lane >> 5;

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
        acc0[r] = p3d_softplus(acc0[r]);
        acc1[r] = p3d_softplus(acc1[r]);
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
    const float sigma = p3d_apply_masks(cfg, px, pz, sa + p3d_partner(sa));
    sigma_out = sigma;
    if constexpr (LAZY) {
        if (__builtin_amdgcn_ballot_w64(live && sigma != P3D_SIGMA_MASKED) == 0) return false;
    }

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
            float sg = p3d_sigmoid(o[r]);
            rgb[r] = fs ? sg : sg * 1.002f - 0.001f;
        }
    }
    return WANT_RGB;
}

// ---- tolerance-mode decode (P3D_FLAG_FAST_COLOR, final pass only) -----------------------------------------------------
// two f16 terms of 8 values: hi = x truncated to f16 (v_cvt_pkrtz), lo = f16(x - hi)
P3D_DEV void p3d_split_f16x8(const float* x, f16x8& hi, f16x8& lo) {
    uint32_t uh[4], ul[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const f16x2 ph = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
        const float ra = a - (float)ph[0], rb = b - (float)ph[1];
        uh[i] = __builtin_bit_cast(uint32_t, ph);
        ul[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(ra, rb));
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    hi = __builtin_bit_cast(f16x8, (u32x4){uh[0], uh[1], uh[2], uh[3]});
    lo = __builtin_bit_cast(f16x8, (u32x4){ul[0], ul[1], ul[2], ul[3]});
}
// softplus / sigmoid on the hardware transcendentals (v_exp_f32 = 2^x, v_log_f32 = log2, v_rcp_f32; ~1 ulp each)
P3D_DEV float p3d_softplus_hw(float x) {
    const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(x) * P3D_LOG2E);
    return p3d_fma(__builtin_amdgcn_logf(1.0f + e), 0x1.62e430p-1f, __builtin_fmaxf(x, 0.0f));
}
P3D_DEV float p3d_sigmoid_hw(float x) {
    const float e = __builtin_amdgcn_exp2f(-x * P3D_LOG2E);  // +inf for x < -88: 1 / inf = 0
    return __builtin_amdgcn_rcpf(1.0f + e);
}

// Same interface and lane layout as p3d_decode_wave<true>; lds must also hold the f16 images (p3d_load_mlp_f16_to_lds).
// Domain: |interpolated feature| and hidden activations below the f16 range (65504); results agree with the exact decode to
// ~1e-6 (sigma, relative to the magnitude of the sum's terms) / ~3e-7 (colours).
// GUARD (k_render's final pass; needs the fp32 layer-1 image W0A in LDS, which the exact coarse pass of the same kernel uses):
// the cull / binarize masks are THRESHOLD decisions, the one place where a 1e-6 difference in sigma can move an output by a whole
// interval weight.  A sample whose opacity comes out within P3D_FAST_MASK_BAND of the threshold has its density re-decoded on the
// exact contract (layer 1 on f32 MFMAs + the sigma row, the coarse pass's decoder: ~1/3 of a colour decode, on the features
// already gathered) and takes the exact sigma and mask — at wave level, for the rare wave-step that holds such a sample
// (surface scene: 0.4 % of the final steps).  With it the tolerance mode makes the SAME mask decisions as the exact contract
// (as long as the tolerance decoder's own error in the opacity stays below the band: |sigma error| < 8e-3, i.e. sigma-row
// products up to ~10^4), and its outputs differ from the exact ones by arithmetic round-off only: a stated, asserted bound
//  instead of "a few rays flip".
#define P3D_FAST_MASK_BAND 2e-3f
template <bool WANT_RGB, bool LAZY = false, bool GUARD = false>
P3D_DEV bool p3d_decode_features_fast(const float* lds, const P3dDecodeCfg& cfg, const f32x16& X, float px, float pz, float& sigma_out,
                                      f32x16& rgb, bool live = true);

template <bool WANT_RGB = true, bool QUADG = false, bool LAZY = false, bool GUARD = false, typename RSRC>
P3D_DEV bool p3d_decode_wave_fast(const float* lds, RSRC rs, const P3dPlaneGeom& g, const P3dDecodeCfg& cfg, float px,
                                  float py, float pz, float& sigma_out, f32x16& rgb, bool live = true) {
    const f32x16 X = p3d_gather_features<QUADG>(rs, g, cfg, px, py, pz, live);
    return p3d_decode_features_fast<WANT_RGB, LAZY, GUARD>(lds, cfg, X, px, pz, sigma_out, rgb, live);
}

template <bool WANT_RGB, bool LAZY, bool GUARD>
P3D_DEV bool p3d_decode_features_fast(const float* lds, const P3dDecodeCfg& cfg, const f32x16& X, float px, float pz, float& sigma_out,
                                      f32x16& rgb, bool live) {
    const int lane = __lane_id();
    const int h = lane >> 5;
    float xs[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) xs[c] = X[c];
    f16x8 xh0, xl0, xh1, xl1;
    p3d_split_f16x8(xs, xh0, xl0);
    p3d_split_f16x8(xs + 8, xh1, xl1);
    // ---- layer 1: acc[t] = b0 + W0 X, K = 32 as two chunks of 16 (lane (j,h) supplies channels 16h + 8q + i)
    const f32x4* b0p = (const f32x4*)(lds + P3D_LDS_B0P + h * 32);
    f32x16 acc0, acc1;
    {
        f32x4 q0 = b0p[0], q1 = b0p[1], q2 = b0p[2], q3 = b0p[3];
        f32x4 r0 = b0p[4], r1 = b0p[5], r2 = b0p[6], r3 = b0p[7];
        acc0 = (f32x16){q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        acc1 = (f32x16){r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
    }
    const f16x8* w0h = (const f16x8*)(lds + P3D_LDS_W0H) + lane;  // [(t*2+q)*2 + hi/lo][64 lanes]
    {
        const f16x8 a00h = w0h[0 * 64], a00l = w0h[1 * 64], a01h = w0h[2 * 64], a01l = w0h[3 * 64];
        const f16x8 a10h = w0h[4 * 64], a10l = w0h[5 * 64], a11h = w0h[6 * 64], a11l = w0h[7 * 64];
        acc0 = P3D_MFMA_H(a00h, xl0, acc0); acc1 = P3D_MFMA_H(a10h, xl0, acc1);
        acc0 = P3D_MFMA_H(a00l, xh0, acc0); acc1 = P3D_MFMA_H(a10l, xh0, acc1);
        acc0 = P3D_MFMA_H(a01h, xl1, acc0); acc1 = P3D_MFMA_H(a11h, xl1, acc1);
        acc0 = P3D_MFMA_H(a01l, xh1, acc0); acc1 = P3D_MFMA_H(a11l, xh1, acc1);
        acc0 = P3D_MFMA_H(a00h, xh0, acc0); acc1 = P3D_MFMA_H(a10h, xh0, acc1);
        acc0 = P3D_MFMA_H(a01h, xh1, acc0); acc1 = P3D_MFMA_H(a11h, xh1, acc1);
    }
    float hs[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // Softplus: triplane.py:524
        hs[r] = p3d_softplus_hw(acc0[r]);
        hs[16 + r] = p3d_softplus_hw(acc1[r]);
    }
    // ---- sigma row on the VALU, fp32 weights and activations (triplane.py:543)
    const f32x4* w1s = (const f32x4*)(lds + P3D_LDS_W1S + h * 32);
    float sa = (h == 0) ? lds[P3D_LDS_B1S] : 0.0f;
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
        f32x4 w = w1s[s4];
#pragma unroll
        for (int e = 0; e < 4; ++e) sa = p3d_fma(w[e], hs[4 * s4 + e], sa);
    }
    float sigma = sa + p3d_partner(sa);
    // ---- masks (hardware transcendentals here too)
    if (cfg.flags & P3D_FLAG_CROP) {
        if (__builtin_fabsf(px) > cfg.crop_limit || __builtin_fabsf(pz) > cfg.crop_limit) sigma = P3D_SIGMA_MASKED;
    }
    if (cfg.flags & (P3D_FLAG_CULL | P3D_FLAG_BINARIZE)) {
        float a = 1.0f - __builtin_amdgcn_exp2f(-p3d_softplus_hw(sigma - 1.0f) * P3D_LOG2E);
        const bool near = live && sigma != P3D_SIGMA_MASKED && __builtin_fabsf(a - cfg.cull_thresh) < P3D_FAST_MASK_BAND;
        if (cfg.flags & P3D_FLAG_BINARIZE)
            sigma = (a < cfg.cull_thresh) ? P3D_SIGMA_MASKED : P3D_SIGMA_SOLID;
        else if (a < cfg.cull_thresh)
            sigma = P3D_SIGMA_MASKED;
        if constexpr (GUARD) {
            if (__builtin_amdgcn_ballot_w64(near) != 0) {  // rare: the exact density (and mask) for the samples on the threshold
                float sx;
                f32x16 unused;
                p3d_decode_features<false>(lds, cfg, X, px, pz, sx, unused);
                sigma = near ? sx : sigma;
            }
        }
    }
    sigma_out = sigma;
    if constexpr (LAZY) {
        if (__builtin_amdgcn_ballot_w64(live && sigma != P3D_SIGMA_MASKED) == 0) return false;
    }
    // ---- layer 2 rows 1..32: K = 64 as four chunks (t, pp): lane (j,h) supplies neurons 32t + rowof(8pp + i) + 4h
    if constexpr (WANT_RGB) {
        const f32x4* b1p = (const f32x4*)(lds + P3D_LDS_B1P + h * 16);
        f32x4 q0 = b1p[0], q1 = b1p[1], q2 = b1p[2], q3 = b1p[3];
        f32x16 o = (f32x16){q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        const f16x8* w1h = (const f16x8*)(lds + P3D_LDS_W1H) + lane;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f16x8 hh, hl;
            p3d_split_f16x8(hs + 8 * c, hh, hl);
            const f16x8 ah = w1h[(2 * c) * 64], al = w1h[(2 * c + 1) * 64];
            o = P3D_MFMA_H(ah, hl, o);
            o = P3D_MFMA_H(al, hh, o);
            o = P3D_MFMA_H(ah, hh, o);
        }
        const bool fs = (cfg.flags & P3D_FLAG_FORCE_SIGMOID) != 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sg = p3d_sigmoid_hw(o[r]);
            rgb[r] = fs ? sg : sg * 1.002f - 0.001f;
        }
    }
    return WANT_RGB;
}
