// p3d_conv_stage.hpp — staging helpers of the register-staged convolution kernels (fp32 operands, f16 / two-term operands on the
// 8 x 16 tile, two-term operands on the 8 x 32 tile, the activation-image DMA of k_modconv_w2): shared by the plain
// (p3d_conv_plain.hip) and the transposed (p3d_conv_up.hip) kernels.
#pragma once
#include "p3d_conv_common.hpp"

// =====================================================================================================================
// The convolution kernels.  The f32 MFMA shares its SIMD with the VALU (tools/ubench/mfma_valu_overlap.hip), so the K loop is
// written to contain ds_reads and MFMAs only:
//   * staging goes through raw buffer loads: a per-thread byte offset computed ONCE (0x80000000 = padding / out of range ->
//     the hardware returns 0, no exec-mask branches), the K-chunk advance lives in the SCALAR base of the buffer resource and
//     the channel tail in its num_records; weights are fetched along the contiguous k axis (thread = output channel x k
//     quarter), so global and LDS addresses are affine in the unrolled index (instruction immediates);
//   * k pairs of one MFMA are (channel 2c, tap t) on lanes 0-31 and (channel 2c+1, same tap) on lanes 32-63: both LDS operand
//     addresses become lane base + immediate;
//   * LDS is double buffered: the next chunk is stored while the other buffer is read -> ONE barrier per chunk; with the plan
//     registers gone three workgroups fit a CU (k_modconv) / two instead of one (k_modconv_up).
// =====================================================================================================================

struct ConvStagePlan {
    int xoff[6];  // byte offset of staged patch value u inside the chunk-relative image slice (CONV_OOB = zero)
    int soff[6];  // byte offset of its style inside the chunk-relative style slice
    int woff;     // byte offset of this thread's first weight inside the chunk-relative weight tensor
};

template <int NT>
DEV ConvStagePlan conv_plan(const ConvParams& p, int tid, int gy0, int gx0, int o0) {
    ConvStagePlan s;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int idx = tid + u * 256;
        const int ic = idx / XS_PLANE, rem = idx - ic * XS_PLANE;
        const int r = rem / XS_ROW, c = rem - r * XS_ROW;
        const int iy = gy0 - 1 + r, ix = gx0 - 1 + c;
        const bool ok = idx < 8 * XS_PLANE && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        s.xoff[u] = ok ? ((ic * p.H + iy) * p.W + ix) * 4 : CONV_OOB;
        s.soff[u] = ok ? ic * 4 : CONV_OOB;
    }
    const int wo = tid >> 2, kq = tid & 3;
    s.woff = (o0 + wo < p.O) ? ((o0 + wo) * p.I * NT + kq * (2 * NT)) * 4 : CONV_OOB;
    return s;
}

// registers of one staged chunk (8 input channels): 6 patch values + their styles, 2*NT weights (k = kq*2*NT .. +2*NT-1 of row wo)
template <int NT>
struct ConvStageRegs { float x[6], s[6], w[2 * NT]; };

template <int NT>
DEV void conv_gload(const ConvParams& p, const ConvStagePlan& pl, const float* xn, const float* sn, int ic0, int ic_end,
                    ConvStageRegs<NT>& r) {
    const int HW = p.H * p.W;
    // channels left in this split-K slice; a slice beyond the last channel (I not a multiple of the slice width) has none:
    // every load is then out of range -> zeros -> the workgroup stores a zero partial sum
    const int left = ic_end > ic0 ? ic_end - ic0 : 0;
    auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xn + (size_t)ic0 * HW), 0, left * HW * 4, CONV_RSRC_FLAGS);
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(sn + ic0), 0, left * 4, CONV_RSRC_FLAGS);
    auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (size_t)ic0 * NT), 0, left ? (p.O * p.I - ic0) * NT * 4 : 0,
                                                CONV_RSRC_FLAGS);
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        r.x[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, pl.xoff[u], 0, 0));
        r.s[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, pl.soff[u], 0, 0));
    }
    // (dword loads: __builtin_amdgcn_raw_buffer_load_b64 of this toolchain returns its first dword twice — seen in the ISA)
#pragma unroll
    for (int v = 0; v < 2 * NT; ++v) r.w[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, pl.woff, 4 * v, 0));
}

template <int NT>
DEV void conv_lstore(float* xs, float* ws, int tid, const ConvStageRegs<NT>& r, int klim /* valid k of this chunk */) {
    constexpr int WROW = 65;
#pragma unroll
    for (int u = 0; u < 6; ++u) xs[tid + u * 256] = r.s[u] * r.x[u];
    const int wo = tid >> 2, kq = tid & 3;
    float* wd = ws + (kq * 2 * NT) * WROW + wo;
    if (klim >= 8 * NT) {
#pragma unroll
        for (int v = 0; v < 2 * NT; ++v) wd[v * WROW] = r.w[v];
    } else {  // channel tail (I not a multiple of 8): k beyond the last channel contributes 0
#pragma unroll
        for (int v = 0; v < 2 * NT; ++v) wd[v * WROW] = (kq * 2 * NT + v < klim) ? r.w[v] : 0.0f;
    }
}

#define CONV_XSZ (6 * 256)  // staged patch values per buffer (8 * XS_PLANE = 1440, padded to the 6 x 256 store pattern)

// =====================================================================================================================
// f16-operand variants (opt-in; the reference runs its super-resolution blocks in fp16 on the GPU, superresolution.py:264-293
// with sr_num_fp16_res = 4).  Activations and outputs stay fp32 in HBM, accumulation is fp32; only the two MFMA operands are
// rounded to f16 (RNE) while they are staged: the modulated input s*x per element, the weights once per layer
// (k_weights_to_f16, layout [O][taps][I]).  v_mfma_f32_32x32x16_f16 does 16x the flops of the f32 instruction per cycle, so
// the tile is re-balanced around LDS bandwidth: a K chunk is 16 input channels = ONE MFMA per tap and N tile; a lane's operand
// is 8 consecutive channels = one ds_read_b128.
//   LDS B: [k half][10 rows][32 px][8 ch] f16  (row pitch 32 px: the 16-lane groups of ds_read_b128 then hit 16 distinct 16-B slots)
//   LDS A: [tap][k half][64 o][8 ch] f16       (lanes = consecutive o -> consecutive slots)
// Requires I % 16 == 0 (the host falls back to the f32 kernels otherwise).
// =====================================================================================================================
#define HX_PITCH 32                           // pixels per patch row in LDS
#define HX_HALF ((CONV_TH + 2) * HX_PITCH * 16)  // bytes of one k half of the patch
#define HX_BYTES (2 * HX_HALF)
#define HX_ITEMS (2 * (CONV_TH + 2) * XS_ROW)  // (k half, pixel) items staged per chunk: 360

struct ConvStagePlanH {
    int xoff[2];   // byte offset of the item's pixel inside the chunk-relative image slice of its first channel (CONV_OOB = zero)
    int xdst[2];   // LDS byte offset of the item
    int soff[2];   // byte offset of the item's 8 styles inside the chunk-relative style slice
    int woff[5];   // byte offset of weight piece q inside the chunk-relative f16 weight tensor
};

template <int NT>
DEV ConvStagePlanH conv_plan_h(const ConvParams& p, int tid, int gy0, int gx0, int o0) {
    ConvStagePlanH s;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int it = tid + u * 256;
        const int h = it / ((CONV_TH + 2) * XS_ROW), px = it - h * ((CONV_TH + 2) * XS_ROW);
        const int r = px / XS_ROW, c = px - r * XS_ROW;
        const int iy = gy0 - 1 + r, ix = gx0 - 1 + c;
        const bool item = it < HX_ITEMS;
        const bool ok = item && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        s.xoff[u] = ok ? ((8 * h * p.H + iy) * p.W + ix) * 4 : CONV_OOB;
        s.soff[u] = ok ? 32 * h : CONV_OOB;
        s.xdst[u] = item ? h * HX_HALF + (r * HX_PITCH + c) * 16 : -1;
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int q = tid + u * 256;  // piece = (tap, k half, o): 16 bytes = 8 channels
        const int t = q >> 7, h = (q >> 6) & 1, o = q & 63;
        const bool ok = q < NT * 128 && o0 + o < p.O;
        s.woff[u] = ok ? (((o0 + o) * NT + t) * p.I + 8 * h) * 2 : CONV_OOB;
    }
    return s;
}

// SPLIT (two-term operands, p3d_modconv2d_f16x2mma_f32): every operand is carried as hi + lo, hi = f16(v) (RNE), lo = f16(v - hi),
// and a product is a_hi*b_hi + a_lo*b_hi + a_hi*b_lo with fp32 accumulation: the dropped a_lo*b_lo term and the rounding of lo are
// ~2^-22 relative, i.e. fp32-class results at 3 f16 MFMAs (96 cycles per 16 channels) instead of 8 f32 ones (512 cycles).
// The weight tensor then holds the hi parts followed by the lo parts (k_weights_to_f16 with split = 1); LDS keeps the lo images
// behind the hi ones, and the weights single-buffered (hi + lo of a chunk are 36 KB for 3x3: two workgroups per CU still fit).
template <int NT, bool SPLIT = false>
struct ConvStageRegsH { float x[2][8]; f32x4 s[2][2]; i32x4 w[NT ? (NT * 128 + 255) / 256 : 1]; i32x4 wl[SPLIT ? (NT * 128 + 255) / 256 : 1]; };  // NT = 0: activations only

template <int NT, bool SPLIT = false>
DEV void conv_gload_h(const ConvParams& p, const ConvStagePlanH& pl, const float* xn, const float* sn, int ic0, int ic_end,
                      ConvStageRegsH<NT, SPLIT>& r) {
    const int HW = p.H * p.W;
    // channels left in this split-K slice; a slice beyond the last channel (I not a multiple of the slice width) has none:
    // every load is then out of range -> zeros -> the workgroup stores a zero partial sum
    const int left = ic_end > ic0 ? ic_end - ic0 : 0;
    auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xn + (size_t)ic0 * HW), 0, left * HW * 4, CONV_RSRC_FLAGS);
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(sn + ic0), 0, left * 4, CONV_RSRC_FLAGS);
    auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.wh + (size_t)ic0 * 2), 0,
                                                left ? (p.O * NT * p.I - ic0) * 2 : 0, CONV_RSRC_FLAGS);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            r.x[u][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, pl.xoff[u], i * HW * 4, 0));
        r.s[u][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, pl.soff[u], 0, 0));
        r.s[u][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, pl.soff[u], 16, 0));
    }
#pragma unroll
    for (int u = 0; u < (NT * 128 + 255) / 256; ++u) r.w[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, pl.woff[u], 0, 0);
    if constexpr (SPLIT) {  // the lo parts: a second tensor of the same shape right behind the hi parts
        auto rl = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.wh + ((size_t)p.O * NT * p.I + ic0) * 2), 0,
                                                    left ? (p.O * NT * p.I - ic0) * 2 : 0, CONV_RSRC_FLAGS);
#pragma unroll
        for (int u = 0; u < (NT * 128 + 255) / 256; ++u) r.wl[u] = __builtin_amdgcn_raw_buffer_load_b128(rl, pl.woff[u], 0, 0);
    }
}

template <int NT, bool SPLIT, typename REGS>
DEV void conv_lstore_hx(char* xs, const ConvStagePlanH& pl, const REGS& r, unsigned int* satp = nullptr) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (pl.xdst[u] < 0) continue;
        f16x8 v, l;
        bool sat = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float m = r.s[u][i >> 2][i & 3] * r.x[u][i];
            if constexpr (SPLIT) {
                m *= HX_SPLIT_SCALE_X;
                sat = sat || !(__builtin_fabsf(m) <= 65504.0f);  // beyond the f16 range (or NaN): hi is clamped, lo = 0
                m = __builtin_fminf(__builtin_fmaxf(m, -65504.0f), 65504.0f);
            }
            v[i] = (_Float16)m;  // RNE
            if constexpr (SPLIT) l[i] = (_Float16)(m - (float)v[i]);
        }
        if constexpr (SPLIT) {
            if (sat && satp) atomicOr(satp, 1u);
        }
        *reinterpret_cast<f16x8*>(xs + pl.xdst[u]) = v;
        if constexpr (SPLIT) *reinterpret_cast<f16x8*>(xs + HX_BYTES + pl.xdst[u]) = l;
    }
}
template <int NT, bool SPLIT>
DEV void conv_lstore_hw(char* ws, int tid, const ConvStageRegsH<NT, SPLIT>& r) {
#pragma unroll
    for (int u = 0; u < (NT * 128 + 255) / 256; ++u) {
        const int q = tid + u * 256;
        if (q < NT * 128) {
            *reinterpret_cast<i32x4*>(ws + q * 16) = r.w[u];
            if constexpr (SPLIT) *reinterpret_cast<i32x4*>(ws + NT * 128 * 16 + q * 16) = r.wl[u];
        }
    }
}
template <int NT, bool SPLIT = false>
DEV void conv_lstore_h(char* xs, char* ws, int tid, const ConvStagePlanH& pl, const ConvStageRegsH<NT, SPLIT>& r) {
    conv_lstore_hx<NT, SPLIT>(xs, pl, r);
    conv_lstore_hw<NT, SPLIT>(ws, tid, r);
}
// the weight pieces of one chunk (hi and lo) straight from L2 into LDS (buffer_load_dwordx4 ... lds: wave-uniform LDS base +
// lane * 16, which is exactly the [piece] order of the image) — no staging registers; out-of-range pieces arrive as zeros
template <int NT>
DEV void conv_glds_w2(const ConvParams& p, const ConvStagePlanH& pl, char* ws, int tid, int ic0, int ic_end, int which = 2 /* 0 hi, 1 lo, 2 both */) {
    const int left = ic_end > ic0 ? ic_end - ic0 : 0;
    auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.wh + (size_t)ic0 * 2), 0,
                                                left ? (p.O * NT * p.I - ic0) * 2 : 0, CONV_RSRC_FLAGS);
    auto rl = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.wh + ((size_t)p.O * NT * p.I + ic0) * 2), 0,
                                                left ? (p.O * NT * p.I - ic0) * 2 : 0, CONV_RSRC_FLAGS);
    typedef __attribute__((address_space(3))) void* lds_ptr;
#pragma unroll
    for (int u = 0; u < (NT * 128 + 255) / 256; ++u) {
        const int q = tid + u * 256;
        if (q < NT * 128) {
            char* dst = ws + ((tid & ~63) + u * 256) * 16;
            if (which != 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)dst, 16, pl.woff[u], 0, 0, 0);
            if (which != 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (lds_ptr)(dst + NT * 128 * 16), 16, pl.woff[u], 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The two-term convolution on a WIDE tile (3x3, maps of 32 columns and more): 64 output channels x 8 rows x 32 columns per
// workgroup, a wave = 64 channels x 2 rows x 32 columns = 2 x 2 MFMA tiles (a B tile = one row).  Measured on the 8 x 16 tile above (256 -> 256
// channels at 256^2): 0.39 ms, of which 0.10 ms weight staging, 0.09 ms activation staging and 0.20 ms the MFMA loop itself —
// one ds_read_b128 per MFMA is the LDS's limit, not the matrix cores'.  Here a tap costs 8 (+2) reads for 12 MFMAs, the
// weights of a chunk are staged once for twice the MFMAs, and both halves of the (single-buffered) weight image are re-loaded
// UNDER MFMAs:   phase 1 = a_hi x (b_hi, b_lo)   | barrier | a_hi(next) -> LDS under phase 2 = a_lo x b_hi | barrier |
//                a_lo(next) -> LDS under the next chunk's phase 1.
//   LDS B: [hi | lo][buffer][k half][10 rows][34 px][8 ch] f16 = 2 x 2 x 10 880 B;  LDS A: [hi | lo][tap][k half][64 o][8 ch] = 36 864 B
//   -> 80 384 B per workgroup, two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------------
#define WX_HALF ((CONV_TH + 2) * WX_ROW * 16)      // bytes of one k half
#define WX_BYTES (2 * WX_HALF)                     // one (hi or lo) patch image: 10 880
#define WX_ITEMS (2 * (CONV_TH + 2) * WX_ROW)      // (k half, pixel) items per chunk: 680
#define WX_ROUNDS ((WX_ITEMS + 255) / 256)         // 3

struct ConvStagePlanW {
    int xoff[WX_ROUNDS], xdst[WX_ROUNDS], soff[WX_ROUNDS];
    int woff[5];
};
DEV ConvStagePlanW conv_plan_w(const ConvParams& p, int tid, int gy0, int gx0, int o0) {
    ConvStagePlanW s;
#pragma unroll
    for (int u = 0; u < WX_ROUNDS; ++u) {
        const int it = tid + u * 256;
        const int h = it / ((CONV_TH + 2) * WX_ROW), px = it - h * ((CONV_TH + 2) * WX_ROW);
        const int r = px / WX_ROW, c = px - r * WX_ROW;
        const int iy = gy0 - 1 + r, ix = gx0 - 1 + c;
        const bool item = it < WX_ITEMS;
        const bool ok = item && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        s.xoff[u] = ok ? ((8 * h * p.H + iy) * p.W + ix) * 4 : CONV_OOB;
        s.soff[u] = ok ? 32 * h : CONV_OOB;
        s.xdst[u] = item ? h * WX_HALF + (r * WX_ROW + c) * 16 : -1;
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int q = tid + u * 256;  // piece = (tap, k half, o): 16 bytes = 8 channels
        const int t = q >> 7, h = (q >> 6) & 1, o = q & 63;
        const bool ok = q < 9 * 128 && o0 + o < p.O;
        s.woff[u] = ok ? (((o0 + o) * 9 + t) * p.I + 8 * h) * 2 : CONV_OOB;
    }
    return s;
}
struct ConvStageRegsW { float x[WX_ROUNDS][8]; f32x4 s[WX_ROUNDS][2]; };
DEV void conv_gload_w(const ConvParams& p, const ConvStagePlanW& pl, const float* xn, const float* sn, int ic0, int ic_end, ConvStageRegsW& r) {
    const int HW = p.H * p.W;
    const int left = ic_end > ic0 ? ic_end - ic0 : 0;
    auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xn + (size_t)ic0 * HW), 0, left * HW * 4, CONV_RSRC_FLAGS);
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(sn + ic0), 0, left * 4, CONV_RSRC_FLAGS);
#pragma unroll
    for (int u = 0; u < WX_ROUNDS; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            r.x[u][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, pl.xoff[u], i * HW * 4, 0));
        r.s[u][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, pl.soff[u], 0, 0));
        r.s[u][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, pl.soff[u], 16, 0));
    }
}
// hi image at xs, lo image at xs + 2 * WX_BYTES (the two buffers of one kind are adjacent)
DEV void conv_lstore_w(char* xs, const ConvStagePlanW& pl, const ConvStageRegsW& r, unsigned int* satp) {
#pragma unroll
    for (int u = 0; u < WX_ROUNDS; ++u) {
        if (pl.xdst[u] < 0) continue;
        f16x8 v, l;
        bool sat = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float m = r.s[u][i >> 2][i & 3] * r.x[u][i] * HX_SPLIT_SCALE_X;
            sat = sat || !(__builtin_fabsf(m) <= 65504.0f);
            m = __builtin_fminf(__builtin_fmaxf(m, -65504.0f), 65504.0f);
            v[i] = (_Float16)m;
            l[i] = (_Float16)(m - (float)v[i]);
        }
        *reinterpret_cast<f16x8*>(xs + pl.xdst[u]) = v;
        *reinterpret_cast<f16x8*>(xs + 2 * WX_BYTES + pl.xdst[u]) = l;
        if (sat && satp) atomicOr(satp, 1u);
    }
}
// one half (hi: which = 0, lo: which = 1) of a chunk's weight image, L2 -> LDS
DEV void conv_glds_wh(const ConvParams& p, const ConvStagePlanW& pl, char* ws, int tid, int ic0, int ic_end, int which) {
    const int left = ic_end > ic0 ? ic_end - ic0 : 0;
    auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.wh + ((size_t)which * p.O * 9 * p.I + ic0) * 2), 0,
                                                left ? (p.O * 9 * p.I - ic0) * 2 : 0, CONV_RSRC_FLAGS);
    typedef __attribute__((address_space(3))) void* lds_ptr;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int q = tid + u * 256;
        if (q < 9 * 128)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(ws + which * (9 * 128 * 16) + ((tid & ~63) + u * 256) * 16), 16, pl.woff[u], 0, 0, 0);
    }
}

// =====================================================================================================================
// The activation IMAGE (round 3).  Between an up-sampling layer (conv0) and the plain 3x3 layer that follows it
// (conv1) the activation travels as what the two-term MFMA kernel consumes: per (sample, group of 8 channels, pixel) one 16-byte
// piece of f16 hi parts and one of lo parts of 16 * s[n][c] * x — the CONSUMER's modulation, applied by the producer
// (k_fir4x4_img: the FIR + bias_act pass that ends conv0) — laid out [hi | lo][N][C/8][H][W][8], 4 bytes per value like the
// fp32 tensor it replaces.  The values are exactly the ones k_modconv_w2 computes when it stages an fp32 tensor (the same
// multiply, scale, clamp, split), so results are bit-identical; what goes away is the work: the consumer stages a K chunk's patch
// with buffer_load ... lds only (the patch's LDS order (k half, row, column) IS ascending item order, so every wave writes 64
// consecutive pieces; padding / channel tail arrive as zeros through the buffer's range check): no staging registers, no
// conversion VALU, and the O/64 channel-tile workgroups no longer each repeat the fp32 -> hi/lo split of the same patch.
// Measured (profiles/history/r03_notes.txt): k_modconv_w2 -4 % .. -16 % per layer, the image-writing FIR pass +2 .. +5 us.
// Variants built on the way and dropped: an UNMODULATED image for every consumer (3x3, transposed 3x3, ToRGB) with the modulation
// on per-sample weights — the weight preparation (30 us per backbone pass, x N) and the slower ToRGB ate the convolutions' gain.
// =====================================================================================================================
DEV void conv_glds_ximg(const ConvParams& p, char* xs_hi, char* xs_lo, const int (&xoff)[WX_ROUNDS], int tid, int n, int ic0, int ic_end) {
    const int HW = p.H * p.W;
    const int left = ic_end > ic0 ? ic_end - ic0 : 0;
    const char* base = (const char*)p.ximg + ((size_t)n * (p.I >> 3) + (ic0 >> 3)) * HW * 16;
    auto rh = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (left >> 3) * HW * 16, CONV_RSRC_FLAGS);
    auto rl = __builtin_amdgcn_make_buffer_rsrc((void*)(base + p.ximg_lo), 0, (left >> 3) * HW * 16, CONV_RSRC_FLAGS);
    typedef __attribute__((address_space(3))) void* lds_ptr;
#pragma unroll
    for (int u = 0; u < WX_ROUNDS; ++u) {
        if (tid + u * 256 < WX_ITEMS) {
            const int slot = ((tid & ~63) + u * 256) * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr)(xs_hi + slot), 16, xoff[u], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (lds_ptr)(xs_lo + slot), 16, xoff[u], 0, 0, 0);
        }
    }
}
