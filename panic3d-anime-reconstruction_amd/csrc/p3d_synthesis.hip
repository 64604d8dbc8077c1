// p3d_synthesis.hip — StyleGAN2 synthesis operators of the triplane backbone on gfx950 (MI355X).
//
//   k_modconv<MODE>   modulated convolution as an implicit GEMM on the matrix cores (v_mfma_f32_32x32x2_f32; exact f32):
//                     D[o][pixel] = sum_k W[o][k] * (s[n,i(k)] * x[n,i(k),y+dy(k),x+dx(k)])
//                     A = weights (LDS tile [k][64 o]), B = modulated input patch (LDS tile [8 ic][10][18] with halo),
//                     4 waves = 2 (32-channel halves) x 2 (64-pixel halves), 2 accumulators per wave; K loop = ds_reads + MFMAs
//                     only, double-buffered LDS (see the comment above the kernels).
//                     MODE 0: 3x3 / pad 1 correlation   (conv1 of every block, networks_stylegan2.py:93 -> conv2d_resample.py:136)
//                     MODE 1: 1x1                       (ToRGB, networks_stylegan2.py:378)
//   k_modconv_up      the stride-2 transposed 3x3 conv of the up-sampling layer (conv0, conv2d_resample.py:114-127): the four
//                     output phases in one workgroup; only the taps that meet non-zero inputs are multiplied (4/2/2/1 of 9),
//                     i.e. no zero-insertion.
//                     The per-sample weights w*s*d of the reference's fused path (networks_stylegan2.py:68-73) are refactored
//                     into shared weights, input scaling by s and output scaling by d (its own non-fused path, :76-85).
//   k_demod           d[n,o] = rsqrt(sum_{i,t} (w[o,i,t] s[n,i])^2 + 1e-8)              (networks_stylegan2.py:70-71)
//   k_upfirdn2d       zero-insert x up, pad/crop, FIR, with an optional fused epilogue d*v + noise -> +bias -> act*gain -> clamp
//                     (upfirdn2d.py:169-213 _upfirdn2d_ref; bias_act.py:93-122 _bias_act_ref)
//   k_bias_act        clamp(act(x + b) * gain)                                            (bias_act.py:93-122)
#include "p3d_conv_common.hpp"

static bool env_no_w3();  // (defined with the other read-once environment switches, above up3_applies)
static int w3_min_w();
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

template <int MODE>
__global__ __launch_bounds__(256, 3) void k_modconv(ConvParams p) {
    using T = ConvTaps<MODE>;
    constexpr int NT = T::N, KC = 8 * NT, NB = CONV_TH / 4, WROW = 65;
    __shared__ float xs[2][CONV_XSZ];
    __shared__ float ws[2][KC * WROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wp = wave >> 1, half = lane >> 5, j = lane & 31;
    const int tiles_x = (p.GW + CONV_TW - 1) / CONV_TW;
    const int gy0 = (blockIdx.x / tiles_x) * CONV_TH, gx0 = (blockIdx.x % tiles_x) * CONV_TW;
    const int o0 = blockIdx.y * 64;
    const int n = blockIdx.z / p.ksplit, kz = blockIdx.z - n * p.ksplit;
    const int ic_per = ((p.I + p.ksplit - 1) / p.ksplit + 31) / 32 * 32;
    const int ic_beg = kz * ic_per, ic_end = (ic_beg + ic_per < p.I) ? ic_beg + ic_per : p.I;
    const float* xn = p.x + (size_t)n * p.I * p.H * p.W;
    const float* sn = p.styles + (size_t)n * p.I;

    f32x16 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int prow0 = (CONV_TH / 2) * wp + (j >> 4), pcol = j & 15;
    // lane bases of the two LDS operands: this lane's pixel (+ the halo origin) and its k half
    const int xlane = (prow0 + 1) * XS_ROW + pcol + 1 + half * XS_PLANE;
    const int wlane = wc * 32 + j + half * NT * WROW;

    const ConvStagePlan pl = conv_plan<NT>(p, tid, gy0, gx0, o0);
    ConvStageRegs<NT> rg;
    conv_gload<NT>(p, pl, xn, sn, ic_beg, ic_end, rg);
    conv_lstore<NT>(xs[0], ws[0], tid, rg, (ic_end - ic_beg) * NT);
    __syncthreads();
    int buf = 0;
    for (int ic0 = ic_beg; ic0 < ic_end; ic0 += 8) {
        const bool more = ic0 + 8 < ic_end;
        if (more) conv_gload<NT>(p, pl, xn, sn, ic0 + 8, ic_end, rg);  // in flight during this chunk's MFMAs
        const float* xb = xs[buf] + xlane;
        const float* wb = ws[buf] + wlane;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float av = wb[((2 * c) * NT + t) * WROW];
                float bv[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) bv[b] = xb[(2 * c) * XS_PLANE + T::dy[t] * XS_ROW + T::dx[t] + 2 * b * XS_ROW];
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[b], acc[b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the operand reads at most one channel pair ahead of their MFMAs
        }
        // the stores wait for the prefetched chunk: they must stay BEHIND the MFMAs (the scheduler would hoist them, and
        // their vmcnt waits, to the top of the MFMA phase)
        __builtin_amdgcn_sched_barrier(0);
        if (more) conv_lstore<NT>(xs[buf ^ 1], ws[buf ^ 1], tid, rg, (ic_end - ic0 - 8) * NT);
        __syncthreads();
        buf ^= 1;
    }
    // ---- epilogue (ksplit > 1: raw partial sums into slice kz of the partial buffer)
    float* yout = p.y + (p.ksplit > 1 ? (size_t)kz * p.N * p.O * p.OH * p.OW : 0);
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const int gy = gy0 + prow0 + 2 * t, gx = gx0 + pcol;
        if (gy >= p.GH || gx >= p.GW) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = o0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (ch >= p.O) continue;
            float v = acc[t][r];
            if (p.epilogue) {
                if (p.dcoef) v = v * p.dcoef[(size_t)n * p.O + ch];
                if (p.noise) v = v + p.noise[(p.noise_per_sample ? (size_t)n * p.OH * p.OW : 0) + (size_t)gy * p.OW + gx];
                if (p.bias) v = v + p.bias[ch];
                v = act_apply(v, p.act, p.alpha, p.gain, p.clamp);
            }
            yout[(((size_t)n * p.O + ch) * p.OH + gy) * p.OW + gx] = v;
        }
    }
}

// Stride-2 transposed 3x3 convolution with ALL FOUR output phases in one workgroup (conv0 of every block,
// conv2d_resample.py:114-127).  T[o][2y+py][2x+px] = sum_i sum_{ky == py, kx == px (mod 2)} w[o][i][ky][kx] * x[i][y - ky/2][x - kx/2]:
// the four phases read the same four input values x[y][x], x[y][x-1], x[y-1][x], x[y-1][x-1] with disjoint subsets of the 9 taps
// (4 / 2 / 2 / 1).  One staging round (8 input channels: the 10x18 input patch and the [72][64] weight slice, exactly the
// MODE 0 tiles) feeds 9 MFMAs per input-channel pair and N tile instead of 4 / 2 / 2 / 1 in four separate launches.
// Grid positions: (H+1) x (W+1); 8 accumulators per wave (4 phases x 2 N tiles of 32 positions).  k pairs = two input channels.
__global__ __launch_bounds__(256, 2) void k_modconv_up(ConvParams p) {
    constexpr int NT = 9, KC = 72, WROW = 65;
    __shared__ float xs[2][CONV_XSZ];
    __shared__ float ws[2][KC * WROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wp = wave >> 1, half = lane >> 5, j = lane & 31;
    const int tiles_x = (p.GW + CONV_TW - 1) / CONV_TW;
    const int gy0 = (blockIdx.x / tiles_x) * CONV_TH, gx0 = (blockIdx.x % tiles_x) * CONV_TW;
    const int o0 = blockIdx.y * 64;
    const int n = blockIdx.z / p.ksplit, kz = blockIdx.z - n * p.ksplit;
    const int ic_per = ((p.I + p.ksplit - 1) / p.ksplit + 31) / 32 * 32;
    const int ic_beg = kz * ic_per, ic_end = (ic_beg + ic_per < p.I) ? ic_beg + ic_per : p.I;
    const float* xn = p.x + (size_t)n * p.I * p.H * p.W;
    const float* sn = p.styles + (size_t)n * p.I;

    f32x16 acc[4][2];  // [phase = 2*py + px][N tile]
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ph][t][r] = 0.0f;
    const int prow0 = 4 * wp + (j >> 4), pcol = j & 15;
    const int xlane = (prow0 + 1) * XS_ROW + pcol + 1 + half * XS_PLANE;
    const int wlane = wc * 32 + j + half * NT * WROW;

    const ConvStagePlan pl = conv_plan<NT>(p, tid, gy0, gx0, o0);
    ConvStageRegs<NT> rg;
    conv_gload<NT>(p, pl, xn, sn, ic_beg, ic_end, rg);
    conv_lstore<NT>(xs[0], ws[0], tid, rg, (ic_end - ic_beg) * NT);
    __syncthreads();
    int buf = 0;
    for (int ic0 = ic_beg; ic0 < ic_end; ic0 += 8) {
        const bool more = ic0 + 8 < ic_end;
        if (more) conv_gload<NT>(p, pl, xn, sn, ic0 + 8, ic_end, rg);
        const float* xb = xs[buf] + xlane;
        const float* wb = ws[buf] + wlane;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* xp = xb + (2 * c) * XS_PLANE;
            const float* wr = wb + (2 * c) * NT * WROW;
            // the four input values per N tile: [dy][dx] with dy, dx in {0, -1}; N tile 1 is two rows below
            float b00[2], b01[2], b10[2], b11[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                b00[t] = xp[2 * t * XS_ROW]; b01[t] = xp[2 * t * XS_ROW - 1];
                b10[t] = xp[2 * t * XS_ROW - XS_ROW]; b11[t] = xp[2 * t * XS_ROW - XS_ROW - 1];
            }
            float a[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) a[t] = wr[t * WROW];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b00[t], acc[0][t], 0, 0, 0);  // phase (0,0): taps (0,0) (0,2) (2,0) (2,2)
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b00[t], acc[1][t], 0, 0, 0);  // phase (0,1): taps (0,1) (2,1)
                acc[2][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b00[t], acc[2][t], 0, 0, 0);  // phase (1,0): taps (1,0) (1,2)
                acc[3][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4], b00[t], acc[3][t], 0, 0, 0);  // phase (1,1): tap (1,1)
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b01[t], acc[0][t], 0, 0, 0);
                acc[2][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[5], b01[t], acc[2][t], 0, 0, 0);
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[6], b10[t], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[7], b10[t], acc[1][t], 0, 0, 0);
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8], b11[t], acc[0][t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the stores wait for the prefetched chunk: they must stay BEHIND the MFMAs (the scheduler would hoist them, and
        // their vmcnt waits, to the top of the MFMA phase)
        __builtin_amdgcn_sched_barrier(0);
        if (more) conv_lstore<NT>(xs[buf ^ 1], ws[buf ^ 1], tid, rg, (ic_end - ic0 - 8) * NT);
        __syncthreads();
        buf ^= 1;
    }
    // ---- raw store of the four phases (ksplit > 1: into slice kz of the partial buffer); the FIR pass applies the epilogue
    float* yout = p.y + (p.ksplit > 1 ? (size_t)kz * p.N * p.O * p.OH * p.OW : 0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int gy = gy0 + prow0 + 2 * t, gx = gx0 + pcol;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int py = ph >> 1, px = ph & 1;
            if (gy > p.H - py || gx > p.W - px) continue;
            const int oy = 2 * gy + py, ox = 2 * gx + px;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = o0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (ch < p.O) yout[(((size_t)n * p.O + ch) * p.OH + oy) * p.OW + ox + p.tox] = acc[ph][t][r];
            }
        }
    }
}

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

template <int MODE, bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_modconv_h(ConvParams p) {
    using T = ConvTaps<MODE>;
    constexpr int NT = T::N, NB = CONV_TH / 4, WBYTES = NT * 128 * 16;
    __shared__ __attribute__((aligned(16))) char xs[2][SPLIT ? 2 * HX_BYTES : HX_BYTES];
    __shared__ __attribute__((aligned(16))) char ws[SPLIT ? 1 : 2][SPLIT ? 2 * WBYTES : WBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wp = wave >> 1, half = lane >> 5, j = lane & 31;
    const int tiles_x = (p.GW + CONV_TW - 1) / CONV_TW;
    const int gy0 = (blockIdx.x / tiles_x) * CONV_TH, gx0 = (blockIdx.x % tiles_x) * CONV_TW;
    const int o0 = blockIdx.y * 64;
    const int n = blockIdx.z / p.ksplit, kz = blockIdx.z - n * p.ksplit;
    const int ic_per = ((p.I + p.ksplit - 1) / p.ksplit + 31) / 32 * 32;
    const int ic_beg = kz * ic_per, ic_end = (ic_beg + ic_per < p.I) ? ic_beg + ic_per : p.I;
    const float* xn = p.x + (size_t)n * p.I * p.H * p.W;
    const float* sn = p.styles + (size_t)n * p.I;

    f32x16 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int prow0 = (CONV_TH / 2) * wp + (j >> 4), pcol = j & 15;
    const int xlane = half * HX_HALF + ((prow0 + 1) * HX_PITCH + pcol + 1) * 16;  // bytes
    const int wlane = (half * 64 + wc * 32 + j) * 16;

    const ConvStagePlanH pl = conv_plan_h<NT>(p, tid, gy0, gx0, o0);
    // SPLIT: only the activations go through registers (the fp32 -> hi / lo conversion); the weights are copied L2 -> LDS
    ConvStageRegsH<SPLIT ? 0 : NT, false> rg;
    if constexpr (SPLIT) {
        conv_gload_h<0, false>(p, pl, xn, sn, ic_beg, ic_end, rg);
        conv_glds_w2<NT>(p, pl, ws[0], tid, ic_beg, ic_end);
        conv_lstore_hx<0, true>(xs[0], pl, rg, p.sat);
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the LDS-direct loads have landed
    } else {
        conv_gload_h<NT>(p, pl, xn, sn, ic_beg, ic_end, rg);
        conv_lstore_h<NT>(xs[0], ws[0], tid, pl, rg);
    }
    __syncthreads();
    int buf = 0;
    for (int ic0 = ic_beg; ic0 < ic_end; ic0 += 16) {
        const bool more = ic0 + 16 < ic_end;
        if (more) conv_gload_h<SPLIT ? 0 : NT, false>(p, pl, xn, sn, ic0 + 16, ic_end, rg);
        const char* xb = xs[buf] + xlane;
        const char* wb = ws[SPLIT ? 0 : buf] + wlane;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f16x8 av = *reinterpret_cast<const f16x8*>(wb + t * 128 * 16);
            f16x8 al;
            if constexpr (SPLIT) al = *reinterpret_cast<const f16x8*>(wb + WBYTES + t * 128 * 16);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int off = ((T::dy[t] + 2 * b) * HX_PITCH + T::dx[t]) * 16;
                const f16x8 bv = *reinterpret_cast<const f16x8*>(xb + off);
                if constexpr (SPLIT) {
                    const f16x8 bl = *reinterpret_cast<const f16x8*>(xb + HX_BYTES + off);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bv, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bl, acc[b], 0, 0, 0);
                }
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[b], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // the stores (and their vmcnt waits) stay behind the MFMAs
        if constexpr (SPLIT) {  // single-buffered weights: everybody has to be done with them first
            if (more) conv_lstore_hx<0, true>(xs[buf ^ 1], pl, rg, p.sat);
            __syncthreads();
            if (more) conv_glds_w2<NT>(p, pl, ws[0], tid, ic0 + 16, ic_end);
            __builtin_amdgcn_s_waitcnt(0);
        } else {
            if (more) conv_lstore_h<NT>(xs[buf ^ 1], ws[buf ^ 1], tid, pl, rg);
        }
        __syncthreads();
        buf ^= 1;
    }
    float* yout = p.y + (p.ksplit > 1 ? (size_t)kz * p.N * p.O * p.OH * p.OW : 0);
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const int gy = gy0 + prow0 + 2 * t, gx = gx0 + pcol;
        if (gy >= p.GH || gx >= p.GW) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = o0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (ch >= p.O) continue;
            float v = acc[t][r];
            if constexpr (SPLIT) v *= HX_SPLIT_UNSCALE;
            if (p.epilogue) {
                if (p.dcoef) v = v * p.dcoef[(size_t)n * p.O + ch];
                if (p.noise) v = v + p.noise[(p.noise_per_sample ? (size_t)n * p.OH * p.OW : 0) + (size_t)gy * p.OW + gx];
                if (p.bias) v = v + p.bias[ch];
                v = act_apply(v, p.act, p.alpha, p.gain, p.clamp);
            }
            yout[(((size_t)n * p.O + ch) * p.OH + gy) * p.OW + gx] = v;
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

template <bool IMG>
__global__ __launch_bounds__(256, 2) void k_modconv_w2(ConvParams p) {
    using T = ConvTaps<0>;
    constexpr int WBYTES = 9 * 128 * 16;
    __shared__ __attribute__((aligned(16))) char xs[2 /*hi, lo*/][2 /*buffer*/][WX_BYTES];
    __shared__ __attribute__((aligned(16))) char ws[2 * WBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int tiles_x = (p.GW + WX_TW - 1) / WX_TW;
    const int gy0 = (blockIdx.x / tiles_x) * CONV_TH, gx0 = (blockIdx.x % tiles_x) * WX_TW;
    const int o0 = blockIdx.y * 64;
    const int n = blockIdx.z / p.ksplit, kz = blockIdx.z - n * p.ksplit;
    const int ic_per = ((p.I + p.ksplit - 1) / p.ksplit + 31) / 32 * 32;
    const int ic_beg = kz * ic_per, ic_end = (ic_beg + ic_per < p.I) ? ic_beg + ic_per : p.I;
    const float* xn = p.x + (size_t)n * p.I * p.H * p.W;
    const float* sn = p.styles + (size_t)n * p.I;

    f32x16 acc[2][2];  // [channel tile][row of the wave's row pair]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    // a B tile is ONE row of 32 columns (lane j = column j): ds_read_b128 serves lanes {0-3, 12-15, 20-27} together, and with
    // 2 rows x 16 columns per tile the 34-pixel row pitch put lanes 20-27 on the slots of lanes 12-13 (35 % conflict cycles)
    const int prow = 2 * wave, pcol = j;
    const int xlane = half * WX_HALF + ((prow + 1) * WX_ROW + pcol + 1) * 16;  // row b adds one row pitch
    const int wlane = (half * 64 + j) * 16;                                     // channel tile a adds 32 o

    ConvStagePlanW pl = conv_plan_w(p, tid, gy0, gx0, o0);
    if constexpr (IMG) {  // piece offsets inside the chunk-relative image slice
#pragma unroll
        for (int u = 0; u < WX_ROUNDS; ++u) {
            const int it = tid + u * 256;
            const int h = it / ((CONV_TH + 2) * WX_ROW), px = it - h * ((CONV_TH + 2) * WX_ROW);
            const int r = px / WX_ROW, c = px - r * WX_ROW;
            const int iy = gy0 - 1 + r, ix = gx0 - 1 + c;
            const bool ok = it < WX_ITEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            pl.xoff[u] = ok ? ((h * p.H + iy) * p.W + ix) * 16 : CONV_OOB;
        }
    }
    ConvStageRegsW rg;
    if constexpr (IMG) conv_glds_ximg(p, xs[0][0], xs[1][0], pl.xoff, tid, n, ic_beg, ic_end);
    else conv_gload_w(p, pl, xn, sn, ic_beg, ic_end, rg);
    conv_glds_wh(p, pl, ws, tid, ic_beg, ic_end, 0);
    conv_glds_wh(p, pl, ws, tid, ic_beg, ic_end, 1);
    if constexpr (!IMG) conv_lstore_w(xs[0][0], pl, rg, p.sat);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    int buf = 0;
    for (int ic0 = ic_beg; ic0 < ic_end; ic0 += 16) {
        const bool more = ic0 + 16 < ic_end;
        if constexpr (IMG) {
            if (more) conv_glds_ximg(p, xs[0][buf ^ 1], xs[1][buf ^ 1], pl.xoff, tid, n, ic0 + 16, ic_end);  // lands under this chunk's MFMAs
        } else {
            if (more) conv_gload_w(p, pl, xn, sn, ic0 + 16, ic_end, rg);
        }
        const char* xh = xs[0][buf] + xlane;
        const char* xl = xs[1][buf] + xlane;
        const char* wb = ws + wlane;
        // phase 1: a_hi x (b_lo, b_hi).  Round 4: the taps run column-major (dx outer, dy inner) and a B tile is a patch ROW — output row
        // b under tap dy reads patch row b + dy, so the wave's two output rows and three dy share FOUR row tiles per dx instead of
        // reading six — and the hi row tiles stay in registers (12 x 4 VGPRs) for phase 2, which then reads weights only:
        // 42 + 18 = 60 ds_read_b128 per wave and chunk instead of 54 + 36 = 90 for the same 108 MFMAs (the LDS port was as busy as the
        // matrix cores: profiles/history/r03_notes.txt).  Same products, another summation order (dx-major).
        f16x8 bh[3][4];
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
            f16x8 bl[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int off = ((r - 1) * WX_ROW + (dxi - 1)) * 16;
                bh[dxi][r] = *reinterpret_cast<const f16x8*>(xh + off);
                bl[r] = *reinterpret_cast<const f16x8*>(xl + off);
            }
#pragma unroll
            for (int dyi = 0; dyi < 3; ++dyi) {
                const int t = dyi * 3 + dxi;
                f16x8 ah[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) ah[a] = *reinterpret_cast<const f16x8*>(wb + t * 128 * 16 + a * 32 * 16);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b + dyi], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[dxi][b + dyi], acc[a][b], 0, 0, 0);
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!IMG) {
            if (more) conv_lstore_w(xs[0][buf ^ 1], pl, rg, p.sat);   // (waits for this chunk's a_lo too: it was requested before phase 1)
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();                                    // a_hi is free, a_lo has landed everywhere
        if (more) conv_glds_wh(p, pl, ws, tid, ic0 + 16, ic_end, 0);
        // phase 2: a_lo x b_hi (the row tiles of phase 1, still in registers)
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
            for (int dyi = 0; dyi < 3; ++dyi) {
                const int t = dyi * 3 + dxi;
                f16x8 al[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) al[a] = *reinterpret_cast<const f16x8*>(wb + WBYTES + t * 128 * 16 + a * 32 * 16);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[dxi][b + dyi], acc[a][b], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();                                    // a_lo and this patch buffer are free, a_hi(next) has landed
        if (more) conv_glds_wh(p, pl, ws, tid, ic0 + 16, ic_end, 1);
        buf ^= 1;
    }
    float* yout = p.y + (p.ksplit > 1 ? (size_t)kz * p.N * p.O * p.OH * p.OW : 0);
    const int gx = gx0 + pcol;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int gy = gy0 + prow + b;
        if (gy >= p.GH || gx >= p.GW) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = o0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (ch >= p.O) continue;
                float v = acc[a][b][r] * HX_SPLIT_UNSCALE;
                if (p.epilogue) {
                    if (p.dcoef) v = v * p.dcoef[(size_t)n * p.O + ch];
                    if (p.noise) v = v + p.noise[(p.noise_per_sample ? (size_t)n * p.OH * p.OW : 0) + (size_t)gy * p.OW + gx];
                    if (p.bias) v = v + p.bias[ch];
                    v = act_apply(v, p.act, p.alpha, p.gain, p.clamp);
                }
                yout[(((size_t)n * p.O + ch) * p.OH + gy) * p.OW + gx] = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_modconv_w3 (round 4): the image-fed plain 3x3 two-term convolution of k_modconv_w2<true> with a REAL software pipeline.
// What the ISA of k_modconv_w2 showed (profiles/r04_notes.txt): the compiler tracks `buffer_load ... lds` as a pending LDS write and
// waits `vmcnt(0)` in front of the FIRST ds_read that follows it — so the next chunk's patch, requested at the top of a chunk "to
// land under this chunk's MFMAs", was waited for before the chunk's first MFMA; a_hi(next) had only the 36 MFMAs of phase 2 to land
// before the `s_waitcnt(0)` of the second barrier; and the epilogue's dcoef / noise / bias loads sat behind uniform branches with a
// `vmcnt(0)` each.  Two exposed L2 round trips per 16-channel chunk: MFMA-busy 0.13-0.42 (profiles/history/r03_mfma_util.json).
// Here (the recipe of cdna_hip_programming.md "Pipelining across barriers"):
//   * every DMA is issued from inline asm (s_mov m0 + buffer_load_dwordx4 ... lds): invisible to the compiler's wait insertion;
//   * counted `s_waitcnt vmcnt(N)` by hand + raw s_barrier: loads stay in flight ACROSS barriers;
//   * the weights of a chunk live in a ring of three column groups (dx = -1, 0, +1: 3 taps x hi|lo = 12 KB each); a chunk = three
//     phases of 36 MFMAs per wave, group g is re-loaded for the next chunk right after phase g and has two phases to land; the
//     patch is double buffered and has a whole chunk;
//   * every wave issues the same number of DMA instructions per chunk (wave w loads the (hi|lo, k half) sub-image w of the patch:
//     5 full + 1 partial instruction; 3 x 3 weight instructions), so the counts are compile-time constants:
//         queue before the barrier after phase 0 / 1:  [W(g+1) 3][W(g+2) 3][patch(next) 6]   -> vmcnt(9)
//         queue before the barrier after phase 2:      [patch(next) 6][W0(next) 3][W1(next) 3] -> vmcnt(3)
//     chunks beyond the slice are "loaded" through a zero-length buffer resource (zeros, no traffic): no tail special cases;
//   * B tiles are patch rows shared by the two output rows and three dy of a column group (8 + 12 reads per 36 MFMAs);
//   * epilogue branch-free: d * 2^-10 and bias of the 64 channels staged in LDS once, stores through a buffer resource.
// LDS (ONE array): weights 3 x 12 288 | patch 2 x [hi|lo][k half][10][34][8] f16 (2 x 21 760) | d, bias 2 x 256 = 80 896 B, two
// workgroups per CU.  Same products as k_modconv_w2, summation order (dx-major) identical to it: bit-identical results.
// Requires O % 64 == 0 (the 3x3 layers of the backbone / super-resolution: 512 .. 64); others take k_modconv_w2<true>.
// ---------------------------------------------------------------------------------------------------------------------
#define W3_GROUP_BYTES (768 * 16)
#define W3_WBYTES (3 * W3_GROUP_BYTES)
#define W3_SUB ((CONV_TH + 2) * WX_ROW * 16)
#define W3_PATCH (4 * W3_SUB)
#define W3_EPI (W3_WBYTES + 2 * W3_PATCH)
#define W3_LDS (W3_EPI + 768)

template <bool RGB>
__global__ __launch_bounds__(256, 2) void k_modconv_w3(ConvParams p) {
    __shared__ __attribute__((aligned(16))) char lds[W3_LDS];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // an SGPR: the LDS destinations of the DMAs (M0) derive from it
    const int tiles_x = (p.GW + WX_TW - 1) / WX_TW;
    const WgOrder wo = p3d_wg_order(p.xcd != 0);
    const int gy0 = (wo.tile / tiles_x) * CONV_TH, gx0 = (wo.tile % tiles_x) * WX_TW;
    const int o0 = wo.otile * 64;
    const int n = wo.z / p.ksplit, kz = wo.z - n * p.ksplit;
    const int ic_per = ((p.I + p.ksplit - 1) / p.ksplit + 31) / 32 * 32;
    const int ic_beg = kz * ic_per, ic_end = (ic_beg + ic_per < p.I) ? ic_beg + ic_per : p.I;
    const int nch = ic_end > ic_beg ? (ic_end - ic_beg) >> 4 : 0;
    const int HW = p.H * p.W;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // ---- epilogue constants into LDS (read after the last barrier of the loop, or after the barrier below when nch == 0)
    float* epi = reinterpret_cast<float*>(lds + W3_EPI);
    if (tid < 64) {
        const int ch = o0 + tid;
        epi[tid] = (p.epilogue && p.dcoef) ? p.dcoef[(size_t)n * p.O + ch] * HX_SPLIT_UNSCALE : HX_SPLIT_UNSCALE;
        epi[64 + tid] = (p.epilogue && p.bias) ? p.bias[ch] : 0.0f;
        epi[128 + tid] = p.yimg ? p.ystyles[(size_t)n * p.O + ch] : 0.0f;
    }
    // RGB: the ToRGB styles and weights of this workgroup's 64 channels, requested here, staged in LDS after the K loop
    float rgb_s = 0.0f, rgb_w[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (RGB) {
        if (tid < 64) {
            rgb_s = p.rgbs[(size_t)n * p.O + o0 + tid];
#pragma unroll
            for (int o = 0; o < 4; ++o) rgb_w[o] = o < p.rgbo ? p.rgbw[(size_t)o * p.O + o0 + tid] : 0.0f;
        }
    }
    // ---- DMA plans.  Patch: wave w owns sub-image w = (hi|lo, k half); item = (row, column) of the 10 x 34 patch
    const int sub_which = wave >> 1, sub_kh = wave & 1;
    int pvoff[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int it = u * 64 + lane;
        const int r = it / WX_ROW, c = it - r * WX_ROW;
        const int iy = gy0 - 1 + r, ix = gx0 - 1 + c;
        const bool ok = it < (CONV_TH + 2) * WX_ROW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        pvoff[u] = ok ? ((sub_kh * p.H + iy) * p.W + ix) * 16 : CONV_OOB;
    }
    const bool last_lanes = lane < (CONV_TH + 2) * WX_ROW - 5 * 64;  // the sixth instruction covers items 320 .. 339
    const char* img_base = (const char*)p.ximg + (sub_which ? p.ximg_lo : 0) + (size_t)n * (p.I >> 3) * HW * 16;
    // chunk >= nch: a zero-length resource (zeros, no traffic, same instruction count)
    auto patch_rsrc = [&](int chunk) {
        const bool in = chunk < nch;
        return w3_rsrc(img_base + (size_t)(in ? (ic_beg + 16 * chunk) >> 3 : 0) * HW * 16, in ? 2u * HW * 16u : 0u);
    };
    auto patch_piece = [&](const i32x4& rs, int buf, int u) {  // u: compile-time after unrolling
        const uint32_t dst = lds0 + W3_WBYTES + buf * W3_PATCH + wave * W3_SUB + u * 1024;
        if (u < 5) w3_dma16(dst, rs, pvoff[u]);
        else if (last_lanes) w3_dma16(dst, rs, pvoff[5]);
    };
    // Weights: piece q = u * 256 + tid of a group = (hi|lo, dy, k half, o); group g (dx = g - 1) adds g * I * 2 bytes
    const int LO = p.O * 9 * p.I * 2;  // bytes of the hi tensor (the lo parts follow it)
    int wvoff[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int q = u * 256 + tid, which = q / 384, rem = q - which * 384;
        const int dyi = rem >> 7, kh = (rem >> 6) & 1, o = rem & 63;
        wvoff[u] = which * LO + (((o0 + o) * 9 + dyi * 3) * p.I + 8 * kh) * 2;
    }
    auto w_rsrc = [&](int chunk) {
        const int ic0 = ic_beg + 16 * chunk;
        const bool in = chunk < nch;
        return w3_rsrc((const char*)p.wh + (size_t)(in ? ic0 : 0) * 2, in ? (uint32_t)(2 * LO - ic0 * 2) : 0u);
    };
    auto w_piece = [&](const i32x4& rs, int g, int u) {
        w3_dma16(lds0 + g * W3_GROUP_BYTES + wave * 1024 + u * 4096, rs, wvoff[u] + g * p.I * 2);
    };

    f32x16 acc[2][2];  // [channel tile][row of the wave's row pair]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const int prow = 2 * wave;
    const int blane = half * W3_SUB + (prow * WX_ROW + j) * 16;  // patch row prow, column j of this lane's k half (hi image)
    const int alane = (half * 64 + j) * 16;

    // ---- prologue: [patch(0) 6][W0(0) 3][W1(0) 3]; the first two must have landed.
    // The loop issues its DMA pieces BETWEEN the MFMAs of a phase (an LDS-DMA costs ~100 issue clocks; back to back after a barrier
    // they were a bubble of the matrix core): phase 0 of chunk k requests patch(k+1) and W2(k) (9 pieces), phase 1 W0(k+1), phase 2
    // W1(k+1) (3 each).  A barrier needs what EARLIER phases requested, so its counted wait leaves this phase's own pieces in flight.
    {
        const i32x4 rp = patch_rsrc(0), rw = w_rsrc(0);
#pragma unroll
        for (int u = 0; u < 6; ++u) patch_piece(rp, 0, u);
#pragma unroll
        for (int u = 0; u < 3; ++u) w_piece(rw, 0, u);
#pragma unroll
        for (int u = 0; u < 3; ++u) w_piece(rw, 1, u);
    }
    W3_VMWAIT(3);
    __builtin_amdgcn_s_barrier();
    for (int k = 0; k < nch; ++k) {
        const char* pb = lds + W3_WBYTES + (k & 1) * W3_PATCH + blane;
        const i32x4 rp = patch_rsrc(k + 1), rw0 = w_rsrc(k), rw1 = w_rsrc(k + 1);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const char* wg = lds + g * W3_GROUP_BYTES + alane;
            f16x8 bh[4], bl[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bh[r] = *reinterpret_cast<const f16x8*>(pb + (r * WX_ROW + g) * 16);
                bl[r] = *reinterpret_cast<const f16x8*>(pb + 2 * W3_SUB + (r * WX_ROW + g) * 16);
            }
#pragma unroll
            for (int dyi = 0; dyi < 3; ++dyi) {
                f16x8 ah[2], al[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    ah[a] = *reinterpret_cast<const f16x8*>(wg + (dyi * 128 + a * 32) * 16);
                    al[a] = *reinterpret_cast<const f16x8*>(wg + 384 * 16 + (dyi * 128 + a * 32) * 16);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b + dyi], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b + dyi], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b + dyi], acc[a][b], 0, 0, 0);
                    }
                    if (a == 0) {  // half of this tap row's MFMAs are queued: the pieces issue under them
                        __builtin_amdgcn_sched_barrier(0);
                        if (g == 0) {
                            patch_piece(rp, (k + 1) & 1, 2 * dyi);
                            patch_piece(rp, (k + 1) & 1, 2 * dyi + 1);
                            w_piece(rw0, 2, dyi);
                        } else {
                            w_piece(rw1, g - 1, dyi);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g == 0) W3_VMWAIT(9); else W3_VMWAIT(3);
            __builtin_amdgcn_s_barrier();  // what the next phase reads has landed; group g (g == 2: and this patch buffer) is free
        }
    }
    W3_VMWAIT(0);  // nothing may land in LDS after this workgroup has given it back
    // RGB: the ToRGB constants go into the patch buffer the last chunk read — every wave is past the loop's last barrier, i.e. done
    // reading it, and none of the requests still in flight (zeros for the chunk after the last) targets it
    float* rgbc = reinterpret_cast<float*>(lds + W3_WBYTES + ((nch + 1) & 1) * W3_PATCH);  // [5][64]: styles, weights of 4 channels
    if constexpr (RGB) {
        if (tid < 64) {
            rgbc[tid] = rgb_s;
#pragma unroll
            for (int o = 0; o < 4; ++o) rgbc[64 + 64 * o + tid] = rgb_w[o];
        }
        __syncthreads();
    }
    // ---- epilogue (branch-free): v = act((acc * d * 2^-10 + noise) + bias) * gain, clamped; raw partials: d = 2^-10, the rest neutral
    const bool ep = p.epilogue != 0;
    const float alpha = (ep && p.act == 1) ? p.alpha : 1.0f, gain = ep ? p.gain : 1.0f;
    const float cl = (ep && p.clamp >= 0.0f) ? p.clamp : __builtin_inff();
    float* yout = p.y + (p.ksplit > 1 ? (size_t)kz * p.N * p.O * p.OH * p.OW : 0) + (size_t)n * p.O * p.OH * p.OW;
    const int OHW = p.OH * p.OW;
    auto ry = __builtin_amdgcn_make_buffer_rsrc((void*)yout, 0, p.O * OHW * 4, CONV_RSRC_FLAGS);
    const int gx = gx0 + j;
    float nz[2];
    int yoff[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int gy = gy0 + prow + b;
        const bool ok = gy < p.GH && gx < p.GW;
        yoff[b] = ok ? ((o0 + 4 * half) * OHW + gy * p.OW + gx) * 4 : CONV_OOB;
        nz[b] = (ep && p.noise && ok) ? p.noise[(p.noise_per_sample ? (size_t)n * OHW : 0) + (size_t)gy * p.OW + gx] : 0.0f;
    }
    const f32x4* dq = reinterpret_cast<const f32x4*>(epi + 4 * half);        // channels a * 32 + 8 * (r >> 2) + 4 * half + (r & 3)
    const f32x4* bq = reinterpret_cast<const f32x4*>(epi + 64 + 4 * half);
    const f32x4* sq = reinterpret_cast<const f32x4*>(epi + 128 + 4 * half);
    // the optional image of the result for the layer that follows (the next block's up-sampling conv0): this lane's four channels
    // of a group of eight are half a 16-byte piece — 8 bytes of hi parts and 8 of lo parts per (pixel, channel group), the two
    // channel halves of the wave fill the piece.  Same arithmetic as k_act_to_image on the fp32 result: (s * v) * 16, clamp, RNE, residual.
    const bool wimg = p.yimg != nullptr;  // (uniform)
    const char* ib = (const char*)p.yimg + (size_t)n * (p.O >> 3) * OHW * 16;
    auto rih = __builtin_amdgcn_make_buffer_rsrc((void*)ib, 0, wimg ? (p.O >> 3) * OHW * 16 : 0, CONV_RSRC_FLAGS);
    auto ril = __builtin_amdgcn_make_buffer_rsrc((void*)(ib + p.yimg_lo), 0, wimg ? (p.O >> 3) * OHW * 16 : 0, CONV_RSRC_FLAGS);
    int ioff[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int gy = gy0 + prow + b;
        ioff[b] = (gy < p.GH && gx < p.GW) ? ((o0 >> 3) * OHW + gy * p.OW + gx) * 16 + 8 * half : CONV_OOB;
    }
    bool bad = false;
    const bool wy = !RGB || p.y != nullptr;  // (uniform)
    float rgba[2][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};  // RGB: [row][ToRGB channel], this lane's 32 channels
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 d4 = dq[(a * 32 + 8 * r4) >> 2], b4 = bq[(a * 32 + 8 * r4) >> 2], s4 = sq[(a * 32 + 8 * r4) >> 2];
            f32x4 ts4, tw4[4];
            if constexpr (RGB) {
                ts4 = *reinterpret_cast<const f32x4*>(rgbc + a * 32 + 8 * r4 + 4 * half);
#pragma unroll
                for (int o = 0; o < 4; ++o) tw4[o] = *reinterpret_cast<const f32x4*>(rgbc + 64 + 64 * o + a * 32 + 8 * r4 + 4 * half);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float vv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[a][b][4 * r4 + e] * d4[e];
                    v = v + nz[b];
                    v = v + b4[e];
                    v = v < 0.0f ? v * alpha : v;
                    v = v * gain;
                    v = __builtin_fminf(__builtin_fmaxf(v, -cl), cl);
                    vv[e] = v;
                    if (wy) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, yoff[b], (a * 32 + 8 * r4 + e) * OHW * 4, 0);
                    if constexpr (RGB) {  // ToRGB's modulated input s * x (its own rounding, networks_stylegan2.py:68), then the 1x1 weights
                        const float m = ts4[e] * v;
#pragma unroll
                        for (int o = 0; o < 4; ++o) rgba[b][o] = __builtin_fmaf(tw4[o][e], m, rgba[b][o]);
                    }
                }
                if (wimg) {
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    typedef int i32x2 __attribute__((ext_vector_type(2)));
                    f16x4 hv, lv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float m = s4[e] * vv[e] * HX_SPLIT_SCALE_X;
                        bad = bad || !(__builtin_fabsf(m) <= 65504.0f);
                        m = __builtin_fminf(__builtin_fmaxf(m, -65504.0f), 65504.0f);
                        hv[e] = (_Float16)m;
                        lv[e] = (_Float16)(m - (float)hv[e]);
                    }
                    const int so = (a * 4 + r4) * OHW * 16;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, hv), rih, ioff[b], so, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, lv), ril, ioff[b], so, 0);
                }
            }
            if constexpr (RGB) __builtin_amdgcn_sched_barrier(0);  // (the ToRGB constants of one channel group at a time: hoisted together they filled the register file)
        }
    if (wimg && bad && p.sat) atomicOr(p.sat, 1u);
    if constexpr (RGB) {
        // the two channel halves of a pixel sit on lanes j and j + 32: lane (half, j) finishes row `half` of the wave's pair (it sends
        // its share of the other row to its partner: a + b == b + a, so both rows are summed in the same order) and stores the
        // workgroup's share of the ToRGB sum; p3d_torgb_combine_f32 adds the channel tiles in tile order
        const int gy = gy0 + prow + half;
        float* dst = p.rgbp + (((size_t)wo.otile * p.N + n) * p.rgbo) * OHW + (size_t)gy * p.OW + gx;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float other = half ? rgba[0][o] : rgba[1][o], own = half ? rgba[1][o] : rgba[0][o];
            const float got = __shfl_xor(other, 32, 64);
            if (o < p.rgbo && gy < p.GH && gx < p.GW) dst[(size_t)o * OHW] = own + got;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_modconv_up3 (round 4): the stride-2 transposed two-term convolution (conv0 of every block) fed from an activation IMAGE, every
// operand by LDS-DMA, everything double buffered, ONE barrier per 16-channel chunk (k_modconv_up_h: fp32 input converted in the
// kernel through registers, two barriers and two exposed DMA round trips per chunk — MFMA-busy 0.07-0.24, profiles/history/r03_mfma_util.json).
//   workgroup = 32 output channels x 8 rows x 32 columns of grid positions ((H+1) x (W+1), four output phases each);
//   wave w    = rows 2w, 2w + 1 (two N tiles of one row x 32 columns: lane j = column j, conflict-free ds_read_b128 at any pitch)
//               x 4 phases = 8 accumulators; 54 MFMAs per chunk (9 taps x 2 rows x 3 two-term products), 30 ds_read_b128
//   LDS       = weights 2 x [hi|lo][9 taps][k half][32 o][8] (2 x 18 432 B) + patch 2 x [hi|lo][k half][9 rows][34 px][8]
//               (2 x 19 584 B) = 76 032 B: two workgroups per CU.  The chunk k + 1 is requested (inline-asm DMA, invisible to the
//               compiler's wait insertion) at the top of chunk k and waited for (vmcnt(0)) at its end.
// Raw store of the four phases into the (2H+1) x (2W+1) intermediate (or split-K partials); the FIR pass applies the epilogue.
// Same products as k_modconv_up_h<true>; fp32 summation order: per tap a_hi*b_lo, a_lo*b_hi, a_hi*b_hi.
// ---------------------------------------------------------------------------------------------------------------------
#define U3_ROWS 9
#define U3_SUB (U3_ROWS * WX_ROW * 16)             // one (hi|lo, k half) sub-image of the patch: 4 896
#define U3_PATCH (4 * U3_SUB)
#define U3_LDS (2 * U3_WB + 2 * U3_PATCH)
// FUSED (unsplit launches whose consumer takes an activation image): the FIR pass and the layer's epilogue run IN this kernel — the
// (2H+1) x (2W+1) fp32 intermediate (135 MB written and read back at 256 -> 128 @256^2 -> 512^2: the transposed convolution was
// bound by that store, not by its MFMAs) never exists.  A workgroup's 8 x 32 grid points are 16 x 64 intermediate values per channel,
// enough for 12 x 60 outputs of the 4x4 filter: tiles advance by 6 x 30 grid points (1.42 x the MFMA work), the accumulators go to LDS
// (the pipeline's buffers, free after the K loop) sixteen channels at a time, and every thread filters 4 pixels x 8 channels and
// stores the consumer's 16-byte pieces — the products, sums and filter order of k_modconv_up3<false> + k_fir4x4_img, bit for bit.
#define U3F_PS (16 * 64 + 8)   // floats per channel plane of the intermediate tile in LDS (16 rows x 64 columns + 8: the four channel pairs a wave reads at once start 16 banks apart)
template <bool FUSED>
__global__ __launch_bounds__(256, 2) void k_modconv_up3(ConvParams p) {
    __shared__ __attribute__((aligned(16))) char lds[U3_LDS];
    static_assert(16 * U3F_PS * 4 <= U3_LDS, "sixteen channels of a 16 x 64 intermediate tile fit the pipeline's buffers");
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = FUSED ? (2 * p.W + 59) / 60 : (p.GW + WX_TW - 1) / WX_TW;
    const WgOrder wo = p3d_wg_order(p.xcd != 0);
    // FUSED: outputs [12 ty, 12 ty + 12) x [60 tx, 60 tx + 60) need intermediate rows 12 ty - 1 .. and columns 60 tx - 1 ..: grid origin -1
    const int gy0 = FUSED ? (wo.tile / tiles_x) * 6 - 1 : (wo.tile / tiles_x) * 8;
    const int gx0 = FUSED ? (wo.tile % tiles_x) * 30 - 1 : (wo.tile % tiles_x) * WX_TW;
    const int o0 = wo.otile * 32;
    const int n = wo.z / p.ksplit, kz = wo.z - n * p.ksplit;
    const int ic_per = ((p.I + p.ksplit - 1) / p.ksplit + 31) / 32 * 32;
    const int ic_beg = kz * ic_per, ic_end = (ic_beg + ic_per < p.I) ? ic_beg + ic_per : p.I;
    const int nch = ic_end > ic_beg ? (ic_end - ic_beg) >> 4 : 0;
    const int HW = p.H * p.W;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    // FUSED: the epilogue's per-channel constants (read after the K loop's barriers) and this thread's twelve noise values, requested
    // here so that no global round trip is left between the K loop and the stores
    __shared__ float epi[FUSED ? 96 : 1];
    float nzv[FUSED ? 12 : 1];
    if constexpr (FUSED) {
        if (tid < 32) {
            const int ch = o0 + tid;
            epi[tid] = p.dcoef ? p.dcoef[(size_t)n * p.O + ch] : 1.0f;
            epi[32 + tid] = p.bias ? p.bias[ch] : 0.0f;
            epi[64 + tid] = p.ystyles[(size_t)n * p.O + ch];
        }
        const int OHo = 2 * p.H, OWo = 2 * p.W, X = 2 * gx0 + 2 + (tid >> 2);
        const float* nz = p.noise ? p.noise + (p.noise_per_sample ? (long long)n * OHo * OWo : 0) : nullptr;
#pragma unroll
        for (int ly = 1; ly < 13; ++ly) {
            const int Y = 2 * gy0 + 1 + ly;
            nzv[ly - 1] = (nz && (tid >> 2) < 60 && X < OWo && Y < OHo) ? nz[(long long)Y * OWo + X] : 0.0f;
        }
    }
    // ---- DMA plans.  Patch: wave w owns sub-image w = (hi|lo, k half): 9 x 34 items, 4 full + 1 partial instruction
    const int sub_which = wave >> 1, sub_kh = wave & 1;
    int pvoff[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int it = u * 64 + lane;
        const int r = it / WX_ROW, c = it - r * WX_ROW;
        const int iy = gy0 - 1 + r, ix = gx0 - 1 + c;
        const bool ok = it < U3_ROWS * WX_ROW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        pvoff[u] = ok ? ((sub_kh * p.H + iy) * p.W + ix) * 16 : CONV_OOB;
    }
    const bool last_lanes = lane < U3_ROWS * WX_ROW - 4 * 64;
    const char* img_base = (const char*)p.ximg + (sub_which ? p.ximg_lo : 0) + (size_t)n * (p.I >> 3) * HW * 16;
    // Weights: 1152 pieces (hi|lo, tap, k half, o) = 18 instructions; wave w issues instructions w, w + 4, ...
    const int LO = p.O * 9 * p.I * 2;
    int wvoff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int q = (wave + 4 * i) * 64 + lane, which = q / 576, rem = q - which * 576;
        const int tap = rem >> 6, kh = (rem >> 5) & 1, o = rem & 31;
        wvoff[i] = (q < 1152 && o0 + o < p.O) ? which * LO + (((o0 + o) * 9 + tap) * p.I + 8 * kh) * 2 : CONV_OOB;
    }
    const bool five = wave < 2;  // instructions 16, 17 exist for waves 0, 1 only
    // piece i of chunk `chunk` into buffer `buf`: 0 .. 4 the patch (4: partial), 5 .. 9 the weights (9: waves 0, 1).  chunk >= nch: a
    // zero-length resource (zeros into the idle buffer, no traffic, the same instruction count)
    struct U3Rs { i32x4 rp, rw; };
    auto rsrcs = [&](int chunk) {
        const bool in = chunk < nch;
        const int ic0 = in ? ic_beg + 16 * chunk : 0;
        U3Rs r;
        r.rp = w3_rsrc(img_base + (size_t)(ic0 >> 3) * HW * 16, in ? 2u * HW * 16u : 0u);
        r.rw = w3_rsrc((const char*)p.wh + (size_t)ic0 * 2, in ? (uint32_t)(2 * LO - ic0 * 2) : 0u);
        return r;
    };
    auto piece = [&](const U3Rs& r, int buf, int i) {  // i: compile-time after unrolling
        const uint32_t pd = lds0 + 2 * U3_WB + buf * U3_PATCH + wave * U3_SUB;
        const uint32_t wd = lds0 + buf * U3_WB + wave * 1024;
        if (i < 4) w3_dma16(pd + i * 1024, r.rp, pvoff[i]);
        else if (i == 4) { if (last_lanes) w3_dma16(pd + 4 * 1024, r.rp, pvoff[4]); }
        else if (i < 9) w3_dma16(wd + (i - 5) * 4096, r.rw, wvoff[i - 5]);
        else if (five) w3_dma16(wd + 4 * 4096, r.rw, wvoff[4]);
    };

    f32x16 acc[4][2];  // [phase = 2 py + px][row of the wave's pair]
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ph][t][r] = 0.0f;
    // patch row 2w + 1 + t is grid row gy0 + 2w + t; column j + 1 is grid column gx0 + j
    const int blane = half * U3_SUB + ((2 * wave) * WX_ROW + j) * 16;
    const int alane = (half * 32 + j) * 16;
    // (phase, tap, input) of the nine products: input 0 = x[y][x], 1 = x[y][x-1], 2 = x[y-1][x], 3 = x[y-1][x-1]
    const int PH[9] = {0, 1, 2, 3, 0, 2, 0, 1, 0}, TP[9] = {0, 1, 3, 4, 2, 5, 6, 7, 8}, BO[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};

    {
        const U3Rs r0 = rsrcs(0);
#pragma unroll
        for (int i = 0; i < 10; ++i) piece(r0, 0, i);
    }
    W3_VMWAIT(0);
    __builtin_amdgcn_s_barrier();
    // QM: the taps (bits of q) this tile needs, NT: its rows per wave, W0: only wave 0 has a valid row.  The grid is (H + 1) x (W + 1):
    // its last column / row is a tile of its own whose lanes see zeros for x[.][W] / x[H][.], i.e. 6 of the 9 taps add exact zeros
    // (never -0: an accumulator that starts at +0 cannot become -0) — those tiles skip them and leave the matrix core to their neighbours.
    auto run = [&](auto QMc, auto NTc, auto W0c) {
        constexpr int QM = decltype(QMc)::value, NT = decltype(NTc)::value;
        constexpr bool W0 = decltype(W0c)::value, FULL = QM == 0x1FF;
        for (int k = 0; k < nch; ++k) {
            const U3Rs rn = rsrcs(k + 1);
            if (!FULL) {
#pragma unroll
                for (int i = 0; i < 10; ++i) piece(rn, (k + 1) & 1, i);
            }
            if (!W0 || wave == 0) {
                const char* pb = lds + 2 * U3_WB + (k & 1) * U3_PATCH + blane;
                const char* wb = lds + (k & 1) * U3_WB + alane;
                // rows 2w, 2w + 1, 2w + 2 of the patch x columns j (dx = -1), j + 1 (dx = 0), hi and lo
                f16x8 bh[3][2], bl[3][2];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        bh[r][c] = *reinterpret_cast<const f16x8*>(pb + (r * WX_ROW + c) * 16);
                        bl[r][c] = *reinterpret_cast<const f16x8*>(pb + 2 * U3_SUB + (r * WX_ROW + c) * 16);
                    }
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    if (!((QM >> q) & 1)) continue;
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(wb + TP[q] * 64 * 16);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(wb + (9 + TP[q]) * 64 * 16);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int r = 1 + t - (BO[q] >> 1), c = 1 - (BO[q] & 1);
                        acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[r][c], acc[PH[q]][t], 0, 0, 0);
                        acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[r][c], acc[PH[q]][t], 0, 0, 0);
                        acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[r][c], acc[PH[q]][t], 0, 0, 0);
                        // a full tile issues the next chunk's ten pieces two at a time under the MFMAs of its first five taps
                        if (FULL && t == 0 && q < 5) {
                            __builtin_amdgcn_sched_barrier(0);
                            piece(rn, (k + 1) & 1, 2 * q);
                            piece(rn, (k + 1) & 1, 2 * q + 1);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            W3_VMWAIT(0);
            __builtin_amdgcn_s_barrier();
        }
    };
    {
        using std::integral_constant;
        const bool col_edge = !FUSED && gx0 == p.W, row_edge = !FUSED && gy0 == p.H;  // (uniform)
        if (!col_edge && !row_edge) run(integral_constant<int, 0x1FF>{}, integral_constant<int, 2>{}, integral_constant<bool, false>{});
        else if (!row_edge) run(integral_constant<int, 0x130>{}, integral_constant<int, 2>{}, integral_constant<bool, false>{});
        else if (!col_edge) run(integral_constant<int, 0x1C0>{}, integral_constant<int, 1>{}, integral_constant<bool, true>{});
        else run(integral_constant<int, 0x100>{}, integral_constant<int, 1>{}, integral_constant<bool, true>{});
    }
    if constexpr (FUSED) {
        // ---- FIR + epilogue.  Output (Y, X) = (2 gy0 + 1 + ly, 2 gx0 + 1 + lx), ly in [1, 13), lx in [1, 61), reads the local
        // intermediate rows ly .. ly + 3, columns lx .. lx + 3 (= T[Y - 1 + fy][X - 1 + fx]); grid points outside the map gave exact
        // zeros (the FIR pass's zero padding).  Sixteen channels at a time through LDS; a thread = (channel pair, output column) and
        // walks the 12 rows with a 4 x 4 window per channel in registers: consecutive lanes = the four channel pairs of a 16-byte
        // piece, then the next pixel — a wave's 4-byte stores are 256 contiguous bytes of the hi (and of the lo) image.
        float* T = reinterpret_cast<float*>(lds);  // [16 channels][16 rows][64] at a plane stride of U3F_PS floats
        const int OHo = 2 * p.H, OWo = 2 * p.W;
        float fs[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) fs[i] = p.fir[i];
        const long long lo_off = (long long)p.N * p.O * OHo * OWo * 2;
        const bool has_nz = p.noise != nullptr;
        const int xq = tid >> 2, cpl = tid & 3;      // output column 1 + xq of the tile, channel pair cpl of its 8-channel group
        const int X = 2 * gx0 + 2 + xq;
        const bool col_ok = xq < 60 && X < OWo;
        bool bad = false;
#pragma unroll 1
        for (int bt = 0; bt < 2; ++bt) {
            if (bt) __builtin_amdgcn_s_barrier();  // (the loop ended on a barrier: every wave is done with the buffers)
            if (bt == 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr)
                            T[((rr & 3) + 8 * (rr >> 2) + 4 * half) * U3F_PS + (2 * (2 * wave + t) + (ph >> 1)) * 64 + 2 * j + (ph & 1)] =
                                acc[ph][t][rr] * HX_SPLIT_UNSCALE;
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr)
                            T[((rr & 3) + 8 * (rr >> 2) + 4 * half) * U3F_PS + (2 * (2 * wave + t) + (ph >> 1)) * 64 + 2 * j + (ph & 1)] =
                                acc[ph][t][8 + rr] * HX_SPLIT_UNSCALE;
            }
            __syncthreads();
#pragma unroll 1
            for (int g2 = 0; g2 < 2; ++g2) {
                const int c8 = (o0 >> 3) + 2 * bt + g2;   // channels 8 c8 .. 8 c8 + 7 of the layer; this thread: 8 c8 + 2 cpl, + 1
                float dc[2], bs[2], ns[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int cl = 16 * bt + 8 * g2 + 2 * cpl + c;  // channel of the workgroup's 32
                    dc[c] = epi[cl]; bs[c] = epi[32 + cl]; ns[c] = epi[64 + cl];
                }
                const float* Tc = T + (g2 * 8 + 2 * cpl) * U3F_PS + 1 + (xq < 60 ? xq : 0);
                float win[2][4][4];  // [channel][row slot = local row & 3][tap column]
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 1; r < 4; ++r)
#pragma unroll
                        for (int fx = 0; fx < 4; ++fx) win[c][r][fx] = Tc[c * U3F_PS + r * 64 + fx];
                char* dst = (char*)p.yimg + (((size_t)n * (p.O >> 3) + c8) * OHo * (size_t)OWo + X) * 16 + cpl * 4;
#pragma unroll
                for (int ly = 1; ly < 13; ++ly) {
                    const int Y = 2 * gy0 + 1 + ly;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int fx = 0; fx < 4; ++fx) win[c][(ly + 3) & 3][fx] = Tc[c * U3F_PS + (ly + 3) * 64 + fx];
                    const bool ok = col_ok && Y < OHo;
                    const float nvv = nzv[ly - 1];
                    _Float16 hh[2], ll[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        float o = 0.0f;
#pragma unroll
                        for (int fy = 0; fy < 4; ++fy)
#pragma unroll
                            for (int fx = 0; fx < 4; ++fx) o = __builtin_fmaf(fs[fy * 4 + fx], win[c][(ly + fy) & 3][fx], o);
                        float a = o * dc[c];
                        a = has_nz ? a + nvv : a;
                        a = a + bs[c];
                        a = ns[c] * act_apply(a, p.act, p.alpha, p.gain, p.clamp) * HX_SPLIT_SCALE_X;  // (k_fir4x4_img's epilogue)
                        bad = bad || (ok && !(__builtin_fabsf(a) <= 65504.0f));
                        a = __builtin_fminf(__builtin_fmaxf(a, -65504.0f), 65504.0f);
                        hh[c] = (_Float16)a;
                        ll[c] = (_Float16)(a - (float)hh[c]);
                    }
                    if (ok) {
                        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<f16x2*>(dst + (size_t)Y * OWo * 16) = (f16x2){hh[0], hh[1]};
                        *reinterpret_cast<f16x2*>(dst + lo_off + (size_t)Y * OWo * 16) = (f16x2){ll[0], ll[1]};
                    }
                    if ((ly & 3) == 0) asm volatile("" ::: "memory");  // four rows of LDS reads in flight, not all twelve in one 100-register block
                }
            }
        }
        if (bad && p.sat) atomicOr(p.sat, 1u);
        return;
    }
    // ---- raw store: a lane owns both column phases (ox = 2 gx, 2 gx + 1) of its grid point: one 8-byte store per (row phase, channel),
    // 32 lanes = 256 contiguous bytes; the last grid column (gx = W) has only px = 0: a 4-byte store of its own
    float* yout = p.y + (p.ksplit > 1 ? (size_t)kz * p.N * p.O * p.OH * p.OW : 0) + (size_t)n * p.O * p.OH * p.OW;
    const int OHW = p.OH * p.OW;
    auto ry = __builtin_amdgcn_make_buffer_rsrc((void*)yout, 0, p.O * OHW * 4, CONV_RSRC_FLAGS);
    const int gx = gx0 + j;
    const bool edge_tile = gx0 + WX_TW > p.W;  // (uniform) this tile holds the column gx = W
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int gy = gy0 + 2 * wave + t;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            const bool row_ok = gy <= p.H - py;
            const int base = ((o0 + 4 * half) * OHW + (2 * gy + py) * p.OW + 2 * gx + p.tox) * 4;
            const int off2 = (row_ok && gx < p.W && o0 + 4 * half < p.O) ? base : CONV_OOB;
            const int off1 = (row_ok && gx == p.W && o0 + 4 * half < p.O) ? base : CONV_OOB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int so = ((r & 3) + 8 * (r >> 2)) * OHW * 4;
                const float v0 = acc[2 * py][t][r] * HX_SPLIT_UNSCALE, v1 = acc[2 * py + 1][t][r] * HX_SPLIT_UNSCALE;
                typedef int i32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64((i32x2){__builtin_bit_cast(int, v0), __builtin_bit_cast(int, v1)}, ry, off2, so, 0);
                if (edge_tile) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v0), ry, off1, so, 0);
            }
        }
    }
}

// the fused four-phase transposed convolution (see k_modconv_up) on f16 operands
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_modconv_up_h(ConvParams p) {
    constexpr int NT = 9, WBYTES = NT * 128 * 16;
    __shared__ __attribute__((aligned(16))) char xs[2][SPLIT ? 2 * HX_BYTES : HX_BYTES];
    __shared__ __attribute__((aligned(16))) char ws[SPLIT ? 1 : 2][SPLIT ? 2 * WBYTES : WBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wp = wave >> 1, half = lane >> 5, j = lane & 31;
    const int tiles_x = (p.GW + CONV_TW - 1) / CONV_TW;
    const int gy0 = (blockIdx.x / tiles_x) * CONV_TH, gx0 = (blockIdx.x % tiles_x) * CONV_TW;
    const int o0 = blockIdx.y * 64;
    const int n = blockIdx.z / p.ksplit, kz = blockIdx.z - n * p.ksplit;
    const int ic_per = ((p.I + p.ksplit - 1) / p.ksplit + 31) / 32 * 32;
    const int ic_beg = kz * ic_per, ic_end = (ic_beg + ic_per < p.I) ? ic_beg + ic_per : p.I;
    const float* xn = p.x + (size_t)n * p.I * p.H * p.W;
    const float* sn = p.styles + (size_t)n * p.I;

    f32x16 acc[4][2];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ph][t][r] = 0.0f;
    const int prow0 = 4 * wp + (j >> 4), pcol = j & 15;
    const int xlane = half * HX_HALF + ((prow0 + 1) * HX_PITCH + pcol + 1) * 16;
    const int wlane = (half * 64 + wc * 32 + j) * 16;

    const ConvStagePlanH pl = conv_plan_h<NT>(p, tid, gy0, gx0, o0);
    // SPLIT: only the activations go through registers (x: the fp32 -> hi / lo conversion); the weights are copied L2 -> LDS
    ConvStageRegsH<SPLIT ? 0 : NT, false> rg;
    if constexpr (SPLIT) {
        conv_gload_h<0, false>(p, pl, xn, sn, ic_beg, ic_end, rg);
        conv_glds_w2<NT>(p, pl, ws[0], tid, ic_beg, ic_end);
        conv_lstore_hx<0, true>(xs[0], pl, rg, p.sat);
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the LDS-direct loads have landed
    } else {
        conv_gload_h<NT>(p, pl, xn, sn, ic_beg, ic_end, rg);
        conv_lstore_h<NT>(xs[0], ws[0], tid, pl, rg);
    }
    __syncthreads();
    int buf = 0;
    // (phase, tap, patch offset) of the nine products of the four output phases
    const int PH[9] = {0, 1, 2, 3, 0, 2, 0, 1, 0}, TP[9] = {0, 1, 3, 4, 2, 5, 6, 7, 8}, BO[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
    for (int ic0 = ic_beg; ic0 < ic_end; ic0 += 16) {
        const bool more = ic0 + 16 < ic_end;
        if (more) conv_gload_h<SPLIT ? 0 : NT, false>(p, pl, xn, sn, ic0 + 16, ic_end, rg);
        const char* xb = xs[buf] + xlane;
        const char* wb = ws[SPLIT ? 0 : buf] + wlane;
        // the four input values per N tile ([dy][dx] with dy, dx in {0, -1}; N tile 1 is two rows below), hi or lo image
        auto load_b = [&](int boff, f16x8 (&bq)[2][4]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const char* xp = xb + boff + (2 * t * HX_PITCH) * 16;
                bq[t][0] = *reinterpret_cast<const f16x8*>(xp);
                bq[t][1] = *reinterpret_cast<const f16x8*>(xp - 16);
                bq[t][2] = *reinterpret_cast<const f16x8*>(xp - HX_PITCH * 16);
                bq[t][3] = *reinterpret_cast<const f16x8*>(xp - HX_PITCH * 16 - 16);
            }
        };
        auto run_pass = [&](int aoff, int boff) {
            f16x8 bq[2][4];
            load_b(boff, bq);
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const f16x8 a = *reinterpret_cast<const f16x8*>(wb + aoff + TP[q] * 128 * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[t][BO[q]], acc[PH[q]][t], 0, 0, 0);
            }
        };
        if constexpr (SPLIT) {
            // as in k_modconv_w2: the two halves of the single-buffered weight image are re-loaded under MFMAs.  Round 4: a_hi is read
            // once for both of its products and the hi input tiles stay in registers for the a_lo pass: 16 + 9 + 9 = 34 ds_read_b128
            // per wave and chunk instead of 3 x 17 = 51 for the same 54 MFMAs.
            f16x8 bhq[2][4];
            {
                f16x8 blq[2][4];
                load_b(0, bhq);
                load_b(HX_BYTES, blq);
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    const f16x8 a = *reinterpret_cast<const f16x8*>(wb + TP[q] * 128 * 16);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, blq[t][BO[q]], acc[PH[q]][t], 0, 0, 0);  // a_hi x b_lo
                        acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bhq[t][BO[q]], acc[PH[q]][t], 0, 0, 0);  // a_hi x b_hi
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) conv_lstore_hx<0, true>(xs[buf ^ 1], pl, rg, p.sat);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();         // a_hi is free, a_lo (requested before this chunk's first pass) has landed everywhere
            if (more) conv_glds_w2<NT>(p, pl, ws[0], tid, ic0 + 16, ic_end, 0);
#pragma unroll
            for (int q = 0; q < 9; ++q) {  // a_lo x b_hi
                const f16x8 a = *reinterpret_cast<const f16x8*>(wb + WBYTES + TP[q] * 128 * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bhq[t][BO[q]], acc[PH[q]][t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();         // a_lo and this patch buffer are free, a_hi(next) has landed
            if (more) conv_glds_w2<NT>(p, pl, ws[0], tid, ic0 + 16, ic_end, 1);
        } else {
            run_pass(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) conv_lstore_h<NT>(xs[buf ^ 1], ws[buf ^ 1], tid, pl, rg);
            __syncthreads();
        }
        buf ^= 1;
    }
    float* yout = p.y + (p.ksplit > 1 ? (size_t)kz * p.N * p.O * p.OH * p.OW : 0);
    // a lane owns both column phases (ox = 2 gx, 2 gx + 1) of its grid point: one 8-byte store per (row phase, channel) — 16 lanes
    // cover 128 contiguous bytes of an output row — instead of two 4-byte stores 8 bytes apart (the last grid column has only
    // px = 0; rows of the odd-width intermediate are 4-byte aligned, which global stores allow)
    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int gy = gy0 + prow0 + 2 * t, gx = gx0 + pcol;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            if (gy > p.H - py || gx > p.W) continue;
            const int oy = 2 * gy + py, ox = 2 * gx;
            const bool both = gx < p.W;  // px = 1 exists
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = o0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (ch >= p.O) continue;
                const float v0 = SPLIT ? acc[2 * py][t][r] * HX_SPLIT_UNSCALE : acc[2 * py][t][r];
                const float v1 = SPLIT ? acc[2 * py + 1][t][r] * HX_SPLIT_UNSCALE : acc[2 * py + 1][t][r];
                float* dst = yout + (((size_t)n * p.O + ch) * p.OH + oy) * p.OW + ox + p.tox;
                if (both) *reinterpret_cast<f32x2u*>(dst) = (f32x2u){v0, v1};
                else dst[0] = v0;
            }
        }
    }
}

// upsample2d (up 2, pad [2,1,2,1], 4x4 filter: upfirdn2d.py:341-350) of one plane xc [H][W] at output pixel (Y, X), polyphase:
// only the 2 x 2 taps that meet non-zero samples of the zero-inserted input, in the generic operator's order (fy, then fx,
// ascending) — the same fma chain as k_upsample2x_add / k_upfirdn2d, bit for bit.
DEV float upsample2x_at(const float* __restrict__ xc, const float* __restrict__ f, int H, int W, int Y, int X) {
    const int fy0 = Y & 1, fx0 = X & 1;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int fy = fy0 + 2 * a, u = (Y + fy - 2) >> 1;  // arithmetic shift: -1 for the row above the image
        if (u < 0 || u >= H) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int fx = fx0 + 2 * b, v = (X + fx - 2) >> 1;
            if (v < 0 || v >= W) continue;
            acc = __builtin_fmaf(f[fy * 4 + fx], xc[(size_t)u * W + v], acc);
        }
    }
    return acc;
}


// =====================================================================================================================
// ToRGB (networks_stylegan2.py:366-380) as what it is — a [O x I] x [I x pixels] GEMM that reads its activation exactly once —
// fused with SynthesisBlock's skip connection `img = upsample2d(img) + y` (:476-478).  Round 3; replaces k_modconv<1> (the 3x3
// kernels' tile machinery: patch staging through LDS, 49 us for 128 -> 96 channels at 256^2 = 0.6 TB/s) + k_splitk_reduce +
// k_upsample2x_add.
//   * B operand (the activation) goes from global memory STRAIGHT into MFMA operand registers: lane (j, h) of a wave owns pixel
//     p0 + j and loads x[n][k0 + 2c + h][p0 + j] — 32 consecutive floats per half wave, every byte of x read once by one wave;
//     the modulation s[n][k] * x is one VALU multiply per operand (same rounding as k_modconv<1>: bit-identical where that
//     kernel ran without split-K).
//   * A operand: the raw (unmodulated, hence per-layer constant) weights, pre-transposed once per layer to [I][O32] (O padded to a
//     multiple of 32), stream L2 -> LDS by buffer_load ... lds in 64-channel chunks, double buffered; lanes read [k][32 t + j].
//   * v_mfma_f32_32x32x2_f32: exact fp32; the C/D layout (pixels on lanes, channels on registers) stores NCHW rows directly,
//     128 B per (channel, half wave), and the skip image's 2 x 2 polyphase taps are neighbouring pixels on neighbouring lanes.
//   * two shapes of the same loop: PX (maps of >= 256^2: a wave = 32 pixels x all K, a workgroup = 128 pixels) and KS (smaller
//     maps: a workgroup = 32 pixels, its four waves split every chunk's channel pairs and add their partial sums through LDS in
//     wave order — deterministic; enough workgroups without a second launch).
// =====================================================================================================================
struct TorgbParams {
    const float* x;       // [N][I][HW]
    const float* wt;      // [I][OP] raw weights, transposed, OP = 32 * MT (zero padded)
    const float* styles;  // [N][I] (already multiplied by ToRGB's weight_gain)
    const float* bias;    // [O] or null
    const float* skip;    // [N][O][H/2][W/2] or null
    const float* skipf;   // [16]
    float* y;             // [N][O][HW]
    int N, I, O, H, W;
    float clamp;
};
#define TG_KC 64
// MS (KS only, MT = 1): the workgroup multiplies ONE of the three 32-channel tiles of a 96-channel layer (blockIdx.z): on the
// 4^2 .. 64^2 maps a launch is a handful of workgroups, each a serial chain of 192 f32 MFMAs per wave (64 clocks each) — three times
// the workgroups, a third of the chain; the same sums in the same order.
// PRE (KS, MT = 1, I <= 512; round 6): on the 4^2 .. 64^2 maps the launch is a few workgroups and the chunk loop below was eight
// exposed round trips (load, wait, barrier: 9-10 us for microseconds of work).  Here a wave requests EVERYTHING it multiplies up front —
// its 64 activation values and its 64 weight values per lane, the weights straight from global memory into MFMA operand registers
// (128 contiguous bytes per half wave; no LDS ring) — and multiplies as the data lands: one exposed round trip per launch.  The same
// products in the same order (chunk, channel pair; then the waves' partial sums in wave order): bit-identical to the loop.
template <int MT, bool KS, bool MS = false, bool PRE = false>
__global__ __launch_bounds__(256, 2) void k_torgb(TorgbParams p) {
    static_assert(!MS || (KS && MT == 1), "the channel-tile split is a variant of the small-map shape");
    static_assert(!PRE || (KS && MT == 1), "the all-up-front variant is a variant of the small-map shape");
    constexpr int OP = 32 * MT, ABUF = TG_KC * OP;  // floats per A chunk
    constexpr int OPW = MS ? 96 : OP;               // floats per row of wt
    const int chb = MS ? 32 * blockIdx.z : 0;       // first output channel of this workgroup
    extern __shared__ __attribute__((aligned(16))) float tg_lds[];
    float* As = tg_lds;                 // [2][TG_KC][OP]
    float* Ss = tg_lds + 2 * ABUF;      // [I] styles of this image (I <= 512... sized by the host)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int n = blockIdx.y, HW = p.H * p.W;
    const int p0 = KS ? blockIdx.x * 32 : blockIdx.x * 128 + wave * 32;
    const int px = p0 + j;
    const bool pvalid = px < HW;
    const int pxc = pvalid ? px : HW - 1;
    for (int i = tid; i < (PRE ? 8 * TG_KC : ((p.I + TG_KC - 1) / TG_KC) * TG_KC); i += 256) Ss[i] = i < p.I ? p.styles[(size_t)n * p.I + i] : 0.0f;  // zero tail: no predicate in the K loop
    // x of this image through a buffer resource: per-lane offset = ((channel pair + h) * HW + pixel) * 4
    auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)n * p.I * HW), 0, p.I * HW * 4, CONV_RSRC_FLAGS);
    auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.I * OPW * 4, CONV_RSRC_FLAGS);
    const int xoff = (h * HW + pxc) * 4;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // one A chunk = TG_KC * OP floats, contiguous in wt: 16 bytes per lane per instruction (channels beyond I arrive as zeros)
    auto load_a = [&](int chunk, int buf) {
        const int base = chunk * ABUF * 4;
        static_assert((ABUF * 4) % 4096 == 0, "a chunk is a whole number of 256-lane x 16-byte rounds");
#pragma unroll
        for (int u = 0; u < ABUF * 4 / 4096; ++u) {
            const int idx = u * 256 + tid;  // 16-byte piece of the chunk; MS: row idx / 8 of wt, 128 bytes from column chb
            const int src = MS ? ((chunk * TG_KC + (idx >> 3)) * OPW + chb) * 4 + (idx & 7) * 16 : base + idx * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)((char*)(As + buf * ABUF) + (u * 256 + (tid & ~63)) * 16), 16, src, 0, 0, 0);
        }
    };
    constexpr int NC = KS ? TG_KC / 8 : TG_KC / 2;  // channel pairs of a chunk this wave multiplies: all 32, or its quarter
    const int c0 = KS ? wave * NC : 0;
    auto load_x = [&](int chunk, float (&xv)[NC]) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            // (the channel pair goes into the VECTOR offset — the one the hardware range-checks: a channel beyond I reads zero)
            xv[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff + (chunk * TG_KC + 2 * (c0 + c)) * HW * 4, 0, 0));
    };
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int nchunks = (p.I + TG_KC - 1) / TG_KC;
    if constexpr (PRE) {
        float xall[8][NC], aall[8][NC];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int c = 0; c < NC; ++c) {  // (rows beyond I: outside the resources, zeros)
                const int k = q * TG_KC + 2 * (c0 + c);
                xall[q][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff + k * HW * 4, 0, 0));
                aall[q][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, ((k + h) * OPW + chb + j) * 4, 0, 0));
            }
        __syncthreads();  // the styles are staged
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int k = q * TG_KC + 2 * (c0 + c);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aall[q][c], Ss[k + h] * xall[q][c], acc[0], 0, 0, 0);
            }
        __syncthreads();  // (the styles' region is part of what the partial sums overwrite below)
    }
    float xa[NC], xb[NC];
    if constexpr (!PRE) {
    load_a(0, 0);
    load_x(0, xa);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    }
    auto chunk_mma = [&](int chunk, int buf, const float (&xv)[NC]) {
        const float* A = As + buf * ABUF + j;
        const float* S = Ss + chunk * TG_KC + h;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = 2 * (c0 + c);  // + h: this lane's channel of the pair (a channel beyond I: style 0, x 0, weights 0)
            const float b = S[k] * xv[c];
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(k + h) * OP + 32 * t], b, acc[t], 0, 0, 0);
        }
    };
    for (int q = 0; q < (PRE ? 0 : nchunks); q += 2) {  // two chunks per iteration: the register prefetch buffers alternate by name
        if (q + 1 < nchunks) { load_a(q + 1, 1); load_x(q + 1, xb); }
        chunk_mma(q, 0, xa);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (q + 1 >= nchunks) break;
        if (q + 2 < nchunks) { load_a(q + 2, 0); load_x(q + 2, xa); }
        chunk_mma(q + 1, 1, xb);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    constexpr int NV = KS ? MT * 4 : MT * 16;  // accumulator elements this wave finishes: a quarter (KS) or all of them
    float vals[NV];
    if constexpr (KS) {
        // the four waves' partial sums meet in LDS (the A buffers are free now) and are added in wave order (0, 1, 2, 3:
        // deterministic); wave w then FINISHES elements w, w + 4, ... — the epilogue is a chain of load latencies (skip taps, bias)
        // and one wave doing all 48 channels of a 32-pixel tile cost ~15 of this kernel's ~19 us on the small maps
        float* red = tg_lds;  // [4 waves][MT * 16][64 lanes]
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * MT + t) * 16 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = wave + 4 * i;
            float v = red[e * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += red[(w * MT * 16 + e) * 64 + lane];
            vals[i] = v;
        }
    } else {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) vals[t * 16 + r] = acc[t][r];
    }
    const int ebase = KS ? wave : 0, estride = KS ? 4 : 1;  // element slot of vals[i] = ebase + i * estride = 16 t + r
    if (!pvalid) return;
    // ---- epilogue.  The skip image's four polyphase taps of this lane's pixel are the same for every channel: offsets and filter
    // weights once per lane; out-of-image taps get weight 0 at a clamped address — fma(0, x, acc) returns acc, the bits of the
    // generic operator that skips them — so the 4 x 48 loads carry no branches and pipeline (a first version kept
    // k_upsample2x_add's `continue`s: one exposed load latency per channel, 80 us instead of 49 + 21 at 256^2).
    const int Y = px / p.W, X = px - Y * p.W;
    float* yn = p.y + (size_t)n * p.O * HW + px;
    const int H2 = p.H >> 1, W2 = p.W >> 1;
    const bool has_skip = p.skip != nullptr, has_bias = p.bias != nullptr;
    // (no skip / no bias: the loads go to some valid address and a select drops them — uniform branches between the unrolled
    // elements would fence their loads exactly like the `continue`s did)
    const float* sk = has_skip ? p.skip + (size_t)n * p.O * (H2 * W2) : p.styles;
    const float* bp = has_bias ? p.bias : p.styles;
    int toff[4];
    float tw[4];
    {
        const int fy0 = Y & 1, fx0 = X & 1;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int fy = fy0 + 2 * a, fx = fx0 + 2 * b;
                const int u = (Y + fy - 2) >> 1, v = (X + fx - 2) >> 1;  // arithmetic shift: -1 above / left of the image
                const bool in = has_skip && u >= 0 && u < H2 && v >= 0 && v < W2;
                toff[2 * a + b] = in ? u * W2 + v : 0;
                tw[2 * a + b] = in ? p.skipf[fy * 4 + fx] : 0.0f;
            }
    }
    const int plane = has_skip ? H2 * W2 : 0;
    auto rsk = __builtin_amdgcn_make_buffer_rsrc((void*)sk, 0, has_skip ? p.O * plane * 4 : 4, CONV_RSRC_FLAGS);
    // groups of 8 channels: 40 loads in flight, then their stores (all 48 channels at once: 240 loads hoisted, 241 spilled VGPRs)
    constexpr int GS = NV == 12 ? 12 : (NV < 8 ? NV : 8);
    static_assert(NV % GS == 0, "whole groups");
#pragma unroll
    for (int g8 = 0; g8 < NV / GS; ++g8) {
        float outv[GS];
#pragma unroll
        for (int e = 0; e < GS; ++e) {  // values (branch-free)
            const int slot = ebase + (g8 * GS + e) * estride, t = slot >> 4, r = slot & 15;
            const int ch = chb + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int chc = ch < p.O ? ch : p.O - 1;
            float v = vals[g8 * GS + e];
            const float bb = bp[has_bias ? chc : 0];
            v = has_bias ? v + bb : v;
            v = act_apply(v, 0, 0.0f, 1.0f, p.clamp);
            float up = 0.0f;  // (buffer loads: a 32-bit offset per tap instead of a 64-bit address pair)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                up = __builtin_fmaf(tw[q], __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsk, (chc * plane + toff[q]) * 4, 0, 0)), up);
            outv[e] = has_skip ? up + v : v;
        }
#pragma unroll
        for (int e = 0; e < GS; ++e) {  // stores
            const int slot = ebase + (g8 * GS + e) * estride, t = slot >> 4, r = slot & 15;
            const int ch = chb + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ch < p.O) yn[(size_t)ch * HW] = outv[e];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The second half of a ToRGB layer whose channel sums came out of its conv1's epilogue (k_modconv_w3<true>): the shares of the
// 64-channel tiles added in tile order, + bias, clamp, + the up-sampled skip image (k_torgb's epilogue: the same four polyphase taps in
// the same order).  part [tiles][N][O][H][W]; one thread per output value.
__global__ __launch_bounds__(256) void k_torgb_combine(const float* __restrict__ part, int tiles, int N, int O, int H, int W,
                                                       const float* __restrict__ bias, float clamp, const float* __restrict__ skip,
                                                       const float* __restrict__ skipf, float* __restrict__ y) {
    const long long slice = (long long)N * O * H * W, idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= slice) return;
    const int HW = H * W, px = (int)(idx % HW), o = (int)((idx / HW) % O);
    const long long no = idx / HW;
    float v = part[idx];
    for (int t = 1; t < tiles; ++t) v += part[(size_t)t * slice + idx];
    if (bias) v = v + bias[o];
    v = act_apply(v, 0, 0.0f, 1.0f, clamp);
    if (skip) {
        const int Y = px / W, X = px - Y * W, H2 = H >> 1, W2 = W >> 1;
        const float* sk = skip + no * (H2 * W2);
        const int fy0 = Y & 1, fx0 = X & 1;
        float up = 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int fy = fy0 + 2 * a, fx = fx0 + 2 * b;
                const int u = (Y + fy - 2) >> 1, w = (X + fx - 2) >> 1;
                const bool in = u >= 0 && u < H2 && w >= 0 && w < W2;
                up = __builtin_fmaf(in ? skipf[fy * 4 + fx] : 0.0f, sk[in ? u * W2 + w : 0], up);
            }
        v = up + v;
    }
    y[idx] = v;
}

// ToRGB weights [O][I] -> [I][OP] (transposed, channels padded with zeros to OP = 32 or 96), once per layer
__global__ void k_torgb_weights(const float* __restrict__ w, int O, int I, int OP, float* __restrict__ wt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= I * OP) return;
    const int i = idx / OP, o = idx - i * OP;
    wt[idx] = o < O ? w[(size_t)o * I + i] : 0.0f;
}

// w [O][I][kk] f32 -> wh [O][kk][I] f16 (RNE), once per layer; split: followed by the lo parts f16(w - hi) in the same layout
__global__ void k_weights_to_f16(const float* __restrict__ w, int O, int I, int kk, _Float16* __restrict__ wh, int split) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x, total = (long long)O * I * kk;
    if (idx >= total) return;
    const int i = (int)(idx % I), t = (int)((idx / I) % kk), o = (int)(idx / ((long long)I * kk));
    float v = w[((long long)o * I + i) * kk + t];
    if (split) v = fminf(fmaxf(v * HX_SPLIT_SCALE_W, -65504.0f), 65504.0f);
    const _Float16 hi = (_Float16)v;
    wh[idx] = hi;
    if (split) wh[total + idx] = (_Float16)(v - (float)hi);
}

// fp32 activation [N][C][H][W] -> image for a consumer with styles s [N][C] (null: 1): split(16 * s * x); C % 8 == 0.  One thread
// per piece.  (Tests, and callers whose producer is not one of ours.)
__global__ void k_act_to_image(const float* __restrict__ x, const float* __restrict__ s, int N, int C, int HW, char* __restrict__ img,
                               long long lo_off, unsigned int* sat) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x, total = (long long)N * (C >> 3) * HW;
    if (idx >= total) return;
    const long long g = idx / HW;  // (n, c8)
    const int pix = (int)(idx - g * HW);
    const float* xp = x + g * 8 * HW + pix;
    f16x8 v, l;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float m = (s ? s[g * 8 + i] : 1.0f) * xp[(size_t)i * HW] * HX_SPLIT_SCALE_X;
        bad = bad || !(__builtin_fabsf(m) <= 65504.0f);
        m = __builtin_fminf(__builtin_fmaxf(m, -65504.0f), 65504.0f);
        v[i] = (_Float16)m;
        l[i] = (_Float16)(m - (float)v[i]);
    }
    *reinterpret_cast<f16x8*>(img + idx * 16) = v;
    *reinterpret_cast<f16x8*>(img + lo_off + idx * 16) = l;
    if (bad && sat) atomicOr(sat, 1u);
}

// sum the split-K partials in slice order (deterministic) and apply the epilogue.  part [KS][N][O][OH][OW]
struct ReduceParams {
    const float* part; float* y; const float* dcoef; const float* noise; const float* bias;
    long long per_slice;  // N*O*OH*OW
    int ksplit, O, OHW, noise_per_sample, act, epilogue;
    float alpha, gain, clamp;
};
__global__ void k_splitk_reduce(ReduceParams p) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.per_slice) return;
    // slice-ordered sum (deterministic); the loads of 8 slices are issued together — a deep split of a tiny map (64 slices of
    // 8192 outputs) is otherwise one exposed load latency per slice
    float v = p.part[idx];
    int k = 1;
    for (; k + 8 <= p.ksplit; k += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p.part[(size_t)(k + u) * p.per_slice + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; k < p.ksplit; ++k) v += p.part[(size_t)k * p.per_slice + idx];
    if (p.epilogue) {
        long long no = idx / p.OHW;
        int pix = (int)(idx - no * p.OHW), ch = (int)(no % p.O);
        long long n = no / p.O;
        if (p.dcoef) v = v * p.dcoef[no];
        if (p.noise) v = v + p.noise[(p.noise_per_sample ? n * p.OHW : 0) + pix];
        if (p.bias) v = v + p.bias[ch];
        v = act_apply(v, p.act, p.alpha, p.gain, p.clamp);
    }
    p.y[idx] = v;
}

// k_splitk_reduce of a plain layer whose result is ALSO wanted as the activation image of the layer that follows (round 6: the
// separate k_act_to_image launch after every split layer of the 4^2 .. 64^2 blocks).  One thread per output value like
// k_splitk_reduce — the same slice-ordered sum, eight slices in flight, the same epilogue — with the eight channels of a 16-byte piece
// on eight CONSECUTIVE lanes (the thread index runs channel-in-group fastest, then pixel): their 2-byte hi and lo parts are adjacent
// stores that fill whole pieces (128 contiguous bytes per eight pixels), the arithmetic of k_act_to_image on the stored value (same bits).
__global__ __launch_bounds__(256) void k_splitk_reduce_img(ReduceParams p, int N, const float* __restrict__ ystyles, _Float16* __restrict__ yimg,
                                                           long long lo_halfs, unsigned int* sat) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= p.per_slice) return;
    const int c = (int)(tid & 7);
    const long long q = tid >> 3;                 // (n, c8, pixel), pixel fastest
    const long long g = q / p.OHW;                // (n, c8)
    const int pix = (int)(q - g * p.OHW);
    const long long no = g * 8 + c, idx = no * p.OHW + pix;
    float v = p.part[idx];
    int k = 1;
    for (; k + 8 <= p.ksplit; k += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p.part[(size_t)(k + u) * p.per_slice + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; k < p.ksplit; ++k) v += p.part[(size_t)k * p.per_slice + idx];
    const long long n = no / p.O;
    if (p.dcoef) v = v * p.dcoef[no];
    if (p.noise) v = v + p.noise[(p.noise_per_sample ? n * p.OHW : 0) + pix];
    if (p.bias) v = v + p.bias[(int)(no % p.O)];
    v = act_apply(v, p.act, p.alpha, p.gain, p.clamp);
    if (p.y) p.y[idx] = v;
    float m = ystyles[no] * v * HX_SPLIT_SCALE_X;
    const bool bad = !(__builtin_fabsf(m) <= 65504.0f);
    m = __builtin_fminf(__builtin_fmaxf(m, -65504.0f), 65504.0f);
    const _Float16 hi = (_Float16)m;
    yimg[q * 8 + c] = hi;
    yimg[lo_halfs + q * 8 + c] = (_Float16)(m - (float)hi);
    if (bad && sat) atomicOr(sat, 1u);
}

// d[n,o] = rsqrt(sum_i (sum_t w[o,i,t]^2) * s[n,i]^2 + 1e-8).  One wave per (n,o).
__global__ void k_demod(const float* __restrict__ w, const float* __restrict__ s, int N, int O, int I, int kk, float* d) {
    int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wid >= N * O) return;
    int n = wid / O, o = wid - n * O;
    float acc = 0.0f;
    for (int i = lane; i < I; i += 64) {
        float sv = s[(size_t)n * I + i];
        const float* wp = w + ((size_t)o * I + i) * kk;
        float q = 0.0f;
        for (int t = 0; t < kk; ++t) { float v = wp[t] * sv; q = __builtin_fmaf(v, v, q); }
        acc += q;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) d[wid] = 1.0f / __builtin_sqrtf(acc + 1e-8f);
}

// Demodulation coefficients of L modulated convolutions in ONE launch, from the per-layer W2[o][i] = sum_taps w[o][i][t]^2 (cached
// by the host: it changes only when the weights do): d[n,o] = rsqrt(sum_i W2[o][i] * s[n][i]^2 + 1e-8)  (networks_stylegan2.py:70-73,
// the same sum with the taps folded first).  table[l] = {w2 offset, styles offset, d offset, O, I, first wave}; styles / d hold the
// layers' [N][I] / [N][O] blocks back to back.  One wave per (layer, n, o).
__global__ void k_demod_plan(const float* __restrict__ w2, const float* __restrict__ styles, const int* __restrict__ table, int L,
                             int N, int total_waves, float* __restrict__ d) {
    const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wid >= total_waves) return;
    int l = 0;
    while (l + 1 < L && table[(l + 1) * 6 + 5] <= wid) ++l;  // L <= 64: a short uniform search
    const int* t = table + l * 6;
    const int O = t[3], I = t[4], local = wid - t[5];
    const int n = local / O, o = local - n * O;
    const float* wp = w2 + t[0] + (size_t)o * I;
    const float* sp = styles + t[1] + (size_t)n * I;
    float acc = 0.0f;
    for (int i = lane; i < I; i += 64) { const float sv = sp[i]; acc = __builtin_fmaf(wp[i], sv * sv, acc); }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) d[t[2] + local] = 1.0f / __builtin_sqrtf(acc + 1e-8f);
}

struct FirParams {
    const float* x;  // [NC][H][W]
    const float* f;  // [fh][fw], already flipped for convolution and multiplied by gain
    float* y;        // [NC][OH][OW]
    const float* dcoef;  // [NC] (= [N][C]) or null
    const float* noise;  // [OH*OW] or [N][OH*OW] or null
    const float* bias;   // [C] or null
    long long NC;
    int C, H, W, OH, OW, fh, fw, up, down, padx0, pady0;
    int noise_per_sample, act, epilogue;
    float alpha, gain, clamp;
    const float* nstyles; // k_fir4x4_img: the consuming layer's styles [N][C] (the image holds split(16 * s * y))
    int ksplit;           // k_fir4x4_tiled: x holds ksplit split-K partial tensors, `slice` elements apart, summed in slice order
    long long slice;      // while the tile is loaded (shallow splits only: see modconv_impl); 1 / 0 otherwise
    int pitch, xoff;      // k_fir4x4_*: x rows are `pitch` floats apart and column v sits at index v + xoff (ConvParams::tox); the generic
                          // operator ignores them (pitch = W, xoff = 0)
};

// y[Y][X] = sum_{fy,fx} f[fy][fx] * xz[Y*down + fy - pady0][X*down + fx - padx0],  xz = zero-inserted x (xz[u*up][v*up] = x[u][v])
__global__ void k_upfirdn2d(FirParams p) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = p.NC * p.OH * p.OW;
    if (idx >= total) return;
    int X = (int)(idx % p.OW);
    int Y = (int)((idx / p.OW) % p.OH);
    long long nc = idx / ((long long)p.OW * p.OH);
    const float* xc = p.x + nc * p.H * p.W;
    float acc = 0.0f;
    for (int fy = 0; fy < p.fh; ++fy) {
        int u = Y * p.down + fy - p.pady0;
        if (u < 0 || u % p.up) continue;
        u /= p.up;
        if (u >= p.H) continue;
        for (int fx = 0; fx < p.fw; ++fx) {
            int v = X * p.down + fx - p.padx0;
            if (v < 0 || v % p.up) continue;
            v /= p.up;
            if (v >= p.W) continue;
            acc = __builtin_fmaf(p.f[fy * p.fw + fx], xc[(size_t)u * p.W + v], acc);
        }
    }
    if (p.epilogue) {
        int c = (int)(nc % p.C);
        long long n = nc / p.C;
        if (p.dcoef) acc = acc * p.dcoef[nc];
        if (p.noise) acc = acc + p.noise[(p.noise_per_sample ? n * p.OH * p.OW : 0) + (long long)Y * p.OW + X];
        if (p.bias) acc = acc + p.bias[c];
        acc = act_apply(acc, p.act, p.alpha, p.gain, p.clamp);
    }
    p.y[idx] = acc;
}

// upsample2d of the skip image (networks_stylegan2.py:476 -> upfirdn2d.py:341-350: up 2, pad [2,1,2,1], 4x4 filter) in polyphase
// form — only the 2x2 taps that meet non-zero samples of the zero-inserted input, in the generic kernel's order (fy, then fx,
// ascending), so the sums are bit-identical to k_upfirdn2d — fused with `img.add_(y)` (:478): out = upsample(x) + add.
// One thread = 4 consecutive output pixels of a row (OW % 4 == 0).
__global__ __launch_bounds__(256) void k_upsample2x_add(const float* __restrict__ x, const float* __restrict__ f, const float* __restrict__ add,
                                                         float* __restrict__ y, long long NC, int H, int W) {
    const int OW = 2 * W, OH = 2 * H, QW = OW >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NC * OH * QW) return;
    const int q = (int)(idx % QW);
    const int Y = (int)((idx / QW) % OH);
    const long long nc = idx / ((long long)QW * OH);
    const float* xc = x + nc * H * W;
    float ff[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ff[i] = f[i];
    float out[4];
    const int fy0 = Y & 1;  // taps fy0, fy0 + 2 meet rows u = (Y + fy - 2) / 2
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int X = 4 * q + j;
        const int fx0 = j & 1;  // X & 1
        float acc = 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int fy = fy0 + 2 * a, u = (Y + fy - 2) >> 1;  // arithmetic shift: -1 for the row above the image
            if (u < 0 || u >= H) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int fx = fx0 + 2 * b, v = (X + fx - 2) >> 1;
                if (v < 0 || v >= W) continue;
                acc = __builtin_fmaf(fy0 ? (fx0 ? ff[(1 + 2 * a) * 4 + 1 + 2 * b] : ff[(1 + 2 * a) * 4 + 2 * b])
                                         : (fx0 ? ff[(2 * a) * 4 + 1 + 2 * b] : ff[(2 * a) * 4 + 2 * b]),
                                     xc[(size_t)u * W + v], acc);
            }
        }
        out[j] = acc;
    }
    const size_t o = ((size_t)nc * OH + Y) * OW + 4 * q;
    if (add) {
        const float4 a4 = *reinterpret_cast<const float4*>(add + o);
        out[0] += a4.x; out[1] += a4.y; out[2] += a4.z; out[3] += a4.w;
    }
    *reinterpret_cast<float4*>(y + o) = make_float4(out[0], out[1], out[2], out[3]);
}

// 4x4 FIR without resampling (the filter pass after the stride-2 transposed conv), LDS-tiled: a 256-thread block produces
// a 32x32 output tile of one (n,c) plane from a 35x35 input tile.  Every input element is read from HBM/L2 once (the generic
// kernel above re-reads each 16 times through L1).
// y[Y][X] = sum_{fy,fx} f[fy][fx] * x[Y + fy - pady0][X + fx - padx0]
// Round 3: a thread computes FOUR consecutive outputs of one row from a 4 x 7 window = 8 ds_read_b128 (was 2 x 2 outputs from a
// 5 x 5 window = 25 ds_read_b32 at a 2-float lane stride: LDS bank-conflict cycles 0.52 of the LDS cycles, VALU-active 0.66;
// profiles/history/r03a_mfma_util.json).  Row pitch 96 floats: consecutive rows start 32 banks apart (of the 64 a b128 read sees), so the
// 16 lanes of every b128 group — 2-4 rows x 4-8 column quads — hit 64 distinct banks.  Same fma order (fy, then fx): same bits.
#define FIR_PITCH 96
__global__ __launch_bounds__(256) void k_fir4x4_tiled(FirParams p) {
    __shared__ __attribute__((aligned(16))) float tile[35 * FIR_PITCH];
    __shared__ float fs[16];
    const int tid = threadIdx.x;
    const int tiles_x = (p.OW + 31) / 32;
    const int X0 = (blockIdx.x % tiles_x) * 32, Y0 = (blockIdx.x / tiles_x) * 32;
    const long long nc = blockIdx.y;
    if (tid < 16) fs[tid] = p.f[tid];
    {   // the 35 x 36 window (columns X0 - padx0 .. + 35; column 35 only pads the b128 reads).  Round 4: when the rows are 16-byte
        // aligned (the padded intermediate of the up-sampling layers: pitch % 4 == 0, xoff == padx0) a thread loads 4 columns at a
        // time — 315 16-byte loads per channel instead of 1260 4-byte ones; same values, same sums
        const bool vec = (p.pitch & 3) == 0 && p.xoff == p.padx0 && (((uintptr_t)p.x | (uintptr_t)(p.slice * 4)) & 15) == 0;
        if (vec) {
            auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + nc * (long long)p.H * p.pitch), 0, p.H * p.pitch * 4, CONV_RSRC_FLAGS);
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int it = tid + ps * 256, r = it / 9, c4 = it - r * 9;
                if (it >= 35 * 9) break;
                const int u = Y0 + r - p.pady0;
                const int off = (u >= 0 && u < p.H) ? (u * p.pitch + X0 + 4 * c4) * 4 : CONV_OOB;
                f32x4 val = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
                for (int k = 1; k < p.ksplit; ++k) {  // split-K partials, slice order (= k_splitk_reduce)
                    auto rk = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + nc * (long long)p.H * p.pitch + (size_t)k * p.slice), 0, p.H * p.pitch * 4, CONV_RSRC_FLAGS);
                    const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0));
                    val.x += t.x; val.y += t.y; val.z += t.z; val.w += t.w;
                }
                const int v0 = X0 + 4 * c4 - p.padx0;  // logical column of val.x
                val.x = (v0 >= 0 && v0 < p.W) ? val.x : 0.0f;
                val.y = (v0 + 1 >= 0 && v0 + 1 < p.W) ? val.y : 0.0f;
                val.z = (v0 + 2 >= 0 && v0 + 2 < p.W) ? val.z : 0.0f;
                val.w = (v0 + 3 >= 0 && v0 + 3 < p.W) ? val.w : 0.0f;
                *reinterpret_cast<f32x4*>(tile + r * FIR_PITCH + 4 * c4) = val;
            }
        } else {
        const float* xc = p.x + nc * (long long)p.H * p.pitch + p.xoff;
        const int r0 = tid / 36, c = tid - r0 * 36;
        const int v = X0 + c - p.padx0;
        const bool cv = tid < 252 && v >= 0 && v < p.W;
#pragma unroll
        for (int ps = 0; ps < 5; ++ps) {
            const int r = ps * 7 + r0, u = Y0 + r - p.pady0;
            float val = 0.0f;
            if (cv && u >= 0 && u < p.H) {
                const float* q = xc + (size_t)u * p.pitch + v;
                val = q[0];
                for (int k = 1; k < p.ksplit; ++k) val += q[(size_t)k * p.slice];  // split-K partials, slice order (= k_splitk_reduce)
            }
            if (tid < 252) tile[r * FIR_PITCH + c] = val;
        }
        }
    }
    __syncthreads();
    const int lx = (tid & 7) * 4, ly = tid >> 3;
    float win[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(tile + (ly + r) * FIR_PITCH + lx);
        const f32x4 b = *reinterpret_cast<const f32x4*>(tile + (ly + r) * FIR_PITCH + lx + 4);
        win[r][0] = a.x; win[r][1] = a.y; win[r][2] = a.z; win[r][3] = a.w;
        win[r][4] = b.x; win[r][5] = b.y; win[r][6] = b.z; win[r][7] = b.w;
    }
    float dco = 1.0f, bias = 0.0f;
    const int ch = (int)(nc % p.C);
    const long long n = nc / p.C;
    if (p.epilogue) {
        if (p.dcoef) dco = p.dcoef[nc];
        if (p.bias) bias = p.bias[ch];
    }
    const int Y = Y0 + ly, Xb = X0 + lx;
    if (Y >= p.OH || Xb >= p.OW) return;
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float acc = 0.0f;
#pragma unroll
        for (int fy = 0; fy < 4; ++fy)
#pragma unroll
            for (int fx = 0; fx < 4; ++fx) acc = __builtin_fmaf(fs[fy * 4 + fx], win[fy][j + fx], acc);
        out[j] = acc;
    }
    const float* nz = (p.epilogue && p.noise) ? p.noise + (p.noise_per_sample ? n * p.OH * p.OW : 0) + (long long)Y * p.OW + Xb : nullptr;
    float* yo = p.y + (nc * p.OH + Y) * p.OW + Xb;
    // a row of 4-aligned width: the four outputs are one 16-byte store (Xb is a multiple of 4) — provided the caller's y (and noise)
    // are 16-byte aligned, which the C ABI does not demand of them: an offset view takes the scalar path
    const bool vec = (p.OW & 3) == 0 && (((uintptr_t)p.y | (uintptr_t)((p.epilogue && p.noise) ? p.noise : nullptr)) & 15) == 0;
    float nv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (nz) {
        if (vec) { const f32x4 t = *reinterpret_cast<const f32x4*>(nz); nv[0] = t.x; nv[1] = t.y; nv[2] = t.z; nv[3] = t.w; }
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) nv[j] = (Xb + j < p.OW) ? nz[j] : 0.0f;
        }
    }
    if (p.epilogue) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = out[j] * dco;
            if (nz) acc = acc + nv[j];
            acc = acc + bias;
            out[j] = act_apply(acc, p.act, p.alpha, p.gain, p.clamp);
        }
    }
    if (vec) *reinterpret_cast<f32x4*>(yo) = (f32x4){out[0], out[1], out[2], out[3]};
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (Xb + j < p.OW) yo[j] = out[j];
    }
}

// k_fir4x4_tiled writing an activation IMAGE for the layer that follows (FirParams::nstyles): a workgroup = a 32 x 32 output tile of EIGHT consecutive channels (blockIdx.y = (n, c8)),
// in 8 / CPS stages of CPS channels through one LDS image (pitch 40; CPS = 2: 11 KB, 125 VGPRs): the next stage's loads are in
// flight while this one is filtered.  A thread's 4 pixels x 8 channels leave as 4 pieces of hi parts + 4 of lo parts, 512
// contiguous bytes per 8 threads.  Always applies the epilogue.  (Measured at 512^2 x 128 channels, whole up-convolution: channel
// by channel through two buffers 381 us, all eight tiles resident (45 KB, 3 workgroups per CU) 342 us, fp32 output 303 us.)
#define FIRI_PITCH 40
// CPS: channels per LDS stage (8 / CPS stages per tile); WPE: waves per SIMD the register budget is held to
template <bool VEC, int CPS, int WPE>  // VEC: the input rows are 16-byte aligned (decided by the host: fir_rows_aligned)
__global__ __launch_bounds__(256, WPE) void k_fir4x4_img(FirParams p, char* __restrict__ yimg, long long lo_off, unsigned int* sat) {
    __shared__ __attribute__((aligned(16))) float tile[CPS][35 * FIRI_PITCH];
    __shared__ float fs[16];
    const int tid = threadIdx.x;
    const int tiles_x = (p.OW + 31) / 32;
    const int X0 = (blockIdx.x % tiles_x) * 32, Y0 = (blockIdx.x / tiles_x) * 32;
    const long long g = blockIdx.y;  // (n, c8)
    const long long n = g / (p.C >> 3);
    const int c0 = (int)(g - n * (p.C >> 3)) * 8;
    if (tid < 16) fs[tid] = p.f[tid];
    const int HP = p.H * p.pitch;  // floats per channel plane of the input
    const float* xg = p.x + (n * p.C + c0) * (long long)HP;
    // Round 4: rows of the padded intermediate are 16-byte aligned (pitch % 4 == 0, column v at index v + xoff, xoff == padx0): the
    // 35 x 36 window is 315 16-byte loads per channel (2 per thread: rows 0-27, then 28-34) instead of 1260 4-byte ones (5 per
    // thread) — same values (columns outside [0, W) are zeroed in registers), same sums
    constexpr bool vec = VEC;
    auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, 8 * HP * 4, CONV_RSRC_FLAGS);
    // scalar plan (odd pitches: callers' own tensors)
    const int r0 = tid / 36, c = tid - r0 * 36;
    const int vcol = X0 + c - p.padx0;
    const bool cv = tid < 252 && vcol >= 0 && vcol < p.W;
    int off[5];
#pragma unroll
    for (int ps = 0; ps < 5; ++ps) {
        const int u = Y0 + ps * 7 + r0 - p.pady0;
        off[ps] = (cv && u >= 0 && u < p.H) ? (u * p.pitch + vcol + p.xoff) * 4 : CONV_OOB;
    }
    // vector plan: item it = tid + 256 ps -> row it / 9, column quad it % 9
    int voff[2], vr[2], vc4[2];
    bool vm[2][4];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int it = tid + ps * 256, r = it / 9, c4 = it - r * 9;
        const int u = Y0 + r - p.pady0;
        vr[ps] = r; vc4[ps] = c4;
        voff[ps] = (it < 35 * 9 && u >= 0 && u < p.H) ? (u * p.pitch + X0 + 4 * c4) * 4 : CONV_OOB;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int v = X0 + 4 * c4 + e - p.padx0; vm[ps][e] = v >= 0 && v < p.W; }
    }
    struct Stage { float s[VEC ? 1 : CPS][5]; f32x4 v[VEC ? CPS : 1][2]; };
    auto fetch = [&](int half, Stage& st) {
        if constexpr (vec) {
#pragma unroll
            for (int ch = 0; ch < CPS; ++ch)
#pragma unroll
                for (int ps = 0; ps < 2; ++ps)
                    st.v[ch][ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff[ps], (half * CPS + ch) * HP * 4, 0));
            for (int k = 1; k < p.ksplit; ++k) {  // split-K partials, slice order (= k_splitk_reduce)
                auto rk = __builtin_amdgcn_make_buffer_rsrc((void*)(xg + (size_t)k * p.slice), 0, 8 * HP * 4, CONV_RSRC_FLAGS);
                f32x4 t[CPS][2];
#pragma unroll
                for (int ch = 0; ch < CPS; ++ch)
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps)
                        t[ch][ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, voff[ps], (half * CPS + ch) * HP * 4, 0));
#pragma unroll
                for (int ch = 0; ch < CPS; ++ch)
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) { st.v[ch][ps].x += t[ch][ps].x; st.v[ch][ps].y += t[ch][ps].y; st.v[ch][ps].z += t[ch][ps].z; st.v[ch][ps].w += t[ch][ps].w; }
            }
        } else {
#pragma unroll
        for (int ch = 0; ch < CPS; ++ch)
#pragma unroll
            for (int ps = 0; ps < 5; ++ps)
                st.s[ch][ps] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off[ps], (half * CPS + ch) * HP * 4, 0));
        for (int k = 1; k < p.ksplit; ++k) {  // split-K partials, slice order (= k_splitk_reduce); 20 independent loads per slice
            auto rk = __builtin_amdgcn_make_buffer_rsrc((void*)(xg + (size_t)k * p.slice), 0, 8 * HP * 4, CONV_RSRC_FLAGS);
            float t[CPS][5];
#pragma unroll
            for (int ch = 0; ch < CPS; ++ch)
#pragma unroll
                for (int ps = 0; ps < 5; ++ps)
                    t[ch][ps] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rk, off[ps], (half * CPS + ch) * HP * 4, 0));
#pragma unroll
            for (int ch = 0; ch < CPS; ++ch)
#pragma unroll
                for (int ps = 0; ps < 5; ++ps) st.s[ch][ps] += t[ch][ps];
        }
        }
    };
    auto put = [&](const Stage& st) {
        if constexpr (vec) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                if (tid + ps * 256 < 35 * 9) {
#pragma unroll
                    for (int ch = 0; ch < CPS; ++ch) {
                        f32x4 v = st.v[ch][ps];
                        v.x = vm[ps][0] ? v.x : 0.0f; v.y = vm[ps][1] ? v.y : 0.0f; v.z = vm[ps][2] ? v.z : 0.0f; v.w = vm[ps][3] ? v.w : 0.0f;
                        *reinterpret_cast<f32x4*>(&tile[ch][vr[ps] * FIRI_PITCH + 4 * vc4[ps]]) = v;
                    }
                }
            }
        } else {
        if (tid < 252) {
#pragma unroll
            for (int ch = 0; ch < CPS; ++ch)
#pragma unroll
                for (int ps = 0; ps < 5; ++ps) tile[ch][(ps * 7 + r0) * FIRI_PITCH + c] = st.s[ch][ps];
        }
        }
    };
    const int lx = (tid & 7) * 4, ly = tid >> 3;
    const int Y = Y0 + ly, Xb = X0 + lx;
    float out[8][4];
    auto filter = [&](int half) {
#pragma unroll
        for (int ch = 0; ch < CPS; ++ch) {
            float win[4][8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&tile[ch][(ly + r) * FIRI_PITCH + lx]);
                const f32x4 b = *reinterpret_cast<const f32x4*>(&tile[ch][(ly + r) * FIRI_PITCH + lx + 4]);
                win[r][0] = a.x; win[r][1] = a.y; win[r][2] = a.z; win[r][3] = a.w;
                win[r][4] = b.x; win[r][5] = b.y; win[r][6] = b.z; win[r][7] = b.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int fy = 0; fy < 4; ++fy)
#pragma unroll
                    for (int fx = 0; fx < 4; ++fx) acc = __builtin_fmaf(fs[fy * 4 + fx], win[fy][j + fx], acc);
                out[half * CPS + ch][j] = acc;
            }
        }
    };
    Stage va;  // ONE staging set: the next stage is requested once this one sits in LDS and lands under its filtering
    fetch(0, va);
    put(va);
    __syncthreads();
#pragma unroll
    for (int part = 0; part < 8 / CPS; ++part) {
        if (part + 1 < 8 / CPS) fetch(part + 1, va);
        filter(part);
        if (part + 1 < 8 / CPS) {
            __syncthreads();
            put(va);
            __syncthreads();
        }
    }
    if (Y >= p.OH || Xb >= p.OW) return;
    float nv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (p.noise) {
        const float* nz = p.noise + (p.noise_per_sample ? n * p.OH * p.OW : 0) + (long long)Y * p.OW + Xb;
#pragma unroll
        for (int j = 0; j < 4; ++j) nv[j] = (Xb + j < p.OW) ? nz[j] : 0.0f;
    }
    float dco[8], bs[8], ns[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        dco[ch] = p.dcoef ? p.dcoef[n * p.C + c0 + ch] : 1.0f;
        bs[ch] = p.bias ? p.bias[c0 + ch] : 0.0f;
        ns[ch] = p.nstyles[n * p.C + c0 + ch];
    }
    bool bad = false;
    const size_t piece0 = ((size_t)g * p.OH + Y) * p.OW + Xb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f16x8 hv, lv;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            float a = out[ch][j] * dco[ch];
            if (p.noise) a = a + nv[j];
            a = a + bs[ch];
            a = ns[ch] * act_apply(a, p.act, p.alpha, p.gain, p.clamp) * HX_SPLIT_SCALE_X;  // = conv_lstore_w's s * x * 16, bit for bit
            bad = bad || !(__builtin_fabsf(a) <= 65504.0f);
            a = __builtin_fminf(__builtin_fmaxf(a, -65504.0f), 65504.0f);
            hv[ch] = (_Float16)a;
            lv[ch] = (_Float16)(a - (float)hv[ch]);
        }
        if (Xb + j < p.OW) {
            *reinterpret_cast<f16x8*>(yimg + (piece0 + j) * 16) = hv;
            *reinterpret_cast<f16x8*>(yimg + lo_off + (piece0 + j) * 16) = lv;
        }
    }
    if (bad && sat) atomicOr(sat, 1u);
}

// x viewed as [outer][C][inner]
__global__ void k_bias_act(const float* __restrict__ x, const float* __restrict__ b, long long total, int C, long long inner,
                           int act, float alpha, float gain, float clamp, float* __restrict__ y) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    float v = x[idx];
    if (b) v = v + b[(idx / inner) % C];
    y[idx] = act_apply(v, act, alpha, gain, clamp);
}

static inline int chk() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? P3D_OK : (int)e;
}

// Tile variants measured on MI355X and rejected (with the first version of these kernels): 128-channel output tiles (two A tiles
// per wave; 256->256 @256^2: 53 vs 66 TF: fewer, fatter workgroups) and 16-row pixel tiles (58.7 vs 66 TF).
template <int MODE>
static void launch_conv(ConvParams p, hipStream_t st) {
    dim3 grid(((p.GW + CONV_TW - 1) / CONV_TW) * ((p.GH + CONV_TH - 1) / CONV_TH), (p.O + 63) / 64, p.N * p.ksplit);
    if (p.wh && p.wsplit && MODE == 0 && p.GW >= w3_min_w()) {  // the wide tile (the split-K factor was chosen for it: modconv_impl)
        dim3 gw(((p.GW + WX_TW - 1) / WX_TW) * ((p.GH + CONV_TH - 1) / CONV_TH), (p.O + 63) / 64, p.N * p.ksplit);
        if (p.ximg && p.O % 64 == 0 && !env_no_w3()) {
            if (p.rgbp) hipLaunchKernelGGL(k_modconv_w3<true>, gw, dim3(256), 0, st, p);
            else hipLaunchKernelGGL(k_modconv_w3<false>, gw, dim3(256), 0, st, p);
        }
        else if (p.ximg) hipLaunchKernelGGL(k_modconv_w2<true>, gw, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(k_modconv_w2<false>, gw, dim3(256), 0, st, p);
        return;
    }
    if (p.wh && p.wsplit) hipLaunchKernelGGL((k_modconv_h<MODE, true>), grid, dim3(256), 0, st, p);
    else if (p.wh) hipLaunchKernelGGL((k_modconv_h<MODE, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_modconv<MODE>), grid, dim3(256), 0, st, p);
}

#ifndef P3D_KSPLIT_TARGET
#define P3D_KSPLIT_TARGET 256  // workgroups a launch is split towards.  Batch-1 backbone, ms: 64 -> 1.31, 128 -> 1.15, 256 -> 1.08, 512 (rounds 1-2) -> 1.15, 1024 -> 1.36 (profiles/history/r03_notes.txt)
#endif
// split-K factor: small feature maps (4^2..64^2) give too few workgroups for 256 CUs; split the K loop until ~512
static int ksplit_target_image() {  // the image-fed kernels (k_modconv_w3 / k_modconv_up3); P3D_KSPLIT_TARGET_IMG in the environment: A/B runs
    static const int v = getenv("P3D_KSPLIT_TARGET_IMG") ? atoi(getenv("P3D_KSPLIT_TARGET_IMG")) : P3D_KSPLIT_TARGET;
    return v;
}
static int choose_ksplit(int N, int I, int O, int GH, int GW, int tw = CONV_TW) {
    long long wgs = (long long)((GW + tw - 1) / tw) * ((GH + CONV_TH - 1) / CONV_TH) * ((O + 63) / 64) * N;
    int ks = 1;
    if (tw == WX_TW) {
        while (ks < 64 && wgs * ks < ksplit_target_image() && I / (ks * 2) >= 8) ks *= 2;
        return ks;
    }
    // down to ONE 8-channel chunk per workgroup: at batch 1 the 4^2..16^2 layers are a weight stream (9.4 MB for 512 -> 512 x 3x3)
    // that 8..16 workgroups cannot pull in; measured at batch 1: b4.conv1 36 -> see profiles/history/r02_notes.txt
    while (ks < 64 && wgs * ks < P3D_KSPLIT_TARGET && I / (ks * 2) >= 8) ks *= 2;
    return ks;
}

// k_modconv_up3 (image-fed, DMA-pipelined transposed convolution): two-term operands, 16-channel chunks, 32-channel output tiles,
// every map from 4^2 up: below W = 32 its 32-column tile is mostly empty, but the 4^2 .. 16^2 layers are latency, not arithmetic, and
// the pipelined kernel (+ the 5 us conversion pass of its input) still beats k_modconv_up_h there (measured, batch-1 backbone as a
// hipGraph replay: W >= 32 only 0.709 ms, >= 16 0.688, >= 8 0.680-0.689, >= 4 0.689; W = 32: 65.2 -> 47.7 + 4.8 us at 512 -> 512)
#ifndef P3D_UP3_MIN_W
#define P3D_UP3_MIN_W 4
#endif
static int up3_min_w() {  // (P3D_UP3_MIN_W in the environment: A/B runs)
    static const int v = getenv("P3D_UP3_MIN_W") ? atoi(getenv("P3D_UP3_MIN_W")) : P3D_UP3_MIN_W;
    return v;
}
// The kernel-selection switches of the environment (A/B runs) are read ONCE per process: what p3d_modconv2d_workspace_bytes
// answered for a shape stays the size the launch of that shape needs (a caller may cache the query).
// narrowest map the pipelined plain 3x3 kernel (k_modconv_w3, a 32-column tile) takes; P3D_W3_MIN_W in the environment: A/B runs
#ifndef P3D_W3_MIN_W
#define P3D_W3_MIN_W 32
#endif
static int w3_min_w() { static const int v = getenv("P3D_W3_MIN_W") ? atoi(getenv("P3D_W3_MIN_W")) : P3D_W3_MIN_W; return v; }
static bool env_no_up3() { static const bool v = getenv("P3D_NO_UP3") != nullptr; return v; }
static bool env_no_w3() { static const bool v = getenv("P3D_NO_W3") != nullptr; return v; }
static bool up3_applies(int I, int O, int W) { return I % 16 == 0 && O % 32 == 0 && W >= up3_min_w() && !env_no_up3(); }
// k_modconv_up4 (p3d_conv_up4.hip)
int p3d_up4_shape(int N, int O, int H, int W);
int p3d_up4_launch(const ConvParams& p, int rpw, hipStream_t st);
// P3D_UP4=0 in the environment: the round-5 kernels (k_modconv_up3 + FIR pass) for every layer; P3D_UP4_MIN_WGS / P3D_UP4_MIN_I: the
// launch size (workgroups of the 8-row tiling) and K depth from which the one-launch form is taken — defaults 384 and 64: measured
// (p3d_conv_up4.hip, above p3d_up4_shape) it wins from the 128^2 -> 256^2 layer of the backbone up, loses on underfilled launches
// (split-K fills the chip) and on two-chunk K loops.  Read per call (tests switch them), like P3D_UP3_FUSED.
static bool up4_applies(int N, int I, int O, int H, int W) {
    const char* e = getenv("P3D_UP4");
    if (e && atoi(e) == 0) return false;
    const char* m = getenv("P3D_UP4_MIN_WGS");
    const char* mi = getenv("P3D_UP4_MIN_I");
    const long long min_wgs = m ? atoll(m) : 384;
    const long long wgs = (long long)((2 * W + 59) / 60) * ((2 * H + 11) / 12) * (O / 32) * N;
    return wgs >= min_wgs && I >= (mi ? atoi(mi) : 64);
}
static int choose_ksplit_up3(int N, int I, int O, int H, int W) {
    long long wgs = (long long)((W + 1 + WX_TW - 1) / WX_TW) * ((H + 1 + 7) / 8) * (O / 32) * N;
    int ks = 1;
    while (ks < 64 && wgs * ks < ksplit_target_image() && I / (ks * 2) >= 16) ks *= 2;
    return ks;
}

static bool rgb_fusable(int N, int I, int O, int H, int W, int rgbo) {
    return rgbo >= 1 && rgbo <= 4 && I % 16 == 0 && O % 64 == 0 && W >= WX_TW && !env_no_w3() && !getenv("P3D_NO_RGB_FUSE") &&
           choose_ksplit(N, I, O, H, W, WX_TW) == 1;
}

extern "C" {

int p3d_conv_fuses_torgb(int N, int I, int O, int H, int W, int rgb_channels) {
    if (N <= 0 || I <= 0 || O <= 0 || H <= 0 || W <= 0) return 0;
    return rgb_fusable(N, I, O, H, W, rgb_channels) ? 1 : 0;
}
size_t p3d_torgb_partial_bytes(int N, int O, int H, int W, int rgb_channels) {
    return (size_t)(O / 64) * N * rgb_channels * H * W * 4;
}

size_t p3d_modconv2d_workspace_bytes(int N, int I, int O, int H, int W, int up) {
    size_t b = (size_t)N * O * 4 + 256;  // demodulation coefficients
    size_t out_elems = (up == 2) ? (size_t)N * O * (2 * H + 1) * (2 * W + 4) : (size_t)N * O * H * W;  // (up = 2: the intermediate's row pitch)
    if (up == 2) b += out_elems * 4;  // transposed-conv intermediate
    int ks = choose_ksplit(N, I, O, up == 2 ? H + 1 : H, up == 2 ? W + 1 : W);
    if (up == 1 && W >= w3_min_w()) {  // the wide tile of the two-term kernel may split deeper
        const int kw = choose_ksplit(N, I, O, H, W, WX_TW);
        ks = kw > ks ? kw : ks;
    }
    if (ks > 1) b += (size_t)ks * out_elems * 4;  // split-K partial sums
    if (up == 1 && W >= w3_min_w() && I % 16 == 0 && O % 64 == 0) b += (size_t)N * I * H * W * 4 + 256;  // the activation image an fp32 input is turned into (k_modconv_w3)
    if (up == 2 && up3_applies(I, O, W)) {  // k_modconv_up3: its own split-K depth, and the activation image of an fp32 input
        const int k3 = choose_ksplit_up3(N, I, O, H, W);
        if (k3 > ks) b += (size_t)(k3 - (ks > 1 ? ks : 0)) * out_elems * 4;
        b += (size_t)N * I * H * W * 4 + 256;
    }
    return b + 256;
}

// the rule by which an image-consuming layer is accepted (p3d_conv_args.x_img): exported so that a binding cannot drift from it
int p3d_conv_takes_image(int I, int O, int W, int up) {
    if (I <= 0 || O <= 0 || W <= 0 || I % 16 != 0) return 0;
    if (up == 1) return W >= w3_min_w() ? 1 : 0;
    if (up == 2) return up3_applies(I, O, W) ? 1 : 0;
    return 0;
}

// the block's ToRGB layer riding on its conv1 launch (p3d_conv_args.rgb_*)
struct RgbFuse { const float* w; const float* styles; float* partial; int channels; };

// The launch that takes the ToRGB layer along: the pipelined plain 3x3 kernel, unsplit (the activation image as input is the caller's
// business: it is asked for p3d_conv_takes_image as well)
static bool rgb_fusable(int N, int I, int O, int H, int W, int rgbo);

static int modconv_impl(const float* x, int N, int I, int H, int W, const float* w, const void* wh, int wsplit, int O, int ks,
                        const float* styles, int demodulate, const float* dcoef_in, const float* noise, int noise_per_sample, const float* bias,
                        int up, int act, float alpha, float gain, float clamp, const float* fir, float* y, void* workspace,
                        size_t workspace_bytes, void* stream, unsigned int* sat = nullptr, const void* ximg = nullptr,
                        void* yimg = nullptr, const float* ystyles = nullptr, const RgbFuse* rgb = nullptr) {
    if (rgb && !rgb->partial) rgb = nullptr;
    if ((!x && !ximg) || !w || (!styles && !ximg) || (!y && !yimg && !rgb) || (y && yimg && up == 2) || (!y && up == 1 && !rgb) || !workspace || N <= 0 ||
        I <= 0 || O <= 0 || H <= 0 || W <= 0)
        return P3D_E_ARG;
    if (rgb) {
        if (!rgb->w || !rgb->styles) return P3D_E_ARG;
        if (up != 1 || ks != 3 || !ximg || !wh || !wsplit || !rgb_fusable(N, I, O, H, W, rgb->channels)) return P3D_E_RANGE;
    }
    if (ximg) {  // an image input (already modulated by its producer): the pipelined two-term kernels; demodulation must be precomputed
        if (!wh || !wsplit || (demodulate && !dcoef_in)) return P3D_E_ARG;
        if (ks != 3 || I % 16 != 0 || ((uintptr_t)ximg & 15)) return P3D_E_RANGE;
        if (up == 1 ? W < w3_min_w() : !up3_applies(I, O, W)) return P3D_E_RANGE;  // (an up-sampling layer reads images only through k_modconv_up3)
    }
    if (yimg) {      // an image output for a consumer with styles ystyles [N][O]: up = 2: written by the FIR pass INSTEAD of y; up = 1: next to y
        if (!ystyles) return P3D_E_ARG;
        if (O % 8 != 0 || ((uintptr_t)yimg & 15)) return P3D_E_RANGE;
    }
    // 32-bit byte offsets inside one image / the weight tensor (raw buffer addressing)
    if ((long long)I * H * W * 4 >= (1ll << 31) || (long long)O * I * ks * ks * 4 >= (1ll << 31)) return P3D_E_RANGE;
    if (!((ks == 3 && (up == 1 || up == 2)) || (ks == 1 && up == 1))) return P3D_E_RANGE;
    if (up == 2 && !fir) return P3D_E_ARG;
    if (workspace_bytes < p3d_modconv2d_workspace_bytes(N, I, O, H, W, up)) return P3D_E_WORKSPACE;
    // the carve-up below rounds its regions to 256 bytes from the BASE: the budget of the query holds for a 256-byte aligned base
    // (hipMalloc and torch allocations are); anything else is refused rather than written past
    if ((uintptr_t)workspace & 255) return P3D_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    float* dco = (float*)workspace;
    float* tmp = dco + (((size_t)N * O + 63) / 64) * 64;
    // up = 2: the intermediate T has 2W + 1 columns, stored at a pitch of 2W + 4 floats with column ox at index ox + 1, so that the
    // FIR pass reads 16-byte aligned windows (k_fir4x4_*); OW below is that PITCH for everything that indexes T
    const int OH = (up == 2) ? 2 * H + 1 : H, OW = (up == 2) ? 2 * W + 4 : W;
    const size_t out_elems = (size_t)N * O * OH * OW;
    float* part = (up == 2) ? tmp + ((out_elems + 63) / 64) * 64 : tmp;
    if (demodulate && dcoef_in) dco = const_cast<float*>(dcoef_in);  // precomputed by p3d_demod_coefs_f32 (one launch per network)
    else if (demodulate) {
        int waves = N * O;
        hipLaunchKernelGGL(k_demod, dim3((waves * 64 + 255) / 256), dim3(256), 0, st, w, styles, N, O, I, ks * ks, dco);
    }
    const bool wide = wh && wsplit && ks == 3 && up == 1 && W >= w3_min_w();  // k_modconv_w3 / k_modconv_up3
    const bool up3 = wh && wsplit && ks == 3 && up == 2 && up3_applies(I, O, W);  // k_modconv_up3 / k_modconv_up4
    // k_modconv_up4 (round 6): transposed convolution + FIR pass + epilogue in one launch, no intermediate and no split-K — every
    // up-sampling layer whose 8-row tiling alone gives the chip enough workgroups (the 64^2 .. 512^2 maps of the backbone and of
    // the super-resolution); smaller maps keep the split-K form (k_modconv_up3 + reduction + FIR pass)
    const bool up4 = up3 && (act == 0 || (alpha >= 0.0f && alpha <= 1.0f)) && up4_applies(N, I, O, H, W);
    const int ksplit = up4 ? 1 : up3 ? choose_ksplit_up3(N, I, O, H, W)
                           : choose_ksplit(N, I, O, up == 2 ? H + 1 : H, up == 2 ? W + 1 : W, wide ? WX_TW : CONV_TW);
    // An fp32 input of a layer the pipelined kernel can run (O % 64 == 0) is first turned into the image that kernel stages from
    // (one pass, 8 bytes per value; the generator's blocks hand over images and never come here): ONE kernel does the arithmetic
    // of a layer whichever way its input arrives, so both ways give the same bits.
    if ((up3 && !ximg) || (wide && !ximg && O % 64 == 0 && I % 16 == 0 && !env_no_w3())) {
        char* img = (char*)(part + (ksplit > 1 ? ((size_t)ksplit * out_elems + 63) / 64 * 64 : 0));
        img = (char*)(((uintptr_t)img + 255) & ~(uintptr_t)255);
        const long long tot = (long long)N * (I / 8) * H * W;
        hipLaunchKernelGGL(k_act_to_image, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, x, styles, N, I, H * W, img, (long long)N * I * H * W * 2, sat);
        ximg = img;
    }
    ConvParams p;
    p.x = x; p.w = w; p.wh = wh; p.wsplit = wsplit; p.styles = styles; p.dcoef = demodulate ? dco : nullptr; p.noise = noise; p.bias = bias;
    p.N = N; p.I = I; p.O = O; p.H = H; p.W = W; p.ks = ks; p.noise_per_sample = noise_per_sample;
    p.act = act; p.alpha = alpha; p.gain = gain; p.clamp = clamp; p.ksplit = ksplit; p.OH = OH; p.OW = OW; p.sat = sat;
    p.ximg = ximg; p.ximg_lo = (long long)N * I * H * W * 2; p.tox = up == 2 ? 1 : 0;
    p.rgbw = rgb ? rgb->w : nullptr; p.rgbs = rgb ? rgb->styles : nullptr; p.rgbp = rgb ? rgb->partial : nullptr; p.rgbo = rgb ? rgb->channels : 0;
    static const bool xcd_order = !getenv("P3D_NO_XCD_ORDER");  // (A/B runs)
    p.xcd = xcd_order ? 1 : 0;
    // up = 1 with an image output: k_modconv_w3 writes it from its epilogue when it runs unsplit; otherwise a pass over y below
    const bool w3_img = up == 1 && yimg && wide && ximg && O % 64 == 0 && ksplit == 1 && !env_no_w3();
    // up = 2 into an image, unsplit, few input channels: the FIR pass and the epilogue run inside k_modconv_up3<true> (no
    // intermediate).  Measured (tools/conv_layers_time.py, us): 32 -> 256 @128^2 -> 256^2 71 -> 56; 256 -> 128 @256^2 -> 512^2 240 -> 254:
    // with a long K loop the filter's VALU work (76 us chip-wide) and the 1.42 x MFMA work of the overlapping tiles cost more than
    // the intermediate's round trip.  P3D_UP3_FUSED=0/1 in the environment overrides the choice (tests, A/B runs).
    const char* fused_env = getenv("P3D_UP3_FUSED");
    const bool up3_fused = up3 && yimg && ksplit == 1 && (fused_env ? atoi(fused_env) != 0 : I <= 64);
    p.yimg = (w3_img || up3_fused) ? yimg : nullptr; p.yimg_lo = (long long)N * O * H * W * 2; p.ystyles = ystyles; p.fir = fir;
    // conv output goes to: y (up 1, no split), tmp (up 2, no split) or the partial buffer (split-K), raw unless final
    float* conv_dst = (ksplit > 1) ? part : (up == 2 ? tmp : y);
    p.y = conv_dst;
    p.epilogue = (up == 1 && ksplit == 1) ? 1 : 0;
    if (up == 1) {
        p.GH = H; p.GW = W;
        if (ks == 3) launch_conv<0>(p, st); else launch_conv<1>(p, st);
    } else {  // stride-2 transposed conv into [N][O][2H+1][2W+1]: all four output phases in one launch
        p.GH = H + 1; p.GW = W + 1;
        dim3 grid(((p.GW + CONV_TW - 1) / CONV_TW) * ((p.GH + CONV_TH - 1) / CONV_TH), (p.O + 63) / 64, p.N * p.ksplit);
        if (up4) {
            p.y = y; p.yimg = yimg;
            return p3d_up4_launch(p, p3d_up4_shape(N, O, H, W), st);
        }
        if (up3) {
            dim3 g3(((p.GW + WX_TW - 1) / WX_TW) * ((p.GH + 7) / 8), p.O / 32, p.N * p.ksplit);
            if (up3_fused) {
                dim3 gf(((2 * W + 59) / 60) * ((2 * H + 11) / 12), p.O / 32, p.N);
                hipLaunchKernelGGL(k_modconv_up3<true>, gf, dim3(256), 0, st, p);
                return chk();
            }
            hipLaunchKernelGGL(k_modconv_up3<false>, g3, dim3(256), 0, st, p);
        } else if (p.wh && p.wsplit) hipLaunchKernelGGL(k_modconv_up_h<true>, grid, dim3(256), 0, st, p);
        else if (p.wh) hipLaunchKernelGGL(k_modconv_up_h<false>, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(k_modconv_up, grid, dim3(256), 0, st, p);
    }
    // Split-K partial sums.  Up-sampling layer with a SHALLOW split (<= 8 slices: the 64^2 .. 256^2 layers at batch 1): the FIR pass
    // below sums the slices while it loads its tiles — one launch and one round trip of the (2H+1)x(2W+1) intermediate less, the
    // same slice-ordered sum.  Deep splits (the 4^2 .. 32^2 layers, up to 64 slices) keep the separate, chip-wide reduction:
    // measured (profiles/history/r03_notes.txt) both a per-element slice loop inside the FIR pass (4.8 + 5.8 -> 37 us at 64 slices) and an
    // in-launch last-arriver reduction of the plain convolutions (release / ticket / acquire: +15 .. +50 us per layer) lose to it.
    // ... unless the FIR launch itself is a handful of workgroups (one per 32 x 32 tile and eight channels: 64 for 512 channels at 32^2,
    // each then pulling eight slices of its tile through one CU: 17.9 us) — there the chip-wide reduction + a plain FIR pass are faster
    static const long long fir_sums_min_wgs = getenv("P3D_FIR_SUMS_MIN_WGS") ? atoll(getenv("P3D_FIR_SUMS_MIN_WGS")) : 0;  // (measured with 128: 6.5 + 10.2 us instead of 18.2 at 32^2 — one more launch for 1.5 us: left off)
    const long long fir_wgs = (long long)((2 * W + 31) / 32) * ((2 * H + 31) / 32) * ((long long)N * O / (yimg ? 8 : 1));
    const bool fir_sums = up == 2 && ksplit > 1 && ksplit <= 8 && fir_wgs >= fir_sums_min_wgs;
    if (ksplit > 1 && !fir_sums) {
        ReduceParams r;
        r.part = part; r.y = (up == 2) ? tmp : y; r.dcoef = p.dcoef; r.noise = noise; r.bias = bias;
        r.per_slice = (long long)out_elems; r.ksplit = ksplit; r.O = O; r.OHW = OH * OW; r.noise_per_sample = noise_per_sample;
        r.act = act; r.epilogue = (up == 1) ? 1 : 0; r.alpha = alpha; r.gain = gain; r.clamp = clamp;
        if (up == 1 && yimg && !w3_img && !getenv("P3D_NO_REDUCE_IMG")) {  // the sums, the epilogue and the next layer's image in one launch
            hipLaunchKernelGGL(k_splitk_reduce_img, dim3((unsigned)((out_elems + 255) / 256)), dim3(256), 0, st, r, N, ystyles, (_Float16*)yimg,
                               (long long)N * O * H * W, sat);
            return chk();
        }
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((out_elems + 255) / 256)), dim3(256), 0, st, r);
    }
    if (up == 1) {
        if (yimg && !w3_img) {  // (split-K layers, channel counts the pipelined kernel does not take: the image from the finished y)
            const long long tot = (long long)N * (O / 8) * H * W;
            hipLaunchKernelGGL(k_act_to_image, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, y, ystyles, N, O, H * W, (char*)yimg,
                               (long long)N * O * H * W * 2, sat);
        }
        return chk();
    }
    // FIR (pad 1; the caller passes the 4x4 filter already flipped and multiplied by up^2, upfirdn2d.py:193-196) + epilogue
    FirParams q;
    q.x = fir_sums ? part : tmp; q.ksplit = fir_sums ? ksplit : 1; q.slice = (long long)out_elems;
    q.f = fir; q.y = y; q.dcoef = demodulate ? dco : nullptr; q.noise = noise; q.bias = bias;
    q.NC = (long long)N * O; q.C = O; q.H = 2 * H + 1; q.W = 2 * W + 1; q.OH = 2 * H; q.OW = 2 * W; q.fh = 4; q.fw = 4; q.pitch = OW; q.xoff = 1;
    q.up = 1; q.down = 1; q.padx0 = 1; q.pady0 = 1; q.noise_per_sample = noise_per_sample; q.act = act; q.epilogue = 1;
    q.alpha = alpha; q.gain = gain; q.clamp = clamp; q.nstyles = ystyles;
    dim3 grid(((q.OW + 31) / 32) * ((q.OH + 31) / 32), (unsigned)q.NC);
    if (yimg) {
        dim3 gi(grid.x, (unsigned)(q.NC / 8));
        const bool rows_aligned = (q.pitch & 3) == 0 && q.xoff == q.padx0 && (((uintptr_t)q.x | (uintptr_t)(q.slice * 4)) & 15) == 0;
        // two channels per stage: 125 VGPRs, four waves per SIMD (four per stage: 195, two; measured 2-5 % slower)
        if (rows_aligned) hipLaunchKernelGGL((k_fir4x4_img<true, 2, 3>), gi, dim3(256), 0, st, q, (char*)yimg, (long long)N * O * q.OH * q.OW * 2, sat);
        else hipLaunchKernelGGL((k_fir4x4_img<false, 4, 2>), gi, dim3(256), 0, st, q, (char*)yimg, (long long)N * O * q.OH * q.OW * 2, sat);
    } else hipLaunchKernelGGL(k_fir4x4_tiled, grid, dim3(256), 0, st, q);
    return chk();
}

int p3d_modconv2d_f32(const float* x, int N, int I, int H, int W, const float* w, int O, int ks, const float* styles,
                      int demodulate, const float* demod_coefs, const float* noise, int noise_per_sample, const float* bias, int up,
                      int act, float alpha, float gain, float clamp, const float* fir, float* y, void* workspace,
                      size_t workspace_bytes, void* stream) {
    return modconv_impl(x, N, I, H, W, w, nullptr, 0, O, ks, styles, demodulate, demod_coefs, noise, noise_per_sample, bias, up, act,
                        alpha, gain, clamp, fir, y, workspace, workspace_bytes, stream);
}

int p3d_demod_coefs_f32(const float* w2, const float* styles, const int32_t* table, int L, int N, int total_waves, float* d,
                        void* stream) {
    if (!w2 || !styles || !table || !d || L <= 0 || N <= 0 || total_waves <= 0) return P3D_E_ARG;
    if (L > 64) return P3D_E_RANGE;
    hipLaunchKernelGGL(k_demod_plan, dim3((unsigned)((total_waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, w2, styles,
                       (const int*)table, L, N, total_waves, d);
    return chk();
}

static int weights_to_f16(const float* w, int O, int I, int ks, void* w_f16, int split, void* stream) {
    if (!w || !w_f16 || O <= 0 || I <= 0) return P3D_E_ARG;
    if (ks != 1 && ks != 3) return P3D_E_RANGE;
    const long long total = (long long)O * I * ks * ks;
    hipLaunchKernelGGL(k_weights_to_f16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, O, I, ks * ks,
                       (_Float16*)w_f16, split);
    return chk();
}
int p3d_conv_weights_to_f16(const float* w, int O, int I, int ks, void* w_f16, void* stream) {
    return weights_to_f16(w, O, I, ks, w_f16, 0, stream);
}
int p3d_conv_weights_to_f16x2(const float* w, int O, int I, int ks, void* w_f16x2, void* stream) {
    return weights_to_f16(w, O, I, ks, w_f16x2, 1, stream);
}
int p3d_modconv2d_f16mma_f32(const float* x, int N, int I, int H, int W, const float* w, const void* w_f16, int O, int ks,
                             const float* styles, int demodulate, const float* demod_coefs, const float* noise, int noise_per_sample, const float* bias,
                             int up, int act, float alpha, float gain, float clamp, const float* fir, float* y, void* workspace,
                             size_t workspace_bytes, void* stream) {
    if (!w_f16) return P3D_E_ARG;
    if (I % 16 != 0 || ((uintptr_t)w_f16 & 15)) return P3D_E_RANGE;  // a K chunk is 16 channels; 16-byte weight pieces
    return modconv_impl(x, N, I, H, W, w, w_f16, 0, O, ks, styles, demodulate, demod_coefs, noise, noise_per_sample, bias, up, act, alpha,
                        gain, clamp, fir, y, workspace, workspace_bytes, stream);
}

int p3d_modconv2d_ex_f32(const p3d_conv_args* a, void* stream) {
    if (!a) return P3D_E_ARG;
    const void* wh = nullptr;
    int wsplit = 0;
    if (a->mma != P3D_CONV_MMA_F32) {
        if (!a->w_f16) return P3D_E_ARG;
        wsplit = a->mma == P3D_CONV_MMA_F16X2;
        if (a->mma != P3D_CONV_MMA_F16 && !wsplit) return P3D_E_RANGE;
        if (a->I % 16 != 0 || ((uintptr_t)a->w_f16 & 15) || (wsplit && ((size_t)a->O * a->I * a->ks * a->ks * 2) % 16 != 0)) return P3D_E_RANGE;
        wh = a->w_f16;
    }
    if (a->x_img && !wsplit) return P3D_E_RANGE;
    const RgbFuse rgb = {a->rgb_w, a->rgb_styles, a->rgb_partial, a->rgb_channels};
    return modconv_impl(a->x, a->N, a->I, a->H, a->W, a->w, wh, wsplit, a->O, a->ks, a->styles, a->demodulate, a->demod_coefs, a->noise,
                        a->noise_per_sample, a->bias, a->up, a->act, a->alpha, a->gain, a->clamp, a->fir, a->y, a->workspace, a->workspace_bytes,
                        stream, (wsplit || a->y_img) ? (unsigned int*)a->saturated : nullptr, a->x_img, a->y_img, a->y_img_styles, &rgb);
}

size_t p3d_act_image_bytes(int N, int C, int H, int W) { return (size_t)N * C * H * W * 4; }

int p3d_act_to_image_f32(const float* x, const float* styles, int N, int C, int H, int W, void* img, uint32_t* saturated, void* stream) {
    if (!x || !img || N <= 0 || C <= 0 || H <= 0 || W <= 0) return P3D_E_ARG;
    if (C % 8 != 0 || ((uintptr_t)img & 15)) return P3D_E_RANGE;
    const long long total = (long long)N * (C / 8) * H * W;
    hipLaunchKernelGGL(k_act_to_image, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, styles, N, C, H * W,
                       (char*)img, (long long)N * C * H * W * 2, (unsigned int*)saturated);
    return chk();
}

int p3d_modconv2d_f16x2mma_f32(const float* x, int N, int I, int H, int W, const float* w, const void* w_f16x2, int O, int ks,
                               const float* styles, int demodulate, const float* demod_coefs, const float* noise, int noise_per_sample,
                               const float* bias, int up, int act, float alpha, float gain, float clamp, const float* fir, float* y,
                               void* workspace, size_t workspace_bytes, uint32_t* saturated, void* stream) {
    if (!w_f16x2) return P3D_E_ARG;
    if (I % 16 != 0 || ((uintptr_t)w_f16x2 & 15) || ((size_t)O * I * ks * ks * 2) % 16 != 0) return P3D_E_RANGE;
    return modconv_impl(x, N, I, H, W, w, w_f16x2, 1, O, ks, styles, demodulate, demod_coefs, noise, noise_per_sample, bias, up, act,
                        alpha, gain, clamp, fir, y, workspace, workspace_bytes, stream, (unsigned int*)saturated);
}


int p3d_torgb_weights_f32(const float* w, int O, int I, float* w_t, void* stream) {
    if (!w || !w_t || O <= 0 || I <= 0) return P3D_E_ARG;
    if (O > 96) return P3D_E_RANGE;
    const int OP = O <= 32 ? 32 : 96;
    hipLaunchKernelGGL(k_torgb_weights, dim3((unsigned)((I * OP + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, O, I, OP, w_t);
    return chk();
}

int p3d_torgb_f32(const float* x, int N, int I, int H, int W, const float* w_t, int O, const float* styles, const float* bias, float clamp,
                  const float* skip, const float* skip_fir, float* y, void* stream) {
    if (!x || !w_t || !styles || !y || N <= 0 || I <= 0 || O <= 0 || H <= 0 || W <= 0) return P3D_E_ARG;
    if ((skip != nullptr) != (skip_fir != nullptr)) return P3D_E_ARG;
    if (O > 96 || I > 1024 || (long long)I * H * W * 4 >= (1ll << 31) || (skip && ((H & 1) || (W & 1)))) return P3D_E_RANGE;
    TorgbParams p;
    p.x = x; p.wt = w_t; p.styles = styles; p.bias = bias; p.skip = skip; p.skipf = skip_fir; p.y = y;
    p.N = N; p.I = I; p.O = O; p.H = H; p.W = W; p.clamp = clamp;
    const int HW = H * W, MT = O <= 32 ? 1 : 3;
    // PX shape (a wave = 32 pixels x all K) once the map alone gives >= 512 workgroups of 128 pixels; KS (a workgroup = 32 pixels,
    // waves split K) below that
    const bool ks = (long long)N * ((HW + 127) / 128) < 512;
    // small maps of a 96-channel layer: one workgroup per 32-channel tile while that still leaves the chip underfilled
    const bool ms = ks && MT == 3 && (long long)N * ((HW + 31) / 32) * 3 <= 1024 && !getenv("P3D_NO_TORGB_MS");
    const bool pre = ms && I <= 8 * TG_KC && !getenv("P3D_NO_TORGB_PRE");  // everything requested up front (k_torgb<..., PRE>)
    const size_t lds = (size_t)(2 * TG_KC * 32 * (ms ? 1 : MT) + (pre ? 8 * TG_KC : ((I + 63) / 64) * 64)) * 4;
    dim3 grid((unsigned)(ks ? (HW + 31) / 32 : (HW + 127) / 128), (unsigned)N, ms ? 3u : 1u);
    if (lds > 64 * 1024) return P3D_E_RANGE;  // (53 KB at I = 1024, O = 96: inside the default dynamic-LDS limit, no per-device attribute to set)
#define P3D_TORGB(MTV, KSV) hipLaunchKernelGGL((k_torgb<MTV, KSV>), grid, dim3(256), lds, (hipStream_t)stream, p)
    if (MT == 1) { if (ks) P3D_TORGB(1, true); else P3D_TORGB(1, false); }
    else if (pre) hipLaunchKernelGGL((k_torgb<1, true, true, true>), grid, dim3(256), lds, (hipStream_t)stream, p);
    else if (ms) hipLaunchKernelGGL((k_torgb<1, true, true>), grid, dim3(256), lds, (hipStream_t)stream, p);
    else { if (ks) P3D_TORGB(3, true); else P3D_TORGB(3, false); }
    return chk();
}

int p3d_torgb_combine_f32(const float* partial, int tiles, int N, int O, int H, int W, const float* bias, float clamp, const float* skip,
                          const float* skip_fir, float* y, void* stream) {
    if (!partial || !y || tiles <= 0 || N <= 0 || O <= 0 || H <= 0 || W <= 0 || (skip && !skip_fir)) return P3D_E_ARG;
    if (skip && ((H | W) & 1)) return P3D_E_RANGE;
    const long long total = (long long)N * O * H * W;
    if (total * tiles >= (1ll << 40)) return P3D_E_RANGE;
    hipLaunchKernelGGL(k_torgb_combine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial, tiles, N, O, H, W, bias,
                       clamp, skip, skip_fir, y);
    return chk();
}

int p3d_upfirdn2d_f32(const float* x, int64_t NC, int H, int W, const float* f, int fh, int fw, int up, int down, int padx0,
                      int padx1, int pady0, int pady1, float* y, void* stream) {
    if (!x || !f || !y || NC <= 0 || H <= 0 || W <= 0) return P3D_E_ARG;
    if (up < 1 || down < 1 || fh < 1 || fw < 1 || fh > 32 || fw > 32) return P3D_E_RANGE;
    FirParams q;
    q.x = x; q.f = f; q.y = y; q.dcoef = nullptr; q.noise = nullptr; q.bias = nullptr;
    q.NC = NC; q.C = 1; q.H = H; q.W = W;
    q.OH = (H * up + pady0 + pady1 - fh) / down + 1;
    q.OW = (W * up + padx0 + padx1 - fw) / down + 1;
    if (q.OH <= 0 || q.OW <= 0) return P3D_E_RANGE;
    q.fh = fh; q.fw = fw; q.up = up; q.down = down; q.padx0 = padx0; q.pady0 = pady0;
    q.noise_per_sample = 0; q.act = 0; q.epilogue = 0; q.alpha = 0; q.gain = 1; q.clamp = -1; q.ksplit = 1; q.slice = 0; q.nstyles = nullptr; q.pitch = W; q.xoff = 0;
    long long total = q.NC * q.OH * q.OW;
    hipLaunchKernelGGL(k_upfirdn2d, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q);
    return chk();
}

int p3d_upsample2d_add_f32(const float* x, int64_t NC, int H, int W, const float* f4x4, const float* add, float* y, void* stream) {
    if (!x || !f4x4 || !y || NC <= 0 || H <= 0 || W <= 0) return P3D_E_ARG;
    if ((2 * W) % 4 != 0 || (((uintptr_t)y | (uintptr_t)add) & 15)) return P3D_E_RANGE;  // float4 rows: use p3d_upfirdn2d_f32 otherwise
    const long long total = (long long)NC * (2 * H) * ((2 * W) / 4);
    hipLaunchKernelGGL(k_upsample2x_add, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, f4x4, add, y,
                       (long long)NC, H, W);
    return chk();
}

int p3d_bias_act_f32(const float* x, const float* b, int64_t outer, int C, int64_t inner, int act, float alpha, float gain,
                     float clamp, float* y, void* stream) {
    if (!x || !y || outer <= 0 || C <= 0 || inner <= 0) return P3D_E_ARG;
    if (act != 0 && act != 1) return P3D_E_RANGE;
    long long total = outer * C * inner;
    hipLaunchKernelGGL(k_bias_act, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, b, total, C,
                       (long long)inner, act, alpha, gain, clamp, y);
    return chk();
}

}  // extern "C"
