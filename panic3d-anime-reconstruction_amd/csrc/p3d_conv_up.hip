// p3d_conv_up.hip — the stride-2 transposed 3x3 convolution of the up-sampling layer (conv0 of every block,
// conv2d_resample.py:114-127) on gfx950: all four output phases in one workgroup, only the taps that meet non-zero inputs multiplied
// (4 / 2 / 2 / 1 of 9: no zero insertion).
//   k_modconv_up          fp32 operands
//   k_modconv_up_h<S>     f16 / two-term operands, register-staged (shapes the image-fed kernels do not take)
//   k_modconv_up3<FUSED>  two-term operands, image-fed, LDS-DMA, one barrier per chunk; split-K; FUSED: FIR pass + epilogue inside
// (k_modconv_up4, the one-launch layer on the 16-row tile: p3d_conv_up4.hip.)  Dispatch: p3d_launch_conv_up.
#include "p3d_conv_stage.hpp"

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
    const bool wlds = p.wlayout == P3D_WLAYOUT_UP;
    int wvoff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int q = (wave + 4 * i) * 64 + lane, which = q / 576, rem = q - which * 576;
        const int tap = rem >> 6, kh = (rem >> 5) & 1, o = rem & 31;
        // P3D_WLAYOUT_UP: the weights arrive as this kernel's LDS image [chunk][O/32][hi|lo][tap][k half][32 o][8] (18 KB of consecutive bytes
        // per chunk and channel tile, a request = 1 KB of them); else 16-byte pieces gathered out of [hi|lo][O][9][I]
        wvoff[i] = q >= 1152 ? CONV_OOB : wlds ? q * 16 : (o0 + o < p.O) ? which * LO + (((o0 + o) * 9 + tap) * p.I + 8 * kh) * 2 : CONV_OOB;
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
        if (wlds) r.rw = w3_rsrc((const char*)p.wh + (size_t)((ic0 >> 4) * (p.O >> 5) + (o0 >> 5)) * U3_WB, in ? (uint32_t)U3_WB : 0u);
        else r.rw = w3_rsrc((const char*)p.wh + (size_t)ic0 * 2, in ? (uint32_t)(2 * LO - ic0 * 2) : 0u);
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

// ---------------------------------------------------------------------------------------------------------------------
// k_modconv_up5 (round 6): k_modconv_up3<false> — the image-fed transposed two-term convolution that stores the raw four phases
// (intermediate or split-K partials) — for what that kernel is used for since k_modconv_up4 took the filled launches: the 4^2 .. 128^2
// maps of a batch-1 pass, launches that leave ONE multiplying wave per SIMD.  There k_modconv_up3's chunk is a serial chain (measured
// per 16-channel chunk, profiles/r06_notes.txt: LDS reads + barrier 0.65 us, + 54 MFMAs 0.58, + DMA 0.30): a buffer_load ... lds costs
// the issuing wave ~150 clocks and ten of them per chunk are as long as the chunk's MFMAs — with a second workgroup on the CU the other
// wave multiplies meanwhile, alone nobody does.  Here the workgroup brings its own second wave per SIMD:
//   * waves 0-3 multiply (k_modconv_up3's tile: wave w = grid rows 2w, 2w + 1, four phases) and never issue a request;
//   * waves 4-7 request (wave 4 + s: sub-image s of the patch, every fourth weight piece), wait for their requests and meet the others
//     at the chunk's one barrier: the patch ring holds FOUR chunks, requested three ahead, the weights THREE, requested two ahead, and the
//     counted wait leaves the chunk's own requests in flight (a request lands ~1 us after its issue: longer than a chunk's MFMAs);
//   * the A operands of tap q + 1 are read under the MFMAs of tap q; patch rows of 33 items, sub-images padded to whole 1 KB pieces.
// Same products, the same per-accumulator summation order as k_modconv_up3 (chunk, tap, a_hi*b_lo, a_lo*b_hi, a_hi*b_hi): bit-identical.
// LDS 3 x 18 432 + 4 x 20 480 = 137 216 B: one workgroup per CU.
// ---------------------------------------------------------------------------------------------------------------------
#define U5_PW 33
#define U5_SUB 5120
#define U5_PSZ (4 * U5_SUB)
#define U5_NP 4
#define U5_NW 3
#define U5_LDS (U5_NW * U3_WB + U5_NP * U5_PSZ)
__global__ __launch_bounds__(512, 2) void k_modconv_up5(ConvParams p) {
    __shared__ __attribute__((aligned(16))) char lds[U5_LDS];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;
    // ---- which tile.  One workgroup per CU: what matters is that the FULL tiles all start in the first round.  The (H + 1) x (W + 1)
    // grid has a last row / column of its own whenever H % 8 == 0 / W % 32 == 0 — light tiles (3 or 1 of 9 taps) that in row-major tile
    // order alternate with the full ones and take half of the first round's CUs (512 -> 512 @32^2 -> 64^2: 128 full + 192 light
    // workgroups on 256 CUs: the full tiles of the second round doubled the launch).  The launch is a 1-D grid; the dispatcher deals
    // workgroup L to XCD L % 8 as that XCD's (L / 8)-th: every XCD first gets its share of the full tiles, then of the light ones, each
    // share a contiguous run of (K slice, tile, channel tile) with the channel tile fastest (the tiles of a run meet in one L2).
    const int tiles_x = (p.GW + WX_TW - 1) / WX_TW, tiles_y = (p.GH + 7) / 8, OT = p.O >> 5, Z = p.N * p.ksplit;
    const int txf = tiles_x - (p.W % WX_TW == 0 ? 1 : 0), tyf = tiles_y - (p.H % 8 == 0 ? 1 : 0);  // full tile columns / rows
    int tile, otile, zz;
    {
        const int T = gridDim.x, L = blockIdx.x, x = L & 7, m = L >> 3;
        const int nH = txf * tyf, Th = nH * OT * Z;                       // full items
        const int cx = T / 8 + (x < T % 8 ? 1 : 0);                       // workgroups of XCD x
        const int hx = Th / 8 + (x < Th % 8 ? 1 : 0);                     // full items of XCD x ...
        const int hpre = x * (Th / 8) + (x < Th % 8 ? x : Th % 8);        // ... and before it
        const int lpre = (x * (T / 8) + (x < T % 8 ? x : T % 8)) - hpre;  // light items before XCD x (every XCD: cx - hx of them)
        (void)cx;
        int u;
        if (m < hx) {
            u = hpre + m;
            otile = u % OT; u /= OT;
            const int th = u % nH;
            zz = u / nH;
            tile = (th / txf) * tiles_x + th % txf;
        } else {
            u = lpre + (m - hx);
            otile = u % OT; u /= OT;
            const int nL = tiles_x * tiles_y - nH;
            const int tl = u % nL;
            zz = u / nL;
            const int ncol = txf < tiles_x ? tiles_y : 0;                 // the light column first (top to bottom), then the light row
            tile = tl < ncol ? tl * tiles_x + txf : tyf * tiles_x + (tl - ncol);
        }
    }
    const int gy0 = (tile / tiles_x) * 8, gx0 = (tile % tiles_x) * WX_TW;
    const int o0 = otile * 32;
    const int n = zz / p.ksplit, kz = zz - n * p.ksplit;
    const int ic_per = ((p.I + p.ksplit - 1) / p.ksplit + 31) / 32 * 32;
    const int ic_beg = kz * ic_per, ic_end = (ic_beg + ic_per < p.I) ? ic_beg + ic_per : p.I;
    const int nch = ic_end > ic_beg ? (ic_end - ic_beg) >> 4 : 0;
    const int HW = p.H * p.W;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const bool col_edge = gx0 == p.W, row_edge = gy0 == p.H;  // (uniform) the light tiles of the (H + 1) x (W + 1) grid
    if (loader) {
        // ---- the requesting waves.  Patch: wave 4 + s owns sub-image s = (hi|lo, k half): 9 x 33 items in five full pieces
        const int sub_which = wave >> 1, sub_kh = wave & 1;
        int pvoff[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int it = u * 64 + lane;
            const int r = it / U5_PW, c = it - r * U5_PW;
            const int iy = gy0 - 1 + r, ix = gx0 - 1 + c;
            pvoff[u] = (it < 9 * U5_PW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? ((sub_kh * p.H + iy) * p.W + ix) * 16 : CONV_OOB;
        }
        const char* img_base = (const char*)p.ximg + (sub_which ? p.ximg_lo : 0) + (size_t)n * (p.I >> 3) * HW * 16;
        // Weights: 1152 pieces (hi|lo, tap, k half, o) = 18 requests; wave 4 + s issues requests s, s + 4, ... (the fifth round: s = 0, 1)
        const int LO = p.O * 9 * p.I * 2;
        const bool wlds = p.wlayout == P3D_WLAYOUT_UP;
        int wvoff[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = (wave + 4 * i) * 64 + lane, which = q / 576, rem = q - which * 576;
            const int tap = rem >> 6, kh = (rem >> 5) & 1, o = rem & 31;
            wvoff[i] = q >= 1152 ? CONV_OOB : wlds ? q * 16 : (o0 + o < p.O) ? which * LO + (((o0 + o) * 9 + tap) * p.I + 8 * kh) * 2 : CONV_OOB;
        }
        const bool five = wave < 2;
        // chunk >= nch: zero-length resources (zeros into an idle buffer, no traffic, the same request count)
        auto patch_rsrc = [&](int chunk) {
            const bool in = chunk < nch;
            const int ic0 = in ? ic_beg + 16 * chunk : 0;
            return w3_rsrc(img_base + (size_t)(ic0 >> 3) * HW * 16, in ? 2u * HW * 16u : 0u);
        };
        auto weight_rsrc = [&](int chunk) {
            const bool in = chunk < nch;
            const int ic0 = in ? ic_beg + 16 * chunk : 0;
            if (wlds) return w3_rsrc((const char*)p.wh + (size_t)((ic0 >> 4) * (p.O >> 5) + (o0 >> 5)) * U3_WB, in ? (uint32_t)U3_WB : 0u);
            return w3_rsrc((const char*)p.wh + (size_t)ic0 * 2, in ? (uint32_t)(2 * LO - ic0 * 2) : 0u);
        };
        auto patch_chunk = [&](int chunk, int buf) {
            const i32x4 rp = patch_rsrc(chunk);
#pragma unroll
            for (int u = 0; u < 5; ++u) w3_dma16(lds0 + U5_NW * U3_WB + buf * U5_PSZ + wave * U5_SUB + u * 1024, rp, pvoff[u]);
        };
        auto weight_chunk = [&](int chunk, int buf) {
            const i32x4 rw = weight_rsrc(chunk);
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (i < 4 || five) w3_dma16(lds0 + buf * U3_WB + (wave + 4 * i) * 1024, rw, wvoff[i]);
        };
        // A request lands ~1 us after it was issued (MI355X_MICROARCH.md: LDS-DMA issued -> landed) — longer than a chunk's 54 MFMAs
        // (0.7 us): whatever a chunk reads is requested at least a whole chunk before the barrier that publishes it.  Chunk k: the
        // weights of chunk k + 2 (three buffers), the patch of chunk k + 3 (four); the counted wait leaves exactly this chunk's own
        // requests (nine or ten) in flight.  Prologue: patch(0), weights(0) must have landed; patch(1), patch(2), weights(1) may fly.
        patch_chunk(0, 0);
        weight_chunk(0, 0);
        patch_chunk(1, 1);
        patch_chunk(2, 2);
        weight_chunk(1, 1);
        if (five) W3_VMWAIT(15); else W3_VMWAIT(14);
        __builtin_amdgcn_s_barrier();
        for (int k = 0; k < nch; ++k) {
            int wb2 = k + 2;
            wb2 = wb2 - (wb2 / U5_NW) * U5_NW;
            weight_chunk(k + 2, wb2);
            patch_chunk(k + 3, (k + 3) & 3);
            if (five) W3_VMWAIT(10); else W3_VMWAIT(9);
            __builtin_amdgcn_s_barrier();
        }
        W3_VMWAIT(0);  // nothing may land in LDS after this workgroup has given it back
        return;
    }

    // ---- the multiplying waves
    f32x16 acc[4][2];  // [phase = 2 py + px][row of the wave's pair]
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ph][t][r] = 0.0f;
    const int blane = half * U5_SUB + ((2 * wave) * U5_PW + j) * 16;
    const int alane = (half * 32 + j) * 16;
    // (phase, tap, input) of the nine products: input 0 = x[y][x], 1 = x[y][x-1], 2 = x[y-1][x], 3 = x[y-1][x-1] (k_modconv_up3's order)
    const int PH[9] = {0, 1, 2, 3, 0, 2, 0, 1, 0}, TP[9] = {0, 1, 3, 4, 2, 5, 6, 7, 8}, BO[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
    __builtin_amdgcn_s_barrier();  // (the prologue's)
    auto run = [&](auto QMc, auto NTc, auto W0c) {
        constexpr int QM = decltype(QMc)::value, NT = decltype(NTc)::value;
        constexpr bool W0 = decltype(W0c)::value;
        constexpr int QF = QM & -QM;  // (lowest valid tap)
        for (int k = 0; k < nch; ++k) {
            if (!W0 || wave == 0) {
                const char* pb = lds + U5_NW * U3_WB + (k & 3) * U5_PSZ + blane;
                const char* wb = lds + (k - (k / U5_NW) * U5_NW) * U3_WB + alane;
                // rows 2w, 2w + 1, 2w + 2 of the patch x columns j (dx = -1), j + 1 (dx = 0), hi and lo
                f16x8 bh[3][2], bl[3][2];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        bh[r][c] = *reinterpret_cast<const f16x8*>(pb + (r * U5_PW + c) * 16);
                        bl[r][c] = *reinterpret_cast<const f16x8*>(pb + 2 * U5_SUB + (r * U5_PW + c) * 16);
                    }
                f16x8 ah[2], al[2];
                int cu = 0;
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    if (!((QM >> q) & 1)) continue;
                    if ((1 << q) == QF) {
                        ah[0] = *reinterpret_cast<const f16x8*>(wb + TP[q] * 64 * 16);
                        al[0] = *reinterpret_cast<const f16x8*>(wb + (9 + TP[q]) * 64 * 16);
                    }
                    int qn = -1;  // the next valid tap: its weights are read under this tap's MFMAs
#pragma unroll
                    for (int z = 8; z > q; --z)
                        if ((QM >> z) & 1) qn = z;
                    if (qn >= 0) {
                        ah[cu ^ 1] = *reinterpret_cast<const f16x8*>(wb + TP[qn] * 64 * 16);
                        al[cu ^ 1] = *reinterpret_cast<const f16x8*>(wb + (9 + TP[qn]) * 64 * 16);
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int r = 1 + t - (BO[q] >> 1), c = 1 - (BO[q] & 1);
                        acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cu], bl[r][c], acc[PH[q]][t], 0, 0, 0);
                        acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cu], bh[r][c], acc[PH[q]][t], 0, 0, 0);
                        acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cu], bh[r][c], acc[PH[q]][t], 0, 0, 0);
                    }
                    cu ^= 1;
                }
            }
            __builtin_amdgcn_s_barrier();
        }
    };
    {
        using std::integral_constant;
        if (!col_edge && !row_edge) run(integral_constant<int, 0x1FF>{}, integral_constant<int, 2>{}, integral_constant<bool, false>{});
        else if (!row_edge) run(integral_constant<int, 0x130>{}, integral_constant<int, 2>{}, integral_constant<bool, false>{});
        else if (!col_edge) run(integral_constant<int, 0x1C0>{}, integral_constant<int, 1>{}, integral_constant<bool, true>{});
        else run(integral_constant<int, 0x100>{}, integral_constant<int, 1>{}, integral_constant<bool, true>{});
    }
    // ---- raw store (k_modconv_up3's): a lane owns both column phases (ox = 2 gx, 2 gx + 1) of its grid point: one 8-byte store per
    // (row phase, channel), 32 lanes = 256 contiguous bytes; the last grid column (gx = W) has only px = 0: a 4-byte store of its own
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

// kind: 0 = the register-staged kernels (by operand mode), 1 = k_modconv_up3<false> (raw intermediate / split-K partials),
// 2 = k_modconv_up3<true> (FIR pass and epilogue inside; p.H, p.W: the layer's input map)
void p3d_launch_conv_up(const ConvParams& p, int kind, hipStream_t st) {
    if (kind == 2) {
        dim3 gf(((2 * p.W + 59) / 60) * ((2 * p.H + 11) / 12), p.O / 32, p.N);
        hipLaunchKernelGGL(k_modconv_up3<true>, gf, dim3(256), 0, st, p);
    } else if (kind == 1) {
        dim3 g3(((p.GW + WX_TW - 1) / WX_TW) * ((p.GH + 7) / 8), p.O / 32, p.N * p.ksplit);
        // k_modconv_up5 (one workgroup per CU, deep prefetch) while the launch leaves the chip under-filled anyway: up to two workgroups
        // per CU in k_modconv_up3's terms; P3D_UP5=0 / 1 in the environment: never / always (tests, A/B runs)
        const char* e5 = getenv("P3D_UP5");  // (read per call: tests switch it)
        const int up5 = e5 ? atoi(e5) : -1;
        const long long wgs = (long long)g3.x * g3.y * g3.z;
        // ... and a K slice is at least eight chunks: the deep ring's prologue requests three patches and two weight chunks before the first
        // MFMA — on the four-chunk slices of the 16^2 -> 32^2 layer it costs more than it hides (23.7 against 19.5 us, same lease);
        // 512 -> 512 @32^2 -> 64^2: 41.5 -> 30.5 us, 512 -> 256 @64^2 -> 128^2: 45.2 -> 42.8
        if (up5 != 0 && p.O % 32 == 0 && (up5 == 1 || (wgs <= 640 && p.I / p.ksplit >= 128))) hipLaunchKernelGGL(k_modconv_up5, dim3((unsigned)wgs), dim3(512), 0, st, p);
        else hipLaunchKernelGGL(k_modconv_up3<false>, g3, dim3(256), 0, st, p);
    } else {
        dim3 grid(((p.GW + CONV_TW - 1) / CONV_TW) * ((p.GH + CONV_TH - 1) / CONV_TH), (p.O + 63) / 64, p.N * p.ksplit);
        if (p.wh && p.wsplit) hipLaunchKernelGGL(k_modconv_up_h<true>, grid, dim3(256), 0, st, p);
        else if (p.wh) hipLaunchKernelGGL(k_modconv_up_h<false>, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(k_modconv_up, grid, dim3(256), 0, st, p);
    }
}
