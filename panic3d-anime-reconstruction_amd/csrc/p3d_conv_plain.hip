// p3d_conv_plain.hip — the plain (stride 1) modulated convolutions of the StyleGAN2 synthesis on gfx950: 3x3 / pad 1 (conv1 of every
// block, networks_stylegan2.py:93 -> conv2d_resample.py:136) and 1x1 (ToRGB's fallback, :378) as implicit GEMMs on the matrix cores.
//   k_modconv<MODE>        fp32 operands (v_mfma_f32_32x32x2_f32, exact)
//   k_modconv_h<MODE, S>   f16 / two-term f16 operands, 8 x 16 tile, register-staged (maps narrower than 32 columns, odd channel counts)
//   k_modconv_w2<IMG>      two-term operands, 8 x 32 tile (O % 64 != 0)
//   k_modconv_w3<RGB>      two-term operands, 8 x 32 tile, image-fed, every operand by LDS-DMA, software pipelined (the hot kernel);
//                          RGB: the block's ToRGB sums from the epilogue (p3d_conv_args.rgb_*)
// Dispatch: p3d_launch_conv_plain (the host's choice of split-K depth and epilogue is modconv_impl's, p3d_synthesis.hip).
#include "p3d_conv_stage.hpp"

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
    // P3D_WLAYOUT_PLAIN: the weights arrive as this kernel's LDS image [chunk][O/64][dx][hi|lo][dy][k half][64 o][8] — a group of a
    // chunk is 12 KB of consecutive bytes and a request 1 KB of them, instead of 64 16-byte pieces 9 * I * 2 bytes apart (64 cache lines
    // a request, each shared with other slices' workgroups on other XCDs): measured -10 % on the 256 -> 256 @256^2 layer, bit-identical
    const bool wlds = p.wlayout == P3D_WLAYOUT_PLAIN;
    int wvoff[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int q = u * 256 + tid, which = q / 384, rem = q - which * 384;
        const int dyi = rem >> 7, kh = (rem >> 6) & 1, o = rem & 63;
        wvoff[u] = wlds ? q * 16 : which * LO + (((o0 + o) * 9 + dyi * 3) * p.I + 8 * kh) * 2;
    }
    auto w_rsrc = [&](int chunk) {
        const int ic0 = ic_beg + 16 * chunk;
        const bool in = chunk < nch;
        if (wlds) return w3_rsrc((const char*)p.wh + (size_t)(((in ? ic0 >> 4 : 0) * (p.O >> 6) + (o0 >> 6)) * 3) * W3_GROUP_BYTES, in ? 3u * W3_GROUP_BYTES : 0u);
        return w3_rsrc((const char*)p.wh + (size_t)(in ? ic0 : 0) * 2, in ? (uint32_t)(2 * LO - ic0 * 2) : 0u);
    };
    auto w_piece = [&](const i32x4& rs, int g, int u) {
        w3_dma16(lds0 + g * W3_GROUP_BYTES + wave * 1024 + u * 4096, rs, wvoff[u] + g * (wlds ? W3_GROUP_BYTES : p.I * 2));
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

// Tile variants measured on MI355X and rejected (with the first version of these kernels): 128-channel output tiles (two A tiles
// per wave; 256->256 @256^2: 53 vs 66 TF: fewer, fatter workgroups) and 16-row pixel tiles (58.7 vs 66 TF).
void p3d_launch_conv_plain(const ConvParams& p, hipStream_t st) {
    const bool k3 = p.ks == 3;
    dim3 grid(((p.GW + CONV_TW - 1) / CONV_TW) * ((p.GH + CONV_TH - 1) / CONV_TH), (p.O + 63) / 64, p.N * p.ksplit);
    if (p.wh && p.wsplit && k3 && p.GW >= p3d_w3_min_w()) {  // the wide tile (the split-K factor was chosen for it: modconv_impl)
        dim3 gw(((p.GW + WX_TW - 1) / WX_TW) * ((p.GH + CONV_TH - 1) / CONV_TH), (p.O + 63) / 64, p.N * p.ksplit);
        if (p.ximg && p.O % 64 == 0 && !p3d_env_no_w3()) {
            if (p.rgbp) hipLaunchKernelGGL(k_modconv_w3<true>, gw, dim3(256), 0, st, p);
            else hipLaunchKernelGGL(k_modconv_w3<false>, gw, dim3(256), 0, st, p);
        }
        else if (p.ximg) hipLaunchKernelGGL(k_modconv_w2<true>, gw, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(k_modconv_w2<false>, gw, dim3(256), 0, st, p);
        return;
    }
    if (k3) {
        if (p.wh && p.wsplit) hipLaunchKernelGGL((k_modconv_h<0, true>), grid, dim3(256), 0, st, p);
        else if (p.wh) hipLaunchKernelGGL((k_modconv_h<0, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_modconv<0>), grid, dim3(256), 0, st, p);
    } else {
        if (p.wh && p.wsplit) hipLaunchKernelGGL((k_modconv_h<1, true>), grid, dim3(256), 0, st, p);
        else if (p.wh) hipLaunchKernelGGL((k_modconv_h<1, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_modconv<1>), grid, dim3(256), 0, st, p);
    }
}
