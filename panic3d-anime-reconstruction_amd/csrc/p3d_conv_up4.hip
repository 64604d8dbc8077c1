// p3d_conv_up4.hip — k_modconv_up4 (round 6): the up-sampling layer of a synthesis block in ONE launch on gfx950 —
// the stride-2 transposed 3x3 two-term convolution (conv2d_resample.py:114-127), the 4x4 FIR pass that follows it
// (upfirdn2d.py:169-213 with the [1,3,3,1] x [1,3,3,1] filter, networks_stylegan2.py:84-94) and the layer's epilogue
// (demodulation, noise, bias, lrelu, gain, clamp: bias_act.py:93-122, networks_stylegan2.py:342-349), for EVERY input width, on an
// 8-wave workgroup.  It replaces k_modconv_up3<false> + k_fir4x4_img (the (2H+1) x (2W+4) fp32 intermediate: 135 MB written and read
// back at 256 -> 128 @256^2 -> 512^2) wherever a launch does not need split-K, and k_modconv_up3<true> (4 waves, 8 x 32 grid points).
//
//   workgroup = NW waves (template <NW, RPW, NP>); tile = GR x 32 grid points ((H+1) x (W+1) grid, four output phases each) x 32 output
//               channels, GR = NW RPW;  wave w = grid rows RPW w .. RPW w + RPW - 1 (lane j = column j) x 4 phases = 4 RPW accumulators
//   <8, 2, 3>   16 x 32 grid points -> a 32 x 64 tile of the intermediate -> 28 x 60 outputs: tiles advance by 14 x 30 grid points,
//               1.22 x the MFMA work of the un-fused form (k_modconv_up3<true>'s 8 x 32 tile: 1.42 x); 152 KB of LDS, one workgroup per CU
//   <4, 2, 2>   8 x 32 grid points (12 x 60 outputs) on four waves, the patch double buffered: 78 KB, two workgroups per CU
//   pipeline    per 16-channel chunk a wave issues 9 taps x RPW rows x 3 two-term products = 27 RPW MFMAs (32x32x16 f16); the weights
//               of a chunk (18 KB: [hi|lo][tap][k half][32 o][8]) are double buffered, the patch ([hi|lo][k half][GR+1][33][8]) sits
//               in a ring of NP = 3 buffers: the weights of chunk k+1 and the patch of chunk k+2 are requested under the MFMAs of
//               chunk k (inline-asm LDS-DMA, one piece after each of the first taps), the ONE barrier per chunk waits with a counted
//               vmcnt that leaves the youngest patch in flight — a patch has more than a whole chunk to land, the weights
//               (L2-resident) two thirds of one.  The A operands of tap q+1 are read while tap q's MFMAs issue (explicitly: the DMA
//               asm statements are scheduling barriers the compiler cannot software-pipeline across).
//   epilogue    the accumulators go to LDS sixteen channels at a time (the pipeline's buffers) as the intermediate tile T, column c
//               at index c - 1 (column 0 is never read) so that every 4-pixel window starts on a 16-byte boundary; a thread filters
//               4 pixels x 8 channels (two ds_read_b128 per tile row and channel), applies the epilogue and stores whole 16-byte
//               pieces of the consumer's activation image (or float4s of the fp32 tensor).
// Same products, the same per-accumulator summation order (chunk, tap order of k_modconv_up3, a_hi*b_lo, a_lo*b_hi, a_hi*b_hi) and
// the same 16-term fma chain of the filter as k_modconv_up3 + k_fir4x4_img / k_fir4x4_tiled: results are BIT-IDENTICAL to the
// two-pass form (tests/test_hip_synthesis.py::test_up4_*).
#include "p3d_conv_common.hpp"

namespace {

template <int NW, int RPW, int NP>
struct Up4 {
    static constexpr int TH = NW * 64;              // threads
    static constexpr int GR = NW * RPW;             // grid rows of a tile
    static constexpr int PR = GR + 1;               // patch rows
    static constexpr int OR = 2 * GR - 4;           // output rows of a tile (12 / 28); 60 output columns
    static constexpr int PW = WX_TW + 1;            // patch columns = LDS row pitch: dx in {-1, 0} only (k_modconv_w3's 34 holds dx = +1 as well)
    static constexpr int ITEMS = PR * PW;           // 16-byte items of one (hi|lo, k half) sub-image of the patch: 297 / 561
    static constexpr int NPS = (ITEMS + 63) / 64;   // DMA pieces per sub-image: 5 / 9; a sub-image is padded to whole pieces, so the last
    static constexpr int SUB = NPS * 1024;          // piece runs with every lane (the lanes past ITEMS request nothing and zero the padding)
    static constexpr int PSZ = 4 * SUB;             // one patch buffer: 20 480 / 36 864
    static constexpr int PARTS = NW / 4;            // waves per sub-image: they take its pieces alternately
    static constexpr int NPW = (NPS + PARTS - 1) / PARTS;  // patch pieces per wave at most
    static constexpr int WPW = (18 + NW - 1) / NW;  // weight pieces per wave at most (18 per chunk)
    static constexpr int PPT = (WPW + NPW + 8) / 9; // pieces issued after each of the first taps
    static constexpr int PD = NP - 1;               // the patch is requested PD chunks ahead
    static constexpr int PIPE = 2 * U3_WB + NP * PSZ;
    static constexpr int TPS = 2 * GR * 64;         // floats per channel plane of T: [2 GR rows][64], column c at index c - 1
    static constexpr int T_OFF = 16;                // (index -1 of the first row stays inside the array)
    static constexpr int TBYTES = T_OFF + 16 * TPS * 4;
    static constexpr int MAIN = PIPE > TBYTES ? PIPE : TBYTES;
    static constexpr int EPI_OFF = MAIN;            // d [32], bias [32], 16 x next styles [32]
    static constexpr int NZ_OFF = EPI_OFF + 96 * 4;
    static constexpr int LDS = NZ_OFF + OR * 64 * 4;  // noise of the tile's outputs [OR][64]
    static constexpr int FIR_ITEMS = 2 * OR * 16;   // (8-channel group, output row, 4-pixel column group; 15 of 16 groups are real)
    static constexpr int FIR_PASSES = (FIR_ITEMS + TH - 1) / TH;
};

template <int NW, int RPW, int NP>
__global__ __launch_bounds__(NW * 64, 2) void k_modconv_up4(ConvParams p) {
    using C = Up4<NW, RPW, NP>;
    static_assert(C::LDS <= 160 * 1024 && (NW == 8 || C::LDS <= 80 * 1024), "LDS of one CU (four waves: two workgroups per CU)");
    static_assert(NW == 4 || NW == 8, "four sub-images of the patch on four or eight waves");
    __shared__ __attribute__((aligned(16))) char lds[C::LDS];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int OHo = 2 * p.H, OWo = 2 * p.W;
    const int tiles_x = (OWo + 59) / 60;
    const WgOrder wo = p3d_wg_order(p.xcd != 0);
    // outputs [OR ty, OR ty + OR) x [60 tx, 60 tx + 60) need the intermediate's rows OR ty - 1 .. and columns 60 tx - 1 ..: grid origin -1
    const int gy0 = (wo.tile / tiles_x) * (C::GR - 2) - 1, gx0 = (wo.tile % tiles_x) * 30 - 1;
    const int o0 = wo.otile * 32;
    const int n = wo.z;
    const int nch = p.I >> 4;
    const int HW = p.H * p.W;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // ---- DMA plans.  Patch: sub-image s = (hi|lo, k half) belongs to waves s and s + 4, which take its pieces alternately
    const int sub = wave & 3, part = wave >> 2;
    const int sub_which = sub >> 1, sub_kh = sub & 1;
    int pvoff[C::NPW];
#pragma unroll
    for (int i = 0; i < C::NPW; ++i) {
        // (eight waves, nine pieces: the second wave of a sub-image requests the last piece once more instead of skipping a turn — the same
        // bytes to the same place, and every wave issues the same number of requests: no branch in the K loop, one counted wait)
        const int pc = part + C::PARTS * i < C::NPS ? part + C::PARTS * i : C::NPS - 1;
        const int it = pc * 64 + lane;
        const int r = it / C::PW, c = it - r * C::PW;
        const int iy = gy0 - 1 + r, ix = gx0 - 1 + c;
        pvoff[i] = (it < C::ITEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? ((sub_kh * p.H + iy) * p.W + ix) * 16 : CONV_OOB;
    }
    const char* img_base = (const char*)p.ximg + (sub_which ? p.ximg_lo : 0) + (size_t)n * (p.I >> 3) * HW * 16;
    // Weights: 1152 pieces (hi|lo, tap, k half, o) = 18 instructions: wave w issues instructions w, w + NW, ... (the last round: waves 0, 1)
    const int LO = p.O * 9 * p.I * 2;
    const bool wlds = p.wlayout == P3D_WLAYOUT_UP;
    int wvoff[C::WPW];
#pragma unroll
    for (int i = 0; i < C::WPW; ++i) {
        const int q = (wave + NW * i) * 64 + lane, which = q / 576, rem = q - which * 576;
        const int tap = rem >> 6, kh = (rem >> 5) & 1, o = rem & 31;
        wvoff[i] = q >= 1152 ? CONV_OOB : wlds ? q * 16 : which * LO + (((o0 + o) * 9 + tap) * p.I + 8 * kh) * 2;  // (P3D_WLAYOUT_UP: k_modconv_up3's image)
    }
    // chunk >= nch: a zero-length resource (zeros into an idle buffer, no traffic, the same instruction count)
    auto patch_rsrc = [&](int chunk) {
        const bool in = chunk < nch;
        return w3_rsrc(img_base + (size_t)(in ? 2 * chunk : 0) * HW * 16, in ? 2u * HW * 16u : 0u);
    };
    auto weight_rsrc = [&](int chunk) {
        const bool in = chunk < nch;
        const int ic0 = in ? 16 * chunk : 0;
        if (wlds) return w3_rsrc((const char*)p.wh + (size_t)((in ? chunk : 0) * (p.O >> 5) + (o0 >> 5)) * U3_WB, in ? (uint32_t)U3_WB : 0u);
        return w3_rsrc((const char*)p.wh + (size_t)ic0 * 2, in ? (uint32_t)(2 * LO - ic0 * 2) : 0u);
    };
    uint32_t pdst[C::NPW];  // (wave-uniform)
#pragma unroll
    for (int i = 0; i < C::NPW; ++i) {
        const int pc = part + C::PARTS * i < C::NPS ? part + C::PARTS * i : C::NPS - 1;
        pdst[i] = lds0 + 2 * U3_WB + sub * C::SUB + pc * 1024;
    }
    auto patch_piece = [&](const i32x4& rs, int buf, int i) {  // i: compile-time after unrolling
        w3_dma16(pdst[i] + buf * C::PSZ, rs, pvoff[i]);
    };
    auto weight_piece = [&](const i32x4& rs, int buf, int i) {
        if (NW * i + NW > 18 && wave + NW * i >= 18) return;   // (wave-uniform; only the last round — pieces 16, 17 — is not shared by all waves)
        w3_dma16(lds0 + buf * U3_WB + (wave + NW * i) * 1024, rs, wvoff[i]);
    };
    // piece s of a chunk's requests: 0 .. WPW-1 the weights of the next chunk, then this wave's patch pieces; PPT of them are issued
    // after each of the first taps
    auto issue1 = [&](const i32x4& rw, int wbuf, const i32x4& rp, int pbuf, int s) {
        if (s < C::WPW) weight_piece(rw, wbuf, s);
        else if (s < C::WPW + C::NPW) patch_piece(rp, pbuf, s - C::WPW);
    };
    auto issue = [&](const i32x4& rw, int wbuf, const i32x4& rp, int pbuf, int q) {
#pragma unroll
        for (int e = 0; e < C::PPT; ++e) issue1(rw, wbuf, rp, pbuf, q * C::PPT + e);
    };

    f32x16 acc[4][RPW];  // [phase = 2 py + px][row of the wave]
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int t = 0; t < RPW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ph][t][r] = 0.0f;
    // patch row RPW w + 1 + t is grid row gy0 + RPW w + t; column j + 1 is grid column gx0 + j
    const int blane = half * C::SUB + ((RPW * wave) * C::PW + j) * 16;
    const int alane = (half * 32 + j) * 16;
    // (phase, tap, input) of the nine products: input 0 = x[y][x], 1 = x[y][x-1], 2 = x[y-1][x], 3 = x[y-1][x-1] (k_modconv_up3's order)
    const int PH[9] = {0, 1, 2, 3, 0, 2, 0, 1, 0}, TP[9] = {0, 1, 3, 4, 2, 5, 6, 7, 8}, BO[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
    // a wave whose grid rows all lie outside [0, H] (the first row of the top tiles, the rows below the map in the bottom tiles)
    // multiplies zeros only: it keeps its DMA duty and the barriers and leaves the matrix core to the others
    bool wave_active = false;
#pragma unroll
    for (int t = 0; t < RPW; ++t) wave_active = wave_active || (gy0 + RPW * wave + t >= 0 && gy0 + RPW * wave + t <= p.H);

    // ---- prologue: patch(0 .. PD-1), weights(0)
    {
        const i32x4 rw = weight_rsrc(0);
#pragma unroll
        for (int c = 0; c < C::PD; ++c) {
            const i32x4 rp = patch_rsrc(c);
#pragma unroll
            for (int i = 0; i < C::NPW; ++i) patch_piece(rp, c, i);
            if (c == 0) {
#pragma unroll
                for (int i = 0; i < C::WPW; ++i) weight_piece(rw, 0, i);
            }
        }
    }
    // ---- the epilogue's constants and the tile's noise into LDS (read after the K loop's barriers)
    {
        float* epi = reinterpret_cast<float*>(lds + C::EPI_OFF);
        if (tid < 32) {
            const int ch = o0 + tid;
            epi[tid] = p.dcoef ? p.dcoef[(size_t)n * p.O + ch] : 1.0f;
            epi[32 + tid] = p.bias ? p.bias[ch] : 0.0f;
            epi[64 + tid] = p.ystyles ? p.ystyles[(size_t)n * p.O + ch] * HX_SPLIT_SCALE_X : 0.0f;  // (x 16 is exact: (s v) 16 == (16 s) v)
        }
        float* nzs = reinterpret_cast<float*>(lds + C::NZ_OFF);
        const float* nz = p.noise ? p.noise + (p.noise_per_sample ? (long long)n * OHo * OWo : 0) : nullptr;
        for (int i = tid; i < C::OR * 64; i += C::TH) {
            const int r = i >> 6, c = i & 63;
            const int Y = 2 * gy0 + 2 + r, X = 2 * gx0 + 2 + c;
            nzs[i] = (nz && c < 60 && X < OWo && Y < OHo) ? nz[(long long)Y * OWo + X] : -0.0f;  // (a + -0 == a for every a, -0 included)
        }
    }
    auto wait_chunk = [&]() {  // everything but the youngest patch has landed (this wave's pieces; the barrier covers the others')
        if constexpr (C::PD < 2) W3_VMWAIT(0);
        else W3_VMWAIT(5);
    };
    static_assert(C::PD < 2 || C::NPW == 5, "the counted wait above leaves one patch = five requests of this wave in flight");
    wait_chunk();
    __builtin_amdgcn_s_barrier();

    // A wave whose rows are all outside the map runs the second loop: its DMA duty and the barriers only (two loops, not one loop
    // with a branch inside: the accumulators then live in the same registers all the way)
    if (wave_active) {
        int pcur = 0;  // patch buffer of chunk k
        for (int k = 0; k < nch; ++k) {
            const i32x4 rw = weight_rsrc(k + 1), rp = patch_rsrc(k + C::PD);
            const int wnext = (k + 1) & 1;
            int pnext = pcur + C::PD;
            pnext = pnext >= NP ? pnext - NP : pnext;
            const char* pb = lds + 2 * U3_WB + pcur * C::PSZ + blane;
            const char* wb = lds + (k & 1) * U3_WB + alane;
            // rows RPW w .. RPW w + RPW of the patch x columns j (dx = -1), j + 1 (dx = 0), hi and lo
            f16x8 bh[RPW + 1][2], bl[RPW + 1][2];
#pragma unroll
            for (int r = 0; r < RPW + 1; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    bh[r][c] = *reinterpret_cast<const f16x8*>(pb + (r * C::PW + c) * 16);
                    bl[r][c] = *reinterpret_cast<const f16x8*>(pb + 2 * C::SUB + (r * C::PW + c) * 16);
                }
            f16x8 ah[2], al[2];
            ah[0] = *reinterpret_cast<const f16x8*>(wb + TP[0] * 64 * 16);
            al[0] = *reinterpret_cast<const f16x8*>(wb + (9 + TP[0]) * 64 * 16);
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int cur = q & 1;
                if (q < 8) {  // the next tap's weights, read under this tap's MFMAs
                    ah[cur ^ 1] = *reinterpret_cast<const f16x8*>(wb + TP[q + 1] * 64 * 16);
                    al[cur ^ 1] = *reinterpret_cast<const f16x8*>(wb + (9 + TP[q + 1]) * 64 * 16);
                }
#pragma unroll
                for (int t = 0; t < RPW; ++t) {
                    const int r = 1 + t - (BO[q] >> 1), c = 1 - (BO[q] & 1);
                    acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bl[r][c], acc[PH[q]][t], 0, 0, 0);
                    acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur], bh[r][c], acc[PH[q]][t], 0, 0, 0);
                    acc[PH[q]][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bh[r][c], acc[PH[q]][t], 0, 0, 0);
                    if (t == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue(rw, wnext, rp, pnext, q);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            wait_chunk();
            __builtin_amdgcn_s_barrier();
            pcur = pcur + 1 == NP ? 0 : pcur + 1;
        }
    } else {
        int pcur = 0;
        for (int k = 0; k < nch; ++k) {
            const i32x4 rw = weight_rsrc(k + 1), rp = patch_rsrc(k + C::PD);
            int pnext = pcur + C::PD;
            pnext = pnext >= NP ? pnext - NP : pnext;
#pragma unroll
            for (int s = 0; s < 9; ++s) issue(rw, (k + 1) & 1, rp, pnext, s);
            wait_chunk();
            __builtin_amdgcn_s_barrier();
            pcur = pcur + 1 == NP ? 0 : pcur + 1;
        }
    }
    W3_VMWAIT(0);  // nothing may land in LDS once the buffers are reused
    __builtin_amdgcn_s_barrier();
    // ---- FIR + epilogue.  Output (Y, X) = (2 gy0 + 1 + ly, 2 gx0 + 1 + lx), ly in [1, OR], lx in [1, 60], reads the local intermediate
    // rows ly .. ly + 3, columns lx .. lx + 3 (= T[Y - 1 + fy][X - 1 + fx]); grid points outside the map gave exact zeros (the FIR
    // pass's zero padding).  Sixteen channels at a time through LDS.
    float* T = reinterpret_cast<float*>(lds + C::T_OFF);
    const float* epi = reinterpret_cast<const float*>(lds + C::EPI_OFF);
    const float* nzs = reinterpret_cast<const float*>(lds + C::NZ_OFF);
    float fs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)  // (scalar registers: sixteen VGPRs back to the windows' two register sets)
        fs[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.fir[i])));
    const bool wimg = p.yimg != nullptr;
    // act_apply with its switches folded into constants (the host sends only 0 <= alpha <= 1 here): lrelu = max(v, v alpha) — v alpha >= v
    // exactly when v < 0 —, linear: alpha = 1; no clamp: +inf
    const float alpha = p.act == 1 ? p.alpha : 1.0f, gain = p.gain, cl = p.clamp >= 0.0f ? p.clamp : __builtin_inff();
    const long long lo_off = (long long)p.N * p.O * OHo * OWo * 2;
    const bool vec_y = (OWo & 3) == 0 && (((uintptr_t)p.y) & 15) == 0;
    uint32_t badbits = 0;
#pragma unroll 1
    for (int bt = 0; bt < 2; ++bt) {
        if (bt) __builtin_amdgcn_s_barrier();  // (every thread is done reading the first sixteen channels)
        {
            float* Tw = T + (4 * half) * C::TPS + (2 * RPW * wave) * 64 + 2 * j - 1;
            if (bt == 0) {
#pragma unroll
                for (int t = 0; t < RPW; ++t)
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr)
                            Tw[((rr & 3) + 8 * (rr >> 2)) * C::TPS + (2 * t + (ph >> 1)) * 64 + (ph & 1)] = acc[ph][t][rr] * HX_SPLIT_UNSCALE;
            } else {
#pragma unroll
                for (int t = 0; t < RPW; ++t)
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr)
                            Tw[((rr & 3) + 8 * (rr >> 2)) * C::TPS + (2 * t + (ph >> 1)) * 64 + (ph & 1)] = acc[ph][t][8 + rr] * HX_SPLIT_UNSCALE;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int pass = 0; pass < C::FIR_PASSES; ++pass) {
            const int it = tid + C::TH * pass;
            const int g2 = it / (C::OR * 16), rem = it - g2 * (C::OR * 16);
            const int ly = 1 + (rem >> 4), m = rem & 15;
            const int Y = 2 * gy0 + 1 + ly, X0 = 2 * gx0 + 2 + 4 * m;
            if (it >= C::FIR_ITEMS || m == 15 || Y >= OHo || X0 >= OWo) continue;
            const int cl0 = 16 * bt + 8 * g2;           // first of this item's eight channels among the workgroup's 32
            const f32x4 nz4 = *reinterpret_cast<const f32x4*>(nzs + (ly - 1) * 64 + 4 * m);
            const float* Tc = T + (g2 * 8) * C::TPS + ly * 64 + 4 * m;
            // the item's per-channel constants (d, bias): six 16-byte broadcast reads in front of the windows
            f32x4 dc4[2], bs4[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                dc4[q] = *reinterpret_cast<const f32x4*>(epi + cl0 + 4 * q);
                bs4[q] = *reinterpret_cast<const f32x4*>(epi + 32 + cl0 + 4 * q);
            }
            // The 4 x 8 window of a channel is eight ds_read_b128; the windows of channel ch + 1 are requested BEFORE the 64 fmas of
            // channel ch (two register sets, the scheduler fenced between request and arithmetic: left to itself hipcc waited for a
            // channel's reads right in front of its arithmetic — one exposed LDS round trip per channel on a machine that holds two
            // waves per SIMD here), and the four pixels' chains advance together (four independent fmas per filter tap).
            f32x4 wa[2][4][2];
            auto request = [&](int set, int ch) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    wa[set][r][0] = *reinterpret_cast<const f32x4*>(Tc + ch * C::TPS + r * 64);
                    wa[set][r][1] = *reinterpret_cast<const f32x4*>(Tc + ch * C::TPS + r * 64 + 4);
                }
            };
            float out[8][4];
            request(0, 0);
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                const int set = ch & 1;
                if (ch < 7) request(set ^ 1, ch + 1);
                __builtin_amdgcn_sched_barrier(0);
                float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int fy = 0; fy < 4; ++fy)
#pragma unroll
                    for (int fx = 0; fx < 4; ++fx)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int c = jj + fx;
                            o[jj] = __builtin_fmaf(fs[fy * 4 + fx], wa[set][fy][c >> 2][c & 3], o[jj]);
                        }
                const float dc = dc4[ch >> 2][ch & 3], bs = bs4[ch >> 2][ch & 3];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float a = o[jj] * dc;
                    a = a + nz4[jj];
                    a = a + bs;
                    a = __builtin_fmaxf(a, a * alpha) * gain;
                    out[ch][jj] = __builtin_fminf(__builtin_fmaxf(a, -cl), cl);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (wimg) {
                const int c8 = (o0 >> 3) + 2 * bt + g2;
                char* dst = (char*)p.yimg + ((((size_t)n * (p.O >> 3) + c8) * OHo + Y) * (size_t)OWo + X0) * 16;
                f32x4 ns4[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) ns4[q] = *reinterpret_cast<const f32x4*>(epi + 64 + cl0 + 4 * q);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    f16x8 hv, lv;
                    const bool inside = X0 + jj < OWo;
                    const uint32_t keep = inside ? 0x7FFFFFFFu : 0u;
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) {
                        float a = ns4[ch >> 2][ch & 3] * out[ch][jj];  // (k_fir4x4_img's ns * act * 16)
                        // out of domain = !(|a| <= 65504) = the bits of |a| above those of 65504 (NaN and inf included): a running maximum
                        const uint32_t ab = __builtin_bit_cast(uint32_t, a) & keep;
                        badbits = ab > badbits ? ab : badbits;
                        a = __builtin_fminf(__builtin_fmaxf(a, -65504.0f), 65504.0f);
                        hv[ch] = (_Float16)a;
                        lv[ch] = (_Float16)(a - (float)hv[ch]);
                    }
                    if (inside) {
                        *reinterpret_cast<f16x8*>(dst + jj * 16) = hv;
                        *reinterpret_cast<f16x8*>(dst + lo_off + jj * 16) = lv;
                    }
                }
            } else {
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) {
                    float* yo = p.y + (((size_t)n * p.O + o0 + cl0 + ch) * OHo + Y) * (size_t)OWo + X0;
                    if (vec_y) *reinterpret_cast<f32x4*>(yo) = (f32x4){out[ch][0], out[ch][1], out[ch][2], out[ch][3]};
                    else {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            if (X0 + jj < OWo) yo[jj] = out[ch][jj];
                    }
                }
            }
        }
    }
    if (badbits > 0x477FE000u && p.sat) atomicOr(p.sat, 1u);  // (0x477FE000 = 65504.0f)
}

}  // namespace

// workgroups of a launch whose tiles hold `rows` grid rows: (spatial tiles, channel tiles, samples)
static dim3 up4_grid(const ConvParams& p, int rows) {
    const int orows = 2 * rows - 4;
    return dim3((unsigned)(((2 * p.W + 59) / 60) * ((2 * p.H + orows - 1) / orows)), (unsigned)(p.O / 32), (unsigned)p.N);
}

// The launch shape the host picks: 2 = eight waves x 2 rows (16 x 32 grid points, one workgroup per CU: 1.22 x instead of 1.42 x the
// MFMA work) once that tiling alone gives every CU two rounds of workgroups; 0 = four waves x 2 rows (8 x 32 grid points, 78 KB of
// LDS: two workgroups per CU) below that.  Measured (tools/up4_ab.py, us, image in / image out, batch 1; r05 = k_modconv_up3 + FIR pass):
//   256 -> 128 @256^2 -> 512^2 (684 / 1548 workgroups):  r05 238   shape 2 191   shape 0 206
//   256 -> 128 @128^2 -> 256^2 (200 / 440):              r05 68    shape 2 65    shape 0 64
//   512 -> 256 @ 64^2 -> 128^2 (120 / 264):              r05 64    shape 2 84    shape 0 101   (the chip is underfilled: split-K wins)
//    32 -> 256 @128^2 -> 256^2 (720 / 1760, 2 chunks):   r05 51    shape 2 62    shape 0 53    (k_modconv_up3<true>: its epilogue alone)
// An eight-wave one-row shape (8 x 32 grid points, one workgroup per CU) was built and dropped: never the fastest.  So was (round 6) a
// four-wave four-row shape — the 16 x 32 tile on ONE wave per SIMD, 256 accumulator + 256 other registers, 108 MFMAs per chunk and
// 0.35 instead of 0.55 LDS reads per MFMA: bit-identical and slower everywhere (256^2 -> 512^2: 209 us against 197; 32 -> 256: 73 / 65;
// the small maps 84-113 against 52-88): without a second wave on the SIMD every LDS round trip of the loop and of the filter is exposed.
// P3D_UP4_RPW = 0 / 2 in the environment forces a shape (A/B runs, tests).
int p3d_up4_shape(int N, int O, int H, int W) {
    const char* e = getenv("P3D_UP4_RPW");
    if (e && (atoi(e) == 0 || atoi(e) == 2)) return atoi(e);
    const long long wg16 = (long long)((2 * W + 59) / 60) * ((2 * H + 27) / 28) * (O / 32) * N;
    return wg16 >= 512 ? 2 : 0;
}

// up = 2, image-fed (p.ximg), unsplit, I % 16 == 0, O % 32 == 0; writes p.yimg (activation image, needs p.ystyles) or p.y (fp32 [N][O][2H][2W])
int p3d_up4_launch(const ConvParams& p0, int shape, hipStream_t st) {
    const ConvParams& p = p0;
    if (shape == 2) hipLaunchKernelGGL((k_modconv_up4<8, 2, 3>), up4_grid(p, 16), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((k_modconv_up4<4, 2, 2>), up4_grid(p, 8), dim3(256), 0, st, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? P3D_OK : (int)e;
}
