// p3d_fir.hip — upfirdn2d (torch_utils/ops/upfirdn2d.py:120-213), its 4x4 special cases and bias_act (bias_act.py:93-122) on gfx950:
//   k_upfirdn2d        the generic operator: zero-insert, pad / crop, FIR, optional fused epilogue d*v + noise -> +bias -> act*gain -> clamp
//   k_upsample2x_add   upsample2d (up 2) in polyphase form + the ToRGB accumulation of SynthesisBlock.forward (networks_stylegan2.py:476-478)
//   k_fir4x4_tiled     the FIR pass that ends an up-sampling layer (4x4 filter, pad 1, LDS-tiled; fp32 output; sums shallow split-K slices)
//   k_fir4x4_img       the same, writing the next layer's activation image
//   k_bias_act         clamp(act(x + b) * gain)
#include "p3d_conv_common.hpp"
#define chk chk_launch

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

// k_fir4x4_img2<RB> (round 6): k_fir4x4_img for the aligned intermediate (pitch % 4 == 0, xoff == padx0 == pady0 == 1, 4x4 filter,
// up = down = 1) with k_modconv_up4's filter stage: a row of the tile is read by SIXTEEN lanes (4-pixel windows 16 bytes apart), so
// that every 16-lane group of a ds_read_b128 covers 256 consecutive bytes of one row: no bank conflicts at any pitch (k_fir4x4_img:
// 8 lanes per 40-float row, two rows per group: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS 3.2-4.4).  A workgroup = 8 channels (one
// 16-byte piece per pixel) x RB output rows x 64 columns: the intermediate's rows Y0 - 1 .. Y0 + RB + 1, columns X0 - 1 .. X0 + 66 (68
// floats per row) — at index X0 + i of a row of the padded intermediate (column v sits at index v + 1): an aligned copy, masked
// outside [0, W) x [0, H).
// Split-K partials are summed in slice order while the tile is loaded (as k_fir4x4_img); the same 16-term fma chain and the same
// epilogue: bit-identical to k_fir4x4_img.  RB = 32 (35 rows x 68 floats x 8 channels = 76 160 B: two workgroups per CU) / 8 (small maps: more
// workgroups).
template <int RB>
__global__ __launch_bounds__(256, 2) void k_fir4x4_img2(FirParams p, char* __restrict__ yimg, long long lo_off, unsigned int* sat) {
    constexpr int TR = RB + 3, RP = 68, C4 = RP / 4, PLANE = TR * RP;  // RP: floats per tile row
    __shared__ __attribute__((aligned(16))) float tile[8 * PLANE];
    const int tid = threadIdx.x;
    const int tiles_x = (p.OW + 63) / 64;
    const int X0 = (blockIdx.x % tiles_x) * 64, Y0 = (blockIdx.x / tiles_x) * RB;
    const long long g = blockIdx.y;  // (n, c8)
    const long long n = g / (p.C >> 3);
    const int c0 = (int)(g - n * (p.C >> 3)) * 8;
    const int HP = p.H * p.pitch;
    const float* xg = p.x + (n * p.C + c0) * (long long)HP;
    float fs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) fs[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.f[i])));
    // ---- the tile: item = (channel, row, 4-column group); rows / columns outside the intermediate are zeros (the filter's padding).
    // Nine items per thread are requested together (and then each further split-K slice of the nine): two rounds of loads per slice
    // instead of one exposed round trip per item
    constexpr int ITEMS = 8 * TR * C4, NIT = (ITEMS + 255) / 256, BATCH = NIT > 10 ? 10 : NIT;
#pragma unroll 1
    for (int b0 = 0; b0 < NIT; b0 += BATCH) {
        f32x4 v[BATCH];
        const float* src[BATCH];
        int dst[BATCH], colb[BATCH];
        bool ok[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int it = tid + (b0 + i) * 256;
            const int ch = it / (TR * C4), rem = it - ch * (TR * C4), r = rem / C4, c4 = rem - r * C4;
            const int u = Y0 - 1 + r;
            ok[i] = it < ITEMS && u >= 0 && u < p.H && X0 + 4 * c4 < p.pitch;
            src[i] = ok[i] ? xg + (size_t)ch * HP + (size_t)u * p.pitch + X0 + 4 * c4 : xg;  // (a valid address either way: no branch around the load)
            dst[i] = it < ITEMS ? ch * PLANE + r * RP + 4 * c4 : -1;
            colb[i] = X0 - 1 + 4 * c4;
            v[i] = *reinterpret_cast<const f32x4*>(src[i]);
        }
        for (int k = 1; k < p.ksplit; ++k) {  // split-K partials, slice order (= k_splitk_reduce)
            f32x4 t[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) t[i] = *reinterpret_cast<const f32x4*>(src[i] + (size_t)k * p.slice);
#pragma unroll
            for (int i = 0; i < BATCH; ++i) { v[i].x += t[i].x; v[i].y += t[i].y; v[i].z += t[i].z; v[i].w += t[i].w; }
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = colb[i] + e;
                if (!ok[i] || col < 0 || col >= p.W) v[i][e] = 0.0f;
            }
            if (dst[i] >= 0) *reinterpret_cast<f32x4*>(&tile[dst[i]]) = v[i];
        }
    }
    __syncthreads();
    bool bad = false;
#pragma unroll 1
    for (int it = tid; it < RB * 16; it += 256) {
        const int ly = it >> 4, m = it & 15;
        const int Y = Y0 + ly, Xb = X0 + 4 * m;
        if (Y >= p.OH || Xb >= p.OW) continue;
        const float* Tc = tile + ly * RP + 4 * m;
        float out[8][4];
        f32x4 wa[2][4][2];
        auto request = [&](int set, int ch) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wa[set][r][0] = *reinterpret_cast<const f32x4*>(Tc + ch * PLANE + r * RP);
                wa[set][r][1] = *reinterpret_cast<const f32x4*>(Tc + ch * PLANE + r * RP + 4);
            }
        };
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
                    for (int j = 0; j < 4; ++j) {
                        const int c = j + fx;
                        o[j] = __builtin_fmaf(fs[fy * 4 + fx], wa[set][fy][c >> 2][c & 3], o[j]);
                    }
#pragma unroll
            for (int j = 0; j < 4; ++j) out[ch][j] = o[j];
            __builtin_amdgcn_sched_barrier(0);
        }
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
        const size_t piece0 = ((size_t)g * p.OH + Y) * p.OW + Xb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f16x8 hv, lv;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                float a = out[ch][j] * dco[ch];
                if (p.noise) a = a + nv[j];
                a = a + bs[ch];
                a = ns[ch] * act_apply(a, p.act, p.alpha, p.gain, p.clamp) * HX_SPLIT_SCALE_X;  // (k_fir4x4_img's epilogue, bit for bit)
                bad = bad || (Xb + j < p.OW && !(__builtin_fabsf(a) <= 65504.0f));
                a = __builtin_fminf(__builtin_fmaxf(a, -65504.0f), 65504.0f);
                hv[ch] = (_Float16)a;
                lv[ch] = (_Float16)(a - (float)hv[ch]);
            }
            if (Xb + j < p.OW) {
                *reinterpret_cast<f16x8*>(yimg + (piece0 + j) * 16) = hv;
                *reinterpret_cast<f16x8*>(yimg + lo_off + (piece0 + j) * 16) = lv;
            }
        }
    }
    if (bad && sat) atomicOr(sat, 1u);
}

// the FIR pass of an up-sampling layer (modconv_impl): q describes the (2H+1) x (2W+1) intermediate; yimg: write the next layer's image
void p3d_launch_fir_pass(const FirParams& q, char* yimg, long long lo_off, unsigned int* sat, hipStream_t st) {
    dim3 grid(((q.OW + 31) / 32) * ((q.OH + 31) / 32), (unsigned)q.NC);
    if (yimg) {
        dim3 gi(grid.x, (unsigned)(q.NC / 8));
        const bool rows_aligned = (q.pitch & 3) == 0 && q.xoff == q.padx0 && (((uintptr_t)q.x | (uintptr_t)(q.slice * 4)) & 15) == 0;
        // two channels per stage: 125 VGPRs, four waves per SIMD (four per stage: 195, two; measured 2-5 % slower)
        // k_fir4x4_img2 where k_fir4x4_img's 32 x 32 tiles are too few workgroups for the chip (512 channels at 32^2: 64 of them, 17.6 us;
        // 8-row tiles: 256, 13.8 us).  On the larger maps the older kernel — which requests its next stage while it filters — stays ahead
        // despite its bank conflicts (64^2: 13.3 against 15.9 us, 128^2: 20.2 against 27.8 - 30.9: measured, profiles/r06_notes.txt).
        // P3D_FIR_IMG2=0 / 8 / 32 in the environment: never / always with that tile height (tests, A/B runs).
        const char* e2 = getenv("P3D_FIR_IMG2");
        const int force = e2 ? atoi(e2) : -1;
        const bool can2 = rows_aligned && q.xoff == 1 && q.pady0 == 1 && q.fh == 4 && q.fw == 4 && force != 0;
        if (can2 && (force == 8 || force == 32 || (long long)gi.x * gi.y < 128)) {
            if (force == 32) hipLaunchKernelGGL((k_fir4x4_img2<32>), dim3((unsigned)(((q.OW + 63) / 64) * ((q.OH + 31) / 32)), (unsigned)(q.NC / 8)), dim3(256), 0, st, q, yimg, lo_off, sat);
            else hipLaunchKernelGGL((k_fir4x4_img2<8>), dim3((unsigned)(((q.OW + 63) / 64) * ((q.OH + 7) / 8)), (unsigned)(q.NC / 8)), dim3(256), 0, st, q, yimg, lo_off, sat);
        } else if (rows_aligned) hipLaunchKernelGGL((k_fir4x4_img<true, 2, 3>), gi, dim3(256), 0, st, q, yimg, lo_off, sat);
        else hipLaunchKernelGGL((k_fir4x4_img<false, 4, 2>), gi, dim3(256), 0, st, q, yimg, lo_off, sat);
    } else hipLaunchKernelGGL(k_fir4x4_tiled, grid, dim3(256), 0, st, q);
}

extern "C" {

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
