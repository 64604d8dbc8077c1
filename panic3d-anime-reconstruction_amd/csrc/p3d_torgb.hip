// p3d_torgb.hip — ToRGBLayer.forward (networks_stylegan2.py:366-380) on gfx950: the 1x1 modulated convolution without demodulation
// as a GEMM that reads its activation once (k_torgb), fused with SynthesisBlock's skip connection (:476-478), and the second half of
// a ToRGB layer whose channel sums came out of conv1's epilogue (k_torgb_combine; k_modconv_w3<true>, p3d_conv_plain.hip).
#include "p3d_conv_common.hpp"
#define chk chk_launch

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

extern "C" {

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

}  // extern "C"
