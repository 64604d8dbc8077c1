// p3d_synthesis.hip — host dispatch of the StyleGAN2 synthesis operators on gfx950 (MI355X) and their small kernels.
//   modconv_impl      modulated_conv2d (networks_stylegan2.py:40-97) + the bias_act that follows it (:350-352): which kernel runs a layer
//                     (the plain kernels: p3d_conv_plain.hip; the transposed ones: p3d_conv_up.hip, p3d_conv_up4.hip; the FIR pass:
//                     p3d_fir.hip), the split-K depth, the workspace carve-up.  The per-sample weights w*s*d of the reference's fused
//                     path (:68-73) are refactored into shared weights, input scaling by s and output scaling by d (its own non-fused
//                     path, :76-85).
//   k_demod / k_demod_plan   d[n,o] = rsqrt(sum_{i,t} (w[o,i,t] s[n,i])^2 + 1e-8)   (networks_stylegan2.py:70-71)
//   k_weights_to_f16, k_act_to_image, k_splitk_reduce, k_splitk_reduce_img
// ToRGB: p3d_torgb.hip.  upfirdn2d / bias_act: p3d_fir.hip.
#include "p3d_conv_common.hpp"
#define chk chk_launch

// w [O][I][kk] f32 -> wh [O][kk][I] f16 (RNE), once per layer; split: followed by the lo parts f16(w - hi) in the same layout.
// layout (split, kk = 9): P3D_WLAYOUT_PLAIN / _UP — the same values in the order the pipelined kernels keep them in LDS
// (include/panic3d_hip.h): element (which, o, tap = 3 dy + dx, i) goes to
//   PLAIN: ((((((c OT + ot) 3 + dx) 2 + which) 3 + dy) 2 + kh) 64 + ol) 8 + e,  c = i / 16, kh = (i / 8) % 2, e = i % 8, ot = o / 64, ol = o % 64
//   UP:    (((((c OT + ot) 2 + which) 9 + tap) 2 + kh) 32 + ol) 8 + e,           ot = o / 32, ol = o % 32
__global__ void k_weights_to_f16(const float* __restrict__ w, int O, int I, int kk, _Float16* __restrict__ wh, int split, int layout) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x, total = (long long)O * I * kk;
    if (idx >= total) return;
    const int i = (int)(idx % I), t = (int)((idx / I) % kk), o = (int)(idx / ((long long)I * kk));
    float v = w[((long long)o * I + i) * kk + t];
    if (split) v = fminf(fmaxf(v * HX_SPLIT_SCALE_W, -65504.0f), 65504.0f);
    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
    if (layout == P3D_WLAYOUT_OIK) {
        wh[idx] = hi;
        if (split) wh[total + idx] = lo;
        return;
    }
    const int c = i >> 4, kh = (i >> 3) & 1, e = i & 7;
    long long d0, dlo;  // destination of the hi part, distance to the lo part
    if (layout == P3D_WLAYOUT_PLAIN) {
        const int ot = o >> 6, ol = o & 63, dy = t / 3, dx = t - 3 * dy;
        d0 = (((((((long long)c * (O >> 6) + ot) * 3 + dx) * 2 + 0) * 3 + dy) * 2 + kh) * 64 + ol) * 8 + e;
        dlo = 3 * 2 * 64 * 8;
    } else {
        const int ot = o >> 5, ol = o & 31;
        d0 = ((((((long long)c * (O >> 5) + ot) * 2 + 0) * 9 + t) * 2 + kh) * 32 + ol) * 8 + e;
        dlo = 9 * 2 * 32 * 8;
    }
    wh[d0] = hi;
    wh[d0 + dlo] = lo;
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

// k_act_to_image fused with the in-place conditioning add of SynthesisNetwork.forward (networks_stylegan2.py:554-560, :600-622:
// `x[:, c0 : c0 + Ca] += t` between two blocks): x <- x + add on the channel range [c0, c0 + Ca) (Ca, c0 multiples of 8; add [Na][Ca][HW],
// Na = 1: shared by the batch), written back, and the image of the UPDATED x for the next block's conv0 — one launch where the
// conditioned generator ran an elementwise add and a conversion pass per block (round 6).  Same arithmetic, same bits.
__global__ void k_act_to_image_add(float* __restrict__ x, const float* __restrict__ s, int N, int C, int HW, const float* __restrict__ add,
                                   int c0, int Ca, int Na, char* __restrict__ img, long long lo_off, unsigned int* sat) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x, total = (long long)N * (C >> 3) * HW;
    if (idx >= total) return;
    const long long g = idx / HW;  // (n, c8)
    const int pix = (int)(idx - g * HW);
    const int n = (int)(g / (C >> 3)), ch0 = (int)(g - (long long)n * (C >> 3)) * 8;
    float* xp = x + g * 8 * HW + pix;
    const bool in = ch0 >= c0 && ch0 < c0 + Ca;  // (whole groups: c0 % 8 == Ca % 8 == 0)
    const float* ap = add + ((size_t)(Na > 1 ? n : 0) * Ca + (in ? ch0 - c0 : 0)) * HW + pix;
    f16x8 v, l;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float xv = xp[(size_t)i * HW];
        if (in) {
            xv = xv + ap[(size_t)i * HW];
            xp[(size_t)i * HW] = xv;
        }
        float m = (s ? s[g * 8 + i] : 1.0f) * xv * HX_SPLIT_SCALE_X;
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
static bool env_no_up3() { static const bool v = getenv("P3D_NO_UP3") != nullptr; return v; }
static bool up3_applies(int I, int O, int W) { return I % 16 == 0 && O % 32 == 0 && W >= up3_min_w() && !env_no_up3(); }
// k_modconv_up4 (p3d_conv_up4.hip)
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
    static const int target_up = getenv("P3D_KSPLIT_TARGET_UP") ? atoi(getenv("P3D_KSPLIT_TARGET_UP")) : 0;  // (A/B runs; 0: the image kernels' target)
    const int target = target_up > 0 ? target_up : ksplit_target_image();
    int ks = 1;
    while (ks < 64 && wgs * ks < target && I / (ks * 2) >= 16) ks *= 2;
    return ks;
}

static bool rgb_fusable(int N, int I, int O, int H, int W, int rgbo) {
    return rgbo >= 1 && rgbo <= 4 && I % 16 == 0 && O % 64 == 0 && W >= WX_TW && !p3d_env_no_w3() && !getenv("P3D_NO_RGB_FUSE") &&
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
    if (up == 1 && W >= p3d_w3_min_w()) {  // the wide tile of the two-term kernel may split deeper
        const int kw = choose_ksplit(N, I, O, H, W, WX_TW);
        ks = kw > ks ? kw : ks;
    }
    if (ks > 1) b += (size_t)ks * out_elems * 4;  // split-K partial sums
    if (up == 1 && W >= p3d_w3_min_w() && I % 16 == 0 && O % 64 == 0) b += (size_t)N * I * H * W * 4 + 256;  // the activation image an fp32 input is turned into (k_modconv_w3)
    if (up == 2 && up3_applies(I, O, W)) {  // k_modconv_up3: its own split-K depth, and the activation image of an fp32 input
        const int k3 = choose_ksplit_up3(N, I, O, H, W);
        if (k3 > ks) b += (size_t)(k3 - (ks > 1 ? ks : 0)) * out_elems * 4;
        b += (size_t)N * I * H * W * 4 + 256;
    }
    return b + 256;
}

// the layout of the two-term weight copy the dispatch of a 3x3 layer can consume (W: the input map's width): the image layouts where
// the layer runs on the pipelined kernels whatever its batch size and split-K depth, OIK elsewhere; P3D_WLAYOUT=0 in the environment
// (read once): OIK everywhere (A/B runs)
int p3d_conv_weight_layout(int I, int O, int W, int up) {
    static const bool off = getenv("P3D_WLAYOUT") && atoi(getenv("P3D_WLAYOUT")) == 0;
    if (off || I <= 0 || O <= 0 || W <= 0 || I % 16 != 0) return P3D_WLAYOUT_OIK;
    if (up == 1 && O % 64 == 0 && W >= p3d_w3_min_w() && !p3d_env_no_w3()) return P3D_WLAYOUT_PLAIN;
    if (up == 2 && up3_applies(I, O, W)) return P3D_WLAYOUT_UP;
    return P3D_WLAYOUT_OIK;
}

// the rule by which an image-consuming layer is accepted (p3d_conv_args.x_img): exported so that a binding cannot drift from it
int p3d_conv_takes_image(int I, int O, int W, int up) {
    if (I <= 0 || O <= 0 || W <= 0 || I % 16 != 0) return 0;
    if (up == 1) return W >= p3d_w3_min_w() ? 1 : 0;
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
                        void* yimg = nullptr, const float* ystyles = nullptr, const RgbFuse* rgb = nullptr, int wlayout = P3D_WLAYOUT_OIK) {
    if (rgb && !rgb->partial) rgb = nullptr;
    // a weight copy in one of the image layouts: only for a layer that runs on the kernel whose LDS image it is (an fp32 input is
    // turned into an activation image first, so the pipelined kernels take those layers whichever way their input arrives)
    if (wlayout != P3D_WLAYOUT_OIK && (!wh || !wsplit || ks != 3 || wlayout != p3d_conv_weight_layout(I, O, W, up))) return P3D_E_RANGE;
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
        if (up == 1 ? W < p3d_w3_min_w() : !up3_applies(I, O, W)) return P3D_E_RANGE;  // (an up-sampling layer reads images only through k_modconv_up3)
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
    const bool wide = wh && wsplit && ks == 3 && up == 1 && W >= p3d_w3_min_w();  // k_modconv_w3 / k_modconv_up3
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
    if ((up3 && !ximg) || (wide && !ximg && O % 64 == 0 && I % 16 == 0 && !p3d_env_no_w3())) {
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
    p.ximg = ximg; p.ximg_lo = (long long)N * I * H * W * 2; p.tox = up == 2 ? 1 : 0; p.wlayout = wlayout;
    p.rgbw = rgb ? rgb->w : nullptr; p.rgbs = rgb ? rgb->styles : nullptr; p.rgbp = rgb ? rgb->partial : nullptr; p.rgbo = rgb ? rgb->channels : 0;
    static const bool xcd_order = !getenv("P3D_NO_XCD_ORDER");  // (A/B runs)
    p.xcd = xcd_order ? 1 : 0;
    // up = 1 with an image output: k_modconv_w3 writes it from its epilogue when it runs unsplit; otherwise a pass over y below
    const bool w3_img = up == 1 && yimg && wide && ximg && O % 64 == 0 && ksplit == 1 && !p3d_env_no_w3();
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
        p3d_launch_conv_plain(p, st);
    } else {  // stride-2 transposed conv into [N][O][2H+1][2W+1]: all four output phases in one launch
        p.GH = H + 1; p.GW = W + 1;
        if (up4) {
            p.y = y; p.yimg = yimg;
            return p3d_up4_launch(p, p3d_up4_shape(N, O, H, W), st);
        }
        if (up3_fused) {
            p3d_launch_conv_up(p, 2, st);
            return chk();
        }
        p3d_launch_conv_up(p, up3 ? 1 : 0, st);
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
    p3d_launch_fir_pass(q, (char*)yimg, (long long)N * O * q.OH * q.OW * 2, sat, st);
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

static int weights_to_f16(const float* w, int O, int I, int ks, void* w_f16, int split, void* stream, int layout = P3D_WLAYOUT_OIK) {
    if (!w || !w_f16 || O <= 0 || I <= 0) return P3D_E_ARG;
    if (ks != 1 && ks != 3) return P3D_E_RANGE;
    if (layout != P3D_WLAYOUT_OIK && (!split || ks != 3 || I % 16 != 0 || (layout == P3D_WLAYOUT_PLAIN ? O % 64 != 0 : layout == P3D_WLAYOUT_UP ? O % 32 != 0 : true)))
        return P3D_E_RANGE;
    const long long total = (long long)O * I * ks * ks;
    hipLaunchKernelGGL(k_weights_to_f16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, O, I, ks * ks,
                       (_Float16*)w_f16, split, layout);
    return chk();
}
int p3d_conv_weights_to_f16x2_layout(const float* w, int O, int I, int ks, int layout, void* w_f16x2, void* stream) {
    return weights_to_f16(w, O, I, ks, w_f16x2, 1, stream, layout);
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
                        stream, (wsplit || a->y_img) ? (unsigned int*)a->saturated : nullptr, a->x_img, a->y_img, a->y_img_styles, &rgb, a->w_f16_layout);
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

int p3d_act_to_image_add_f32(float* x, const float* styles, int N, int C, int H, int W, const float* add, int c0, int Ca, int Na, void* img,
                             uint32_t* saturated, void* stream) {
    if (!x || !img || !add || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Ca <= 0 || c0 < 0 || (Na != 1 && Na != N)) return P3D_E_ARG;
    if (C % 8 != 0 || c0 % 8 != 0 || Ca % 8 != 0 || c0 + Ca > C || ((uintptr_t)img & 15)) return P3D_E_RANGE;
    const long long total = (long long)N * (C / 8) * H * W;
    hipLaunchKernelGGL(k_act_to_image_add, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, styles, N, C, H * W, add,
                       c0, Ca, Na, (char*)img, (long long)N * C * H * W * 2, (unsigned int*)saturated);
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

}  // extern "C"
