// p3d_conv_common.hpp — what the translation units of the StyleGAN2 synthesis operators share: the launch parameters of the
// convolution kernels, the two-term operand scaling, the XCD-aware workgroup order and the inline-asm LDS-DMA helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/panic3d_hip.h"


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__

#define CONV_TH 8
#define CONV_TW 16
#define XS_ROW (CONV_TW + 2)
#define XS_PLANE ((CONV_TH + 2) * XS_ROW)
// a K chunk = 8 input channels (72 k values for 3x3, 8 for 1x1)
template <int MODE> struct ConvTaps;
// N taps; dy, dx: input offset of tap t relative to the output position (tap t is element ky*3+kx of the 3x3 kernel)
template <> struct ConvTaps<0> { static constexpr int N = 9; static constexpr int dy[9] = {-1,-1,-1,0,0,0,1,1,1}; static constexpr int dx[9] = {-1,0,1,-1,0,1,-1,0,1}; };
template <> struct ConvTaps<1> { static constexpr int N = 1; static constexpr int dy[1] = {0}; static constexpr int dx[1] = {0}; };

struct ConvParams {
    const float* x;       // [N][I][H][W]
    const float* w;       // [O][I][ks][ks]
    const void* wh;       // f16 copy [O][ks*ks][I] (f16-operand kernels) or null
    int wsplit;           // wh holds hi parts followed by lo parts (two-term operands)
    int wlayout;          // P3D_WLAYOUT_*: how the two-term copy of the 3x3 weights is laid out (round 6): 0 = [hi|lo][O][9][I]; 1 / 2 = the
                          // consuming kernel's own LDS image per (16-channel chunk, channel tile), see include/panic3d_hip.h
    const float* styles;  // [N][I]
    const float* dcoef;   // [N][O] or null
    const float* noise;   // [OH*OW] (shared) or [N][OH*OW] or null; already multiplied by noise_strength
    const float* bias;    // [O] or null
    float* y;             // [N][O][OH][OW]
    int N, I, O, H, W;    // input dims
    int GH, GW;           // output grid of this launch (phase grid for MODE >= 2)
    int OH, OW;           // output tensor dims
    int ks;               // kernel size of w (1 or 3)
    int noise_per_sample;
    int act;              // 0 linear, 1 lrelu
    float alpha, gain, clamp;
    int epilogue;         // 1: dcoef/noise/bias/act applied here; 0: raw store (transposed-conv intermediate)
    int tox;              // up = 2: the intermediate T [N][O][2H+1][OW = pitch] stores column ox at index ox + tox (tox = 1, pitch = 2W + 4: the
                          // FIR pass reads its 36-column windows — columns X0 - 1 .. X0 + 34 — as aligned 16-byte loads); 0 elsewhere
    int ksplit;           // input channels split over ksplit workgroups (blockIdx.z = n*ksplit + kz); > 1 => raw partials
    int xcd;              // k_modconv_w3 / k_modconv_up3: XCD-aware workgroup order (p3d_wg_order)
    unsigned int* sat;    // caller-owned device word, OR-ed with 1 when a two-term operand left its domain (or null: not reported)
    // ---- the activation IMAGE path (the producer prepares the consumer's operand; see "activation IMAGE" below)
    const void* ximg;     // input as an image [hi | lo][N][I/8][H][W] of 16-byte pieces, or null (then x + styles are used)
    long long ximg_lo;    // byte offset of the lo half of ximg (= N*I*H*W*2)
    // k_modconv_w3 only: ALSO write the result as the image of a following layer with styles ystyles [N][O] (next to the fp32 y)
    void* yimg;
    long long yimg_lo;    // = N*O*OH*OW*2
    const float* ystyles;
    const float* fir;     // k_modconv_up3<true>: the 4x4 filter of the FIR pass it contains (flipped, times up^2)
    // k_modconv_w3<true> only: the block's ToRGB layer (networks_stylegan2.py:366-380, <= 4 output channels) applied to the result in
    // the epilogue — each 64-channel workgroup adds its channels' share into rgbp [O/64][N][rgbo][H][W] (p3d_torgb_combine_f32 sums
    // the shares, adds the bias and the up-sampled skip image); y may then be null (nobody reads the fp32 activation)
    const float* rgbw;    // [rgbo][O] ToRGB weights
    const float* rgbs;    // [N][O] ToRGB styles (already multiplied by the layer's weight_gain)
    float* rgbp;
    int rgbo;
};

DEV float act_apply(float v, int act, float alpha, float gain, float clamp) {
    if (act == 1) v = v < 0.0f ? v * alpha : v;
    v = v * gain;
    if (clamp >= 0.0f) v = __builtin_fminf(__builtin_fmaxf(v, -clamp), clamp);
    return v;
}


#define CONV_OOB ((int)0x80000000)
#define CONV_RSRC_FLAGS 0x00020000

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// The matrix cores flush f16 subnormals, so the operands are scaled by a power of two before they are split — the weights by
// 2^6 (lo parts of |w| >= 2^-8 stay normal), the modulated activations s*x by 2^4 (|s*x| >= 2^-6; more headroom at the top:
// hi saturates, it does not overflow, at |s*x| = 65504 / 16 = 4094) — and the accumulators are scaled back by 2^-10
// when they are stored; all exact.  A value below those thresholds loses its lo part (absolute error <= 2^-11 |v|, i.e.
// below 8e-6 / 2e-6): rare and small next to the 2^-22 relative rounding of the ordinary terms.
#define HX_SPLIT_SCALE_X 16.0f
#define HX_SPLIT_SCALE_W 64.0f
#define HX_SPLIT_UNSCALE (1.0f / 1024.0f)
// Out of domain: a scaled operand beyond the f16 range (|s*x| > 65504 / 16 = 4094, or NaN) is clamped to +-65504 — finite, wrong —
// and the CALLER's flag word (ConvParams::sat, the `saturated` argument of p3d_modconv2d_f16x2mma_f32) is OR-ed with 1: no state
// lives in the library.
#define WX_TW 32
#define WX_ROW (WX_TW + 2)                         // patch columns = LDS row pitch (px)

#define U3_WB (2 * 9 * 64 * 16)                    // one buffer of weights: 18 432

// Workgroup order of the image-fed kernels.  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, each with an L2
// of its own, and in (tile, channel tile) order the four channel tiles of a spatial tile — which read the SAME patch — landed on four
// different XCDs at four different times: 256 -> 256 @256^2 staged 356 MB of patches out of a 67 MB image, all of it past the L2s.
// Here XCD x is given a CONTIGUOUS range of the (slice, tile, channel tile) sequence, channel tile fastest: the channel tiles of a
// tile, and neighbouring tiles with their shared halos, run back to back on one XCD and meet in its L2.
struct WgOrder { int tile, otile, z; };
DEV WgOrder p3d_wg_order(bool xcd) {
    const int T = gridDim.x * gridDim.y * gridDim.z;
    int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    WgOrder r;
    if (!xcd) { r.tile = blockIdx.x; r.otile = blockIdx.y; r.z = blockIdx.z; return r; }
    const int q = T >> 3, rem = T & 7, x = L & 7, m = L >> 3;
    L = x * q + (x < rem ? x : rem) + m;
    r.otile = L % gridDim.y;
    L /= gridDim.y;
    r.tile = L % gridDim.x;
    r.z = L / gridDim.x;
    return r;
}

DEV i32x4 w3_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = CONV_RSRC_FLAGS;
    return r;
}
// 64 lanes x 16 bytes from rsrc[voff] to LDS [lds_addr + lane * 16] (lds_addr wave-uniform); one wait state between the M0 write
// and the LDS-DMA (what the compiler inserts for its own: s_nop 0)
DEV void w3_dma16(uint32_t lds_addr, i32x4 rsrc, int voff) {
    lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);  // ("s" alone does not make a value uniform: s_mov_b32 m0, v75 was emitted)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
#define W3_VMWAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

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

// ---- host side: shared by the translation units of the synthesis operators
static inline int chk_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? P3D_OK : (int)e;
}
// The kernel-selection switches of the environment (A/B runs) are read ONCE per process: what p3d_modconv2d_workspace_bytes
// answered for a shape stays the size the launch of that shape needs (a caller may cache the query).
// narrowest map the pipelined plain 3x3 kernel (k_modconv_w3, a 32-column tile) takes; P3D_W3_MIN_W in the environment: A/B runs
#ifndef P3D_W3_MIN_W
#define P3D_W3_MIN_W 32
#endif
static inline int p3d_w3_min_w() { static const int v = getenv("P3D_W3_MIN_W") ? atoi(getenv("P3D_W3_MIN_W")) : P3D_W3_MIN_W; return v; }
static inline bool p3d_env_no_w3() { static const bool v = getenv("P3D_NO_W3") != nullptr; return v; }
void p3d_launch_conv_plain(const ConvParams& p, hipStream_t st);            // p3d_conv_plain.hip
void p3d_launch_conv_up(const ConvParams& p, int kind, hipStream_t st);     // p3d_conv_up.hip
void p3d_launch_fir_pass(const FirParams& q, char* yimg, long long lo_off, unsigned int* sat, hipStream_t st);  // p3d_fir.hip
int p3d_up4_shape(int N, int O, int H, int W);                               // p3d_conv_up4.hip
int p3d_up4_launch(const ConvParams& p, int rpw, hipStream_t st);
