// p3d_kernels.hip — HIP kernels + the C ABI (include/panic3d_hip.h) of libpanic3d_hip.so.  gfx950 only.
//
// Kernels:
//   k_planes_to_nhwc   LDS-tiled transpose of the reference's NCHW planes to channels-last.
//   k_decode_points    run_model on a point cloud (32 points per wavefront, see p3d_decode.hpp).
//   k_render           ImportanceRenderer.forward fused per wavefront: 32 rays per wave (lane pair = ray x channel half);
//                      coarse density pass -> weights -> importance resampling -> merge -> final decode + compositing,
//                      all per-ray state in registers / LDS; no intermediate tensor ever reaches HBM.
//   k_render_pair      the same algorithm for small launches: 16 rays x 2 samples per wave (bit-identical results).
//   k_render_finish    the one cross-ray dependency: depth clamp to the global [min t, max t] (ray_marcher.py:49-50).
//   k_sigma2density    get_eg3d_volume's activation + crop / cull masks in one pass.
//   k_stratified / k_composite / k_importance / k_unify_perm   operator-level stand-alone stages (one thread per ray).
//
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see panic3d-anime-reconstruction_amd/_build.py).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "p3d_decode.hpp"

#define P3D_WAVES_PER_WG 4
#define P3D_RENDER_WAVES 4  // k_render workgroup (two workgroups per CU -> two waves per SIMD)
// Quad-cooperative gathers (p3d_decode.hpp) in k_render: every decode of both kernels (measured with the 76-instruction transpose,
// 512^2 x (48+48), canonical / surface ms: tolerance kernel 1.99 / 2.62 -> 1.61 / 2.13 with them in the final AND the coarse pass
// (final only: 1.78 / 2.22); exact kernel 2.32 / 3.33 -> 1.93 / 3.09 (final only 2.09 / 3.16); every sample decoded 3.39 / 5.16).
// Not in the exact point / grid query (density-only decoder: 12.75 vs 12.65 ms at 512^3); yes in the tolerance one (10.2 -> 7.7).
#ifndef P3D_QUAD_COARSE
#define P3D_QUAD_COARSE 1
#endif
#ifndef P3D_QUAD_EXACT
#define P3D_QUAD_EXACT 1
#endif
#ifndef P3D_QUAD_PAIR
#define P3D_QUAD_PAIR 1     // ... and in the small-launch kernel k_render_pair
#endif
#ifndef P3D_RENDER_OCC
#define P3D_RENDER_OCC 2    // waves per SIMD the register allocation of k_render is held to (launch_bounds) and the host packs for
#endif
#define P3D_WG (64 * P3D_WAVES_PER_WG)

// =====================================================================================================================
// planes NCHW -> NHWC
// =====================================================================================================================
// grid: (ceil(H*W/64), n3); block 256.  Each block moves a [32 ch][64 px] tile through LDS.
__global__ __launch_bounds__(256) void k_planes_to_nhwc(const float* __restrict__ src, float* __restrict__ dst, int HW) {
    __shared__ float tile[32][65];
    const int p0 = blockIdx.x * 64;
    const size_t img = (size_t)blockIdx.y * 32 * HW;
    {
        int px = threadIdx.x & 63, c0 = threadIdx.x >> 6;  // 4 channels per pass
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int c = c0 + 4 * k;
            tile[c][px] = (p0 + px < HW) ? src[img + (size_t)c * HW + p0 + px] : 0.0f;
        }
    }
    __syncthreads();
    {
        int c = threadIdx.x & 31, q0 = threadIdx.x >> 5;  // 8 pixels per pass
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int px = q0 + 8 * k;
            if (p0 + px < HW) dst[img + (size_t)(p0 + px) * 32 + c] = tile[c][px];
        }
    }
}

// =====================================================================================================================
// run_model on a point cloud
// =====================================================================================================================
struct DecodeParams {
    const float* planes;  // [N][3][H][W][32]
    const float* coords;  // [N][M][3]
    const float *w0, *b0, *w1, *b1;
    float* out_sigma;  // [N][M]
    float* out_rgb;    // [N][M][32] or null
    unsigned char* out_cropmask;  // grid mode only: [M] 1 where |x| or |z| > mask_limit (triplane_crop_mask), or null
    float mask_limit;
    long long M;
    long long tiles_per_img;  // ceil(M/32)
    long long ntiles;
    int H, W;
    // coords == nullptr: points of the reference's regular grid (create_samples, _util/eg3d_metrics3d.py:70-92), flat index
    // grid_lo + m: column 2 = idx % n, column 1 = fmod(float(idx) / n, n), column 0 = fmod(float(idx) / n / n, n), each * vsize + goff
    int grid_n;
    long long grid_lo;
    float vsize, goff0, goff1, goff2;
    P3dDecodeCfg cfg;
};

// fmodf(a, b) for finite a >= 0, b > 0 with a / b < 2^24 — exactly C's fmodf there, without its generic exponent loop:
// q = trunc(RN(a/b)) is floor(a/b) or one more (rounding is monotonic and integers are representable); a - q*b is then exactly
// representable (a multiple of ulp(b), magnitude < b), so the fma returns it without rounding and one conditional add fixes q+1.
P3D_DEV float p3d_fmod_pos(float a, float b) {
    const float q = __builtin_truncf(a / b);
    const float r = __builtin_fmaf(-q, b, a);
    return r < 0.0f ? r + b : r;
}

// STAGED (grid mode, density only): every wave-step first tries to park the texel boxes its 32 points touch in LDS and gathers
// the taps from there (p3d_gather_features_staged); tiles whose taps do not fit (a tile that straddles two grid rows) take the
// direct path.  Dynamic LDS: the MLP image + P3D_BOX_FLOATS_PER_WAVE floats per wave.
// FASTD (P3D_FLAG_FAST_COLOR on the grid query): the tolerance-mode decoder (f16 two-term MFMA + hardware transcendentals).
template <bool WANT_RGB, bool STAGED, bool FASTD>
__global__ __launch_bounds__(P3D_WG, 2) void k_decode_points(DecodeParams p) {
    static_assert(!(FASTD && WANT_RGB), "the tolerance-mode point decoder is density-only");
    extern __shared__ __attribute__((aligned(16))) float lds_alloc[];
    // FASTD (density only) needs neither of the fp32 weight images nor the f16 colour weights: the LDS image starts at the biases
    // (P3D_LDS_B0P) and `lds` points that many floats before the allocation, so that every P3D_LDS_* offset still applies
    float* lds = FASTD ? lds_alloc - P3D_LDS_B0P : lds_alloc;
    p3d_load_mlp_to_lds(lds, p.w0, p.b0, p.w1, p.b1, !FASTD, !FASTD);
    if constexpr (FASTD) p3d_load_mlp_f16_to_lds(lds, p.w0, p.w1, false);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    float* box = lds + (FASTD ? P3D_LDS_FAST_FLOATS : P3D_LDS_MLP_FLOATS) + 4 + (STAGED ? wave * P3D_BOX_FLOATS_PER_WAVE : 0);
    (void)box;
    P3dPlaneGeom g;
    g.halfW = 0.5f * (float)p.W; g.halfH = 0.5f * (float)p.H; g.fW = (float)p.W; g.fH = (float)p.H; g.W = p.W;
    g.plane_bytes = (uint32_t)p.H * (uint32_t)p.W * 128u;
    const long long stride = (long long)gridDim.x * P3D_WAVES_PER_WG;

    // everything a tile needs before its decode: sample position, flags, plane resource
    struct Tile {
        long long n, m;
        bool active, skip, any;
        float px, py, pz;
    };
    auto setup = [&](long long tile, Tile& t) {
        long long n = 0, tl = tile;
        if (p.tiles_per_img != p.ntiles) { n = tile / p.tiles_per_img; tl = tile - n * p.tiles_per_img; }  // uniform; 1 image: no 64-bit division
        t.n = n;
        t.m = tl * 32 + j;
        t.active = t.m < p.M;
        const long long mc = t.active ? t.m : p.M - 1;
        t.skip = false;
        if (p.coords) {
            const float* c = p.coords + ((size_t)n * p.M + mc) * 3;
            t.px = c[0]; t.py = c[1]; t.pz = c[2];
        } else {  // the same float arithmetic as the reference's create_samples (float division: fractional carries are kept)
            // idx < grid_n^3 <= 2^31 (the host checks grid_n <= 1290): 32-bit index arithmetic; (float)idx rounds like torch's
            const unsigned idx = (unsigned)(p.grid_lo + mc);
            const float fn = (float)p.grid_n, f = (float)idx;
            const float s2 = (float)(idx % (unsigned)p.grid_n);
            const float q1 = f / fn;
            const float s1 = p3d_fmod_pos(q1, fn);
            const float s0 = p3d_fmod_pos(q1 / fn, fn);
            t.px = s0 * p.vsize + p.goff0; t.py = s1 * p.vsize + p.goff1; t.pz = s2 * p.vsize + p.goff2;
            if (p.out_cropmask) {
                const bool cropped = __builtin_fabsf(t.px) > p.mask_limit || __builtin_fabsf(t.pz) > p.mask_limit;
                if (t.active && h == 0) p.out_cropmask[t.m] = cropped ? 1 : 0;
                if (p.cfg.flags & P3D_FLAG_SKIP_CROPPED) t.skip = cropped;
            }
        }
        // P3D_FLAG_SKIP_CROPPED (grid mode): a cropped point's density is -1000 whatever the network says
        // (eg3d_metrics3d.py:155-159), so it is not decoded: whole wavefronts are skipped, single lanes fetch nothing
        t.any = __builtin_amdgcn_ballot_w64(t.active && !t.skip) != 0;
    };
    auto resource = [&](long long n) {
        const unsigned nlo = __builtin_amdgcn_readfirstlane((unsigned)n);
        const float* base = p.planes + ((p.cfg.flags & P3D_FLAG_SHARED_PLANES) ? (size_t)0 : (size_t)nlo * 3 * (g.plane_bytes / 4));
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 3 * g.plane_bytes, 0x00020000);
    };
    auto finish = [&](const Tile& t, float sigma, const f32x16& rgb) {
        if (t.skip) sigma = -1000.0f;
        if (t.active) {
            const size_t o = (size_t)t.n * p.M + t.m;
            if (h == 0) p.out_sigma[o] = sigma;
            if (WANT_RGB) {
                float* dst = p.out_rgb + o * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(f32x4*)(dst + 8 * q) = (f32x4){rgb[4 * q], rgb[4 * q + 1], rgb[4 * q + 2], rgb[4 * q + 3]};
            }
        }
    };

    // grid-stride over tiles of 32 points; a tile never straddles two images
    long long tile = (long long)blockIdx.x * P3D_WAVES_PER_WG + wave;
    if constexpr (!STAGED) {
        for (; tile < p.ntiles; tile += stride) {
            Tile t;
            setup(tile, t);
            float sigma = -1000.0f;
            f32x16 rgb;
            if (t.any) {
                if constexpr (FASTD) p3d_decode_wave_fast<WANT_RGB, true>(lds, resource(t.n), g, p.cfg, t.px, t.py, t.pz, sigma, rgb, !t.skip);
                else p3d_decode_wave<WANT_RGB>(lds, resource(t.n), g, p.cfg, t.px, t.py, t.pz, sigma, rgb, !t.skip);
            }
            finish(t, sigma, rgb);
        }
    } else {
        // software pipeline: the NEXT tile's texel boxes are being loaded (into registers) while this tile is decoded
        if (tile >= p.ntiles) return;
        Tile cur, nxt;
        P3dStage st;
        setup(tile, cur);
        bool cur_staged = cur.any && p3d_stage_plan(resource(cur.n), g, p.cfg, cur.px, cur.py, cur.pz, st);
        for (;;) {
            P3dBox cb[3] = {st.b[0], st.b[1], st.b[2]};
            if (cur_staged) p3d_stage_commit(box, st);  // waits for the loads issued one iteration ago
            const long long next = tile + stride;
            bool nxt_staged = false;
            if (next < p.ntiles) {
                setup(next, nxt);
                nxt_staged = nxt.any && p3d_stage_plan(resource(nxt.n), g, p.cfg, nxt.px, nxt.py, nxt.pz, st);
            }
            float sigma = -1000.0f;
            f32x16 rgb;
            if (cur.any) {
                if (cur_staged) {
                    const f32x16 X = p3d_gather_features_boxed(box, cb, g, p.cfg, cur.px, cur.py, cur.pz, !cur.skip);
                    if constexpr (FASTD) p3d_decode_features_fast<WANT_RGB>(lds, p.cfg, X, cur.px, cur.pz, sigma, rgb);
                    else p3d_decode_features<WANT_RGB>(lds, p.cfg, X, cur.px, cur.pz, sigma, rgb);
                } else if constexpr (FASTD) {
                    p3d_decode_wave_fast<WANT_RGB>(lds, resource(cur.n), g, p.cfg, cur.px, cur.py, cur.pz, sigma, rgb, !cur.skip);
                } else {
                    p3d_decode_wave<WANT_RGB>(lds, resource(cur.n), g, p.cfg, cur.px, cur.py, cur.pz, sigma, rgb, !cur.skip);
                }
            }
            finish(cur, sigma, rgb);
            if (next >= p.ntiles) break;
            tile = next;
            cur = nxt;
            cur_staged = nxt_staged;
        }
    }
}

// OSGDecoder.forward (training/triplane.py:528-544) on ALREADY SAMPLED features — the module's own call surface, for callers that
// hold `sampled_features` [N][3][M][32] (the reference's renderer hands the decoder exactly that, renderer.py:271-273).  The plane
// mean in the contract's order ((f0 + f1) + f2) * (1/3), then the same MFMA decoder as every other kernel here.  No masks: the
// decoder knows no positions.  HBM-bound: 384 B in, 132 B out per sample.
struct DecodeFeatParams {
    const float* feats;  // [N][3][M][32]
    const float *w0, *b0, *w1, *b1;
    float* out_sigma;    // [N][M]
    float* out_rgb;      // [N][M][32]
    long long M, tiles_per_img, ntiles;
    P3dDecodeCfg cfg;    // flags: P3D_FLAG_FORCE_SIGMOID only
};
__global__ __launch_bounds__(P3D_WG, 2) void k_decode_features(DecodeFeatParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    p3d_load_mlp_to_lds(lds, p.w0, p.b0, p.w1, p.b1);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const long long stride = (long long)gridDim.x * P3D_WAVES_PER_WG;
    for (long long tile = (long long)blockIdx.x * P3D_WAVES_PER_WG + wave; tile < p.ntiles; tile += stride) {
        const long long n = tile / p.tiles_per_img, m = (tile - n * p.tiles_per_img) * 32 + j;
        const bool active = m < p.M;
        const long long mc = active ? m : p.M - 1;
        f32x16 f[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const f32x4* src = (const f32x4*)(p.feats + (((size_t)n * 3 + pl) * p.M + mc) * 32 + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = src[q];
                f[pl][4 * q] = v.x; f[pl][4 * q + 1] = v.y; f[pl][4 * q + 2] = v.z; f[pl][4 * q + 3] = v.w;
            }
        }
        f32x16 X, rgb;
#pragma unroll
        for (int c = 0; c < 16; ++c) X[c] = ((f[0][c] + f[1][c]) + f[2][c]) * P3D_THIRD;  // triplane.py:530 mean(1)
        float sigma;
        p3d_decode_features<true>(lds, p.cfg, X, 0.0f, 0.0f, sigma, rgb);
        if (active) {
            const size_t o = (size_t)n * p.M + m;
            if (h == 0) p.out_sigma[o] = sigma;
            float* dst = p.out_rgb + o * 32 + 4 * h;  // register r holds channel rowof(r) + 4h (p3d_decode.hpp)
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 8 * q) = (f32x4){rgb[4 * q], rgb[4 * q + 1], rgb[4 * q + 2], rgb[4 * q + 3]};
        }
    }
}

// =====================================================================================================================
// fused ImportanceRenderer.forward
// =====================================================================================================================
struct RenderParams {
    const float* planes;  // [N][3][H][W][32]
    const float *rays_o, *rays_d;  // [N][R][3]
    const float* jitter;  // [N][R][Sc]
    const float* u;       // [N*R][Sf]
    const float *w0, *b0, *w1, *b1;
    float *out_feat, *out_depth, *out_wsum, *out_xyz;
    uint32_t* gminmax;  // [0..1] order-mapped min / max of all depths, [2..3] decode-step count, [4 + 2n ..] per-view min / max
    int per_view_clamp;  // P3D_FLAG_PER_VIEW_CLAMP: the N views are N calls of the reference -> one clamp range each
    p3d_dumps dumps;
    long long R;
    long long tiles_per_img, ntiles;
    int tile_w;       // >0: 8x4 screen tiles over a tile_w-wide image
    int tiles_x;      // tile_w / 8
    int H, W;
    int Sc, Sf;
    float ray_start, ray_end, depth_delta;
    const float *ray_start_arr, *ray_end_arr;  // per-ray limits [N][R] ('auto', renderer.py:165-171) or null
    int disparity;                             // P3D_FLAG_DISPARITY
    int white_back;
    int lds_rows;     // rows (of 32 floats) of per-wave LDS
    int rng;          // p3d_render_rng_f32: the two draws come from the counter-based generator of include/p3d_numerics.h
    uint32_t seed_lo, seed_hi;
    int swz;          // XCD swizzle run length (blocks)
    int blocked;      // k_render: tiles ordered by 16 x 16-tile super-tiles, one per XCD run (see the kernel)
    P3dDecodeCfg cfg;
};

// per-ray running state of MipRayMarcher2 (ray_marcher.py:25-57)
struct MarchState {
    double Td;
    float W, D;
    float prev_t, prev_sigma;
};

// The counter-based generator of p3d_render_rng_f32 (include/p3d_numerics.h "device draws"): stream 0 = the jitter of
// sample_stratified (renderer.py:324), stream 1 = the u of sample_pdf (:371); a pure function of (seed, stream, ray, index).
P3D_DEV uint32_t p3d_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
P3D_DEV float p3d_draw(uint32_t seed_lo, uint32_t seed_hi, uint32_t stream, size_t ray, int idx) {
    uint32_t h = p3d_fmix32(seed_lo ^ ((uint32_t)ray * 0x9E3779B1u));
    h = p3d_fmix32(h ^ seed_hi ^ ((uint32_t)((unsigned long long)ray >> 32) * 0x7FEB352Du) ^ (((uint32_t)idx * 2u + stream) * 0x846CA68Bu));
    return (float)(h >> 8) * 0x1p-24f;  // 24 random bits in [0, 1): exact in binary32
}

// weight of interval (prev, cur); advances the transmittance.  ray_marcher.py:26-42
P3D_DEV float p3d_march_weight(MarchState& st, float t, float sigma, float& tm_out) {
    float dl = t - st.prev_t;
    float sm = (st.prev_sigma + sigma) * 0.5f;
    float tm = (st.prev_t + t) * 0.5f;
    float rho = p3d_softplus(sm - 1.0f);
    float dd = rho * dl;
    float alpha = 1.0f - p3d_exp(-dd);
    float T = (float)st.Td;
    float w = alpha * T;
    st.Td = st.Td * (double)((1.0f - alpha) + 1e-10f);
    tm_out = tm;
    return w;
}

// Batcher odd-even merge sort network on NR register-resident keys; fully unrolled at compile time.  The network is the one for
// N = the next power of two with keys NR..N-1 = +inf: a comparator writes min to the lower and max to the upper index, so one
// whose upper index is >= NR never changes anything and is simply not emitted (NR = 48: 543 -> 384 comparators, NR = 96: 1471 ->
// 1056).
P3D_DEV constexpr int p3d_pow2_ceil(int n) { int p = 1; while (p < n) p <<= 1; return p; }
template <int NR>
P3D_DEV void p3d_sort_network(float (&a)[NR]) {
    constexpr int N = p3d_pow2_ceil(NR);
#pragma unroll
    for (int pp = 1; pp < N; pp <<= 1) {
#pragma unroll
        for (int k = pp; k >= 1; k >>= 1) {
#pragma unroll
            for (int jj = k % pp; jj + k < N; jj += 2 * k) {
#pragma unroll
                for (int i = 0; i < k; ++i) {
                    if (i + jj + k < NR && (i + jj) / (2 * pp) == (i + jj + k) / (2 * pp)) {
                        float x = a[i + jj], y = a[i + jj + k];
                        a[i + jj] = __builtin_fminf(x, y);
                        a[i + jj + k] = __builtin_fmaxf(x, y);
                    }
                }
            }
        }
    }
}

// insertion sort of rows [0, n) of a per-wave LDS column (generic / rare path)
template <int RS = 32>  // RS: floats per LDS row (32 rays per wave; 8 in k_render_quad)
P3D_DEV void p3d_lds_insertion_sort(float* A, int n, int j) {
    for (int i = 1; i < n; ++i) {
        float key = A[i * RS + j];
        int q = i - 1;
        while (q >= 0) {
            float v = A[q * RS + j];
            if (!(v > key)) break;
            A[(q + 1) * RS + j] = v;
            --q;
        }
        A[(q + 1) * RS + j] = key;
    }
}

// one inverse-CDF draw: renderer.py:371-386.  cdf rows [0, Ns], coarse depths tc rows [0, Sc)
template <int RS = 32>
P3D_DEV float p3d_inverse_cdf(const float* cdfA, const float* tcA, int Ns, int j, float ui, int& k_out) {
    // k = #{q in 0..Ns : cdf[q] <= u}  (searchsorted right=True): branchless binary search on the non-decreasing cdf
    const int n = Ns + 1;
    int pos = 0;
#pragma unroll
    for (int step = 128; step >= 1; step >>= 1) {
        int np = pos + step;
        if (step <= n) {  // wave-uniform
            bool ok = (np <= n) && (cdfA[((np <= n) ? np - 1 : 0) * RS + j] <= ui);
            pos = ok ? np : pos;
        }
    }
    int k = pos;
    int below = k - 1 > 0 ? k - 1 : 0;
    int above = k < Ns ? k : Ns;
    float cb = cdfA[below * RS + j], ca = cdfA[above * RS + j];
    float den = ca - cb;
    if (den < 1e-5f) den = 1.0f;
    float bb = 0.5f * (tcA[below * RS + j] + tcA[(below + 1) * RS + j]);
    float ba = 0.5f * (tcA[above * RS + j] + tcA[(above + 1) * RS + j]);
    k_out = k;
    return bb + ((ui - cb) / den) * (ba - bb);
}

// B independent draws at once: the same arithmetic as p3d_inverse_cdf, with the B LDS reads of every search step in flight
// together (one draw is a chain of 8 dependent LDS round trips; 48 of them back to back were ~9 % of k_render).
template <int B, int RS = 32>
P3D_DEV void p3d_inverse_cdf_batch(const float* cdfA, const float* tcA, int Ns, int j, const float (&ui)[B], float (&out)[B], int (&k_out)[B]) {
    const int n = Ns + 1;
    int pos[B];
#pragma unroll
    for (int q = 0; q < B; ++q) pos[q] = 0;
#pragma unroll
    for (int step = 128; step >= 1; step >>= 1) {
        if (step <= n) {  // wave-uniform
            float c[B];
#pragma unroll
            for (int q = 0; q < B; ++q) c[q] = cdfA[((pos[q] + step <= n) ? pos[q] + step - 1 : 0) * RS + j];
#pragma unroll
            for (int q = 0; q < B; ++q) pos[q] = ((pos[q] + step <= n) && (c[q] <= ui[q])) ? pos[q] + step : pos[q];
        }
    }
    float cb[B], ca[B], t0[B], t1[B], t2[B], t3[B];
#pragma unroll
    for (int q = 0; q < B; ++q) {
        const int k = pos[q], below = k - 1 > 0 ? k - 1 : 0, above = k < Ns ? k : Ns;
        k_out[q] = k;
        cb[q] = cdfA[below * RS + j]; ca[q] = cdfA[above * RS + j];
        t0[q] = tcA[below * RS + j]; t1[q] = tcA[(below + 1) * RS + j];
        t2[q] = tcA[above * RS + j]; t3[q] = tcA[(above + 1) * RS + j];
    }
#pragma unroll
    for (int q = 0; q < B; ++q) {
        float den = ca[q] - cb[q];
        if (den < 1e-5f) den = 1.0f;
        const float bb = 0.5f * (t0[q] + t1[q]), ba = 0.5f * (t2[q] + t3[q]);
        out[q] = bb + ((ui[q] - cb[q]) / den) * (ba - bb);
        // the result exists HERE: without this the optimiser sinks the lerp (and keeps its six LDS operands alive) down to the
        // first use — the sorting network behind the last batch: 6 x Sf live values, 82-169 spilled VGPRs in the NF = 48 / 96 kernels
        asm volatile("" : "+v"(out[q]));
    }
}

// The same with the coarse depths behind a functor tc(index) instead of an LDS column (k_render's TCG instantiations recompute
// them from the jitter tensor).  Identical arithmetic.
template <int B, typename TCF>
P3D_DEV void p3d_inverse_cdf_batch_f(const float* cdfA, TCF tc, int Ns, int j, const float (&ui)[B], float (&out)[B], int (&k_out)[B]) {
    const int n = Ns + 1;
    int pos[B];
#pragma unroll
    for (int q = 0; q < B; ++q) pos[q] = 0;
#pragma unroll
    for (int step = 128; step >= 1; step >>= 1) {
        if (step <= n) {  // wave-uniform
            float c[B];
#pragma unroll
            for (int q = 0; q < B; ++q) c[q] = cdfA[((pos[q] + step <= n) ? pos[q] + step - 1 : 0) * 32 + j];
#pragma unroll
            for (int q = 0; q < B; ++q) pos[q] = ((pos[q] + step <= n) && (c[q] <= ui[q])) ? pos[q] + step : pos[q];
        }
    }
    float cb[B], ca[B], t0[B], t1[B], t2[B], t3[B];
#pragma unroll
    for (int q = 0; q < B; ++q) {
        const int k = pos[q], below = k - 1 > 0 ? k - 1 : 0, above = k < Ns ? k : Ns;
        k_out[q] = k;
        cb[q] = cdfA[below * 32 + j]; ca[q] = cdfA[above * 32 + j];
        t0[q] = tc(below); t1[q] = tc(below + 1);
        t2[q] = tc(above); t3[q] = tc(above + 1);
    }
#pragma unroll
    for (int q = 0; q < B; ++q) {
        float den = ca[q] - cb[q];
        if (den < 1e-5f) den = 1.0f;
        const float bb = 0.5f * (t0[q] + t1[q]), ba = 0.5f * (t2[q] + t3[q]);
        out[q] = bb + ((ui[q] - cb[q]) / den) * (ba - bb);
        asm volatile("" : "+v"(out[q]));  // (see p3d_inverse_cdf_batch)
    }
}

// NF: register capacity for the fine depths (sorted by a network); NF == 0: generic path, fine depths sorted in LDS.
//   NF = 48 / 96: Sf == NF exactly (the trainer's 48+48 and the eval-faithful 96+96 of eg3dc_v0.py:30-31): no padding keys, no
//   `i < Sf` predicates (64 uniform predicates held in SGPR pairs were the 102-107 SGPR spills of round 2's NF = 64 kernels);
//   NF = 64: any Sf <= 64, padded with +inf.  Sf in (64, 128] other than 96 takes the LDS path.
//   NF >= 96 is compiled for ONE wave per SIMD (512 registers): its LDS rows (Sc + Sf + bit rows, 26 KB per wave at 96+96) cap a
//   CU at 4-5 waves anyway, and round 2's 128-key variant under the 256-register cap spilled 31-39 VGPRs to scratch.
// DUMP: per-stage dumps (parity tests; disables the early-outs so that every dumped density is a real decode).
//
// Exact early-outs (EARLY = !DUMP && !(flags & P3D_FLAG_NO_EARLY_OUT)), both decided per wavefront:
//   * dead rays: once the binary64 transmittance of a ray is below 1e-60 every later weight alpha * (float)Td is exactly 0
//     (alpha <= 1; Td can grow by at most (1 + 1e-10) per step), so its remaining samples cannot change any output;
//   * cropped samples: triplane_crop masks by POSITION (renderer.py:138-149), so sigma = -1000 is known without a decode.
//   A step whose 32 rays are all dead or cropped skips gather + MLP; in a mixed step the dead / cropped LANES get
//   out-of-bounds gather offsets, i.e. they issue no L1 lookups.
//   In the final pass a skipped sample's colour is needed only if one of its two interval weights is non-zero (sigma of
//   the neighbour >= ~794); that is checked on the exact weights and, if it ever happens, the sample is decoded after
//   all — results are bit-identical by construction.
// the host picks the workgroup size (1, 2 or 4 waves) that fills the CU's 160 KB of LDS best
// FAST (P3D_FLAG_FAST_COLOR): the FINAL pass decodes in tolerance mode (p3d_decode_wave_fast); the coarse pass, and with it
// the importance resampling (inverse-CDF indices, fine depths, merged depth order), stays on the exact contract.
// EARLY: the exact early-outs (compile-time, so that the measurement / dump variant is the plain uniform loop).
// TCG ("coarse depths from global"): the production (EARLY) 96-key kernel keeps NO coarse-depth rows in LDS — Sc + Sf = 192 depth
// rows are 24.5 KB per wave, which capped a CU at 4 waves = ONE per SIMD (round 3: 512^2 x (96+96) every sample decoded took 1.39x
// twice the 48+48 time).  A coarse depth is a pure function of (index, jitter value): t_i = lin_i + jitter[ray][i] * delta, so it is
// recomputed wherever it is needed — streamed in index order in the coarse pass and the merge pre-pass (the jitter row is read
// like the stratified phase reads it), fetched per lane (one 4-byte load from the L2-resident jitter tensor, or the in-kernel
// generator) in the inverse-CDF lerp and in the final walk, where the NEXT coarse depth of every lane is prefetched under the
// current decode.  Rows per wave: max(Sc, Sf) + 12 bit rows = 108 -> 13.8 KB -> 8 waves per CU, two per SIMD, also with the
// tolerance-mode LDS image; the known-masked bits of the coarse samples live in three registers.  With jitter in [0, 1), as
// torch.rand_like produces it, the stratified depths can be out of order only between NEIGHBOURS (rounding), which the sorted
// accessor resolves with two more reads; any other disorder takes an exact O(Sc^2) selection (tc_sorted below).
#define P3D_NF_EXACT(NF) ((NF) == 48 || (NF) == 96)
#define P3D_TCG(NF, DUMP, EARLY) ((NF) == 96 && (EARLY) && !(DUMP))
#define P3D_NF_OCC(NF, TCGV) (((NF) >= 96 && !(TCGV)) ? 1 : P3D_RENDER_OCC)
// TCG is a template parameter of its own (default: on wherever it can be): the host turns it OFF for the launches whose coarse
// depths are not the plain stratified spacing — per-ray limits (ray_start = 'auto') and disparity spacing — which depth_of() below
// does not know (with TCG derived from (NF, DUMP, EARLY) alone those launches silently rendered the fixed spacing).
template <int NF, bool DUMP, bool FAST, bool EARLY, bool TCG = P3D_TCG(NF, DUMP, EARLY)>
__global__ __launch_bounds__(64 * P3D_RENDER_WAVES, P3D_NF_OCC(NF, TCG)) void k_render(RenderParams p) {
    static_assert(!(DUMP && EARLY), "dumps need every sample decoded");
    static_assert(!TCG || P3D_TCG(NF, DUMP, EARLY), "TCG exists only for the production 96-key kernel");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    p3d_load_mlp_to_lds(lds, p.w0, p.b0, p.w1, p.b1, !FAST);
    if constexpr (FAST) p3d_load_mlp_f16_to_lds(lds, p.w0, p.w1);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    // XCD-aware block swizzle: hardware places block b on XCD b % 8; give each XCD a contiguous range of tiles so that
    // the plane texels its rays touch stay in that XCD's L2.
    // Each XCD gets runs of `swz` consecutive blocks, the runs interleaved round-robin over the 8 XCDs: contiguous enough for
    // L2 reuse, fine enough that the XCDs finish together although the early-outs make the work per tile uneven.
    long long nblk = gridDim.x, b = blockIdx.x;
    const long long swz = p.swz;
    long long full = (nblk / (8 * swz)) * (8 * swz);
    long long bs = b;
    if (b < full) {
        long long slot = b >> 3, x = b & 7;  // slot-th block of XCD x
        // p.blocked (round 5): a run is one 16 x 16-tile SUPER-TILE (below); XCD x takes element (x + g) % 8 of group g of eight runs,
        // so that over its runs it visits every column of super-tiles (the subject sits in the middle columns: a fixed column per
        // XCD would leave the XCDs of the outer columns idle early)
        const long long g = slot / swz, xr = p.blocked ? ((x + g) & 7) : x;
        bs = g * (8 * swz) + xr * swz + (slot % swz);
    }
    const int nwaves = blockDim.x >> 6;
    long long tile = bs * nwaves + wave;
    if (tile >= p.ntiles) return;  // no workgroup barrier below this line
    float* wl = lds + (FAST ? P3D_LDS_FAST_FLOATS : P3D_LDS_MLP_FLOATS) + 4 + (size_t)wave * p.lds_rows * 32;  // per-wave rows, 16-B aligned

    const int Sc = p.Sc, Sf = P3D_NF_EXACT(NF) ? NF : p.Sf, S = Sc + Sf;
    long long n = tile / p.tiles_per_img, tl = tile - n * p.tiles_per_img;
    long long r;
    if (p.tile_w > 0) {
        long long ty = tl / p.tiles_x, tx = tl - ty * p.tiles_x;
        if (p.blocked) {
            // Blocked tile order (round 5): consecutive tiles fill a SUPER-TILE of 16 x 16 tiles (128 x 64 pixels)
            // before the next one starts, and the XCD swizzle above hands an XCD whole super-tiles — so the ~256 waves resident on an
            // XCD march through a compact screen block whose footprint in the three planes (x-extent 1/4, y-extent 1/8 of the image:
            // ~0.3 + 2.1 + 1.0 MB) fits that XCD's 4 MB L2.  Row-major order gave an XCD four full-width tile rows: the whole xz plane
            // (8.4 MB) streamed through every L2 (measured round 4: the 25 MB of planes fetched ~5 x per XCD).  A pure permutation of
            // which wave renders which tile: results are bit-identical.  The host sets it when the tile grid divides into super-tiles.
            const long long st = tl >> 8;
            const int q = (int)(tl & 255), SX = p.tiles_x >> 4;
            tx = (st % SX) * 16 + (q & 15);
            ty = (st / SX) * 16 + (q >> 4);
        }
        // 8x4 pixel tile in Morton-like lane order: every lane quad is a 2x2 pixel block, so the quad's four gathers of a
        // tap fall into 1-2 cache lines (the L1 coalesces within a quad; rocprof: TCP accesses/instr 48 -> see DESIGN.md)
        const int lx = (j & 1) | ((j >> 1) & 6), ly = ((j >> 1) & 1) | ((j >> 3) & 2);
        r = (ty * 4 + ly) * p.tile_w + tx * 8 + lx;
    } else {
        r = tl * 32 + j;
    }
    const bool active = r < p.R;
    const long long rc = active ? r : p.R - 1;
    const size_t ray = (size_t)n * p.R + rc;

    P3dPlaneGeom g;
    g.halfW = 0.5f * (float)p.W; g.halfH = 0.5f * (float)p.H; g.fW = (float)p.W; g.fH = (float)p.H; g.W = p.W;
    g.plane_bytes = (uint32_t)p.H * (uint32_t)p.W * 128u;
    unsigned nlo = __builtin_amdgcn_readfirstlane((unsigned)n);
    const float* pbase = p.planes + ((p.cfg.flags & P3D_FLAG_SHARED_PLANES) ? (size_t)0 : (size_t)nlo * 3 * (g.plane_bytes / 4));
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)pbase, 0, 3 * g.plane_bytes, 0x00020000);
    const P3dDecodeCfg cfg = p.cfg;
    constexpr bool early = EARLY;
    const bool f_crop = (cfg.flags & P3D_FLAG_CROP) != 0;
    int ndec = 0;  // decode steps this wave executed (statistics)

    const float ox = p.rays_o[ray * 3], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
    const float dx = p.rays_d[ray * 3], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];

    // LDS rows of this wave: row(i)[j]
    float* tcA = wl;             // [Sc]            coarse depths (TCG: no such rows)
    float* wcA = TCG ? wl : tcA + Sc * 32;  // [max(Sc,Sf)]    coarse weights -> pdf/cdf (row 0 = cdf[0]) -> (NF path) sorted fine depths
    float* tfA = (NF > 0) ? wcA : wcA + Sc * 32;  // [Sf] sorted fine depths
    uint32_t* mkA = (uint32_t*)(wl + (size_t)(p.lds_rows - ((Sc + 31) >> 5)) * 32);  // [ceil(Sc/32)] known-masked bits of the coarse samples
    const int nmw = (S + 31) >> 5;                         // words of a bit row over the merged list
    uint32_t* knA = TCG ? (uint32_t*)(wl + (size_t)(Sc > Sf ? Sc : Sf) * 32) : mkA - (size_t)2 * nmw * 32;  // [ceil(S/32)] merged sample q: sigma = -1000 known without a decode
    uint32_t* slA = knA + (size_t)nmw * 32;                // [ceil(S/32)] merged sample q comes from the coarse list
    // ---- the stratified depth of coarse sample i from its jitter value (renderer.py:324-326): the plain branch of the loop
    // below as a function (the host launches a TCG instantiation only for it: fixed ray_start / ray_end, no disparity spacing)
    const float sd_step = (p.ray_end - p.ray_start) / (float)(Sc - 1);
    auto depth_of = [&](int i, float jv) -> float {
        const float lin = (i < Sc / 2) ? p3d_fma(sd_step, (float)i, p.ray_start) : p3d_fma(-sd_step, (float)(Sc - 1 - i), p.ray_end);
        return lin + jv * p.depth_delta;
    };
    const float* jitp = p.jitter + ray * Sc;  // (not dereferenced with the in-kernel generator)
    auto jit_at = [&](int i) -> float { return p.rng ? p3d_draw(p.seed_lo, p.seed_hi, 0u, ray, i) : jitp[i]; };
    auto tc_raw = [&](int i) -> float { return depth_of(i, jit_at(i)); };  // coarse depth i in DRAW order (bins of the pdf)
    bool wave_unsorted = false;  // TCG: some ray of this wave has two neighbouring stratified depths out of order
    bool bad_order = false;      // TCG: disorder beyond neighbours somewhere in this RAY (jitter outside [0, 1))
    bool wave_bad = false;       //      ... somewhere in this wave
    // coarse depth of sorted RANK i (what the stable sort of renderer.py:289-301 puts there): the draw-order value itself unless
    // the wave saw a reversed pair — then the neighbour that rounding swapped in, or (jitter outside [0, 1): never from
    // torch.rand_like) the rank-i element of the row by counting, O(Sc^2) jitter reads per access: slow, exact, and not a
    // path any renderer.py call reaches
    auto tc_sorted = [&](int i) -> float {
        float a = tc_raw(i);
        if (wave_unsorted) {  // wave-uniform
            if (!wave_bad) {
                const float lo = i > 0 ? tc_raw(i - 1) : -__builtin_inff(), hi = i < Sc - 1 ? tc_raw(i + 1) : __builtin_inff();
                a = lo > a ? lo : (a > hi ? hi : a);
            } else {
                for (int c = 0; c < Sc; ++c) {
                    const float tv = tc_raw(c);
                    int r = 0;
                    for (int x = 0; x < Sc; ++x) {
                        const float tx = tc_raw(x);
                        r += (tx < tv || (tx == tv && x < c)) ? 1 : 0;
                    }
                    a = (r == i) ? tv : a;
                }
            }
        }
        return a;
    };
    uint32_t mw0 = 0u, mw1 = 0u, mw2 = 0u;  // TCG: known-masked bits of coarse samples 0-31 / 32-63 / 64-95 (registers, not LDS rows)
    uint32_t fw0 = 0u, fw1 = 0u, fw2 = 0u;  // TCG: sorted fine sample k is cropped (sigma = -1000 by position)
    const bool dump = DUMP && active && h == 0;

    // ---- sample_stratified: renderer.py:320-324
    bool unsorted = false;
    float tcmin = __builtin_inff(), tcmax = -__builtin_inff();  // extrema of the coarse depths
    if constexpr (!TCG) {
        const float step = (p.ray_end - p.ray_start) / (float)(Sc - 1);
        const float* jit = p.jitter + ray * Sc;
        // per-ray limits (ray_start = ray_end = 'auto'): math_utils.linspace + per-ray depth_delta (renderer.py:317-319)
        const bool limits = p.ray_start_arr != nullptr;
        const float rs = limits ? p.ray_start_arr[ray] : 0.0f, span = limits ? p.ray_end_arr[ray] - rs : 0.0f;
        const float rdelta = span / (float)(Sc - 1);
        float prev = -__builtin_inff();
        for (int i0 = 0; i0 < Sc; i0 += 8) {  // eight loads of the jitter row in flight (one at a time exposed a global-load latency per sample)
            float jv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int iq = i0 + q < Sc ? i0 + q : Sc - 1;
                jv[q] = p.rng ? p3d_draw(p.seed_lo, p.seed_hi, 0u, ray, iq) : jit[iq];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = i0 + q;
                if (i < Sc) {  // wave-uniform
                    float lin = (i < Sc / 2) ? p3d_fma(step, (float)i, p.ray_start) : p3d_fma(-step, (float)(Sc - 1 - i), p.ray_end);
                    float t = lin + jv[q] * p.depth_delta;
                    if (limits) {  // wave-uniform
                        const float prod = ((float)i / (float)(Sc - 1)) * span;
                        t = (rs + prod) + jv[q] * rdelta;
                    } else if (p.disparity) {  // renderer.py:309-316: uniform in 1 / depth; p.ray_start / ray_end hold the reciprocals
                        const float s01 = 1.0f / (float)(Sc - 1);
                        const float l01 = (i < Sc / 2) ? p3d_fma(s01, (float)i, 0.0f) : p3d_fma(-s01, (float)(Sc - 1 - i), 1.0f);
                        const float dd = l01 + jv[q] * p.depth_delta;
                        const float ta_ = p.ray_start * (1.0f - dd), tb_ = p.ray_end * dd;
                        t = 1.0f / (ta_ + tb_);
                    }
                    tcA[i * 32 + j] = t;
                    unsorted |= (t < prev);
                    prev = t;
                    tcmin = __builtin_fminf(tcmin, t);
                    tcmax = __builtin_fmaxf(tcmax, t);
                    if constexpr (DUMP) if (dump && p.dumps.depths_coarse) p.dumps.depths_coarse[ray * Sc + i] = t;
                }
            }
        }
    }
    float tmin = __builtin_inff(), tmax = -__builtin_inff();
    if (Sf > 0) {
        // ---- coarse pass, densities only -> ray-marcher weights: renderer.py:179-211
        MarchState st;
        st.Td = 1.0; st.W = 0.0f; st.D = 0.0f; st.prev_t = 0.0f; st.prev_sigma = 0.0f;
        uint32_t mword = 0;  // bit i & 31: coarse sample i is KNOWN to carry sigma = -1000 (cropped, or really decoded and masked)
        if constexpr (TCG) {
            // stratified depths and coarse pass in one loop: the jitter row is read eight values at a time, the depths never
            // reach LDS
            float prev = -__builtin_inff(), pmax = -__builtin_inff();  // previous depth; maximum of all depths before it
            for (int i0 = 0; i0 < Sc; i0 += 8) {
                float jv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) jv[q] = jit_at(i0 + q < Sc ? i0 + q : Sc - 1);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = i0 + q;
                    if (i >= Sc) break;  // wave-uniform
                    const float t = depth_of(i, jv[q]);
                    unsorted |= (t < prev);
                    bad_order |= (t < pmax);
                    pmax = __builtin_fmaxf(pmax, prev);
                    prev = t;
                    tcmin = __builtin_fminf(tcmin, t);
                    tcmax = __builtin_fmaxf(tcmax, t);
                    const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;  // renderer.py:179
                    float sigma = P3D_SIGMA_MASKED;
                    const bool cropped = f_crop && (__builtin_fabsf(px) > cfg.crop_limit || __builtin_fabsf(pz) > cfg.crop_limit);
                    const bool live = !(cropped || st.Td < 1e-60);
                    const bool skip = __builtin_amdgcn_ballot_w64(live) == 0;
                    if (!skip) {
                        f32x16 dummy;
                        p3d_decode_wave<false, (FAST || P3D_QUAD_EXACT != 0) && (P3D_QUAD_COARSE != 0)>(lds, rs, g, cfg, px, py, pz, sigma, dummy, live);
                    }
                    mword |= (cropped || (live && sigma == P3D_SIGMA_MASKED)) ? (1u << (i & 31)) : 0u;
                    if ((i & 31) == 31 || i == Sc - 1) {  // wave-uniform
                        if ((i >> 5) == 0) mw0 = mword; else if ((i >> 5) == 1) mw1 = mword; else mw2 = mword;
                        mword = 0;
                    }
                    if (i > 0) {
                        float tm;
                        wcA[(i - 1) * 32 + j] = p3d_march_weight(st, t, sigma, tm);
                    }
                    st.prev_t = t; st.prev_sigma = sigma;
                    ndec += skip ? 0 : 1;
                }
            }
        } else
        for (int i = 0; i < Sc; ++i) {
            float t = tcA[i * 32 + j];
            float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;  // renderer.py:179
            float sigma = P3D_SIGMA_MASKED;
            bool skip = false, live = true, cropped = false;
            if (early) {
                cropped = f_crop && (__builtin_fabsf(px) > cfg.crop_limit || __builtin_fabsf(pz) > cfg.crop_limit);
                live = !(cropped || st.Td < 1e-60);  // a cropped sample is -1000 by position; a dead ray's weights are 0
                skip = __builtin_amdgcn_ballot_w64(live) == 0;
            }
            if (!skip) {
                f32x16 dummy;
                p3d_decode_wave<false, (FAST || P3D_QUAD_EXACT != 0) && (P3D_QUAD_COARSE != 0)>(lds, rs, g, cfg, px, py, pz, sigma, dummy, live);
            }
            if (early) {  // a lane whose gathers were suppressed (dead ray) decoded garbage: its sample is NOT known to be masked
                mword |= (cropped || (live && sigma == P3D_SIGMA_MASKED)) ? (1u << (i & 31)) : 0u;
                if ((i & 31) == 31 || i == Sc - 1) { mkA[(i >> 5) * 32 + j] = mword; mword = 0; }
            }
            if constexpr (DUMP) if (dump && p.dumps.sigma_coarse) p.dumps.sigma_coarse[ray * Sc + i] = sigma;
            if (i > 0) {
                float tm;
                float w = p3d_march_weight(st, t, sigma, tm);
                wcA[(i - 1) * 32 + j] = w;
                if constexpr (DUMP) if (dump && p.dumps.weights_coarse) p.dumps.weights_coarse[ray * (Sc - 1) + i - 1] = w;
            }
            st.prev_t = t; st.prev_sigma = sigma;
            if constexpr (!DUMP) ndec += skip ? 0 : 1;
        }
        // ---- sample_importance / sample_pdf: renderer.py:328-387 (per ray; both lanes of a pair compute the same)
        const int Ns = Sc - 3;
        {
            // v[jj] = ws[jj+1] + 1e-5, ws[q] = (max(w[q-1],w[q]) + max(w[q],w[q+1])) * 0.5 + 0.01 ; stored at row jj+1
            double sum = 0.0;
            float wa = wcA[0 * 32 + j], wb = wcA[1 * 32 + j];
            for (int jj = 0; jj < Ns; ++jj) {
                float wc = wcA[(jj + 2) * 32 + j];
                float m1 = __builtin_fmaxf(wa, wb), m2 = __builtin_fmaxf(wb, wc);
                float v = ((m1 + m2) * 0.5f + 0.01f) + 1e-5f;
                sum += (double)v;
                wcA[(jj + 1) * 32 + j] = v;
                wa = wb; wb = wc;
            }
            float fsum = (float)sum;
            double acc = 0.0;
            wcA[j] = 0.0f;  // cdf[0]
            for (int jj = 0; jj < Ns; ++jj) {
                float pdf = wcA[(jj + 1) * 32 + j] / fsum;
                acc += (double)pdf;
                wcA[(jj + 1) * 32 + j] = (float)acc;  // cdf[jj+1]
            }
        }
        const float* uu = p.u + ray * Sf;
        constexpr int DB = 8;  // draws in flight
        if constexpr (NF > 0) {
            float tf[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i)  // every load of the row issued before the first search
                tf[i] = (i < Sf) ? (p.rng ? p3d_draw(p.seed_lo, p.seed_hi, 1u, ray, i) : uu[i]) : 0.0f;
#pragma unroll
            for (int i0 = 0; i0 < NF; i0 += DB) {
                if (i0 < Sf) {  // wave-uniform
                    float ub[DB], vb[DB];
                    int kb[DB];
#pragma unroll
                    for (int q = 0; q < DB; ++q) ub[q] = tf[i0 + q];
                    if constexpr (TCG) p3d_inverse_cdf_batch_f<DB>(wcA, tc_raw, Ns, j, ub, vb, kb);
                    else p3d_inverse_cdf_batch<DB>(wcA, tcA, Ns, j, ub, vb, kb);
#pragma unroll
                    for (int q = 0; q < DB; ++q) {
                        tf[i0 + q] = (i0 + q < Sf) ? vb[q] : __builtin_inff();
                        if constexpr (DUMP) {
                            if (dump && i0 + q < Sf && p.dumps.depths_fine) p.dumps.depths_fine[ray * Sf + i0 + q] = vb[q];
                            if (dump && i0 + q < Sf && p.dumps.inds) p.dumps.inds[ray * Sf + i0 + q] = kb[q];
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < DB; ++q) tf[i0 + q] = __builtin_inff();
                }
            }
            p3d_sort_network<NF>(tf);
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if (i < Sf) {
                    tfA[i * 32 + j] = tf[i];  // over the cdf rows: every search is done
                    if constexpr (TCG) {
                        const float px = ox + tf[i] * dx, pz = oz + tf[i] * dz;
                        const uint32_t b = (f_crop && (__builtin_fabsf(px) > cfg.crop_limit || __builtin_fabsf(pz) > cfg.crop_limit)) ? (1u << (i & 31)) : 0u;
                        if (i < 32) fw0 |= b; else if (i < 64) fw1 |= b; else fw2 |= b;
                    }
                }
        } else {
            for (int i0 = 0; i0 < Sf; i0 += DB) {
                float ub[DB], vb[DB];
                int kb[DB];
#pragma unroll
                for (int q = 0; q < DB; ++q) {
                    const int iq = i0 + q < Sf ? i0 + q : Sf - 1;
                    ub[q] = p.rng ? p3d_draw(p.seed_lo, p.seed_hi, 1u, ray, iq) : uu[iq];
                }
                p3d_inverse_cdf_batch<DB>(wcA, tcA, Ns, j, ub, vb, kb);
#pragma unroll
                for (int q = 0; q < DB; ++q) {
                    if (i0 + q < Sf) {
                        tfA[(i0 + q) * 32 + j] = vb[q];
                        if constexpr (DUMP) {
                            if (dump && p.dumps.depths_fine) p.dumps.depths_fine[ray * Sf + i0 + q] = vb[q];
                            if (dump && p.dumps.inds) p.dumps.inds[ray * Sf + i0 + q] = kb[q];
                        }
                    }
                }
            }
            p3d_lds_insertion_sort(tfA, Sf, j);
        }
        // unify_samples (renderer.py:289-301) merges two sorted lists; the stratified list is sorted unless rounding
        // reversed two neighbours (practically never) — then sort it too.
        if constexpr (TCG) {
            wave_unsorted = __builtin_amdgcn_ballot_w64(unsorted) != 0;  // from here on tc_sorted() looks at the neighbours
            wave_bad = __builtin_amdgcn_ballot_w64(bad_order) != 0;
            if (wave_unsorted) { mw0 = 0u; mw1 = 0u; mw2 = 0u; }         // (bits indexed by draw order: forget them, as below)
        } else
        if (__builtin_amdgcn_ballot_w64(unsorted) != 0) {
            p3d_lds_insertion_sort(tcA, Sc, j);
            if (early) {  // the known-masked bits are indexed by ORIGINAL coarse index: forget them (practically never taken)
                for (int i = 0; i < ((Sc + 31) >> 5); ++i) mkA[i * 32 + j] = 0u;
            }
        }
    }
    // ---- final pass: merge on the fly (ties: coarse first = stable), decode, composite [rgb | xyz]:
    //      renderer.py:243-259, ray_marcher.py:25-57.
    // Every lane walks ITS OWN merged list.  With the early-outs on, a lane first consumes — without a decode — every sample
    // whose sigma = -1000 is already known (cropped by position, or a coarse sample the coarse pass really decoded and found
    // masked) while the previous sample's sigma is <= 602: the interval's softplus argument is then <= -200, rho = alpha = w = 0
    // and Td * (double)(1.0f + 1e-10f) = Td, i.e. the marcher update is exactly the identity (include/p3d_numerics.h); only
    // prev_t / prev_sigma move.  A dead ray (Td < 1e-60: every later weight is exactly 0) drops all its remaining samples.
    // The merge itself runs ONCE, ahead of the walk, as a uniform loop that leaves two bit rows per lane in LDS (merged sample q
    // is coarse / is known); the walk then jumps over a whole run of known samples with a count-trailing-ones — consuming them
    // one at a time was a chain of two LDS round trips per sample and a quarter of the kernel (profiles/history/r02_notes.txt).
    // What is left — the samples that can matter — is decoded one per wave-step until every lane has finished, so a tile of
    // rays that miss the subject runs ~Sf/2 steps instead of Sc + Sf.  A consumed sample's colour is fetched after all if one
    // of its interval weights turns out non-zero (sigma of a neighbour >= ~794): results are bit-identical by construction.
    MarchState st;
    st.Td = 1.0; st.W = 0.0f; st.D = 0.0f; st.prev_t = 0.0f; st.prev_sigma = 0.0f;
    f32x16 C, prev_rgb;
    float Cx = 0.0f, Cy = 0.0f, Cz = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { C[c] = 0.0f; prev_rgb[c] = 0.0f; }
    {
        int ci = 0, fi = 0;
        float ta = TCG ? 0.0f : tcA[j], tb = (Sf > 0) ? tfA[j] : __builtin_inff();
        // the extrema of all depths of the ray (the global clamp range, ray_marcher.py:50); the fine list is sorted
        tmin = __builtin_fminf(tcmin, tb);
        tmax = __builtin_fmaxf(tcmax, (Sf > 0) ? tfA[(Sf - 1) * 32 + j] : -__builtin_inff());
        bool prev_skipped = false, first = true, done = false;
        int m = 0;  // samples of this lane consumed so far (position in the merged list)
        auto is_cropped = [&](float t) {
            const float px = ox + t * dx, pz = oz + t * dz;
            return f_crop && (__builtin_fabsf(px) > cfg.crop_limit || __builtin_fabsf(pz) > cfg.crop_limit);
        };
        auto advance = [&](bool take_c) {  // pop the head of the coarse or of the fine list
            ci += take_c ? 1 : 0;
            fi += take_c ? 0 : 1;
            // (one LDS read at a selected address and two selects: `if (take_c) ta = ...; else tb = ...;` is turned into a store
            // through a selected POINTER, which parks ta / tb in scratch memory — 1.5-3 k clocks per merged sample)
            const int cq = ci < Sc ? ci : Sc - 1, fq = fi < Sf ? fi : (Sf > 0 ? Sf - 1 : 0);
            const float nv = (take_c ? tcA : tfA)[(take_c ? cq : fq) * 32 + j];
            ta = take_c ? nv : ta;
            tb = take_c ? tb : nv;
        };
        float tcn = 0.0f;  // TCG: the head of the coarse list, tc_sorted(min(ci, Sc - 1)), fetched ahead of its use
        if constexpr (TCG) {
            // the merge without a coarse column: coarse rank i lands at merged position i + #{fine < t_i} (ties: coarse first),
            // found by a search in the sorted fine column, eight ranks in flight; the two bit rows are built in registers (six
            // words each: Sc + Sf <= 192) and stored once
            uint32_t slw[6], knw[6];
#pragma unroll
            for (int w = 0; w < 6; ++w) { slw[w] = 0u; knw[w] = 0u; }
            for (int i0 = 0; i0 < Sc; i0 += 8) {
                float tv[8];
                int pos[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { tv[q] = tc_sorted(i0 + q < Sc ? i0 + q : Sc - 1); pos[q] = 0; }
#pragma unroll
                for (int step = 64; step >= 1; step >>= 1) {
                    float c[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) c[q] = tfA[((pos[q] + step <= Sf) ? pos[q] + step - 1 : 0) * 32 + j];
#pragma unroll
                    for (int q = 0; q < 8; ++q) pos[q] = ((pos[q] + step <= Sf) && (c[q] < tv[q])) ? pos[q] + step : pos[q];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = i0 + q;
                    if (i >= Sc) break;  // wave-uniform
                    const uint32_t mwv = (i < 32) ? mw0 : (i < 64 ? mw1 : mw2);  // (i is uniform)
                    const bool known = is_cropped(tv[q]) || ((mwv >> (i & 31)) & 1u);
                    const int P = i + pos[q], pw = P >> 5;
                    const uint32_t b = 1u << (P & 31);
#pragma unroll
                    for (int w = 0; w < 6; ++w) {
                        slw[w] |= (pw == w) ? b : 0u;
                        knw[w] |= (pw == w && known) ? b : 0u;
                    }
                }
            }
            // cropped fine samples: fine k sits at the k-th zero of the is-coarse row
            if (__builtin_amdgcn_ballot_w64((fw0 | fw1 | fw2) != 0u) != 0) {
                int fk = 0;
#pragma unroll
                for (int w = 0; w < 6; ++w) {
                    if (w * 32 < S) {  // wave-uniform
                        uint32_t kk = 0u;
                        for (int b = 0; b < 32; ++b) {
                            const bool isf = !((slw[w] >> b) & 1u) && (w * 32 + b < S);
                            const uint32_t fwv = (fk < 32) ? fw0 : (fk < 64 ? fw1 : fw2);
                            kk |= (isf && ((fwv >> (fk & 31)) & 1u)) ? (1u << b) : 0u;
                            fk += isf ? 1 : 0;
                        }
                        knw[w] |= kk;
                    }
                }
            }
#pragma unroll
            for (int w = 0; w < 6; ++w)
                if (w < nmw) { knA[w * 32 + j] = knw[w]; slA[w * 32 + j] = slw[w]; }
            tcn = tc_sorted(0);
        } else
        if constexpr (EARLY) {
            // the merge, once: bit q of slA = merged sample q is the head of the coarse list, bit q of knA = its sigma is known
            uint32_t kw = 0u, sw = 0u;
            for (int q = 0; q < S; ++q) {
                const bool take_c = (ci < Sc) && (fi >= Sf || ta <= tb);
                bool known = is_cropped(take_c ? ta : tb);
                if (take_c && Sf > 0) known = known || ((mkA[(ci >> 5) * 32 + j] >> (ci & 31)) & 1u);  // Sf == 0: no coarse pass ran
                kw |= known ? (1u << (q & 31)) : 0u;
                sw |= take_c ? (1u << (q & 31)) : 0u;
                advance(take_c);
                if ((q & 31) == 31 || q == S - 1) {
                    knA[(q >> 5) * 32 + j] = kw; slA[(q >> 5) * 32 + j] = sw;
                    kw = 0u; sw = 0u;
                }
            }
            ci = 0;  // from here on: coarse samples among the first m merged ones (the fine index is m - ci)
        }
        for (int it = 0;; ++it) {
            bool take_c, known = false;  // known: sigma = -1000 without a decode (reached here only behind a sigma > 602)
            float t;
            if constexpr (!EARLY) {
                if (it >= S) break;  // every lane takes exactly one sample per step: the plain uniform loop
                take_c = (ci < Sc) && (fi >= Sf || ta <= tb);
                t = take_c ? ta : tb;
                advance(take_c);
            } else {
                // exact mode: below 1e-60 every later weight alpha * (float)Td is exactly 0.  Tolerance mode: the transmittance
                // bounds everything the rest of the ray can still add (sum of the remaining weights <= Td): stop at 2e-6,
                // i.e. <= 4e-6 on a colour, <= 6e-6 on depth / weight sum — early ray termination, inside the 2e-5 budget
                if (!done && st.Td < (FAST ? 2e-6 : 1e-60)) done = true;
                if (!done && (first || st.prev_sigma <= 602.0f)) {
                    // jump over the run of known samples that starts at m (one iteration per 32-bit word the run touches)
                    bool any = false, last_c = false;
                    while (m < S) {
                        const int sh = m & 31, left = (32 - sh < S - m) ? 32 - sh : S - m;
                        const uint32_t kw = knA[(m >> 5) * 32 + j] >> sh, sw = slA[(m >> 5) * 32 + j] >> sh;
                        int run = (~kw != 0u) ? __builtin_ctz(~kw) : 32;
                        run = run < left ? run : left;
                        if (run > 0) {
                            const uint32_t inrun = run >= 32 ? 0xffffffffu : ((1u << run) - 1u);
                            ci += __builtin_popcount(sw & inrun);
                            last_c = (sw >> (run - 1)) & 1u;
                            any = true;
                            m += run;
                        }
                        if (run < left) break;  // stopped in front of a sample that has to be decoded
                    }
                    if (any) {
                        if constexpr (TCG) {
                            // the two coarse depths a jump needs, requested together: the last consumed one and the new head
                            const float pc = tc_sorted(ci > 0 ? ci - 1 : 0);
                            tcn = tc_sorted(ci < Sc ? ci : Sc - 1);
                            st.prev_t = last_c ? pc : tfA[(m - ci - 1 > 0 ? m - ci - 1 : 0) * 32 + j];
                        } else
                        st.prev_t = last_c ? tcA[(ci - 1) * 32 + j] : tfA[(m - ci - 1) * 32 + j];
                        st.prev_sigma = P3D_SIGMA_MASKED;
                        prev_skipped = true; first = false;
                    }
                    done = m >= S;
                }
                if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
                const int mm = done ? S - 1 : m;  // finished lanes: any valid address
                const uint32_t kb = knA[(mm >> 5) * 32 + j] >> (mm & 31), sb = slA[(mm >> 5) * 32 + j] >> (mm & 31);
                take_c = (sb & 1u) != 0u;
                known = !done && (kb & 1u) != 0u;
                const int cq = ci < Sc ? ci : Sc - 1, fq = (m - ci < Sf) ? m - ci : (Sf > 0 ? Sf - 1 : 0);
                if constexpr (TCG) t = take_c ? tcn : tfA[fq * 32 + j];
                else t = (take_c || Sf == 0) ? tcA[cq * 32 + j] : tfA[fq * 32 + j];
                if (!done) { ++m; ci += take_c ? 1 : 0; }
                if constexpr (TCG) {
                    if (!done && take_c) tcn = tc_sorted(ci < Sc ? ci : Sc - 1);  // used one step later at the earliest: under the decode
                }
            }
            const bool have = EARLY ? !done : true;
            const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
            float sigma = P3D_SIGMA_MASKED;
            f32x16 rgb;
#pragma unroll
            for (int c = 0; c < 16; ++c) rgb[c] = 0.0f;
            const bool live = have && !known;
            bool skipped = true;
            if (__builtin_amdgcn_ballot_w64(live) != 0) {
                bool have_rgb;  // EARLY: colour on demand (p3d_decode.hpp, LAZY): a step whose live samples are all masked has none
                if constexpr (FAST) have_rgb = p3d_decode_wave_fast<true, true, EARLY, true>(lds, rs, g, cfg, px, py, pz, sigma, rgb, live);
                else have_rgb = p3d_decode_wave<true, P3D_QUAD_EXACT != 0, EARLY>(lds, rs, g, cfg, px, py, pz, sigma, rgb, live);
                if constexpr (!DUMP) ndec += 1;
                skipped = !live || !have_rgb;  // per lane: a lane whose gathers were suppressed has no colour
                if (known) sigma = P3D_SIGMA_MASKED;
            }
            if constexpr (DUMP) {
                if (dump && p.dumps.depths_sorted) p.dumps.depths_sorted[ray * S + m] = t;
                if (dump && p.dumps.sigma_sorted) p.dumps.sigma_sorted[ray * S + m] = sigma;
            }
            // (the guard decodes below are wave-level operations: every lane must reach them, also the ones without a sample)
            const bool marching = have && !first;
            float w = 0.0f, tm = 0.0f;
            if (marching) w = p3d_march_weight(st, t, sigma, tm);
            const float ppx = ox + st.prev_t * dx, ppy = oy + st.prev_t * dy, ppz = oz + st.prev_t * dz;
            if (early) {  // exactness guard: a skipped endpoint whose interval weight is non-zero needs its real colour
                if (__builtin_amdgcn_ballot_w64(marching && prev_skipped && w != 0.0f) != 0) {
                    float s2;
                    f32x16 c2;
                    if constexpr (FAST) p3d_decode_wave_fast<true, true>(lds, rs, g, cfg, ppx, ppy, ppz, s2, c2);
                    else p3d_decode_wave<true>(lds, rs, g, cfg, ppx, ppy, ppz, s2, c2);
                    if (prev_skipped) prev_rgb = c2;
                    prev_skipped = false;
                }
                if (__builtin_amdgcn_ballot_w64(marching && skipped && w != 0.0f) != 0) {
                    float s2;
                    if constexpr (FAST) p3d_decode_wave_fast<true, true>(lds, rs, g, cfg, px, py, pz, s2, rgb);
                    else p3d_decode_wave<true>(lds, rs, g, cfg, px, py, pz, s2, rgb);
                    skipped = false;
                }
            }
            if (have) {
                if (!first) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) C[c] = p3d_fma(w, (prev_rgb[c] + rgb[c]) * 0.5f, C[c]);
                    Cx = p3d_fma(w, (ppx + px) * 0.5f, Cx);
                    Cy = p3d_fma(w, (ppy + py) * 0.5f, Cy);
                    Cz = p3d_fma(w, (ppz + pz) * 0.5f, Cz);
                    st.W = st.W + w;
                    st.D = p3d_fma(w, tm, st.D);
                }
                st.prev_t = t; st.prev_sigma = sigma;
                prev_rgb = rgb;
                prev_skipped = skipped;
                first = false;
                if constexpr (!EARLY) ++m;
            }
        }
    }
    // ---- outputs.  white_back and the [-1,1] rescale are per ray (ray_marcher.py:52-55); the depth clamp is global.
    {
        const float Wt = st.W;
        float d = st.D / Wt;
        if (d != d) d = __builtin_inff();  // nan_to_num(nan=inf): ray_marcher.py:49
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float v = C[c];
            if (p.white_back) v = (v + 1.0f) - Wt;
            C[c] = v * 2.0f - 1.0f;
        }
        if (p.white_back) { Cx = (Cx + 1.0f) - Wt; Cy = (Cy + 1.0f) - Wt; Cz = (Cz + 1.0f) - Wt; }
        Cx = Cx * 2.0f - 1.0f; Cy = Cy * 2.0f - 1.0f; Cz = Cz * 2.0f - 1.0f;
        if (active) {
            float* dst = p.out_feat + ray * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 8 * q) = (f32x4){C[4 * q], C[4 * q + 1], C[4 * q + 2], C[4 * q + 3]};
            if (h == 0) {
                p.out_depth[ray] = d;  // clamped by k_render_finish
                p.out_wsum[ray] = Wt;  // weights.sum(2): renderer.py:264
                p.out_xyz[ray * 3] = Cx; p.out_xyz[ray * 3 + 1] = Cy; p.out_xyz[ray * 3 + 2] = Cz;
                if constexpr (DUMP) if (p.dumps.depth_unclamped) p.dumps.depth_unclamped[ray] = st.D / Wt;
            }
        }
    }
    // global min / max of all depths of the call (torch.min/max(depths): ray_marcher.py:50)
    if (!active) { tmin = __builtin_inff(); tmax = -__builtin_inff(); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        tmin = __builtin_fminf(tmin, __shfl_xor(tmin, o));
        tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, o));
    }
    if (lane == 0) {
        atomicMin(p.gminmax, p3d_f2ord(tmin));
        atomicMax(p.gminmax + 1, p3d_f2ord(tmax));
        if (p.per_view_clamp) {  // a tile never straddles two views
            atomicMin(p.gminmax + 4 + 2 * nlo, p3d_f2ord(tmin));
            atomicMax(p.gminmax + 5 + 2 * nlo, p3d_f2ord(tmax));
        }
        if constexpr (!DUMP) atomicAdd((unsigned long long*)(p.gminmax + 2), (unsigned long long)ndec);  // wave-level decode steps
    }
}

// =====================================================================================================================
// k_render_pair: the same algorithm for SMALL launches (fewer 32-ray tiles than the chip has SIMDs, e.g. the pipeline's single
// 128^2-ray views: 512 tiles on 1024 SIMDs, each wave alone on its SIMD and ALU-bound at ~4.7 us per decode step).  A wave owns
// 16 rays and decodes TWO consecutive samples of every ray per step: lane j = ray (j & 15) x sample slot (j >> 4) x channel half,
// so a launch makes twice as many waves, each with half as many decode steps.  Everything per ray (depth rows in LDS, marcher,
// cdf, inverse-CDF draws, sort, merge, compositing) is executed identically by both slots of a ray — same inputs, same order,
// same results, so the arithmetic contract and the accumulation order are untouched — and only the decode differs: after
// it the two slots exchange sigma / skipped flag / their 16 colour channels (ds_bpermute), then both consume sample A and
// sample B in order.  Slot 0 writes the outputs.  No dumps on this path (the host falls back to k_render for them).
// =====================================================================================================================
// FAST (P3D_FLAG_FAST_COLOR): the final pass decodes in tolerance mode exactly as k_render<…, FAST = true> does (two-term f16 MLP
// operands, hardware transcendentals, the exact mask guard, rays dropped below a transmittance of 2e-6); the coarse pass, and
// with it every importance draw, stays on the exact contract.
template <int NF, bool FAST>
__global__ __launch_bounds__(64 * P3D_RENDER_WAVES, 1) void k_render_pair(RenderParams p) {  // 1 workgroup per CU is all a small launch has: up to 512 VGPRs
    extern __shared__ __attribute__((aligned(16))) float lds[];
    p3d_load_mlp_to_lds(lds, p.w0, p.b0, p.w1, p.b1, !FAST);
    if constexpr (FAST) p3d_load_mlp_f16_to_lds(lds, p.w0, p.w1);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int jr = j & 15, slot = j >> 4;
    const int nwaves = blockDim.x >> 6;
    long long tile = (long long)blockIdx.x * nwaves + wave;  // 16-ray tiles
    if (tile >= p.ntiles) return;  // no workgroup barrier below this line
    float* wl = lds + (FAST ? P3D_LDS_FAST_FLOATS : P3D_LDS_MLP_FLOATS) + 4 + (size_t)wave * p.lds_rows * 32;

    const int Sc = p.Sc, Sf = P3D_NF_EXACT(NF) ? NF : p.Sf, S = Sc + Sf;
    long long n = tile / p.tiles_per_img, tl = tile - n * p.tiles_per_img;
    long long r;
    if (p.tile_w > 0) {  // 4x4 pixel tile, Morton lane order
        long long ty = tl / p.tiles_x, tx = tl - ty * p.tiles_x;
        const int lx = (jr & 1) | ((jr >> 1) & 2), ly = ((jr >> 1) & 1) | ((jr >> 2) & 2);
        r = (ty * 4 + ly) * p.tile_w + tx * 4 + lx;
    } else {
        r = tl * 16 + jr;
    }
    const bool active = r < p.R;
    const long long rc = active ? r : p.R - 1;
    const size_t ray = (size_t)n * p.R + rc;

    P3dPlaneGeom g;
    g.halfW = 0.5f * (float)p.W; g.halfH = 0.5f * (float)p.H; g.fW = (float)p.W; g.fH = (float)p.H; g.W = p.W;
    g.plane_bytes = (uint32_t)p.H * (uint32_t)p.W * 128u;
    unsigned nlo = __builtin_amdgcn_readfirstlane((unsigned)n);
    const float* pbase = p.planes + ((p.cfg.flags & P3D_FLAG_SHARED_PLANES) ? (size_t)0 : (size_t)nlo * 3 * (g.plane_bytes / 4));
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)pbase, 0, 3 * g.plane_bytes, 0x00020000);
    const P3dDecodeCfg cfg = p.cfg;
    const bool early = !(cfg.flags & P3D_FLAG_NO_EARLY_OUT);
    const bool f_crop = (cfg.flags & P3D_FLAG_CROP) != 0;
    int ndec = 0;
    // value of the partner slot (lane ^ 16): ds_swizzle bit mode and 0x1f, or 0, xor 0x10 — the crossbar, no address VGPR
    auto partner = [](float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401f));
    };

    const float ox = p.rays_o[ray * 3], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
    const float dx = p.rays_d[ray * 3], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];

    // LDS rows of this wave: row(i)[jr]; both slots of a ray write the same values
    float* tcA = wl;
    float* wcA = tcA + Sc * 32;
    float* tfA = (NF > 0) ? wcA : wcA + Sc * 32;

    bool unsorted = false;
    {
        const float step = (p.ray_end - p.ray_start) / (float)(Sc - 1);
        const float* jit = p.jitter + ray * Sc;
        const bool limits = p.ray_start_arr != nullptr;  // per-ray limits: as in k_render
        const float rs = limits ? p.ray_start_arr[ray] : 0.0f, span = limits ? p.ray_end_arr[ray] - rs : 0.0f;
        const float rdelta = span / (float)(Sc - 1);
        float prev = -__builtin_inff();
        for (int i = 0; i < Sc; ++i) {
            float lin = (i < Sc / 2) ? p3d_fma(step, (float)i, p.ray_start) : p3d_fma(-step, (float)(Sc - 1 - i), p.ray_end);
            const float ji = p.rng ? p3d_draw(p.seed_lo, p.seed_hi, 0u, ray, i) : jit[i];
            float t = lin + ji * p.depth_delta;
            if (limits) {
                const float prod = ((float)i / (float)(Sc - 1)) * span;
                t = (rs + prod) + ji * rdelta;
            } else if (p.disparity) {
                const float s01 = 1.0f / (float)(Sc - 1);
                const float l01 = (i < Sc / 2) ? p3d_fma(s01, (float)i, 0.0f) : p3d_fma(-s01, (float)(Sc - 1 - i), 1.0f);
                const float dd = l01 + ji * p.depth_delta;
                const float ta_ = p.ray_start * (1.0f - dd), tb_ = p.ray_end * dd;
                t = 1.0f / (ta_ + tb_);
            }
            tcA[i * 32 + jr] = t;
            unsorted |= (t < prev);
            prev = t;
        }
    }
    float tmin = __builtin_inff(), tmax = -__builtin_inff();
    auto is_cropped = [&](float px, float pz) {
        return f_crop && (__builtin_fabsf(px) > cfg.crop_limit || __builtin_fabsf(pz) > cfg.crop_limit);
    };
    if (Sf > 0) {
        // ---- coarse pass, two samples per step
        MarchState st;
        st.Td = 1.0; st.W = 0.0f; st.D = 0.0f; st.prev_t = 0.0f; st.prev_sigma = 0.0f;
        for (int i = 0; i < Sc; i += 2) {
            const bool haveB = i + 1 < Sc;  // uniform
            const float tA = tcA[i * 32 + jr], tB = tcA[(haveB ? i + 1 : i) * 32 + jr];
            const float t = slot ? tB : tA;
            const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
            float sigma = P3D_SIGMA_MASKED;
            bool skip = false, live = true;
            if (early) {
                live = !(is_cropped(px, pz) || st.Td < 1e-60);  // slot 1: Td before sample A's interval — conservative, still exact
                skip = __builtin_amdgcn_ballot_w64(live) == 0;
            }
            if (!skip) {
                f32x16 dummy;
                p3d_decode_wave<false, P3D_QUAD_PAIR != 0>(lds, rs, g, cfg, px, py, pz, sigma, dummy, live);
                ndec += 1;
            }
            const float so = partner(sigma);
            const float sA = slot ? so : sigma, sB = slot ? sigma : so;
            if (i > 0) {
                float tm;
                wcA[(i - 1) * 32 + jr] = p3d_march_weight(st, tA, sA, tm);
            }
            st.prev_t = tA; st.prev_sigma = sA;
            if (haveB) {
                float tm;
                wcA[i * 32 + jr] = p3d_march_weight(st, tB, sB, tm);
                st.prev_t = tB; st.prev_sigma = sB;
            }
        }
        const int Ns = Sc - 3;
        {
            double sum = 0.0;
            float wa = wcA[0 * 32 + jr], wb = wcA[1 * 32 + jr];
            for (int jj = 0; jj < Ns; ++jj) {
                float wc = wcA[(jj + 2) * 32 + jr];
                float m1 = __builtin_fmaxf(wa, wb), m2 = __builtin_fmaxf(wb, wc);
                float v = ((m1 + m2) * 0.5f + 0.01f) + 1e-5f;
                sum += (double)v;
                wcA[(jj + 1) * 32 + jr] = v;
                wa = wb; wb = wc;
            }
            float fsum = (float)sum;
            double acc = 0.0;
            wcA[jr] = 0.0f;
            for (int jj = 0; jj < Ns; ++jj) {
                float pdf = wcA[(jj + 1) * 32 + jr] / fsum;
                acc += (double)pdf;
                wcA[(jj + 1) * 32 + jr] = (float)acc;
            }
        }
        const float* uu = p.u + ray * Sf;
        if constexpr (NF > 0) {
            // as in k_render: every load of the row issued before the first search, eight draws in lock-step (the eight LDS
            // reads of a search step in flight together; one draw at a time was a chain of 8 dependent LDS round trips x Sf)
            constexpr int DB = 8;
            float tf[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) tf[i] = (i < Sf) ? (p.rng ? p3d_draw(p.seed_lo, p.seed_hi, 1u, ray, i) : uu[i]) : 0.0f;
#pragma unroll
            for (int i0 = 0; i0 < NF; i0 += DB) {
                if (i0 < Sf) {  // wave-uniform
                    float ub[DB], vb[DB];
                    int kb[DB];
#pragma unroll
                    for (int q = 0; q < DB; ++q) ub[q] = tf[i0 + q];
                    p3d_inverse_cdf_batch<DB>(wcA, tcA, Ns, jr, ub, vb, kb);
#pragma unroll
                    for (int q = 0; q < DB; ++q) tf[i0 + q] = (i0 + q < Sf) ? vb[q] : __builtin_inff();
                } else {
#pragma unroll
                    for (int q = 0; q < DB; ++q) tf[i0 + q] = __builtin_inff();
                }
            }
            p3d_sort_network<NF>(tf);
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if (i < Sf) tfA[i * 32 + jr] = tf[i];
        } else {
            // the generic path sorts in LDS: only slot 0 may move the keys (both slots of a ray share the column)
            float* tmpA = tfA;
            for (int i = 0; i < Sf; ++i) {
                int k;
                float v = p3d_inverse_cdf(wcA, tcA, Ns, jr, p.rng ? p3d_draw(p.seed_lo, p.seed_hi, 1u, ray, i) : uu[i], k);
                tmpA[i * 32 + jr] = v;
            }
            if (slot == 0 && h == 0) p3d_lds_insertion_sort(tfA, Sf, jr);
        }
        if (__builtin_amdgcn_ballot_w64(unsorted) != 0 && slot == 0 && h == 0) p3d_lds_insertion_sort(tcA, Sc, jr);
    }
    // ---- final pass: two merged samples per step
    MarchState st;
    st.Td = 1.0; st.W = 0.0f; st.D = 0.0f; st.prev_t = 0.0f; st.prev_sigma = 0.0f;
    f32x16 C, prev_rgb;
    float Cx = 0.0f, Cy = 0.0f, Cz = 0.0f, ppx = 0.0f, ppy = 0.0f, ppz = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { C[c] = 0.0f; prev_rgb[c] = 0.0f; }
    {
        int ci = 0, fi = 0;
        float ta = tcA[jr], tb = (Sf > 0) ? tfA[jr] : __builtin_inff();
        bool prev_skipped = false;
        auto next_depth = [&]() {  // unify_samples' merge: ties take the coarse sample first (stable sort)
            const bool take_c = (ci < Sc) && (fi >= Sf || ta <= tb);
            const float t = take_c ? ta : tb;
            ci += take_c ? 1 : 0;
            fi += take_c ? 0 : 1;
            const int cq = ci < Sc ? ci : Sc - 1, fq = fi < Sf ? fi : (Sf > 0 ? Sf - 1 : 0);
            const float nv = (take_c ? tcA : tfA)[(take_c ? cq : fq) * 32 + jr];  // (selects, not a store through a selected pointer)
            ta = take_c ? nv : ta;
            tb = take_c ? tb : nv;
            tmin = __builtin_fminf(tmin, t);
            tmax = __builtin_fmaxf(tmax, t);
            return t;
        };
        // the second half of k_render's loop body, for one sample whose decode (or skip) has already happened
        auto consume = [&](int m, float t, float px, float py, float pz, float sigma, f32x16 rgb, bool skipped) {
            if (m > 0) {
                float tm;
                float w = p3d_march_weight(st, t, sigma, tm);
                if (early) {
                    if (__builtin_amdgcn_ballot_w64(prev_skipped && w != 0.0f) != 0) {
                        float s2;
                        f32x16 c2;
                        if constexpr (FAST) p3d_decode_wave_fast<true, P3D_QUAD_PAIR != 0>(lds, rs, g, cfg, ppx, ppy, ppz, s2, c2);
                        else p3d_decode_wave<true>(lds, rs, g, cfg, ppx, ppy, ppz, s2, c2);
                        if (prev_skipped) prev_rgb = c2;
                        prev_skipped = false;
                    }
                    if (__builtin_amdgcn_ballot_w64(skipped && w != 0.0f) != 0) {
                        float s2;
                        if constexpr (FAST) p3d_decode_wave_fast<true, P3D_QUAD_PAIR != 0>(lds, rs, g, cfg, px, py, pz, s2, rgb);
                        else p3d_decode_wave<true>(lds, rs, g, cfg, px, py, pz, s2, rgb);
                        skipped = false;
                    }
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) C[c] = p3d_fma(w, (prev_rgb[c] + rgb[c]) * 0.5f, C[c]);
                Cx = p3d_fma(w, (ppx + px) * 0.5f, Cx);
                Cy = p3d_fma(w, (ppy + py) * 0.5f, Cy);
                Cz = p3d_fma(w, (ppz + pz) * 0.5f, Cz);
                st.W = st.W + w;
                st.D = p3d_fma(w, tm, st.D);
            }
            st.prev_t = t; st.prev_sigma = sigma;
            prev_rgb = rgb;
            prev_skipped = skipped;
            ppx = px; ppy = py; ppz = pz;
        };
        for (int m = 0; m < S; m += 2) {
            const bool haveB = m + 1 < S;  // uniform
            const float tA = next_depth();
            const float tB = haveB ? next_depth() : tA;
            const float pxA = ox + tA * dx, pyA = oy + tA * dy, pzA = oz + tA * dz;
            const float pxB = ox + tB * dx, pyB = oy + tB * dy, pzB = oz + tB * dz;
            const float px = slot ? pxB : pxA, py = slot ? pyB : pyA, pz = slot ? pzB : pzA;
            float sigma = P3D_SIGMA_MASKED;
            f32x16 rgb;
#pragma unroll
            for (int c = 0; c < 16; ++c) rgb[c] = 0.0f;
            bool skipped = false, live = true;
            if (early) {
                // exact mode: below 1e-60 every later weight is exactly 0.  Tolerance mode: the transmittance bounds what the rest
                // of the ray can still add — stop at 2e-6, inside the 2e-5 budget (as in k_render)
                live = !(is_cropped(px, pz) || st.Td < (FAST ? 2e-6 : 1e-60));
                skipped = __builtin_amdgcn_ballot_w64(live) == 0;
            }
            if (!skipped) {
                if constexpr (FAST) p3d_decode_wave_fast<true, P3D_QUAD_PAIR != 0, false, true>(lds, rs, g, cfg, px, py, pz, sigma, rgb, live);
                else p3d_decode_wave<true, P3D_QUAD_PAIR != 0>(lds, rs, g, cfg, px, py, pz, sigma, rgb, live);
                ndec += 1;
                skipped = !live;
            }
            // exchange: every lane gets both samples of its ray (its own channel half)
            const float so = partner(sigma);
            const float sA = slot ? so : sigma, sB = slot ? sigma : so;
            const int ko = __builtin_amdgcn_ds_swizzle((int)skipped, 0x401f);
            const int kA = slot ? ko : (int)skipped, kB = slot ? (int)skipped : ko;
            f32x16 rgbA, rgbB;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float other = partner(rgb[c]);
                rgbA[c] = slot ? other : rgb[c];
                rgbB[c] = slot ? rgb[c] : other;
            }
            consume(m, tA, pxA, pyA, pzA, sA, rgbA, kA != 0);
            if (haveB) consume(m + 1, tB, pxB, pyB, pzB, sB, rgbB, kB != 0);
        }
    }
    {
        const float Wt = st.W;
        float d = st.D / Wt;
        if (d != d) d = __builtin_inff();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float v = C[c];
            if (p.white_back) v = (v + 1.0f) - Wt;
            C[c] = v * 2.0f - 1.0f;
        }
        if (p.white_back) { Cx = (Cx + 1.0f) - Wt; Cy = (Cy + 1.0f) - Wt; Cz = (Cz + 1.0f) - Wt; }
        Cx = Cx * 2.0f - 1.0f; Cy = Cy * 2.0f - 1.0f; Cz = Cz * 2.0f - 1.0f;
        if (active && slot == 0) {
            float* dst = p.out_feat + ray * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 8 * q) = (f32x4){C[4 * q], C[4 * q + 1], C[4 * q + 2], C[4 * q + 3]};
            if (h == 0) {
                p.out_depth[ray] = d;
                p.out_wsum[ray] = Wt;
                p.out_xyz[ray * 3] = Cx; p.out_xyz[ray * 3 + 1] = Cy; p.out_xyz[ray * 3 + 2] = Cz;
            }
        }
    }
    if (!active) { tmin = __builtin_inff(); tmax = -__builtin_inff(); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        tmin = __builtin_fminf(tmin, __shfl_xor(tmin, o));
        tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, o));
    }
    if (lane == 0) {
        atomicMin(p.gminmax, p3d_f2ord(tmin));
        atomicMax(p.gminmax + 1, p3d_f2ord(tmax));
        if (p.per_view_clamp) {
            atomicMin(p.gminmax + 4 + 2 * nlo, p3d_f2ord(tmin));
            atomicMax(p.gminmax + 5 + 2 * nlo, p3d_f2ord(tmax));
        }
        atomicAdd((unsigned long long*)(p.gminmax + 2), (unsigned long long)ndec);
    }
}

// =====================================================================================================================
// k_render_quad (round 4): the small-launch kernel with 8 rays x 4 sample slots per wave — lane = ray (j & 7) x slot (j >> 3) x
// channel half — so that a 128^2-ray view makes 2048 waves = TWO per SIMD (k_render_pair: 1024, one per SIMD, every step's
// bookkeeping exposed latency: ~4 us per step whether it decodes or not, profiles/history/r03_notes.txt).  Per-wave LDS rows hold 8 rays
// (32 bytes per row instead of 128: 6.6 KB per wave at 96+96).  Everything per ray is executed identically by the ray's eight
// lanes, as in k_render_pair; a step decodes four consecutive samples of every ray; sigma and the skipped flag are exchanged to
// all lanes of the ray, the 16 colour channels only towards slot 0, whose lanes write the outputs.  Bit-identical to k_render.
// =====================================================================================================================
// WO ("weights only", P3D_FLAG_WEIGHTS_ONLY; tolerance-mode instantiations only): the launch is asked for the accumulated opacity
// (wsum) and depth alone — the occlusion pass of paste_front (training/triplane.py:565-578 reads `image_weights` of a second render and
// nothing else).  A ray's weights depend on depths and densities only, so the final pass decodes DENSITIES (layer 1 + the sigma row, the
// coarse pass's cost), exchanges no colours, runs no colour guards and composites nothing but W and D: wsum and depth are bit-identical
// to the full launch's, feat / xyz are not written.
template <int NF, bool FAST, bool WO = false>
// Registers: the tolerance-mode instantiations are compiled for two waves per SIMD (<= 256 registers; 128^2 rays = 2048 waves).  The
// EXACT ones are compiled for one (512): under the 256-register cap they spilled 108-127 VGPRs (164-176 B of scratch per lane, round 4),
// and the host picks the exact quad kernel only for launches of <= 8192 rays = <= 1024 waves — one per SIMD whatever the cap.
__global__ __launch_bounds__(64 * P3D_RENDER_WAVES, FAST ? 2 : 1) void k_render_quad(RenderParams p) {
    static_assert(!WO || FAST, "the weights-only launch exists for the tolerance mode (the renderer class's default, what the paste runs)");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    p3d_load_mlp_to_lds(lds, p.w0, p.b0, p.w1, p.b1, !FAST);
    if constexpr (FAST) p3d_load_mlp_f16_to_lds(lds, p.w0, p.w1);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int jr = j & 7, slot = j >> 3;  // ray of the wave (8), sample slot (4)
    constexpr int RS = 8;                 // floats per LDS row
    const int nwaves = blockDim.x >> 6;
    long long tile = (long long)blockIdx.x * nwaves + wave;  // 8-ray tiles
    if (tile >= p.ntiles) return;  // no workgroup barrier below this line
    float* wl = lds + (FAST ? P3D_LDS_FAST_FLOATS : P3D_LDS_MLP_FLOATS) + 4 + (size_t)wave * p.lds_rows * RS;

    const int Sc = p.Sc, Sf = P3D_NF_EXACT(NF) ? NF : p.Sf, S = Sc + Sf;
    long long n = tile / p.tiles_per_img, tl = tile - n * p.tiles_per_img;
    long long r;
    if (p.tile_w > 0) {  // 4x2 pixel tile, Morton lane order (a lane quad = a 2x2 pixel block: the quad-cooperative gathers)
        long long ty = tl / p.tiles_x, tx = tl - ty * p.tiles_x;
        const int lx = (jr & 1) | ((jr >> 1) & 2), ly = (jr >> 1) & 1;
        r = (ty * 2 + ly) * p.tile_w + tx * 4 + lx;
    } else {
        r = tl * 8 + jr;
    }
    const bool active = r < p.R;
    const long long rc = active ? r : p.R - 1;
    const size_t ray = (size_t)n * p.R + rc;

    P3dPlaneGeom g;
    g.halfW = 0.5f * (float)p.W; g.halfH = 0.5f * (float)p.H; g.fW = (float)p.W; g.fH = (float)p.H; g.W = p.W;
    g.plane_bytes = (uint32_t)p.H * (uint32_t)p.W * 128u;
    unsigned nlo = __builtin_amdgcn_readfirstlane((unsigned)n);
    const float* pbase = p.planes + ((p.cfg.flags & P3D_FLAG_SHARED_PLANES) ? (size_t)0 : (size_t)nlo * 3 * (g.plane_bytes / 4));
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)pbase, 0, 3 * g.plane_bytes, 0x00020000);
    const P3dDecodeCfg cfg = p.cfg;
    const bool early = !(cfg.flags & P3D_FLAG_NO_EARLY_OUT);
    const bool f_crop = (cfg.flags & P3D_FLAG_CROP) != 0;
    int ndec = 0;
    // value of the lane whose slot differs by X (lane ^ 8X): ds_swizzle bit mode (and 0x1f, or 0, xor 8X) — the crossbar, no address VGPR
    auto swz8 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x201f)); };
    auto swz16 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401f)); };
    auto swz24 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x601f)); };
    // the value of sample slot q of this lane's ray, on EVERY lane of the ray: the lane whose slot is q sits at lane ^ 8 (slot ^ q)
    auto of_slot = [&](int qq, float own, float x8, float x16, float x24) {
        const int k = slot ^ qq;
        return (k & 1) ? ((k & 2) ? x24 : x8) : ((k & 2) ? x16 : own);
    };

    const float ox = p.rays_o[ray * 3], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
    const float dx = p.rays_d[ray * 3], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];

    // LDS rows of this wave: row(i)[jr]; both slots of a ray write the same values
    float* tcA = wl;
    float* wcA = tcA + Sc * RS;
    float* tfA = (NF > 0) ? wcA : wcA + Sc * RS;

    bool unsorted = false;
    {
        const float step = (p.ray_end - p.ray_start) / (float)(Sc - 1);
        const float* jit = p.jitter + ray * Sc;
        const bool limits = p.ray_start_arr != nullptr;  // per-ray limits: as in k_render
        const float rs = limits ? p.ray_start_arr[ray] : 0.0f, span = limits ? p.ray_end_arr[ray] - rs : 0.0f;
        const float rdelta = span / (float)(Sc - 1);
        float prev = -__builtin_inff();
        for (int i = 0; i < Sc; ++i) {
            float lin = (i < Sc / 2) ? p3d_fma(step, (float)i, p.ray_start) : p3d_fma(-step, (float)(Sc - 1 - i), p.ray_end);
            const float ji = p.rng ? p3d_draw(p.seed_lo, p.seed_hi, 0u, ray, i) : jit[i];
            float t = lin + ji * p.depth_delta;
            if (limits) {
                const float prod = ((float)i / (float)(Sc - 1)) * span;
                t = (rs + prod) + ji * rdelta;
            } else if (p.disparity) {
                const float s01 = 1.0f / (float)(Sc - 1);
                const float l01 = (i < Sc / 2) ? p3d_fma(s01, (float)i, 0.0f) : p3d_fma(-s01, (float)(Sc - 1 - i), 1.0f);
                const float dd = l01 + ji * p.depth_delta;
                const float ta_ = p.ray_start * (1.0f - dd), tb_ = p.ray_end * dd;
                t = 1.0f / (ta_ + tb_);
            }
            tcA[i * RS + jr] = t;
            unsorted |= (t < prev);
            prev = t;
        }
    }
    float tmin = __builtin_inff(), tmax = -__builtin_inff();
    auto is_cropped = [&](float px, float pz) {
        return f_crop && (__builtin_fabsf(px) > cfg.crop_limit || __builtin_fabsf(pz) > cfg.crop_limit);
    };
    if (Sf > 0) {
        // ---- coarse pass, four samples per step
        MarchState st;
        st.Td = 1.0; st.W = 0.0f; st.D = 0.0f; st.prev_t = 0.0f; st.prev_sigma = 0.0f;
        for (int i = 0; i < Sc; i += 4) {
            float tq[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) tq[qq] = tcA[(i + qq < Sc ? i + qq : Sc - 1) * RS + jr];
            const float t = (slot & 2) ? ((slot & 1) ? tq[3] : tq[2]) : ((slot & 1) ? tq[1] : tq[0]);
            const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
            float sigma = P3D_SIGMA_MASKED;
            bool skip = false, live = true;
            if (early) {
                live = !(is_cropped(px, pz) || st.Td < 1e-60);  // slots 1-3: Td before the step's first interval — conservative, still exact
                skip = __builtin_amdgcn_ballot_w64(live) == 0;
            }
            if (!skip) {
                f32x16 dummy;
                p3d_decode_wave<false, P3D_QUAD_PAIR != 0>(lds, rs, g, cfg, px, py, pz, sigma, dummy, live);
                ndec += 1;
            }
            const float s8 = swz8(sigma), s16 = swz16(sigma), s24 = swz24(sigma);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                if (i + qq < Sc) {  // uniform
                    const float sq = of_slot(qq, sigma, s8, s16, s24);
                    if (i + qq > 0) {
                        float tm;
                        wcA[(i + qq - 1) * RS + jr] = p3d_march_weight(st, tq[qq], sq, tm);
                    }
                    st.prev_t = tq[qq]; st.prev_sigma = sq;
                }
            }
        }
        const int Ns = Sc - 3;
        {
            double sum = 0.0;
            float wa = wcA[0 * RS + jr], wb = wcA[1 * RS + jr];
            for (int jj = 0; jj < Ns; ++jj) {
                float wc = wcA[(jj + 2) * RS + jr];
                float m1 = __builtin_fmaxf(wa, wb), m2 = __builtin_fmaxf(wb, wc);
                float v = ((m1 + m2) * 0.5f + 0.01f) + 1e-5f;
                sum += (double)v;
                wcA[(jj + 1) * RS + jr] = v;
                wa = wb; wb = wc;
            }
            float fsum = (float)sum;
            double acc = 0.0;
            wcA[jr] = 0.0f;
            for (int jj = 0; jj < Ns; ++jj) {
                float pdf = wcA[(jj + 1) * RS + jr] / fsum;
                acc += (double)pdf;
                wcA[(jj + 1) * RS + jr] = (float)acc;
            }
        }
        const float* uu = p.u + ray * Sf;
        if constexpr (NF > 0) {
            // as in k_render: every load of the row issued before the first search, eight draws in lock-step (the eight LDS
            // reads of a search step in flight together; one draw at a time was a chain of 8 dependent LDS round trips x Sf)
            constexpr int DB = 8;
            float tf[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) tf[i] = (i < Sf) ? (p.rng ? p3d_draw(p.seed_lo, p.seed_hi, 1u, ray, i) : uu[i]) : 0.0f;
#pragma unroll
            for (int i0 = 0; i0 < NF; i0 += DB) {
                if (i0 < Sf) {  // wave-uniform
                    float ub[DB], vb[DB];
                    int kb[DB];
#pragma unroll
                    for (int q = 0; q < DB; ++q) ub[q] = tf[i0 + q];
                    p3d_inverse_cdf_batch<DB, RS>(wcA, tcA, Ns, jr, ub, vb, kb);
#pragma unroll
                    for (int q = 0; q < DB; ++q) tf[i0 + q] = (i0 + q < Sf) ? vb[q] : __builtin_inff();
                } else {
#pragma unroll
                    for (int q = 0; q < DB; ++q) tf[i0 + q] = __builtin_inff();
                }
            }
            p3d_sort_network<NF>(tf);
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if (i < Sf) tfA[i * RS + jr] = tf[i];
        } else {
            // the generic path sorts in LDS: only slot 0 may move the keys (both slots of a ray share the column)
            float* tmpA = tfA;
            for (int i = 0; i < Sf; ++i) {
                int k;
                float v = p3d_inverse_cdf<RS>(wcA, tcA, Ns, jr, p.rng ? p3d_draw(p.seed_lo, p.seed_hi, 1u, ray, i) : uu[i], k);
                tmpA[i * RS + jr] = v;
            }
            if (slot == 0 && h == 0) p3d_lds_insertion_sort<RS>(tfA, Sf, jr);
        }
        if (__builtin_amdgcn_ballot_w64(unsorted) != 0 && slot == 0 && h == 0) p3d_lds_insertion_sort<RS>(tcA, Sc, jr);
    }
    // ---- final pass: four merged samples per step
    MarchState st;
    st.Td = 1.0; st.W = 0.0f; st.D = 0.0f; st.prev_t = 0.0f; st.prev_sigma = 0.0f;
    f32x16 C, prev_rgb;
    float Cx = 0.0f, Cy = 0.0f, Cz = 0.0f, ppx = 0.0f, ppy = 0.0f, ppz = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { C[c] = 0.0f; prev_rgb[c] = 0.0f; }
    {
        int ci = 0, fi = 0;
        float ta = tcA[jr], tb = (Sf > 0) ? tfA[jr] : __builtin_inff();
        bool prev_skipped = false;
        auto next_depth = [&]() {  // unify_samples' merge: ties take the coarse sample first (stable sort)
            const bool take_c = (ci < Sc) && (fi >= Sf || ta <= tb);
            const float t = take_c ? ta : tb;
            ci += take_c ? 1 : 0;
            fi += take_c ? 0 : 1;
            const int cq = ci < Sc ? ci : Sc - 1, fq = fi < Sf ? fi : (Sf > 0 ? Sf - 1 : 0);
            const float nv = (take_c ? tcA : tfA)[(take_c ? cq : fq) * RS + jr];  // (selects, not a store through a selected pointer)
            ta = take_c ? nv : ta;
            tb = take_c ? tb : nv;
            tmin = __builtin_fminf(tmin, t);
            tmax = __builtin_fmaxf(tmax, t);
            return t;
        };
        // the second half of k_render's loop body, for one sample whose decode (or skip) has already happened
        auto consume = [&](int m, float t, float px, float py, float pz, float sigma, f32x16 rgb, bool skipped) {
            if (m > 0) {
                float tm;
                float w = p3d_march_weight(st, t, sigma, tm);
                if (early && !WO) {
                    if (__builtin_amdgcn_ballot_w64(prev_skipped && w != 0.0f) != 0) {
                        float s2;
                        f32x16 c2;
                        if constexpr (FAST) p3d_decode_wave_fast<true, P3D_QUAD_PAIR != 0>(lds, rs, g, cfg, ppx, ppy, ppz, s2, c2);
                        else p3d_decode_wave<true>(lds, rs, g, cfg, ppx, ppy, ppz, s2, c2);
                        if (prev_skipped) prev_rgb = c2;
                        prev_skipped = false;
                    }
                    if (__builtin_amdgcn_ballot_w64(skipped && w != 0.0f) != 0) {
                        float s2;
                        if constexpr (FAST) p3d_decode_wave_fast<true, P3D_QUAD_PAIR != 0>(lds, rs, g, cfg, px, py, pz, s2, rgb);
                        else p3d_decode_wave<true>(lds, rs, g, cfg, px, py, pz, s2, rgb);
                        skipped = false;
                    }
                }
                if constexpr (!WO) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) C[c] = p3d_fma(w, (prev_rgb[c] + rgb[c]) * 0.5f, C[c]);
                    Cx = p3d_fma(w, (ppx + px) * 0.5f, Cx);
                    Cy = p3d_fma(w, (ppy + py) * 0.5f, Cy);
                    Cz = p3d_fma(w, (ppz + pz) * 0.5f, Cz);
                }
                st.W = st.W + w;
                st.D = p3d_fma(w, tm, st.D);
            }
            st.prev_t = t; st.prev_sigma = sigma;
            prev_rgb = rgb;
            prev_skipped = skipped;
            ppx = px; ppy = py; ppz = pz;
        };
        for (int m = 0; m < S; m += 4) {
            float tq[4], pxq[4], pyq[4], pzq[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                tq[qq] = (m + qq < S) ? next_depth() : tq[qq > 0 ? qq - 1 : 0];  // (uniform condition)
                pxq[qq] = ox + tq[qq] * dx; pyq[qq] = oy + tq[qq] * dy; pzq[qq] = oz + tq[qq] * dz;
            }
            const float px = (slot & 2) ? ((slot & 1) ? pxq[3] : pxq[2]) : ((slot & 1) ? pxq[1] : pxq[0]);
            const float py = (slot & 2) ? ((slot & 1) ? pyq[3] : pyq[2]) : ((slot & 1) ? pyq[1] : pyq[0]);
            const float pz = (slot & 2) ? ((slot & 1) ? pzq[3] : pzq[2]) : ((slot & 1) ? pzq[1] : pzq[0]);
            float sigma = P3D_SIGMA_MASKED;
            f32x16 rgb;
#pragma unroll
            for (int c = 0; c < 16; ++c) rgb[c] = 0.0f;
            bool skipped = false, live = true;
            if (early) {
                live = !(is_cropped(px, pz) || st.Td < (FAST ? 2e-6 : 1e-60));
                skipped = __builtin_amdgcn_ballot_w64(live) == 0;
            }
            if (!skipped) {
                if constexpr (FAST) p3d_decode_wave_fast<!WO, P3D_QUAD_PAIR != 0, false, true>(lds, rs, g, cfg, px, py, pz, sigma, rgb, live);
                else p3d_decode_wave<true, P3D_QUAD_PAIR != 0>(lds, rs, g, cfg, px, py, pz, sigma, rgb, live);
                ndec += 1;
                skipped = !live;
            }
            // exchange.  sigma and the skipped flag: every lane of the ray gets all four (the marcher state — transmittance, weights,
            // the guards' decisions — is identical on the ray's lanes).  The colours: lane ^ 8q holds sample q only for the slot-0
            // lanes, which are the ones that write the ray's outputs; the other slots accumulate colours in another order and never
            // store them (48 crossbar moves per step instead of 48 + 192 selects).
            const float s8 = swz8(sigma), s16 = swz16(sigma), s24 = swz24(sigma);
            const float kf = skipped ? 1.0f : 0.0f;
            const float k8 = swz8(kf), k16 = swz16(kf), k24 = swz24(kf);
            f32x16 r8 = rgb, r16 = rgb, r24 = rgb;
            if constexpr (!WO) {
#pragma unroll
                for (int c = 0; c < 16; ++c) { r8[c] = swz8(rgb[c]); r16[c] = swz16(rgb[c]); r24[c] = swz24(rgb[c]); }
            }
            consume(m, tq[0], pxq[0], pyq[0], pzq[0], of_slot(0, sigma, s8, s16, s24), rgb, of_slot(0, kf, k8, k16, k24) != 0.0f);
            if (m + 1 < S) consume(m + 1, tq[1], pxq[1], pyq[1], pzq[1], of_slot(1, sigma, s8, s16, s24), r8, of_slot(1, kf, k8, k16, k24) != 0.0f);
            if (m + 2 < S) consume(m + 2, tq[2], pxq[2], pyq[2], pzq[2], of_slot(2, sigma, s8, s16, s24), r16, of_slot(2, kf, k8, k16, k24) != 0.0f);
            if (m + 3 < S) consume(m + 3, tq[3], pxq[3], pyq[3], pzq[3], of_slot(3, sigma, s8, s16, s24), r24, of_slot(3, kf, k8, k16, k24) != 0.0f);
        }
    }
    {
        const float Wt = st.W;
        float d = st.D / Wt;
        if (d != d) d = __builtin_inff();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float v = C[c];
            if (p.white_back) v = (v + 1.0f) - Wt;
            C[c] = v * 2.0f - 1.0f;
        }
        if (p.white_back) { Cx = (Cx + 1.0f) - Wt; Cy = (Cy + 1.0f) - Wt; Cz = (Cz + 1.0f) - Wt; }
        Cx = Cx * 2.0f - 1.0f; Cy = Cy * 2.0f - 1.0f; Cz = Cz * 2.0f - 1.0f;
        if (active && slot == 0) {
            if constexpr (!WO) {
                float* dst = p.out_feat + ray * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 8 * q) = (f32x4){C[4 * q], C[4 * q + 1], C[4 * q + 2], C[4 * q + 3]};
            }
            if (h == 0) {
                p.out_depth[ray] = d;
                p.out_wsum[ray] = Wt;
                if constexpr (!WO) { p.out_xyz[ray * 3] = Cx; p.out_xyz[ray * 3 + 1] = Cy; p.out_xyz[ray * 3 + 2] = Cz; }
            }
        }
    }
    if (!active) { tmin = __builtin_inff(); tmax = -__builtin_inff(); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        tmin = __builtin_fminf(tmin, __shfl_xor(tmin, o));
        tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, o));
    }
    if (lane == 0) {
        atomicMin(p.gminmax, p3d_f2ord(tmin));
        atomicMax(p.gminmax + 1, p3d_f2ord(tmax));
        if (p.per_view_clamp) {
            atomicMin(p.gminmax + 4 + 2 * nlo, p3d_f2ord(tmin));
            atomicMax(p.gminmax + 5 + 2 * nlo, p3d_f2ord(tmax));
        }
        atomicAdd((unsigned long long*)(p.gminmax + 2), (unsigned long long)ndec);
    }
}

__global__ void k_minmax_init(uint32_t* g, int nviews) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        g[0] = 0xffffffffu;
        g[1] = 0u;
        g[2] = 0u;  // [2..3]: 64-bit count of wave-level decode steps of the launch (statistics for bench.py)
        g[3] = 0u;
    }
    if (i < nviews) { g[4 + 2 * i] = 0xffffffffu; g[5 + 2 * i] = 0u; }  // per-view ranges (P3D_FLAG_PER_VIEW_CLAMP)
}

// rays_per_view > 0: ray i is clamped to the range of ITS view (g + 4 + 2 * (i / rays_per_view)); 0: one range for the call
__global__ void k_render_finish(float* depth, long long n, const uint32_t* g, float* dump_tminmax, long long rays_per_view) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && dump_tminmax) { dump_tminmax[0] = p3d_ord2f(g[0]); dump_tminmax[1] = p3d_ord2f(g[1]); }
    if (i < n) {
        const uint32_t* mm = rays_per_view > 0 ? g + 4 + 2 * (i / rays_per_view) : g;
        const float lo = p3d_ord2f(mm[0]), hi = p3d_ord2f(mm[1]);
        float d = depth[i];
        d = d < lo ? lo : d;  // torch.clamp(x, min, max) = min(max(x, min), max)
        d = d > hi ? hi : d;
        depth[i] = d;
    }
}

// =====================================================================================================================
// operator-level stand-alone stages (one thread per ray)
// =====================================================================================================================
__global__ void k_stratified(float start, float end, float delta, int S, const float* __restrict__ jitter, long long NR,
                             float* __restrict__ out) {
    long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= NR) return;
    const float step = (end - start) / (float)(S - 1);
    for (int i = 0; i < S; ++i) {
        float lin = (i < S / 2) ? p3d_fma(step, (float)i, start) : p3d_fma(-step, (float)(S - 1 - i), end);
        out[r * S + i] = lin + jitter[r * S + i] * delta;
    }
}

__global__ void k_minmax_decode(const uint32_t* g, float* out) {
    out[0] = p3d_ord2f(g[0]);
    out[1] = p3d_ord2f(g[1]);
}

__global__ void k_depth_minmax(const float* __restrict__ d, long long n, uint32_t* g) {
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        lo = __builtin_fminf(lo, d[i]);
        hi = __builtin_fmaxf(hi, d[i]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo = __builtin_fminf(lo, __shfl_xor(lo, o));
        hi = __builtin_fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(g, p3d_f2ord(lo));
        atomicMax(g + 1, p3d_f2ord(hi));
    }
}

// MipRayMarcher2.run_forward, one thread per (ray, channel-slice): thread handles channels k = c0, c0+KS, ...
// To keep the weights bit-identical across the slices every thread recomputes them (cheap) — ray_marcher.py:25-57.
__global__ void k_composite(const float* __restrict__ colors, const float* __restrict__ sigma,
                            const float* __restrict__ depths, long long NR, int S, int K, int white_back,
                            float* __restrict__ out_rgb, float* __restrict__ out_depth, float* __restrict__ out_w) {
    long long r = (long long)blockIdx.x * blockDim.y + threadIdx.y;
    if (r >= NR) return;
    const int c0 = threadIdx.x, KS = blockDim.x;
    const float* col = colors + r * S * K;
    const float* sg = sigma + r * S;
    const float* t = depths + r * S;
    MarchState st;
    st.Td = 1.0; st.W = 0.0f; st.D = 0.0f; st.prev_t = t[0]; st.prev_sigma = sg[0];
    float C[8];  // up to 8 channels per thread (K <= 8*KS)
#pragma unroll
    for (int q = 0; q < 8; ++q) C[q] = 0.0f;
    for (int i = 1; i < S; ++i) {
        float tm;
        float w = p3d_march_weight(st, t[i], sg[i], tm);
        if (out_w && c0 == 0) out_w[r * (S - 1) + i - 1] = w;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int k = c0 + q * KS;
            if (k < K) C[q] = p3d_fma(w, (col[(i - 1) * K + k] + col[i * K + k]) * 0.5f, C[q]);
        }
        st.W = st.W + w;
        st.D = p3d_fma(w, tm, st.D);
        st.prev_t = t[i]; st.prev_sigma = sg[i];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int k = c0 + q * KS;
        if (k < K) {
            float v = C[q];
            if (white_back) v = (v + 1.0f) - st.W;
            out_rgb[r * K + k] = v * 2.0f - 1.0f;
        }
    }
    if (c0 == 0) {
        float d = st.D / st.W;
        if (d != d) d = __builtin_inff();
        out_depth[r] = d;
    }
}

// sample_importance + sample_pdf, one thread per ray, scratch in registers/local memory — renderer.py:328-387
__global__ void k_importance(const float* __restrict__ depths, const float* __restrict__ weights, long long NR, int Sc,
                             int Sf, const float* __restrict__ u, float* __restrict__ out_depths,
                             int* __restrict__ out_inds) {
    long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= NR) return;
    const float* t = depths + r * Sc;
    const float* w = weights + r * (Sc - 1);
    const int Ns = Sc - 3;
    float cdf[P3D_MAX_S];
    double sum = 0.0;
    for (int jj = 0; jj < Ns; ++jj) {
        float m1 = __builtin_fmaxf(w[jj], w[jj + 1]), m2 = __builtin_fmaxf(w[jj + 1], w[jj + 2]);
        float v = ((m1 + m2) * 0.5f + 0.01f) + 1e-5f;
        sum += (double)v;
        cdf[jj + 1] = v;
    }
    float fsum = (float)sum;
    double acc = 0.0;
    cdf[0] = 0.0f;
    for (int jj = 0; jj < Ns; ++jj) {
        float pdf = cdf[jj + 1] / fsum;
        acc += (double)pdf;
        cdf[jj + 1] = (float)acc;
    }
    for (int i = 0; i < Sf; ++i) {
        float ui = u[r * Sf + i];
        int lo = 0, hi = Ns + 1;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (cdf[mid] <= ui) lo = mid + 1; else hi = mid;
        }
        int k = lo;
        int below = k - 1 > 0 ? k - 1 : 0;
        int above = k < Ns ? k : Ns;
        float den = cdf[above] - cdf[below];
        if (den < 1e-5f) den = 1.0f;
        float bb = 0.5f * (t[below] + t[below + 1]);
        float ba = 0.5f * (t[above] + t[above + 1]);
        out_depths[r * Sf + i] = bb + ((ui - cdf[below]) / den) * (ba - bb);
        if (out_inds) out_inds[r * Sf + i] = k;
    }
}

// unify_samples permutation (stable ascending), one thread per ray, rank counting — renderer.py:289-301
__global__ void k_unify_perm(const float* __restrict__ tc, const float* __restrict__ tf, long long NR, int Sc, int Sf,
                             int* __restrict__ perm) {
    long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= NR) return;
    const int S = Sc + Sf;
    const float* a = tc + r * Sc;
    const float* b = tf + r * Sf;
    for (int i = 0; i < S; ++i) {
        float ki = i < Sc ? a[i] : b[i - Sc];
        int rank = 0;
        for (int q = 0; q < S; ++q) {
            float kq = q < Sc ? a[q] : b[q - Sc];
            rank += (kq < ki || (kq == ki && q < i)) ? 1 : 0;
        }
        perm[r * S + rank] = i;
    }
}

// sigma2density + the two masks of get_eg3d_volume (_util/eg3d_metrics3d.py:65-69,153-163), one HBM pass:
//   d = 1 - exp(-softplus(sigma - 1));  cropmask -> d = -1000;  cull (sic, evaluated on the DENSITIES, renderer.py:150-153):
//   1 - exp(-softplus(d - 1)) < cull_thresh -> d = -1000.   Contract math (include/p3d_numerics.h).
P3D_DEV float p3d_density_of(float sigma, bool cropped, float cull_thresh) {
    float d = 1.0f - p3d_exp_nonpos(-p3d_softplus(sigma - 1.0f));
    if (cropped) d = -1000.0f;
    if (cull_thresh >= 0.0f) {
        const float a2 = 1.0f - p3d_exp_nonpos(-p3d_softplus(d - 1.0f));
        if (a2 < cull_thresh) d = -1000.0f;
    }
    return d;
}
// four values per thread (16-byte loads / stores); the host passes vec = 1 only when sigma / out are 16-byte and cropmask
// 4-byte aligned
__global__ void k_sigma2density(const float* __restrict__ sigma, const unsigned char* __restrict__ cropmask, long long M,
                                float cull_thresh, float* __restrict__ out, int vec) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= M) return;
    if (vec && i + 3 < M) {
        const float4 s4 = *reinterpret_cast<const float4*>(sigma + i);
        uchar4 c4 = make_uchar4(0, 0, 0, 0);
        if (cropmask) c4 = *reinterpret_cast<const uchar4*>(cropmask + i);
        float4 d4;
        d4.x = p3d_density_of(s4.x, c4.x != 0, cull_thresh);
        d4.y = p3d_density_of(s4.y, c4.y != 0, cull_thresh);
        d4.z = p3d_density_of(s4.z, c4.z != 0, cull_thresh);
        d4.w = p3d_density_of(s4.w, c4.w != 0, cull_thresh);
        *reinterpret_cast<float4*>(out + i) = d4;
    } else {
        for (long long k = i; k < i + 4 && k < M; ++k) out[k] = p3d_density_of(sigma[k], cropmask && cropmask[k], cull_thresh);
    }
}

// =====================================================================================================================
// C ABI
// =====================================================================================================================
static inline int p3d_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? P3D_OK : (int)e;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is sticky per (kernel, device): raise it only when a launch needs more than any
// earlier one did (one call per kernel instantiation in steady state instead of one per launch).
template <typename K>
static hipError_t p3d_ensure_dynamic_lds(K kernel, size_t bytes) {
    static std::atomic<size_t> granted[64];  // per instantiation, indexed by device ordinal
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::atomic<size_t>& g = granted[dev & 63];
    if (bytes <= g.load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) {
        size_t cur = g.load(std::memory_order_relaxed);
        while (cur < bytes && !g.compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
    }
    return e;
}

static P3dDecodeCfg make_cfg(const p3d_opts* o) {
    P3dDecodeCfg c;
    c.coord_scale = o->coord_scale;
    c.crop_limit = o->crop_limit;
    c.cull_thresh = o->cull_thresh;
    c.plane_mode = o->plane_mode;
    c.flags = o->flags;
    return c;
}

extern "C" {

// (p3d_build_info lives in p3d_paste.hip: the small translation unit that is recompiled when any source changes)
int p3d_abi_version(void) { return P3D_ABI_VERSION; }

int p3d_struct_layout(int which, size_t* out, int cap) {
    if (!out || cap <= 0) return P3D_E_ARG;
    int n = 0;
#define P3D_PUT(v) do { if (n >= cap) return P3D_E_RANGE; out[n++] = (size_t)(v); } while (0)
#define P3D_OFF(T, f) P3D_PUT(offsetof(T, f))
    switch (which) {
    case P3D_STRUCT_OPTS:
        P3D_PUT(sizeof(p3d_opts));
        P3D_OFF(p3d_opts, coord_scale); P3D_OFF(p3d_opts, ray_start); P3D_OFF(p3d_opts, ray_end); P3D_OFF(p3d_opts, depth_delta);
        P3D_OFF(p3d_opts, crop_limit); P3D_OFF(p3d_opts, cull_thresh); P3D_OFF(p3d_opts, Sc); P3D_OFF(p3d_opts, Sf);
        P3D_OFF(p3d_opts, plane_mode); P3D_OFF(p3d_opts, flags);
        break;
    case P3D_STRUCT_DUMPS:
        P3D_PUT(sizeof(p3d_dumps));
        P3D_OFF(p3d_dumps, depths_coarse); P3D_OFF(p3d_dumps, sigma_coarse); P3D_OFF(p3d_dumps, weights_coarse);
        P3D_OFF(p3d_dumps, depths_fine); P3D_OFF(p3d_dumps, inds); P3D_OFF(p3d_dumps, depths_sorted); P3D_OFF(p3d_dumps, sigma_sorted);
        P3D_OFF(p3d_dumps, depth_unclamped); P3D_OFF(p3d_dumps, tminmax);
        break;
    case P3D_STRUCT_PASTE_ARGS:
        P3D_PUT(sizeof(p3d_paste_args));
        P3D_OFF(p3d_paste_args, weights); P3D_OFF(p3d_paste_args, xyz); P3D_OFF(p3d_paste_args, occ); P3D_OFF(p3d_paste_args, rays_o);
        P3D_OFF(p3d_paste_args, rays_d); P3D_OFF(p3d_paste_args, front); P3D_OFF(p3d_paste_args, image);
        P3D_OFF(p3d_paste_args, out_image); P3D_OFF(p3d_paste_args, out_paste); P3D_OFF(p3d_paste_args, out_mask);
        P3D_OFF(p3d_paste_args, out_mask_weights); P3D_OFF(p3d_paste_args, out_mask_edges); P3D_OFF(p3d_paste_args, out_mask_occ);
        P3D_OFF(p3d_paste_args, out_mask_dxyz);
        P3D_OFF(p3d_paste_args, N); P3D_OFF(p3d_paste_args, r); P3D_OFF(p3d_paste_args, S); P3D_OFF(p3d_paste_args, front_shared);
        P3D_OFF(p3d_paste_args, normalize_images);
        P3D_OFF(p3d_paste_args, thresh_weight); P3D_OFF(p3d_paste_args, thresh_edges); P3D_OFF(p3d_paste_args, thresh_occ);
        P3D_OFF(p3d_paste_args, thresh_dxyz); P3D_OFF(p3d_paste_args, box_warp);
        break;
    case P3D_STRUCT_CONV_ARGS:
        P3D_PUT(sizeof(p3d_conv_args));
        P3D_OFF(p3d_conv_args, x); P3D_OFF(p3d_conv_args, w); P3D_OFF(p3d_conv_args, w_f16); P3D_OFF(p3d_conv_args, styles);
        P3D_OFF(p3d_conv_args, demod_coefs); P3D_OFF(p3d_conv_args, noise); P3D_OFF(p3d_conv_args, bias); P3D_OFF(p3d_conv_args, fir);
        P3D_OFF(p3d_conv_args, y); P3D_OFF(p3d_conv_args, workspace); P3D_OFF(p3d_conv_args, saturated); P3D_OFF(p3d_conv_args, x_img);
        P3D_OFF(p3d_conv_args, y_img); P3D_OFF(p3d_conv_args, y_img_styles);
        P3D_OFF(p3d_conv_args, rgb_w); P3D_OFF(p3d_conv_args, rgb_styles); P3D_OFF(p3d_conv_args, rgb_partial); P3D_OFF(p3d_conv_args, workspace_bytes);
        P3D_OFF(p3d_conv_args, N); P3D_OFF(p3d_conv_args, I); P3D_OFF(p3d_conv_args, H); P3D_OFF(p3d_conv_args, W); P3D_OFF(p3d_conv_args, O);
        P3D_OFF(p3d_conv_args, ks); P3D_OFF(p3d_conv_args, up); P3D_OFF(p3d_conv_args, demodulate); P3D_OFF(p3d_conv_args, noise_per_sample);
        P3D_OFF(p3d_conv_args, act); P3D_OFF(p3d_conv_args, mma);
        P3D_OFF(p3d_conv_args, alpha); P3D_OFF(p3d_conv_args, gain); P3D_OFF(p3d_conv_args, clamp); P3D_OFF(p3d_conv_args, rgb_channels); P3D_OFF(p3d_conv_args, w_f16_layout);
        break;
    default:
        return P3D_E_RANGE;
    }
#undef P3D_OFF
#undef P3D_PUT
    return n;
}

int p3d_planes_to_nhwc_f32(const float* src, int n3, int C, int H, int W, float* dst, void* stream) {
    if (!src || !dst || n3 <= 0 || H <= 0 || W <= 0) return P3D_E_ARG;
    if (C != P3D_C) return P3D_E_RANGE;
    int HW = H * W;
    dim3 grid((HW + 63) / 64, n3);
    hipLaunchKernelGGL(k_planes_to_nhwc, grid, dim3(256), 0, (hipStream_t)stream, src, dst, HW);
    return p3d_check_launch();
}

int p3d_triplane_decode_f32(const float* planes, int N, int H, int W, const float* coords, int64_t M, const float* w0,
                            const float* b0, const float* w1, const float* b1, const p3d_opts* opts, float* out_sigma,
                            float* out_rgb, void* stream) {
    if (!planes || !coords || !w0 || !b0 || !w1 || !b1 || !opts || !out_sigma || N <= 0 || M <= 0) return P3D_E_ARG;
    if (H <= 0 || W <= 0 || (long long)H * W * 128 * 3 >= 0x7ffffff0LL) return P3D_E_RANGE;
    DecodeParams p;
    p.planes = planes; p.coords = coords; p.w0 = w0; p.b0 = b0; p.w1 = w1; p.b1 = b1;
    p.out_sigma = out_sigma; p.out_rgb = out_rgb; p.M = M; p.H = H; p.W = W;
    p.tiles_per_img = (M + 31) / 32;
    p.ntiles = p.tiles_per_img * N;
    p.cfg = make_cfg(opts);
    p.grid_n = 0; p.grid_lo = 0; p.vsize = p.goff0 = p.goff1 = p.goff2 = 0.0f; p.out_cropmask = nullptr; p.mask_limit = 0.0f;
    long long blocks = (p.ntiles + P3D_WAVES_PER_WG - 1) / P3D_WAVES_PER_WG;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 workgroups per CU
    const size_t lds_bytes = (size_t)(P3D_LDS_MLP_FLOATS + 4) * 4;
    if (out_rgb)
        hipLaunchKernelGGL((k_decode_points<true, false, false>), dim3((unsigned)blocks), dim3(P3D_WG), lds_bytes, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((k_decode_points<false, false, false>), dim3((unsigned)blocks), dim3(P3D_WG), lds_bytes, (hipStream_t)stream, p);
    return p3d_check_launch();
}

int p3d_grid_density_f32(const float* planes, int H, int W, int grid_n, int64_t lo, int64_t hi, float voxel_size, float off0,
                         float off1, float off2, const float* w0, const float* b0, const float* w1, const float* b1,
                         const p3d_opts* opts, float* out_sigma, unsigned char* out_cropmask, float mask_limit, void* stream) {
    if (!planes || !w0 || !b0 || !w1 || !b1 || !opts || !out_sigma || grid_n <= 1 || lo < 0 || hi <= lo) return P3D_E_ARG;
    if (H <= 0 || W <= 0 || (long long)H * W * 128 * 3 >= 0x7ffffff0LL || hi > (int64_t)grid_n * grid_n * grid_n || grid_n > 1290)
        return P3D_E_RANGE;  // grid_n^3 < 2^31: the kernel's index arithmetic is 32-bit
    DecodeParams p;
    p.planes = planes; p.coords = nullptr; p.w0 = w0; p.b0 = b0; p.w1 = w1; p.b1 = b1;
    p.out_sigma = out_sigma; p.out_rgb = nullptr; p.out_cropmask = out_cropmask; p.mask_limit = mask_limit; p.M = hi - lo; p.H = H; p.W = W;
    p.tiles_per_img = (p.M + 31) / 32;
    p.ntiles = p.tiles_per_img;
    p.cfg = make_cfg(opts);
    p.grid_n = grid_n; p.grid_lo = lo; p.vsize = voxel_size; p.goff0 = off0; p.goff1 = off1; p.goff2 = off2;
    long long blocks = (p.ntiles + P3D_WAVES_PER_WG - 1) / P3D_WAVES_PER_WG;
    // grid-stride beyond 16 workgroups per CU; an ODD workgroup count, so that the stride (4 tiles per workgroup) is not a
    // multiple of the tiles per grid row: with P3D_FLAG_SKIP_CROPPED the masked ends of every row would otherwise always fall
    // on the same waves (measured: 10.7 ms instead of the expected ~7.5 at 512^3 with half of the grid masked)
    if (blocks > 256 * 16 - 1) blocks = 256 * 16 - 1;
    // four variants: exact / tolerance-mode decoder (P3D_FLAG_FAST_COLOR), texel boxes staged through LDS or direct gathers
    // (measurements: DESIGN.md §4.3)
    const bool fastd = (opts->flags & P3D_FLAG_FAST_COLOR) != 0;
    // LDS-staged texel boxes only on request: with quad-cooperative gathers the direct tolerance query is faster (512^3: 7.7 ms
    // direct, 10.0 staged; exact: 12.7 direct, 22.8 staged); P3D_FLAG_NO_STAGING is accepted and has nothing left to switch off
    const bool staged = (opts->flags & P3D_FLAG_FORCE_STAGING) != 0;
    const size_t lds_bytes = (size_t)((fastd ? P3D_LDS_FAST_FLOATS - P3D_LDS_B0P : P3D_LDS_MLP_FLOATS) + 4 +
                                      (staged ? P3D_WAVES_PER_WG * P3D_BOX_FLOATS_PER_WAVE : 0)) * 4;
    hipError_t e = hipSuccess;
#define P3D_LAUNCH_GRID(SV, FV)                                                                                       \
    do {                                                                                                              \
        e = p3d_ensure_dynamic_lds(k_decode_points<false, SV, FV>, lds_bytes);                                        \
        if (e == hipSuccess)                                                                                          \
            hipLaunchKernelGGL((k_decode_points<false, SV, FV>), dim3((unsigned)blocks), dim3(P3D_WG), lds_bytes, (hipStream_t)stream, p); \
    } while (0)
    if (staged) { if (fastd) P3D_LAUNCH_GRID(true, true); else P3D_LAUNCH_GRID(true, false); }
    else { if (fastd) P3D_LAUNCH_GRID(false, true); else P3D_LAUNCH_GRID(false, false); }
    if (e != hipSuccess) return (int)e;
    return p3d_check_launch();
}

int p3d_decode_features_f32(const float* feats, int N, int64_t M, const float* w0, const float* b0, const float* w1, const float* b1,
                            int force_sigmoid, float* out_sigma, float* out_rgb, void* stream) {
    if (!feats || !w0 || !b0 || !w1 || !b1 || !out_sigma || !out_rgb || N <= 0 || M <= 0) return P3D_E_ARG;
    if ((((uintptr_t)feats | (uintptr_t)out_rgb) & 15) != 0) return P3D_E_RANGE;  // 16-byte rows of 32 floats
    DecodeFeatParams p;
    p.feats = feats; p.w0 = w0; p.b0 = b0; p.w1 = w1; p.b1 = b1; p.out_sigma = out_sigma; p.out_rgb = out_rgb;
    p.M = M; p.tiles_per_img = (M + 31) / 32; p.ntiles = p.tiles_per_img * N;
    memset(&p.cfg, 0, sizeof(p.cfg));
    p.cfg.flags = force_sigmoid ? P3D_FLAG_FORCE_SIGMOID : 0;
    const size_t lds_bytes = (size_t)(P3D_LDS_MLP_FLOATS + 4) * 4;
    long long blocks = (p.ntiles + P3D_WAVES_PER_WG - 1) / P3D_WAVES_PER_WG;
    if (blocks > 256 * 8) blocks = 256 * 8;  // grid-stride: the weight image is loaded once per workgroup
    hipError_t e = p3d_ensure_dynamic_lds(k_decode_features, lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_decode_features, dim3((unsigned)blocks), dim3(P3D_WG), lds_bytes, (hipStream_t)stream, p);
    return p3d_check_launch();
}

size_t p3d_render_workspace_bytes(int N, int64_t R, int Sc, int Sf) {
    (void)R; (void)Sc; (void)Sf;
    return 256 + (size_t)(N > 0 ? N : 0) * 8;  // global min / max + decode-step count, then one min / max pair per view
}

int p3d_render_f32(const float* planes, int N, int H, int W, const float* rays_o, const float* rays_d, int64_t R,
                   int ray_tile_w, const float* jitter, const float* u, const float* w0, const float* b0,
                   const float* w1, const float* b1, const p3d_opts* opts, float* out_feat, float* out_depth,
                   float* out_wsum, float* out_xyz, void* workspace, size_t workspace_bytes, const p3d_dumps* dumps,
                   void* stream) {
    return p3d_render_limits_f32(planes, N, H, W, rays_o, rays_d, R, ray_tile_w, jitter, u, w0, b0, w1, b1, nullptr, nullptr, opts,
                                 out_feat, out_depth, out_wsum, out_xyz, workspace, workspace_bytes, dumps, stream);
}

static int render_impl(const float* planes, int N, int H, int W, const float* rays_o, const float* rays_d, int64_t R,
                       int ray_tile_w, const float* jitter, const float* u, int rng, uint64_t seed, const float* w0, const float* b0,
                       const float* w1, const float* b1, const float* ray_start, const float* ray_end, const p3d_opts* opts,
                       float* out_feat, float* out_depth, float* out_wsum, float* out_xyz, void* workspace,
                       size_t workspace_bytes, const p3d_dumps* dumps, void* stream);

int p3d_render_limits_f32(const float* planes, int N, int H, int W, const float* rays_o, const float* rays_d, int64_t R,
                          int ray_tile_w, const float* jitter, const float* u, const float* w0, const float* b0,
                          const float* w1, const float* b1, const float* ray_start, const float* ray_end, const p3d_opts* opts,
                          float* out_feat, float* out_depth, float* out_wsum, float* out_xyz, void* workspace,
                          size_t workspace_bytes, const p3d_dumps* dumps, void* stream) {
    return render_impl(planes, N, H, W, rays_o, rays_d, R, ray_tile_w, jitter, u, 0, 0, w0, b0, w1, b1, ray_start, ray_end, opts, out_feat,
                       out_depth, out_wsum, out_xyz, workspace, workspace_bytes, dumps, stream);
}

int p3d_render_rng_f32(const float* planes, int N, int H, int W, const float* rays_o, const float* rays_d, int64_t R, int ray_tile_w,
                       uint64_t seed, const float* w0, const float* b0, const float* w1, const float* b1, const float* ray_start,
                       const float* ray_end, const p3d_opts* opts, float* out_feat, float* out_depth, float* out_wsum, float* out_xyz,
                       void* workspace, size_t workspace_bytes, const p3d_dumps* dumps, void* stream) {
    return render_impl(planes, N, H, W, rays_o, rays_d, R, ray_tile_w, nullptr, nullptr, 1, seed, w0, b0, w1, b1, ray_start, ray_end, opts,
                       out_feat, out_depth, out_wsum, out_xyz, workspace, workspace_bytes, dumps, stream);
}

static int render_impl(const float* planes, int N, int H, int W, const float* rays_o, const float* rays_d, int64_t R,
                       int ray_tile_w, const float* jitter, const float* u, int rng, uint64_t seed, const float* w0, const float* b0,
                       const float* w1, const float* b1, const float* ray_start, const float* ray_end, const p3d_opts* opts,
                       float* out_feat, float* out_depth, float* out_wsum, float* out_xyz, void* workspace,
                       size_t workspace_bytes, const p3d_dumps* dumps, void* stream) {
    if ((ray_start == nullptr) != (ray_end == nullptr)) return P3D_E_ARG;  // both or neither
    if (!planes || !rays_o || !rays_d || (!jitter && !rng) || !w0 || !b0 || !w1 || !b1 || !opts || !out_feat || !out_depth ||
        !out_wsum || !out_xyz || !workspace || N <= 0 || R <= 0)
        return P3D_E_ARG;
    const int Sc = opts->Sc, Sf = opts->Sf;
    if (Sc < 4 || Sc > P3D_MAX_S || Sf < 0 || Sf > P3D_MAX_S) return P3D_E_RANGE;
    if (Sf > 0 && !u && !rng) return P3D_E_ARG;
    if (H <= 0 || W <= 0 || (long long)H * W * 128 * 3 >= 0x7ffffff0LL) return P3D_E_RANGE;
    if (workspace_bytes < p3d_render_workspace_bytes(N, R, Sc, Sf)) return P3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    RenderParams p;
    p.blocked = 0;
    p.planes = planes; p.rays_o = rays_o; p.rays_d = rays_d; p.jitter = jitter; p.u = u;
    p.rng = rng; p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
    p.w0 = w0; p.b0 = b0; p.w1 = w1; p.b1 = b1;
    p.out_feat = out_feat; p.out_depth = out_depth; p.out_wsum = out_wsum; p.out_xyz = out_xyz;
    p.gminmax = (uint32_t*)workspace;
    p.per_view_clamp = (opts->flags & P3D_FLAG_PER_VIEW_CLAMP) ? 1 : 0;
    if (dumps) p.dumps = *dumps; else memset(&p.dumps, 0, sizeof(p.dumps));
    p.R = R; p.H = H; p.W = W; p.Sc = Sc; p.Sf = Sf;
    p.ray_start = opts->ray_start; p.ray_end = opts->ray_end; p.depth_delta = opts->depth_delta;
    p.ray_start_arr = ray_start; p.ray_end_arr = ray_end;
    p.disparity = (opts->flags & P3D_FLAG_DISPARITY) ? 1 : 0;
    if (p.disparity && ray_start) return P3D_E_RANGE;  // not with per-ray limits
    p.white_back = (opts->flags & P3D_FLAG_WHITE_BACK) ? 1 : 0;
    p.cfg = make_cfg(opts);
    if (ray_tile_w > 0 && R % ray_tile_w == 0 && ray_tile_w % 8 == 0 && (R / ray_tile_w) % 4 == 0) {
        p.tile_w = ray_tile_w;
        p.tiles_x = ray_tile_w / 8;
        p.tiles_per_img = (long long)p.tiles_x * (R / ray_tile_w / 4);
    } else {
        p.tile_w = 0;
        p.tiles_x = 0;
        p.tiles_per_img = (R + 31) / 32;
    }
    p.ntiles = p.tiles_per_img * N;
    // per-wave LDS rows: tc (Sc) + wc/cdf/sorted-fine (max(Sc,Sf)) [+ tf (Sf) on the generic path]
    // register-resident fine depths: 48 / 96 exactly (the trainer's and the eval-faithful rates), any other Sf <= 64 padded to 64;
    // the rest sorts in LDS
    const int nf = (Sf == 48) ? 48 : (Sf == 96 && Sc <= 96) ? 96 : (Sf <= 64 ? 64 : 0);
    const bool dmp = dumps != nullptr;
    const bool pair = !dmp && !(opts->flags & P3D_FLAG_NO_PAIR) && p.ntiles <= 512;  // (the small-launch kernel, below)
    // the production 96-key kernel keeps no coarse-depth rows (P3D_TCG) and runs two waves per SIMD like the others
    // (for the plain stratified spacing: per-ray limits and disparity spacing keep the LDS-resident 96-key kernel)
    const bool tcg = nf == 96 && !dmp && !pair && !(opts->flags & P3D_FLAG_NO_EARLY_OUT) && !ray_start && !p.disparity;
    const int occ = (nf >= 96 && !tcg) ? 1 : P3D_RENDER_OCC;  // waves per SIMD the instantiation is compiled for (P3D_NF_OCC)
    // + two bit rows over the merged list (is-coarse / known-masked) + the known-masked bits of the coarse samples
    p.lds_rows = Sc + (Sc > Sf ? Sc : Sf) + (nf == 0 ? Sf : 0) + 2 * ((Sc + Sf + 31) >> 5) + ((Sc + 31) >> 5);
    if (tcg) p.lds_rows = (Sc > Sf ? Sc : Sf) + 2 * ((Sc + Sf + 31) >> 5);
    int nwaves = P3D_RENDER_WAVES;
    // small ray counts (e.g. the pipeline's 128^2 rays = 512 tiles): shrink the workgroup so that every CU gets work
    while (nwaves > 1 && p.ntiles / nwaves < 2 * 256) nwaves >>= 1;
    size_t lds_bytes;
    const bool fast = (opts->flags & P3D_FLAG_FAST_COLOR) != 0 && Sf > 0;
    const size_t lds_fixed = (size_t)((fast ? P3D_LDS_FAST_FLOATS : P3D_LDS_MLP_FLOATS) + 4) * 4, lds_wave = (size_t)p.lds_rows * 128;
    if (nwaves == P3D_RENDER_WAVES) {
        // large launch: the workgroup shape (4, 2 or 1 waves, as many workgroups as fit) that puts most waves on a CU, at
        // most 8 (two per SIMD: the register cap); ties go to the LARGER workgroup.  48+48: 2 x 4 waves; 64+64: 3 x 2 instead of
        // 1 x 4 (measured 6.85 -> 6.53 ms at 512^2); 96+96: 2 x 4 with the production kernel (no coarse-depth rows: P3D_TCG), 1 x 4
        // with the LDS-resident instantiations (measured there: 2 x 2 waves 13.3 ms, 1 x 5 12.6, 1 x 4 11.2)
        int best = 0, best_waves = 0;
        for (int w = 4; w >= 1; w >>= 1) {
            const size_t per_wg = lds_fixed + w * lds_wave;
            if (per_wg > 160 * 1024) continue;
            int wgs = (int)((160 * 1024) / per_wg);
            int waves = wgs * w > 4 * occ ? 4 * occ / w * w : wgs * w;
            if (waves > best_waves) { best_waves = waves; best = w; }
        }
        if (best == 0) return P3D_E_RANGE;
        nwaves = best;
    }
    for (;; nwaves >>= 1) {
        lds_bytes = lds_fixed + (size_t)nwaves * lds_wave;
        if (lds_bytes <= 160 * 1024) break;
        if (nwaves == 1) return P3D_E_RANGE;
    }
    hipLaunchKernelGGL(k_minmax_init, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, p.gminmax, p.per_view_clamp ? N : 0);
    // small launches: 16 rays x 2 samples per wave (k_render_pair) while its waves still fit in ONE round on the 1024 SIMDs
    // (measured at 48+48: 128^2 rays 0.74 -> 0.49 ms, but 192^2 = 1152 tiles 0.99 -> 1.28 ms: its steps are ~30 % dearer)
    // ... and of those, 8 rays x 4 samples per wave (k_render_quad) where it measured faster (profiles/r04_notes.txt): launches of at
    // most 8192 rays — fewer 16-ray waves than SIMDs: 64^2 x (96+96) 0.80 -> 0.60 ms exact, 0.60 -> 0.45 tolerance — and the
    // tolerance mode at 96+96 (128^2: 0.63 -> 0.57 ms).  NOT the 128^2 exact launches (0.80 -> 0.88 at 96+96, 0.44 -> 0.49 at 48+48):
    // a decode step is ~4k MFMA clocks + ~1.5k VALU instructions of ISSUE, which ONE wave per SIMD already saturates; a second wave
    // per SIMD has nothing to hide and the per-ray work (draws, sort, marcher) is then done by twice as many waves.
    // P3D_FLAG_QUAD8 / P3D_FLAG_PAIR16 force one of the two (tests, A/B timing).
    // Round 5 (profiles/r05_notes.txt): the tolerance-mode quad kernels spill 30 VGPRs instead of 109 (the fold no longer copies its
    // partial sums) and win at every sample count of a 128^2 view (48+48: 0.42 -> 0.325 ms vs 0.354 for the pair kernel; 96+96: 0.559 vs
    // 0.618): the tolerance mode takes the quad kernel for every small launch.  The exact quad kernels are compiled for one wave per
    // SIMD (no spills: 64^2 x (96+96) 0.60 -> 0.50 ms) and stay the choice for <= 8192 rays only.
    const bool quad = pair && !(opts->flags & P3D_FLAG_PAIR16) &&
                      ((opts->flags & P3D_FLAG_QUAD8) || (long long)N * R <= 8192 || (fast && (nf == 96 || nf == 48)));
    if (quad) {
        if (p.tile_w > 0) { p.tiles_x = ray_tile_w / 4; p.tiles_per_img = (long long)p.tiles_x * (R / ray_tile_w / 2); }
        else p.tiles_per_img = (R + 7) / 8;
        p.ntiles = p.tiles_per_img * N;
        nwaves = P3D_RENDER_WAVES;
        for (;; nwaves >>= 1) {
            lds_bytes = lds_fixed + (size_t)nwaves * p.lds_rows * 32;  // rows of 8 rays
            if (2 * lds_bytes <= 160 * 1024) break;                    // two workgroups per CU
            if (nwaves == 1) { if (lds_bytes <= 160 * 1024) break; return P3D_E_RANGE; }
        }
        dim3 grid4((unsigned)((p.ntiles + nwaves - 1) / nwaves)), blk4(64 * nwaves);
        hipError_t e4 = hipSuccess;
#define P3D_LAUNCH4F(NFV, FV)                                                                                        \
    do {                                                                                                             \
        e4 = p3d_ensure_dynamic_lds(k_render_quad<NFV, FV>, lds_bytes);                                              \
        if (e4 == hipSuccess) hipLaunchKernelGGL((k_render_quad<NFV, FV>), grid4, blk4, lds_bytes, st, p);          \
    } while (0)
#define P3D_LAUNCH4(NFV) do { if (fast) P3D_LAUNCH4F(NFV, true); else P3D_LAUNCH4F(NFV, false); } while (0)
#define P3D_LAUNCH4WO(NFV)                                                                                           \
    do {                                                                                                             \
        e4 = p3d_ensure_dynamic_lds(k_render_quad<NFV, true, true>, lds_bytes);                                      \
        if (e4 == hipSuccess) hipLaunchKernelGGL((k_render_quad<NFV, true, true>), grid4, blk4, lds_bytes, st, p);  \
    } while (0)
        // P3D_FLAG_WEIGHTS_ONLY: honoured by the tolerance-mode quad kernels at 48 / 96 fine samples (what paste_front's occlusion pass runs)
        const bool wo = fast && (opts->flags & P3D_FLAG_WEIGHTS_ONLY) != 0;
#ifdef P3D_ONLY_NF
        P3D_LAUNCH4(P3D_ONLY_NF);
#else
        if (wo && nf == 48) P3D_LAUNCH4WO(48);
        else if (wo && nf == 96) P3D_LAUNCH4WO(96);
        else if (nf == 48) P3D_LAUNCH4(48);
        else if (nf == 64) P3D_LAUNCH4(64);
        else if (nf == 96) P3D_LAUNCH4(96);
        else P3D_LAUNCH4(0);
#endif
        if (e4 != hipSuccess) return (int)e4;
        int rc4 = p3d_check_launch();
        if (rc4) return rc4;
        long long NR4 = (long long)N * R;
        hipLaunchKernelGGL(k_render_finish, dim3((unsigned)((NR4 + 255) / 256)), dim3(256), 0, st, out_depth, NR4, p.gminmax,
                           (float*)nullptr, p.per_view_clamp ? (long long)R : 0LL);
        return p3d_check_launch();
    }
    if (pair) {
        if (p.tile_w > 0) { p.tiles_x = ray_tile_w / 4; p.tiles_per_img = (long long)p.tiles_x * (R / ray_tile_w / 4); }
        else p.tiles_per_img = (R + 15) / 16;
        p.ntiles = p.tiles_per_img * N;
        nwaves = P3D_RENDER_WAVES;
        for (;; nwaves >>= 1) {
            lds_bytes = lds_fixed + (size_t)nwaves * p.lds_rows * 128;
            if (lds_bytes <= 160 * 1024) break;
            if (nwaves == 1) return P3D_E_RANGE;
        }
        dim3 grid2((unsigned)((p.ntiles + nwaves - 1) / nwaves)), blk2(64 * nwaves);
        hipError_t e2 = hipSuccess;
#define P3D_LAUNCH2F(NFV, FV)                                                                                        \
    do {                                                                                                             \
        e2 = p3d_ensure_dynamic_lds(k_render_pair<NFV, FV>, lds_bytes);                                              \
        if (e2 == hipSuccess) hipLaunchKernelGGL((k_render_pair<NFV, FV>), grid2, blk2, lds_bytes, st, p);          \
    } while (0)
#define P3D_LAUNCH2(NFV) do { if (fast) P3D_LAUNCH2F(NFV, true); else P3D_LAUNCH2F(NFV, false); } while (0)
#ifdef P3D_ONLY_NF
        P3D_LAUNCH2(P3D_ONLY_NF);
#else
        if (nf == 48) P3D_LAUNCH2(48);
        else if (nf == 64) P3D_LAUNCH2(64);
        else if (nf == 96) P3D_LAUNCH2(96);
        else P3D_LAUNCH2(0);
#endif
        if (e2 != hipSuccess) return (int)e2;
        int rc2 = p3d_check_launch();
        if (rc2) return rc2;
        long long NR2 = (long long)N * R;
        hipLaunchKernelGGL(k_render_finish, dim3((unsigned)((NR2 + 255) / 256)), dim3(256), 0, st, out_depth, NR2, p.gminmax,
                           (float*)nullptr, p.per_view_clamp ? (long long)R : 0LL);
        return p3d_check_launch();
    }
    long long blocks = (p.ntiles + nwaves - 1) / nwaves;
    p.swz = 16;  // measured: 8..64 within 0.5 %, 1..4 and >= 256 about 1-3 % slower
    p.blocked = 0;
    {   // blocked tile order (k_render): whole 16 x 16-tile super-tiles per XCD run when the tile grid divides into them
        static const int tile_order = getenv("P3D_TILE_ORDER") ? atoi(getenv("P3D_TILE_ORDER")) : 1;  // 0: row-major (A/B runs)
        const long long tiles_y = p.tile_w > 0 ? p.tiles_per_img / p.tiles_x : 0;
        // ... and into a multiple of 8 of them: 384^2 (18 super-tiles on 8 XCDs) loses 2-3 % to the imbalance, every shape with whole
        // super-tiles per XCD is equal or up to 3 % better (profiles/r05_tile_order_shapes.json)
        if (tile_order == 1 && p.tile_w > 0 && p.tiles_x % 16 == 0 && tiles_y % 16 == 0 && 256 % nwaves == 0 && (p.ntiles / 256) % 8 == 0) {
            p.blocked = 1;
            p.swz = 256 / nwaves;
        }
    }
    dim3 grid((unsigned)blocks), blk(64 * nwaves);
    hipError_t e = hipSuccess;
#define P3D_LAUNCH(NFV, DV, FV, EV)                                                                                  \
    do {                                                                                                             \
        e = p3d_ensure_dynamic_lds(k_render<NFV, DV, FV, EV>, lds_bytes);                                            \
        if (e == hipSuccess) hipLaunchKernelGGL((k_render<NFV, DV, FV, EV>), grid, blk, lds_bytes, st, p);          \
    } while (0)
#define P3D_LAUNCH_NOTCG(NFV, FV)                                                                                   \
    do {                                                                                                             \
        e = p3d_ensure_dynamic_lds(k_render<NFV, false, FV, true, false>, lds_bytes);                                \
        if (e == hipSuccess) hipLaunchKernelGGL((k_render<NFV, false, FV, true, false>), grid, blk, lds_bytes, st, p); \
    } while (0)
#define P3D_LAUNCH_F(NFV, FV)                                                                                        \
    do {                                                                                                             \
        if (dmp) P3D_LAUNCH(NFV, true, FV, false);                                                                   \
        else if (opts->flags & P3D_FLAG_NO_EARLY_OUT) P3D_LAUNCH(NFV, false, FV, false);                             \
        else if (tcg) P3D_LAUNCH(NFV, false, FV, true);  /* (NFV == 96 only: the default TCG of that instantiation) */ \
        else P3D_LAUNCH_NOTCG(NFV, FV);  /* per-ray limits / disparity spacing at 96+96: the LDS-resident coarse column */ \
    } while (0)
#define P3D_LAUNCH_P(NFV) do { if (fast) P3D_LAUNCH_F(NFV, true); else P3D_LAUNCH_F(NFV, false); } while (0)
#ifdef P3D_ONLY_NF  // development builds (tools/resource_usage.py -D P3D_ONLY_NF=48): compile ONE fine-depth capacity
    P3D_LAUNCH_P(P3D_ONLY_NF);
#else
    if (nf == 48) P3D_LAUNCH_P(48);
    else if (nf == 64) P3D_LAUNCH_P(64);
    else if (nf == 96) P3D_LAUNCH_P(96);
    else P3D_LAUNCH_P(0);
#endif
    if (e != hipSuccess) return (int)e;
    int rc = p3d_check_launch();
    if (rc) return rc;
    long long NR = (long long)N * R;
    hipLaunchKernelGGL(k_render_finish, dim3((unsigned)((NR + 255) / 256)), dim3(256), 0, st, out_depth, NR, p.gminmax,
                       dumps ? dumps->tminmax : nullptr, p.per_view_clamp ? (long long)R : 0LL);
    return p3d_check_launch();
}

int p3d_sample_stratified_f32(float ray_start, float ray_end, float depth_delta, int S, const float* jitter, int64_t NR,
                              float* out, void* stream) {
    if (!jitter || !out || NR <= 0) return P3D_E_ARG;
    if (S < 2 || S > P3D_MAX_S) return P3D_E_RANGE;
    hipLaunchKernelGGL(k_stratified, dim3((unsigned)((NR + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ray_start,
                       ray_end, depth_delta, S, jitter, (long long)NR, out);
    return p3d_check_launch();
}

size_t p3d_composite_workspace_bytes(int64_t NR, int S, int K) {
    (void)NR; (void)S; (void)K;
    return 16;  // order-mapped global depth min / max (+ padding)
}

int p3d_depth_minmax_f32(const float* depths, int64_t n, float* out_minmax, void* workspace, size_t workspace_bytes, void* stream) {
    if (!depths || !out_minmax || !workspace || n <= 0) return P3D_E_ARG;
    if (workspace_bytes < 16) return P3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    uint32_t* g = (uint32_t*)workspace;
    hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(64), 0, st, g, 0);
    hipLaunchKernelGGL(k_depth_minmax, dim3(1024), dim3(256), 0, st, depths, (long long)n, g);
    hipLaunchKernelGGL(k_minmax_decode, dim3(1), dim3(1), 0, st, (const uint32_t*)g, out_minmax);
    return p3d_check_launch();
}

int p3d_composite_f32(const float* colors, const float* sigma, const float* depths, int64_t NR, int S, int K,
                      int white_back, float* out_rgb, float* out_depth, float* out_weights, void* workspace,
                      void* stream) {
    if (!colors || !sigma || !depths || !out_rgb || !out_depth || !workspace || NR <= 0) return P3D_E_ARG;
    if (S < 2 || K < 1 || K > 64) return P3D_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    uint32_t* g = (uint32_t*)workspace;
    hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(64), 0, st, g, 0);
    hipLaunchKernelGGL(k_depth_minmax, dim3(1024), dim3(256), 0, st, depths, (long long)NR * S, g);
    dim3 blk(8, 32);
    hipLaunchKernelGGL(k_composite, dim3((unsigned)((NR + 31) / 32)), blk, 0, st, colors, sigma, depths, (long long)NR, S,
                       K, white_back, out_rgb, out_depth, out_weights);
    hipLaunchKernelGGL(k_render_finish, dim3((unsigned)((NR + 255) / 256)), dim3(256), 0, st, out_depth, (long long)NR, g,
                       (float*)nullptr, 0LL);
    return p3d_check_launch();
}

int p3d_importance_f32(const float* depths, const float* weights, int64_t NR, int Sc, int Sf, const float* u,
                       float* out_depths, int32_t* out_inds, void* stream) {
    if (!depths || !weights || !u || !out_depths || NR <= 0) return P3D_E_ARG;
    if (Sc < 4 || Sc > P3D_MAX_S || Sf < 1) return P3D_E_RANGE;
    hipLaunchKernelGGL(k_importance, dim3((unsigned)((NR + 63) / 64)), dim3(64), 0, (hipStream_t)stream, depths, weights,
                       (long long)NR, Sc, Sf, u, out_depths, out_inds);
    return p3d_check_launch();
}

int p3d_unify_perm_f32(const float* tc, const float* tf, int64_t NR, int Sc, int Sf, int32_t* perm, void* stream) {
    if (!tc || !tf || !perm || NR <= 0) return P3D_E_ARG;
    if (Sc < 1 || Sf < 0) return P3D_E_RANGE;
    hipLaunchKernelGGL(k_unify_perm, dim3((unsigned)((NR + 63) / 64)), dim3(64), 0, (hipStream_t)stream, tc, tf,
                       (long long)NR, Sc, Sf, perm);
    return p3d_check_launch();
}

int p3d_sigma2density_f32(const float* sigma, const unsigned char* cropmask, int64_t M, float cull_thresh, float* out_density,
                          void* stream) {
    if (!sigma || !out_density || M <= 0) return P3D_E_ARG;
    const int vec = (((uintptr_t)sigma | (uintptr_t)out_density) & 15) == 0 && ((uintptr_t)cropmask & 3) == 0;
    hipLaunchKernelGGL(k_sigma2density, dim3((unsigned)((M + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, sigma, cropmask,
                       (long long)M, cull_thresh, out_density, vec);
    return p3d_check_launch();
}

}  // extern "C"
