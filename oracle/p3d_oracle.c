/* p3d_oracle.c — CPU ORACLE for the PAniC-3D triplane volumetric-rendering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (panic3d-anime-reconstruction_amd/) never imports, links or calls anything in oracle/.
 *
 * What it is: a plain-C restatement of the reference's PyTorch algorithm, written against the
 * arithmetic contract in include/p3d_numerics.h (fixed operation order, explicit fma, polynomial
 * exp/log1p).  Each function cites the reference file:line it follows (paths relative to
 * /root/reference/_train/eg3dc/src/training/).
 *
 * Parity status: PINNED.  the .npz files under tests/golden/ hold outputs of the reference itself (imported from
 * /root/reference on CPU by tests/golden/make_golden.py); tests/test_oracle_golden.py checks this
 * oracle against them (floats within the fp32 tolerance written there, indices by exact-match
 * count).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma -fopenmp).  -ffp-contract=off matters:
 * every fused multiply-add below is an explicit fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/p3d_numerics.h"

#define OR_C 32      /* channels per plane (triplane.py:41 triplane_width=32) */
#define OR_HID 64    /* OSGDecoder hidden_dim (triplane.py:519) */
#define OR_OUT 33    /* 1 + decoder_output_dim (triplane.py:525) */

/* flag bits — same values as include/panic3d_hip.h P3D_FLAG_* */
#define OR_FLAG_CROP 1
#define OR_FLAG_CULL 2
#define OR_FLAG_BINARIZE 4
#define OR_FLAG_FORCE_SIGMOID 8
#define OR_FLAG_WHITE_BACK 16
#define OR_FLAG_DISPARITY 4096

/* ------------------------------------------------------------------ scalar math of the contract */
static inline float or_exp(float x) {
    if (x != x) return x;
    if (x < P3D_EXP_LO) return 0.0f;
    if (x > P3D_EXP_HI) return INFINITY;
    float n = rintf(x * P3D_LOG2E);
    float r = fmaf(n, -P3D_LN2_HI, x);
    r = fmaf(n, -P3D_LN2_LO, r);
    float p = P3D_EXP_C6;
    p = fmaf(p, r, P3D_EXP_C5);
    p = fmaf(p, r, P3D_EXP_C4);
    p = fmaf(p, r, P3D_EXP_C3);
    p = fmaf(p, r, P3D_EXP_C2);
    p = fmaf(p, r, P3D_EXP_C1);
    p = fmaf(p, r, P3D_EXP_C0);
    return ldexpf(p, (int)n);
}

static inline float or_log1p01(float z) {
    float q = P3D_L1P_C8;
    q = fmaf(q, z, P3D_L1P_C7);
    q = fmaf(q, z, P3D_L1P_C6);
    q = fmaf(q, z, P3D_L1P_C5);
    q = fmaf(q, z, P3D_L1P_C4);
    q = fmaf(q, z, P3D_L1P_C3);
    q = fmaf(q, z, P3D_L1P_C2);
    q = fmaf(q, z, P3D_L1P_C1);
    q = fmaf(q, z, P3D_L1P_C0);
    return q * z;
}

/* arguments known to be <= 0 */
static inline float or_exp_nonpos(float x) { return or_exp(fmaxf(x, P3D_EXP_LO)); }

/* torch.nn.Softplus(beta=1, threshold=20): triplane.py:524; F.softplus: ray_marcher.py:33, renderer.py:151.
 * Above the threshold the expression equals x in binary32 (see include/p3d_numerics.h), so there is no branch. */
static inline float or_softplus(float x) {
    float z = or_exp_nonpos(-fabsf(x));
    return fmaxf(x, 0.0f) + or_log1p01(z);
}

/* reciprocal of d in [1,2] by the contract's fixed Newton sequence */
static inline float or_rcp12(float d) {
    float r = fmaf(d, -P3D_RCP_A, P3D_RCP_B);
    for (int i = 0; i < 3; ++i) {
        float e = fmaf(-d, r, 1.0f);
        r = fmaf(r, e, r);
    }
    return r;
}

/* torch.sigmoid: triplane.py:540 */
static inline float or_sigmoid(float x) {
    float z = or_exp_nonpos(-fabsf(x));
    float r = or_rcp12(1.0f + z);
    return (x >= 0.0f) ? r : z * r;
}

/* ------------------------------------------------------------------ triplane sample + decoder */

/* One plane, all 32 channels: F.grid_sample(bilinear, zeros, align_corners=False), renderer.py:80.
 * plane points at [C][H][W] (the reference's NCHW layout). */
static void or_sample_plane(const float* plane, int H, int W, float gx, float gy, float* f) {
    float ix = (gx + 1.0f) * (0.5f * (float)W) - 0.5f;
    float iy = (gy + 1.0f) * (0.5f * (float)H) - 0.5f;
    if (!(ix > -1.0f && ix < (float)W && iy > -1.0f && iy < (float)H)) {
        for (int c = 0; c < OR_C; ++c) f[c] = 0.0f;
        return;
    }
    float fx0 = floorf(ix), fy0 = floorf(iy);
    float wx1 = ix - fx0, wy1 = iy - fy0;
    float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    float nw = wy0 * wx0, ne = wy0 * wx1, sw = wy1 * wx0, se = wy1 * wx1;
    int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    int vx0 = (x0 >= 0 && x0 < W), vx1 = (x1 >= 0 && x1 < W);
    int vy0 = (y0 >= 0 && y0 < H), vy1 = (y1 >= 0 && y1 < H);
    const long HW = (long)H * W;
    for (int c = 0; c < OR_C; ++c) {
        const float* pc = plane + c * HW;
        float v00 = (vy0 && vx0) ? pc[(long)y0 * W + x0] : 0.0f;
        float v01 = (vy0 && vx1) ? pc[(long)y0 * W + x1] : 0.0f;
        float v10 = (vy1 && vx0) ? pc[(long)y1 * W + x0] : 0.0f;
        float v11 = (vy1 && vx1) ? pc[(long)y1 * W + x1] : 0.0f;
        float a = nw * v00;
        a = fmaf(ne, v01, a);
        a = fmaf(sw, v10, a);
        a = fmaf(se, v11, a);
        f[c] = a;
    }
}

typedef struct {
    const float *w0, *b0, *w1, *b1; /* pre-scaled: [64][32], [64], [33][64], [33] (networks_stylegan2.py:121-127) */
} or_mlp;

/* OSGDecoder.forward (triplane.py:528-544) on the three sampled feature vectors of ONE point + the crop / cull masks
 * (renderer.py:138-153,187-198; px, pz: the point's position for the crop mask).  rgb may be NULL (density only). */
static void or_decode_features(const float* f0, const float* f1, const float* f2, float px, float pz, const or_mlp* m, int flags,
                               float crop_limit, float cull_thresh, float* sigma_out, float* rgb) {
    float X[OR_C], h[OR_HID];
    for (int c = 0; c < OR_C; ++c) X[c] = ((f0[c] + f1[c]) + f2[c]) * P3D_THIRD; /* triplane.py:530 mean(1) */
    for (int n = 0; n < OR_HID; ++n) { /* net[0] + Softplus: triplane.py:522-524 */
        float a = m->b0[n];
        const float* w = m->w0 + n * OR_C;
        for (int s = 0; s < 16; ++s) {
            a = fmaf(w[s], X[s], a);
            a = fmaf(w[16 + s], X[16 + s], a);
        }
        h[n] = or_softplus(a);
    }
    float alo = m->b1[0], ahi = 0.0f; /* net[2] row 0 -> sigma: triplane.py:543 */
    for (int t = 0; t < 2; ++t)
        for (int s = 0; s < 16; ++s) {
            int nlo = 32 * t + (s & 3) + 8 * (s >> 2), nhi = nlo + 4;
            alo = fmaf(m->w1[nlo], h[nlo], alo);
            ahi = fmaf(m->w1[nhi], h[nhi], ahi);
        }
    float sigma = alo + ahi;
    if (rgb) {
        for (int o = 1; o < OR_OUT; ++o) { /* net[2] rows 1..32 -> rgb: triplane.py:539-542 */
            float a = m->b1[o];
            const float* w = m->w1 + o * OR_HID;
            for (int t = 0; t < 2; ++t)
                for (int s = 0; s < 16; ++s) {
                    int nlo = 32 * t + (s & 3) + 8 * (s >> 2), nhi = nlo + 4;
                    a = fmaf(w[nlo], h[nlo], a);
                    a = fmaf(w[nhi], h[nhi], a);
                }
            float sg = or_sigmoid(a);
            rgb[o - 1] = (flags & OR_FLAG_FORCE_SIGMOID) ? sg : sg * 1.002f - 0.001f;
        }
    }
    if (flags & OR_FLAG_CROP) { /* triplane_crop_mask: renderer.py:138-149 (the allow_bottom term is a no-op) */
        if (fabsf(px) > crop_limit || fabsf(pz) > crop_limit) sigma = P3D_SIGMA_MASKED;
    }
    if (flags & (OR_FLAG_CULL | OR_FLAG_BINARIZE)) { /* cull_clouds_mask: renderer.py:150-153 */
        float a = 1.0f - or_exp_nonpos(-or_softplus(sigma - 1.0f));
        if (flags & OR_FLAG_BINARIZE)
            sigma = (a < cull_thresh) ? P3D_SIGMA_MASKED : P3D_SIGMA_SOLID; /* renderer.py:190-193 */
        else if (a < cull_thresh)
            sigma = P3D_SIGMA_MASKED; /* renderer.py:194-196 */
    }
    *sigma_out = sigma;
}

/* run_model for ONE point: sample_from_planes (renderer.py:68-81) + OSGDecoder.forward (triplane.py:528-544)
 * + crop/cull masks (renderer.py:138-153,187-198).  planes_n: this image's [3][C][H][W].
 * rgb may be NULL (density only). */
static void or_decode_point(const float* planes_n, int H, int W, float px, float py, float pz, const or_mlp* m,
                            float coord_scale, int plane_mode, int flags, float crop_limit, float cull_thresh,
                            float* sigma_out, float* rgb) {
    const long plane_sz = (long)OR_C * H * W;
    float qx = px * coord_scale, qy = py * coord_scale, qz = pz * coord_scale; /* renderer.py:77 */
    float f0[OR_C], f1[OR_C], f2[OR_C];
    or_sample_plane(planes_n + 0 * plane_sz, H, W, qx, qy, f0); /* generate_planes: renderer.py:26-50 */
    or_sample_plane(planes_n + 1 * plane_sz, H, W, qx, qz, f1);
    if (plane_mode)
        or_sample_plane(planes_n + 2 * plane_sz, H, W, qy, qz, f2);
    else
        or_sample_plane(planes_n + 2 * plane_sz, H, W, qz, qx, f2);
    or_decode_features(f0, f1, f2, px, pz, m, flags, crop_limit, cull_thresh, sigma_out, rgb);
}

/* OSGDecoder.forward over already sampled features (triplane.py:528-544): feats [N][3][M][32] -> sigma [N][M], rgb [N][M][32].
 * No masks (the decoder knows no positions). */
void p3d_oracle_decode_features(const float* feats, int N, long M, const float* w0, const float* b0, const float* w1,
                                const float* b1, int flags, float* out_sigma, float* out_rgb) {
    or_mlp m = {w0, b0, w1, b1};
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N * M; ++i) {
        const long n = i / M, q = i - n * M;
        const float* base = feats + (n * 3 * M + q) * OR_C;
        or_decode_features(base, base + M * OR_C, base + 2 * M * OR_C, 0.0f, 0.0f, &m, flags & OR_FLAG_FORCE_SIGMOID, 0.0f, 0.0f,
                           out_sigma + i, out_rgb ? out_rgb + 32 * i : NULL);
    }
}

/* ImportanceRenderer.run_model over a point cloud (renderer.py:266-280), used by sample_mixed (triplane.py:273-298)
 * and get_eg3d_volume (_util/eg3d_metrics3d.py:140).  planes [N][3][32][H][W], coords [N][M][3]. */
void p3d_oracle_decode(const float* planes, int N, int H, int W, const float* coords, long M, const float* w0,
                       const float* b0, const float* w1, const float* b1, float coord_scale, int plane_mode, int flags,
                       float crop_limit, float cull_thresh, float* out_sigma, float* out_rgb) {
    or_mlp m = {w0, b0, w1, b1};
    const long img = 3L * OR_C * H * W;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N * M; ++i) {
        long n = i / M;
        const float* p = coords + 3 * i;
        or_decode_point(planes + n * img, H, W, p[0], p[1], p[2], &m, coord_scale, plane_mode, flags, crop_limit,
                        cull_thresh, out_sigma + i, out_rgb ? out_rgb + 32 * i : NULL);
    }
}

/* ------------------------------------------------------------------ sample_stratified */
/* renderer.py:320-324 (numeric ray_start / ray_end, disparity_space_sampling False).
 * torch.linspace on CPU is symmetric about the midpoint and fused (verified bit-for-bit). */
static void or_stratified(float start, float end, float delta, int S, const float* jitter, float* t) {
    float step = (end - start) / (float)(S - 1);
    for (int i = 0; i < S; ++i) {
        float lin = (i < S / 2) ? fmaf(step, (float)i, start) : fmaf(-step, (float)(S - 1 - i), end);
        t[i] = lin + jitter[i] * delta;
    }
}

/* renderer.py:317-319 (ray_start / ray_end tensors: the 'auto' limits of :165-171): math_utils.linspace (:101-118) is
 * start + (arange(S) / (S - 1)) * (stop - start) — an fp32 division, a multiplication and an addition, each rounded — and
 * depth_delta = (ray_end - ray_start) / (S - 1) per ray. */
static void or_stratified_limits(float start, float end, int S, const float* jitter, float* t) {
    const float span = end - start;
    const float delta = span / (float)(S - 1);
    for (int i = 0; i < S; ++i) {
        const float step = (float)i / (float)(S - 1);
        const float prod = step * span;
        const float lin = start + prod;
        t[i] = lin + jitter[i] * delta;
    }
}

/* renderer.py:309-316 (disparity_space_sampling): d = torch.linspace(0, 1, S) + rand * (1 / (S - 1));
 * t = 1. / (1. / ray_start * (1. - d) + 1. / ray_end * d) with the two reciprocals formed in binary64 by Python and used as
 * binary32 scalars: inv_start = (float)(1 / ray_start), inv_end = (float)(1 / ray_end), delta = (float)(1 / (S - 1)). */
static void or_stratified_disparity(float inv_start, float inv_end, float delta, int S, const float* jitter, float* t) {
    float step = (1.0f - 0.0f) / (float)(S - 1);
    for (int i = 0; i < S; ++i) {
        float lin = (i < S / 2) ? fmaf(step, (float)i, 0.0f) : fmaf(-step, (float)(S - 1 - i), 1.0f);
        float d = lin + jitter[i] * delta;
        float a = inv_start * (1.0f - d);
        float b = inv_end * d;
        t[i] = 1.0f / (a + b);
    }
}

void p3d_oracle_sample_stratified(float start, float end, float delta, int S, const float* jitter, long NR, float* out) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < NR; ++r) or_stratified(start, end, delta, S, jitter + r * S, out + r * S);
}

/* ------------------------------------------------------------------ MipRayMarcher2.run_forward */
/* ray_marcher.py:25-57 for ONE ray.  colors [S][K], sigma [S], t [S] (sorted).  Outputs: C[K] (before white-back /
 * rescale), Wsum, Dsum, weights [S-1]. */
static void or_march_ray(const float* colors, const float* sigma, const float* t, int S, int K, float* C, float* Wsum,
                         float* Dsum, float* weights) {
    double Td = 1.0; /* torch.cumprod accumulates in double on CPU; alpha_shifted[0] = 1 (ray_marcher.py:41-42) */
    float W = 0.0f, D = 0.0f;
    for (int k = 0; k < K; ++k) C[k] = 0.0f;
    for (int i = 0; i < S - 1; ++i) {
        float dl = t[i + 1] - t[i];                  /* :26 */
        float sm = (sigma[i] + sigma[i + 1]) * 0.5f; /* :28 */
        float tm = (t[i] + t[i + 1]) * 0.5f;         /* :29 */
        float rho = or_softplus(sm - 1.0f);          /* :33 */
        float dd = rho * dl;                         /* :37 */
        float alpha = 1.0f - or_exp(-dd);            /* :39 */
        float T = (float)Td;
        float w = alpha * T; /* :42 */
        Td = Td * (double)((1.0f - alpha) + 1e-10f);
        if (weights) weights[i] = w;
        for (int k = 0; k < K; ++k) {
            float cm = (colors[i * K + k] + colors[(i + 1) * K + k]) * 0.5f; /* :27 */
            C[k] = fmaf(w, cm, C[k]);                                        /* :44 */
        }
        W = W + w;          /* :45 */
        D = fmaf(w, tm, D); /* :46 */
    }
    *Wsum = W;
    *Dsum = D;
}

/* finish: depth division / nan_to_num / clamp (ray_marcher.py:46-50), white_back and rescale (:52-55) */
static void or_finish_ray(float* C, int K, float W, float D, float tmin, float tmax, int white_back, float* depth) {
    float d = D / W;
    if (d != d) d = INFINITY; /* nan_to_num(nan=inf); +-inf stay (then clamped) */
    if (d < tmin) d = tmin;
    if (d > tmax) d = tmax;
    *depth = d;
    for (int k = 0; k < K; ++k) {
        float c = C[k];
        if (white_back) c = (c + 1.0f) - W;
        C[k] = c * 2.0f - 1.0f;
    }
}

/* Standalone marcher: colors [NR][S][K], sigma [NR][S], depths [NR][S] -> rgb [NR][K], depth [NR], weights [NR][S-1]. */
void p3d_oracle_composite(const float* colors, const float* sigma, const float* depths, long NR, int S, int K,
                          int white_back, float* out_rgb, float* out_depth, float* out_weights) {
    float tmin = INFINITY, tmax = -INFINITY;
    for (long i = 0; i < NR * S; ++i) { /* torch.min/max(depths): ray_marcher.py:50 */
        if (depths[i] < tmin) tmin = depths[i];
        if (depths[i] > tmax) tmax = depths[i];
    }
#pragma omp parallel for schedule(static)
    for (long r = 0; r < NR; ++r) {
        float W, D;
        float* C = out_rgb + r * K;
        or_march_ray(colors + r * S * K, sigma + r * S, depths + r * S, S, K, C, &W, &D,
                     out_weights ? out_weights + r * (S - 1) : NULL);
        or_finish_ray(C, K, W, D, tmin, tmax, white_back, out_depth + r);
    }
}

/* ------------------------------------------------------------------ sample_importance / sample_pdf */
/* renderer.py:328-387 for ONE ray. t [Sc], w [Sc-1], u [Sf] -> t_fine [Sf], inds [Sf] (searchsorted result). */
static void or_importance_ray(const float* t, const float* w, int Sc, const float* u, int Sf, float* t_fine,
                              int32_t* inds) {
    const int L = Sc - 1, Ns = Sc - 3;
    float m[256], ws[256], b[256], cdf[256];
    m[0] = w[0]; /* max_pool1d(2,1,padding=1): -inf padding, renderer.py:339 */
    for (int j = 1; j < L; ++j) m[j] = fmaxf(w[j - 1], w[j]);
    m[L] = w[L - 1];
    for (int j = 0; j < L; ++j) ws[j] = (m[j] + m[j + 1]) * 0.5f + 0.01f; /* avg_pool1d(2,1) + 0.01: :340-341 */
    for (int j = 0; j < L; ++j) b[j] = 0.5f * (t[j] + t[j + 1]);          /* z_vals_mid: :343 */
    double sum = 0.0;
    for (int j = 0; j < Ns; ++j) sum += (double)(ws[j + 1] + 1e-5f); /* weights[:,1:-1] + eps; torch.sum: :361-362 */
    float fsum = (float)sum;
    double acc = 0.0;
    cdf[0] = 0.0f; /* :364 */
    for (int j = 0; j < Ns; ++j) {
        float pdf = (ws[j + 1] + 1e-5f) / fsum; /* :362 */
        acc += (double)pdf;                     /* torch.cumsum accumulates in double on CPU: :363 */
        cdf[j + 1] = (float)acc;
    }
    for (int i = 0; i < Sf; ++i) {
        float ui = u[i];
        int k = 0;
        while (k <= Ns && cdf[k] <= ui) ++k; /* searchsorted(cdf, u, right=True): :374 */
        int below = k - 1 > 0 ? k - 1 : 0;   /* :375 */
        int above = k < Ns ? k : Ns;         /* :376 (N_samples_ = Ns) */
        float den = cdf[above] - cdf[below]; /* :382 */
        if (den < 1e-5f) den = 1.0f;         /* :383 */
        t_fine[i] = b[below] + ((ui - cdf[below]) / den) * (b[above] - b[below]); /* :386 */
        if (inds) inds[i] = k;
    }
}

void p3d_oracle_importance(const float* depths, const float* weights, long NR, int Sc, int Sf, const float* u,
                           float* out_depths, int32_t* out_inds) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < NR; ++r)
        or_importance_ray(depths + r * Sc, weights + r * (Sc - 1), Sc, u + r * Sf, Sf, out_depths + r * Sf,
                          out_inds ? out_inds + r * Sf : NULL);
}

/* ------------------------------------------------------------------ unify_samples */
/* renderer.py:289-301: concat coarse ++ fine, sort ascending by depth (stable), return permutation. */
static void or_merge_perm(const float* tc, int Sc, const float* tf, int Sf, int32_t* perm) {
    int S = Sc + Sf;
    float key[512];
    for (int i = 0; i < Sc; ++i) key[i] = tc[i];
    for (int i = 0; i < Sf; ++i) key[Sc + i] = tf[i];
    for (int i = 0; i < S; ++i) perm[i] = i;
    for (int i = 1; i < S; ++i) { /* insertion sort = stable */
        int32_t p = perm[i];
        float kp = key[p];
        int j = i - 1;
        while (j >= 0 && key[perm[j]] > kp) {
            perm[j + 1] = perm[j];
            --j;
        }
        perm[j + 1] = p;
    }
}

void p3d_oracle_unify_perm(const float* depths_coarse, const float* depths_fine, long NR, int Sc, int Sf, int32_t* perm) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < NR; ++r) or_merge_perm(depths_coarse + r * Sc, Sc, depths_fine + r * Sf, Sf, perm + r * (Sc + Sf));
}

/* ------------------------------------------------------------------ ImportanceRenderer.forward */
typedef struct {
    float coord_scale;  /* (float)(2.0/box_warp)                    renderer.py:77 */
    float ray_start;    /* rendering_options['ray_start']            renderer.py:177 */
    float ray_end;      /* rendering_options['ray_end'] */
    float depth_delta;  /* (float)((ray_end-ray_start)/(Sc-1)) computed in double: renderer.py:323 */
    float crop_limit;   /* (float)(box_warp/2 - triplane_crop)       renderer.py:139-142 */
    float cull_thresh;  /* cull_clouds or binarize_clouds value      renderer.py:190-196 */
    int32_t Sc;         /* depth_resolution */
    int32_t Sf;         /* depth_resolution_importance (0: single pass, renderer.py:254-259) */
    int32_t plane_mode; /* use_triplane                              renderer.py:41-49 */
    int32_t flags;      /* OR_FLAG_* */
} p3d_oracle_opts;

typedef struct { /* optional per-stage dumps; any pointer may be NULL */
    float* depths_coarse;   /* [N*R][Sc] */
    float* sigma_coarse;    /* [N*R][Sc]  (after masks) */
    float* rgb_coarse;      /* [N*R][Sc][32] */
    float* weights_coarse;  /* [N*R][Sc-1] */
    float* depths_fine;     /* [N*R][Sf] */
    int32_t* inds;          /* [N*R][Sf] */
    float* sigma_fine;      /* [N*R][Sf] */
    int32_t* perm;          /* [N*R][Sc+Sf] */
    float* depth_unclamped; /* [N*R] */
    float* tminmax;         /* [2] */
} p3d_oracle_dumps;

/* renderer.py:162-264.  planes [N][3][32][H][W]; rays_o/rays_d [N][R][3]; jitter [N][R][Sc]; u [N*R][Sf].
 * Outputs: feat [N][R][32], depth [N][R], wsum [N][R], xyz [N][R][3]. */
/* ray_start / ray_end: NULL, or per-ray limits [N][R] (renderer.py:165-171 'auto': get_ray_limits_box + the patching of the
 * rays that miss the box, both done by the caller) — the scalar limits of `o` are then unused. */
int p3d_oracle_render_limits(const float* planes, int N, int H, int W, const float* rays_o, const float* rays_d, long R,
                             const float* jitter, const float* u, const float* w0, const float* b0, const float* w1,
                             const float* b1, const float* ray_start, const float* ray_end, const p3d_oracle_opts* o,
                             float* out_feat, float* out_depth, float* out_wsum, float* out_xyz, const p3d_oracle_dumps* dmp);

int p3d_oracle_render(const float* planes, int N, int H, int W, const float* rays_o, const float* rays_d, long R,
                      const float* jitter, const float* u, const float* w0, const float* b0, const float* w1,
                      const float* b1, const p3d_oracle_opts* o, float* out_feat, float* out_depth, float* out_wsum,
                      float* out_xyz, const p3d_oracle_dumps* dmp) {
    return p3d_oracle_render_limits(planes, N, H, W, rays_o, rays_d, R, jitter, u, w0, b0, w1, b1, NULL, NULL, o, out_feat,
                                    out_depth, out_wsum, out_xyz, dmp);
}

int p3d_oracle_render_limits(const float* planes, int N, int H, int W, const float* rays_o, const float* rays_d, long R,
                             const float* jitter, const float* u, const float* w0, const float* b0, const float* w1,
                             const float* b1, const float* ray_start, const float* ray_end, const p3d_oracle_opts* o,
                             float* out_feat, float* out_depth, float* out_wsum, float* out_xyz, const p3d_oracle_dumps* dmp) {
    const int Sc = o->Sc, Sf = o->Sf, S = Sc + Sf, K = 35;
    if (Sc < 4 || Sc > 192 || Sf < 0 || Sf > 192) return -1;
    or_mlp m = {w0, b0, w1, b1};
    const long img = 3L * OR_C * H * W;
    const long NR = (long)N * R;
    float* Wtot = (float*)malloc(sizeof(float) * NR);
    float* Dtot = (float*)malloc(sizeof(float) * NR);
    float gmin = INFINITY, gmax = -INFINITY;
    const int white_back = (o->flags & OR_FLAG_WHITE_BACK) != 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(min : gmin) reduction(max : gmax)
    for (long r = 0; r < NR; ++r) {
        const float* planes_n = planes + (r / R) * img;
        const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
        const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
        float t[384], sg[384], col[384 * 35], wts[384];
        /* coarse pass: renderer.py:174-206 */
        if (ray_start && ray_end) or_stratified_limits(ray_start[r], ray_end[r], Sc, jitter + r * Sc, t);
        else if (o->flags & OR_FLAG_DISPARITY) or_stratified_disparity(o->ray_start, o->ray_end, o->depth_delta, Sc, jitter + r * Sc, t);
        else or_stratified(o->ray_start, o->ray_end, o->depth_delta, Sc, jitter + r * Sc, t);
        for (int i = 0; i < Sc; ++i) {
            float px = ox + t[i] * dx, py = oy + t[i] * dy, pz = oz + t[i] * dz; /* :179 (mul, then add) */
            or_decode_point(planes_n, H, W, px, py, pz, &m, o->coord_scale, o->plane_mode, o->flags, o->crop_limit,
                            o->cull_thresh, &sg[i], &col[i * K]);
            col[i * K + 32] = px; /* out['xyz'] = sample_coordinates: :279 */
            col[i * K + 33] = py;
            col[i * K + 34] = pz;
        }
        if (dmp) {
            if (dmp->depths_coarse) memcpy(dmp->depths_coarse + r * Sc, t, sizeof(float) * Sc);
            if (dmp->sigma_coarse) memcpy(dmp->sigma_coarse + r * Sc, sg, sizeof(float) * Sc);
            if (dmp->rgb_coarse)
                for (int i = 0; i < Sc; ++i) memcpy(dmp->rgb_coarse + (r * Sc + i) * 32, &col[i * K], sizeof(float) * 32);
        }
        float C[35], Wsum, Dsum;
        float* tm = t; /* merged arrays (in place when Sf == 0) */
        float* sgm = sg;
        float* colm = col;
        float t2[384], sg2[384], col2[384 * 35];
        int Sm = Sc;
        if (Sf > 0) {
            /* coarse marcher for the weights only: renderer.py:211 */
            or_march_ray(col, sg, t, Sc, 0, C, &Wsum, &Dsum, wts); /* K=0: colours are not needed here */
            if (dmp && dmp->weights_coarse) memcpy(dmp->weights_coarse + r * (Sc - 1), wts, sizeof(float) * (Sc - 1));
            /* importance depths: renderer.py:213 */
            int32_t inds[192];
            or_importance_ray(t, wts, Sc, u + r * Sf, Sf, t + Sc, inds);
            if (dmp && dmp->depths_fine) memcpy(dmp->depths_fine + r * Sf, t + Sc, sizeof(float) * Sf);
            if (dmp && dmp->inds) memcpy(dmp->inds + r * Sf, inds, sizeof(int32_t) * Sf);
            /* fine pass: renderer.py:215-241 */
            for (int i = Sc; i < S; ++i) {
                float px = ox + t[i] * dx, py = oy + t[i] * dy, pz = oz + t[i] * dz;
                or_decode_point(planes_n, H, W, px, py, pz, &m, o->coord_scale, o->plane_mode, o->flags, o->crop_limit,
                                o->cull_thresh, &sg[i], &col[i * K]);
                col[i * K + 32] = px;
                col[i * K + 33] = py;
                col[i * K + 34] = pz;
            }
            if (dmp && dmp->sigma_fine) memcpy(dmp->sigma_fine + r * Sf, sg + Sc, sizeof(float) * Sf);
            /* unify_samples: renderer.py:243-246 */
            int32_t perm[384];
            or_merge_perm(t, Sc, t + Sc, Sf, perm);
            if (dmp && dmp->perm) memcpy(dmp->perm + r * S, perm, sizeof(int32_t) * S);
            for (int i = 0; i < S; ++i) {
                t2[i] = t[perm[i]];
                sg2[i] = sg[perm[i]];
                memcpy(&col2[i * K], &col[perm[i] * K], sizeof(float) * K);
            }
            tm = t2;
            sgm = sg2;
            colm = col2;
            Sm = S;
        }
        /* final marcher on [rgb | xyz]: renderer.py:250-259 */
        or_march_ray(colm, sgm, tm, Sm, K, C, &Wsum, &Dsum, NULL);
        for (int i = 0; i < Sm; ++i) { /* torch.min/max(depths) over the whole call: ray_marcher.py:50 */
            if (tm[i] < gmin) gmin = tm[i];
            if (tm[i] > gmax) gmax = tm[i];
        }
        memcpy(out_feat + r * 32, C, sizeof(float) * 32);
        memcpy(out_xyz + r * 3, C + 32, sizeof(float) * 3);
        Wtot[r] = Wsum;
        Dtot[r] = Dsum;
    }
    if (dmp && dmp->tminmax) {
        dmp->tminmax[0] = gmin;
        dmp->tminmax[1] = gmax;
    }
#pragma omp parallel for schedule(static)
    for (long r = 0; r < NR; ++r) {
        float C[35];
        memcpy(C, out_feat + r * 32, sizeof(float) * 32);
        memcpy(C + 32, out_xyz + r * 3, sizeof(float) * 3);
        if (dmp && dmp->depth_unclamped) dmp->depth_unclamped[r] = Dtot[r] / Wtot[r];
        or_finish_ray(C, K, Wtot[r], Dtot[r], gmin, gmax, white_back, out_depth + r);
        memcpy(out_feat + r * 32, C, sizeof(float) * 32);
        memcpy(out_xyz + r * 3, C + 32, sizeof(float) * 3);
        out_wsum[r] = Wtot[r]; /* weights.sum(2): renderer.py:264 */
    }
    free(Wtot);
    free(Dtot);
    return 0;
}

/* expose the scalar functions for unit tests of the contract */
/* sigma2density + crop / cull masks of get_eg3d_volume (_util/eg3d_metrics3d.py:65-69,153-163; cull_clouds_mask renderer.py:150-153
 * applied to the densities, as the reference does) */
void p3d_oracle_sigma2density(const float* sigma, const unsigned char* cropmask, long n, float cull_thresh, float* out) {
    for (long i = 0; i < n; ++i) {
        float d = 1.0f - or_exp_nonpos(-or_softplus(sigma[i] - 1.0f));
        if (cropmask && cropmask[i]) d = -1000.0f;
        if (cull_thresh >= 0.0f) {
            float a2 = 1.0f - or_exp_nonpos(-or_softplus(d - 1.0f));
            if (a2 < cull_thresh) d = -1000.0f;
        }
        out[i] = d;
    }
}

void p3d_oracle_math(const float* x, long n, int which, float* y) {
    for (long i = 0; i < n; ++i)
        y[i] = which == 0 ? or_exp(x[i]) : which == 1 ? or_log1p01(x[i]) : which == 2 ? or_softplus(x[i]) : or_sigmoid(x[i]);
}
