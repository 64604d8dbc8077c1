/* panic3d_hip.h — C ABI of libpanic3d_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for PAniC-3D's triplane volumetric-rendering hot path.  The reference has NO native code for
 * this path (it is a composition of stock torch ops); its own FFI convention for native operators is
 * the pybind11 plugin surface of torch_utils/custom_ops.py:61 (`get_plugin`) as used by torch_utils/ops/bias_act.py:40-50
 * and bias_act.cpp:36-101: stateless functions, inputs validated, launched on the caller's current stream, nothing
 * allocated behind the caller's back.  This header keeps that convention in plain C:
 *
 *   - every entry point is `extern "C"`, takes raw DEVICE pointers + sizes + a hipStream_t (passed as void*), and
 *     returns 0 on success, a negative P3D_E_* code for an argument error, or a positive hipError_t;
 *   - no allocation, no global mutable state, re-entrant; the caller owns every buffer including the workspace
 *     (size from p3d_render_workspace_bytes);
 *   - all arithmetic follows include/p3d_numerics.h (the arithmetic contract), so results are reproducible bit-for-bit.
 *
 * Each function cites the reference code it replaces (paths relative to /root/reference/_train/eg3dc/src/).
 * The Python binding a maintainer adds is in INTEGRATION.md; the in-tree one is panic3d-anime-reconstruction_amd/_lib.py.
 */
#ifndef PANIC3D_HIP_H
#define PANIC3D_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct layout, an argument list or a workspace size changes: the binding checks p3d_abi_version() against
 * the value it was written for, so that a stale libpanic3d_hip.so is refused instead of being called with the wrong layout. */
#define P3D_ABI_VERSION 9  /* 9: p3d_conv_args rgb_* (a block's ToRGB on its conv1 launch), p3d_conv_fuses_torgb, p3d_torgb_partial_bytes, p3d_torgb_combine_f32; p3d_conv_args.w_f16_layout, p3d_conv_weight_layout, p3d_conv_weights_to_f16x2_layout; 8: p3d_conv_takes_image, the convolution workspace must be 256-byte aligned; 7: p3d_struct_layout, p3d_decode_features_f32; 6: p3d_conv_args x_img / y_img / y_img_styles, p3d_act_to_image_f32; 5: caller-owned saturation flag, p3d_modconv2d_ex_f32, p3d_torgb_f32 */

#define P3D_OK 0
#define P3D_E_ARG (-1)       /* null pointer / non-positive size */
#define P3D_E_RANGE (-2)     /* Sc/Sf/H/W/C outside what the kernels support */
#define P3D_E_WORKSPACE (-3) /* workspace too small */

/* flag bits of p3d_opts.flags */
#define P3D_FLAG_CROP 1          /* triplane_crop given          (renderer.py:187-188) */
#define P3D_FLAG_CULL 2          /* cull_clouds given            (renderer.py:194-196) */
#define P3D_FLAG_BINARIZE 4      /* binarize_clouds given        (renderer.py:190-193) */
#define P3D_FLAG_FORCE_SIGMOID 8 /* OSGDecoder.force_sigmoid     (training/triplane.py:539-542) */
#define P3D_FLAG_WHITE_BACK 16   /* rendering_options.white_back (ray_marcher.py:52-53) */
#define P3D_FLAG_NO_EARLY_OUT 32 /* p3d_render_f32: disable the exact early-outs (decode every sample; measurement / tests) */
#define P3D_FLAG_NO_PAIR 256 /* p3d_render_f32: never use a small-launch kernel (8 rays x 4 samples / 16 rays x 2 samples per wave); tests */
#define P3D_FLAG_PAIR16 16384 /* p3d_render_f32: small launches take the 16 rays x 2 samples kernel (k_render_pair) ... */
#define P3D_FLAG_QUAD8 32768  /* ... / the 8 rays x 4 samples kernel (k_render_quad), whatever the size heuristic says (default: quad for
                                 launches of <= 8192 rays and for the tolerance mode at 96+96 samples, pair otherwise); tests, A/B timing */
#define P3D_FLAG_SKIP_CROPPED 128 /* p3d_grid_density_f32 with out_cropmask: points whose crop mask fires are not decoded and get
                                     out_sigma = -1000 (get_eg3d_volume overwrites their density anyway, eg3d_metrics3d.py:155-159) */
#define P3D_FLAG_FAST_COLOR 512 /* p3d_render_f32: the caller ACCEPTS fp32-tolerance colours, so the final pass may run in its
                                   tolerance mode: both decoder layers on f16 MFMA with two-term operand splits (~2^-21 per
                                   product) and hardware exp2 / log2 / rcp activations (both render kernels have the variant).  The
                                   coarse pass — hence the inverse-CDF indices, the fine depths and the merged depth order —
                                   stays on the exact contract; feat / depth / wsum / xyz agree with it to fp32 tolerance
                                   (tests/test_hip_parity.py::test_fast_color_*). */
#define P3D_FLAG_PER_VIEW_CLAMP 1024 /* p3d_render_f32: clamp each of the N images' depths to that image's own [min t, max t] instead
                                        of the call's (ray_marcher.py:49-50 takes torch.min/max over the whole batch).  For callers that
                                        batch what the reference renders as N separate calls (generate.py's view loop): the batched
                                        launch then reproduces the per-view results bit for bit, depth included. */
#define P3D_FLAG_NO_STAGING 2048 /* p3d_grid_density_f32: accepted, no effect (direct gathers are the default in every mode since the
                                    quad-cooperative gathers: 512^3 tolerance query 7.7 ms direct, 10.0 ms staged) */
#define P3D_FLAG_FORCE_STAGING 8192 /* p3d_grid_density_f32: stage each wave's texel boxes through LDS and take the taps from there
                                       (north_star's "LDS-staged plane tiles"; measurement / tests; same bits as the direct gathers) */
#define P3D_FLAG_DISPARITY 4096 /* p3d_render_f32: rendering_options.disparity_space_sampling (renderer.py:309-316): samples uniform in
                                   1 / depth.  opts->ray_start / ray_end then hold (float)(1 / ray_start), (float)(1 / ray_end) and
                                   depth_delta (float)(1 / (Sc - 1)): d = linspace(0, 1, Sc) + jitter * depth_delta;
                                   t = 1 / (ray_start * (1 - d) + ray_end * d).  Not with per-ray limits. */
#define P3D_FLAG_WEIGHTS_ONLY 65536 /* p3d_render_f32: a HINT that the caller reads out_wsum and out_depth only (the occlusion pass of
                                     * paste_front, training/triplane.py:565-578, reads `image_weights` of its second render and nothing
                                     * else).  Where the library has a weights-only instantiation (small launches in the tolerance mode at
                                     * 48 / 96 fine samples) the final pass decodes densities only: wsum / depth bit-identical to the full
                                     * launch's, out_feat / out_xyz LEFT UNWRITTEN (still valid pointers: elsewhere the hint is ignored and
                                     * the full kernel fills them). */
#define P3D_FLAG_SHARED_PLANES 64 /* planes holds ONE image [1][3][H][W][32] shared by all N batches of rays / points (many
                                    views of one subject in one launch; the reference would pass planes.expand(N, ...)) */

#define P3D_C 32        /* channels per plane (triplane_width, training/triplane.py:41) */
#define P3D_HID 64      /* OSGDecoder hidden_dim (training/triplane.py:519) */
#define P3D_MAX_S 192   /* max depth_resolution / depth_resolution_importance */

/* rendering_options + forward() arguments of ImportanceRenderer (renderer.py:162-174), already converted to binary32
 * by the host exactly as include/p3d_numerics.h states. */
typedef struct {
    float coord_scale;  /* (float)(2.0 / box_warp)                              renderer.py:77  */
    float ray_start;    /* rendering_options['ray_start']                       renderer.py:177 */
    float ray_end;      /* rendering_options['ray_end']                                          */
    float depth_delta;  /* (float)((ray_end - ray_start) / (Sc - 1)), in double renderer.py:323 */
    float crop_limit;   /* (float)(box_warp / 2 - triplane_crop)                renderer.py:139-142 */
    float cull_thresh;  /* cull_clouds or binarize_clouds value                 renderer.py:190-196 */
    int32_t Sc;         /* depth_resolution                                                      */
    int32_t Sf;         /* depth_resolution_importance (0: single pass, renderer.py:254-259)    */
    int32_t plane_mode; /* use_triplane: plane 2 = (y,z) if 1 else (z,x)        renderer.py:41-49 */
    int32_t flags;      /* P3D_FLAG_*                                                            */
} p3d_opts;

/* Optional per-stage dumps of p3d_render_f32 (parity tests).  Any pointer may be NULL. */
typedef struct {
    float* depths_coarse;   /* [N*R][Sc]      sample_stratified output                 */
    float* sigma_coarse;    /* [N*R][Sc]      after crop/cull masks                    */
    float* weights_coarse;  /* [N*R][Sc-1]    coarse ray-marcher weights               */
    float* depths_fine;     /* [N*R][Sf]      sample_importance output (draw order)    */
    int32_t* inds;          /* [N*R][Sf]      searchsorted(cdf, u, right=True)         */
    float* depths_sorted;   /* [N*R][Sc+Sf]   unify_samples depths                     */
    float* sigma_sorted;    /* [N*R][Sc+Sf]   unify_samples densities (after masks)    */
    float* depth_unclamped; /* [N*R]          composite depth before the global clamp  */
    float* tminmax;         /* [2]            global min / max of all depths           */
} p3d_dumps;

/* Layout change [n3][C][H][W] -> [n3][H][W][C] (n3 = N*3 planes).  The reference keeps planes NCHW
 * (training/triplane.py:200-206) and lets grid_sample gather 32 lines per tap (renderer.py:80); every kernel below
 * wants one 128-byte line per tap. */
int p3d_planes_to_nhwc_f32(const float* planes_nchw, int n3, int C, int H, int W, float* planes_nhwc, void* stream);

/* OSGDecoder.forward (training/triplane.py:528-544) on already sampled features — the decoder module's own call surface:
 * feats [N][3][M][32] (`sampled_features`, renderer.py:271-273), mean over the three planes, FC 32 -> 64, Softplus, FC 64 -> 33;
 * out_sigma [N][M] = row 0, out_rgb [N][M][32] = sigmoid(rows 1..32) (force_sigmoid != 0) or sigmoid * 1.002 - 0.001.  Weights
 * pre-scaled as for p3d_triplane_decode_f32; feats / out_rgb 16-byte aligned.  Same arithmetic contract as the fused kernels. */
int p3d_decode_features_f32(const float* feats, int N, int64_t M, const float* w0, const float* b0, const float* w1, const float* b1,
                            int force_sigmoid, float* out_sigma, float* out_rgb, void* stream);

/* ImportanceRenderer.run_model (renderer.py:266-280) = sample_from_planes (:68-81) + OSGDecoder.forward
 * (training/triplane.py:528-544) + crop/cull masks (renderer.py:138-153,187-198) on a point cloud.
 * planes_nhwc [N][3][H][W][32]; coords [N][M][3]; w0 [64][32], b0 [64], w1 [33][64], b1 [33] pre-scaled by the host
 * (networks_stylegan2.py:121-127).  out_sigma [N][M]; out_rgb [N][M][32] or NULL (density only: get_eg3d_volume,
 * _util/eg3d_metrics3d.py:140-150).  Uses coord_scale, plane_mode, flags, crop_limit, cull_thresh of opts. */
int p3d_triplane_decode_f32(const float* planes_nhwc, int N, int H, int W, const float* coords, int64_t M,
                            const float* w0, const float* b0, const float* w1, const float* b1, const p3d_opts* opts,
                            float* out_sigma, float* out_rgb, void* stream);

/* Density-only run_model on the reference's regular grid (create_samples + the chunk loop of get_eg3d_volume,
 * _util/eg3d_metrics3d.py:70-92,124-151) without materialising the points: flat grid indices [lo, hi) of a grid_n^3 grid;
 * column c of a point = index component * voxel_size + off_c (off0 = voxel_origin[2], off1 = [1], off2 = [0] there), computed
 * with the reference's float arithmetic.  planes_nhwc [1][3][H][W][32]; out_sigma [hi - lo]; out_cropmask [hi - lo] bytes or
 * NULL: 1 where |x| > mask_limit or |z| > mask_limit (triplane_crop_mask, renderer.py:138-149, which get_eg3d_volume applies
 * to the densities AFTER activation, eg3d_metrics3d.py:158-160 — so it is returned, not applied). */
int p3d_grid_density_f32(const float* planes_nhwc, int H, int W, int grid_n, int64_t lo, int64_t hi, float voxel_size,
                         float off0, float off1, float off2, const float* w0, const float* b0, const float* w1,
                         const float* b1, const p3d_opts* opts, float* out_sigma, unsigned char* out_cropmask,
                         float mask_limit, void* stream);

/* ImportanceRenderer.forward (renderer.py:162-264), fused: stratified depths, coarse density pass, ray-marcher weights,
 * importance resampling, depth merge, final decode + compositing.  rays_o/rays_d [N][R][3]; jitter [N][R][Sc] (the
 * torch.rand_like draw of renderer.py:324); u [N*R][Sf] (the torch.rand draw of :371; may be NULL when Sf == 0).
 * ray_tile_w: image width in rays if the R rays are a row-major ray_tile_w x (R/ray_tile_w) image (enables 8x4 screen
 * tiles per wavefront), 0 for an unstructured ray list.  Outputs feat [N][R][32], depth [N][R], wsum [N][R],
 * xyz [N][R][3].  workspace: p3d_render_workspace_bytes bytes of device memory: u32[0..1] = order-mapped global depth
 * min / max, u64 at byte 8 = number of wave-level decode steps the launch executed (statistics; 32 samples each), then one
 * min / max pair per view.  Depth clamp scope (ray_marcher.py:49-50): the whole call, like the reference's batch of N images;
 * with P3D_FLAG_PER_VIEW_CLAMP each image is clamped to its own range (N views batched by the caller = N calls of the reference). */
size_t p3d_render_workspace_bytes(int N, int64_t R, int Sc, int Sf);
int p3d_render_f32(const float* planes_nhwc, int N, int H, int W, const float* rays_o, const float* rays_d, int64_t R,
                   int ray_tile_w, const float* jitter, const float* u, const float* w0, const float* b0,
                   const float* w1, const float* b1, const p3d_opts* opts, float* out_feat, float* out_depth,
                   float* out_wsum, float* out_xyz, void* workspace, size_t workspace_bytes, const p3d_dumps* dumps,
                   void* stream);

/* The same with PER-RAY depth limits: rendering_options['ray_start'] == ['ray_end'] == 'auto' (renderer.py:165-171).  ray_start,
 * ray_end [N][R]: the box-intersection limits after the reference's patching of the rays that miss the box (the caller
 * computes them: cameras.ray_limits_box + renderer.py:167-170); the coarse depths are then math_utils.linspace (:101-118:
 * start + (i / (Sc - 1)) * (end - start), each operation rounded) + jitter * ((end - start) / (Sc - 1)), and
 * opts->ray_start / ray_end / depth_delta are not read.  Both NULL = p3d_render_f32. */
int p3d_render_limits_f32(const float* planes_nhwc, int N, int H, int W, const float* rays_o, const float* rays_d, int64_t R,
                          int ray_tile_w, const float* jitter, const float* u, const float* w0, const float* b0,
                          const float* w1, const float* b1, const float* ray_start, const float* ray_end, const p3d_opts* opts,
                          float* out_feat, float* out_depth, float* out_wsum, float* out_xyz, void* workspace,
                          size_t workspace_bytes, const p3d_dumps* dumps, void* stream);

/* The same with the two random draws made INSIDE the kernel (opt-in): instead of reading jitter [N][R][Sc] and u [N*R][Sf] — 200 MB
 * written by torch.rand and read back per 512^2 x (48+48) frame, 59 % of the launch's compulsory HBM bytes, and two extra launches —
 * every draw is a pure function of (seed, stream, flat ray index n*R + r, sample index): the counter-based generator of
 * include/p3d_numerics.h ("device draws"), 24 random bits in [0, 1) like torch.rand's float32 output.  Another random stream
 * than torch's, the same distribution; the CPU oracle restates the generator, so results stay bit-exact checkable.  ray_start /
 * ray_end as in p3d_render_limits_f32 (may both be NULL). */
int p3d_render_rng_f32(const float* planes_nhwc, int N, int H, int W, const float* rays_o, const float* rays_d, int64_t R,
                       int ray_tile_w, uint64_t seed, const float* w0, const float* b0, const float* w1, const float* b1,
                       const float* ray_start, const float* ray_end, const p3d_opts* opts, float* out_feat, float* out_depth,
                       float* out_wsum, float* out_xyz, void* workspace, size_t workspace_bytes, const p3d_dumps* dumps,
                       void* stream);

/* sample_stratified (renderer.py:303-326, numeric ray_start/ray_end branch).  jitter, out [NR][S]. */
int p3d_sample_stratified_f32(float ray_start, float ray_end, float depth_delta, int S, const float* jitter, int64_t NR,
                              float* out_depths, void* stream);

/* torch.min(depths), torch.max(depths) of ray_marcher.py:50 — the one cross-ray dependency of the path — as its own entry
 * point: depths [n] -> out_minmax [2] (device).  workspace: >= 16 bytes of device memory. */
int p3d_depth_minmax_f32(const float* depths, int64_t n, float* out_minmax, void* workspace, size_t workspace_bytes, void* stream);

/* MipRayMarcher2.run_forward (ray_marcher.py:25-57).  colors [NR][S][K], sigma [NR][S], depths [NR][S] ->
 * out_rgb [NR][K], out_depth [NR], out_weights [NR][S-1] (may be NULL).  workspace: p3d_composite_workspace_bytes bytes. */
size_t p3d_composite_workspace_bytes(int64_t NR, int S, int K);
int p3d_composite_f32(const float* colors, const float* sigma, const float* depths, int64_t NR, int S, int K,
                      int white_back, float* out_rgb, float* out_depth, float* out_weights, void* workspace,
                      void* stream);

/* sample_importance + sample_pdf (renderer.py:328-387).  depths [NR][Sc], weights [NR][Sc-1], u [NR][Sf] ->
 * out_depths [NR][Sf], out_inds [NR][Sf] (may be NULL). */
int p3d_importance_f32(const float* depths, const float* weights, int64_t NR, int Sc, int Sf, const float* u,
                       float* out_depths, int32_t* out_inds, void* stream);

/* unify_samples' sort (renderer.py:289-301): stable ascending permutation of [coarse ++ fine] depths. perm [NR][Sc+Sf]. */
int p3d_unify_perm_f32(const float* depths_coarse, const float* depths_fine, int64_t NR, int Sc, int Sf, int32_t* perm,
                       void* stream);

/* ---- StyleGAN2 synthesis operators of the triplane backbone (training/networks_stylegan2.py) ------------------------ */

/* modulated_conv2d (networks_stylegan2.py:40-97) + the bias_act that follows it in SynthesisLayer.forward (:350-352) /
 * ToRGBLayer.forward (:376-380), fused.  x [N][I][H][W]; w [O][I][ks][ks] (ks 3 or 1); styles [N][I];
 * demodulate: multiply the output by rsqrt(sum (w*s)^2 + 1e-8) (:70-73); noise: NULL, [OH*OW] (noise_const * strength)
 * or [N][OH*OW] (noise_per_sample = 1), added before the bias (:95-96); bias [O] or NULL; up 1 or 2 (up = 2: stride-2
 * transposed conv + 4x4 FIR `fir` = setup_filter([1,3,3,1]) flipped and multiplied by up^2, conv2d_resample.py:114-128);
 * act 0 linear / 1 lrelu(alpha); then *gain and clamp (< 0: none) as bias_act.py:93-122.  y [N][O][H*up][W*up].
 * demod_coefs: NULL, or the [N][O] coefficients already computed by p3d_demod_coefs_f32 (then no per-call reduction over the
 * weights is launched); ignored unless demodulate.
 * Alignment: x, w, styles, noise, bias and y need only their natural 4-byte alignment (an offset view is fine); when up = 2 and
 * y / noise happen to be 16-byte aligned the FIR pass stores four outputs at a time, otherwise one by one — same values.
 * The WORKSPACE must start on a 256-byte boundary (hipMalloc and torch allocations do; a sub-allocator must keep it): its regions are
 * carved by rounding up from the base, and p3d_modconv2d_workspace_bytes budgets for an aligned base (P3D_E_RANGE otherwise).  The
 * query's answer for a shape is fixed for the life of the process (the environment switches that select kernels are read once). */
size_t p3d_modconv2d_workspace_bytes(int N, int I, int O, int H, int W, int up);
int p3d_modconv2d_f32(const float* x, int N, int I, int H, int W, const float* w, int O, int ks, const float* styles,
                      int demodulate, const float* demod_coefs, const float* noise, int noise_per_sample, const float* bias,
                      int up, int act, float alpha, float gain, float clamp, const float* fir, float* y, void* workspace,
                      size_t workspace_bytes, void* stream);

/* The demodulation coefficients (networks_stylegan2.py:70-73) of L modulated convolutions in one launch:
 * d[n][o] = rsqrt(sum_i W2[o][i] * styles[n][i]^2 + 1e-8) with W2[o][i] = sum over the taps of w[o][i][.]^2, which the caller
 * caches per layer (it changes only with the weights).  w2 / styles / d hold the layers' [O][I] / [N][I] / [N][O] blocks back to
 * back; table (DEVICE int32 [L][6]) = {w2 offset, styles offset, d offset, O, I, first wave} per layer, where a layer owns N*O
 * waves and total_waves is their sum.  L <= 64. */
int p3d_demod_coefs_f32(const float* w2, const float* styles, const int32_t* table, int L, int N, int total_waves, float* d,
                        void* stream);

/* f16-operand variant of p3d_modconv2d_f32 (opt-in; the reference runs its super-resolution blocks in fp16 on the GPU:
 * superresolution.py:264-293 with sr_num_fp16_res = 4, networks_stylegan2.py:52-60).  Same arguments and semantics; x, y and
 * the accumulation stay fp32, the two MFMA operands (the modulated input s*x and the weights) are rounded to f16 (RNE),
 * the demodulation coefficients come from the fp32 weights `w`.  w_f16: [O][ks*ks][I] f16 made ONCE per layer by
 * p3d_conv_weights_to_f16 (16-byte aligned).  Requires I % 16 == 0 (P3D_E_RANGE otherwise: use the fp32 function). */
int p3d_conv_weights_to_f16(const float* w, int O, int I, int ks, void* w_f16, void* stream);
int p3d_modconv2d_f16mma_f32(const float* x, int N, int I, int H, int W, const float* w, const void* w_f16, int O, int ks,
                             const float* styles, int demodulate, const float* demod_coefs, const float* noise, int noise_per_sample,
                             const float* bias, int up, int act, float alpha, float gain, float clamp, const float* fir, float* y,
                             void* workspace, size_t workspace_bytes, void* stream);

/* Two-term f16 operands: every MFMA operand is carried as hi + lo (hi = f16(v) RNE, lo = f16(v - hi)) and a product is
 * a_hi*b_hi + a_lo*b_hi + a_hi*b_lo, accumulated in fp32 — fp32-class results (what is dropped is ~2^-22 relative, the size of
 * the fp32 kernel's own accumulation-order noise; measured in tests/test_hip_synthesis.py) on the f16 matrix cores: 3 MFMAs of
 * 32 cycles per 16 channels instead of 8 of 64.  w_f16x2: 2 x [O][ks*ks][I] f16, the hi parts followed by the lo parts, made
 * once per layer by p3d_conv_weights_to_f16x2 (16-byte aligned, O*I*ks*ks*2 bytes a multiple of 16; the weights are stored
 * scaled by 2^6, the kernel scales s*x by 2^4 and the accumulators back by 2^-10, because the matrix cores flush f16
 * subnormals).  Domain: |s*x| <= 4094 (= 65504 / 2^4), |w| < 1023: a modulated activation beyond that (or NaN) is clamped to the
 * f16 range — finite, wrong — and reported through `saturated`: a CALLER-OWNED device word (or null: not reported) that the
 * kernels OR with 1, stream-ordered like every other output; the caller zeroes it and reads it when it wants to know (ABI 5:
 * the library keeps no flag of its own — "no global mutable state").  Same arguments and semantics as
 * p3d_modconv2d_f32 otherwise; I % 16 == 0. */
int p3d_conv_weights_to_f16x2(const float* w, int O, int I, int ks, void* w_f16x2, void* stream);
int p3d_modconv2d_f16x2mma_f32(const float* x, int N, int I, int H, int W, const float* w, const void* w_f16x2, int O, int ks,
                               const float* styles, int demodulate, const float* demod_coefs, const float* noise, int noise_per_sample,
                               const float* bias, int up, int act, float alpha, float gain, float clamp, const float* fir, float* y,
                               void* workspace, size_t workspace_bytes, uint32_t* saturated, void* stream);

/* The same operator with every argument in one POD (what the Python host calls: one pointer through the FFI instead of 26
 * scalars).  mma: P3D_CONV_MMA_F32 (w_f16 unused), _F16 (w_f16 from p3d_conv_weights_to_f16) or _F16X2 (from
 * p3d_conv_weights_to_f16x2; `saturated` as in p3d_modconv2d_f16x2mma_f32, may be null). */
#define P3D_CONV_MMA_F32 0
#define P3D_CONV_MMA_F16 1
#define P3D_CONV_MMA_F16X2 2
typedef struct p3d_conv_args {
    const float* x;            /* [N][I][H][W] */
    const float* w;            /* [O][I][ks][ks] */
    const void* w_f16;         /* f16 operand copy of w, or null (mma = F32) */
    const float* styles;       /* [N][I] */
    const float* demod_coefs;  /* [N][O] from p3d_demod_coefs_f32, or null */
    const float* noise;        /* null, [OH*OW] or [N][OH*OW] */
    const float* bias;         /* [O] or null */
    const float* fir;          /* up = 2: the 4x4 filter, flipped, times up^2 */
    float* y;                  /* [N][O][H*up][W*up] */
    void* workspace;           /* p3d_modconv2d_workspace_bytes(N, I, O, H, W, up) */
    uint32_t* saturated;       /* see p3d_modconv2d_f16x2mma_f32, or null */
    const void* x_img;         /* the input as an activation image (already modulated by its producer), or null: then x + styles */
    void* y_img;               /* the output as an activation image for the layer that follows, or null: up = 2 INSTEAD of y (y null), up = 1 NEXT TO y */
    const float* y_img_styles; /* with y_img: that layer's styles [N][O] */
    const float* rgb_w;        /* ABI 9, all three null or all set (only where p3d_conv_fuses_torgb says 1): the block's ToRGB weights [rgb_channels][O] ... */
    const float* rgb_styles;   /* ... its styles [N][O], already multiplied by the layer's weight_gain (networks_stylegan2.py:377) ... */
    float* rgb_partial;        /* ... and p3d_torgb_partial_bytes(N, O, H, W, rgb_channels) bytes that receive the ToRGB sums of this layer's result, one
                                * share per 64-channel tile [O/64][N][rgb_channels][H][W]; p3d_torgb_combine_f32 finishes the layer.  y may then be null. */
    size_t workspace_bytes;
    int32_t N, I, H, W, O, ks, up, demodulate, noise_per_sample, act, mma;
    float alpha, gain, clamp;
    int32_t rgb_channels;      /* ToRGB output channels (1 .. 4) with rgb_partial */
    int32_t w_f16_layout;      /* P3D_WLAYOUT_* of w_f16 (mma = F16X2, ks = 3); 0 = P3D_WLAYOUT_OIK */
} p3d_conv_args;
int p3d_modconv2d_ex_f32(const p3d_conv_args* args, void* stream);

/* ToRGBLayer.forward (networks_stylegan2.py:366-380) of a block with <= 4 image channels, riding on the block's conv1 launch (the
 * super-resolution blocks: 3 channels): conv1's epilogue holds the finished activation in registers, multiplies it by the ToRGB styles
 * (the reference's own rounding of the modulated input) and by the 1x1 weights and stores, per 64-channel tile, that tile's share of the
 * sum — the 134 MB activation of the 512^2 block is never read back (k_torgb: 44 us) and, when nobody else reads it, never written.
 * p3d_conv_fuses_torgb: 1 exactly when p3d_modconv2d_ex_f32 accepts rgb_* for a plain 3x3 layer of that shape (two-term operands, an
 * activation image as input, the pipelined kernel unsplit); p3d_torgb_combine_f32 adds the shares in tile order, the bias, the clamp
 * and the up-sampled skip image (p3d_torgb_f32's epilogue: same taps, same order): y [N][O][H][W].  The channel sum runs in another
 * order than p3d_torgb_f32's (fp32-class agreement, not bit equality). */
int p3d_conv_fuses_torgb(int N, int I, int O, int H, int W, int rgb_channels);

/* Layouts of the two-term f16 copy of a 3x3 layer's weights (ABI 9).  P3D_WLAYOUT_OIK: [hi | lo][O][9][I] (p3d_conv_weights_to_f16x2;
 * every kernel reads it).  The pipelined kernels stage a 16-channel chunk of weights with buffer_load ... lds; out of the OIK layout a
 * request gathers 64 16-byte pieces 9 * I * 2 bytes apart — 64 cache lines, each shared with other slices' workgroups.  The two image
 * layouts store, per (chunk, channel tile), exactly the bytes the consuming kernel keeps in LDS, consecutively, so a request is 1 KB of
 * consecutive memory (measured: -10 % on the 256 -> 256 @256^2 layer, -5 .. -20 % on the smaller ones; bit-identical results):
 *   P3D_WLAYOUT_PLAIN  [I/16][O/64][dx 3][hi | lo][dy 3][k half 2][64 o][8]   plain 3x3 layers on k_modconv_w3 (O % 64 == 0)
 *   P3D_WLAYOUT_UP     [I/16][O/32][hi | lo][tap 9][k half 2][32 o][8]        up-sampling layers on k_modconv_up3 / _up4 (O % 32 == 0)
 * p3d_conv_weight_layout(I, O, W, up): the layout the library wants for a layer of that shape (W: the INPUT map's width) — the one its
 * dispatch can consume; a copy in another layout than OIK handed to a layer that does not run on the matching kernel is P3D_E_RANGE.
 * p3d_conv_weights_to_f16x2_layout converts into any of the three (same bytes, another order; same size). */
#define P3D_WLAYOUT_OIK 0
#define P3D_WLAYOUT_PLAIN 1
#define P3D_WLAYOUT_UP 2
int p3d_conv_weight_layout(int I, int O, int W, int up);
int p3d_conv_weights_to_f16x2_layout(const float* w, int O, int I, int ks, int layout, void* w_f16x2, void* stream);
size_t p3d_torgb_partial_bytes(int N, int O, int H, int W, int rgb_channels);
int p3d_torgb_combine_f32(const float* partial, int tiles, int N, int O, int H, int W, const float* bias, float clamp, const float* skip,
                          const float* skip_fir, float* y, void* stream);

/* The activation IMAGE of the two-term convolution path (csrc/p3d_synthesis.hip, "activation IMAGE"): the operand of a plain 3x3
 * layer prepared by the layer in front of it — 16-byte pieces of f16 hi parts and of lo parts of 16 * s[n][c] * x[n][c][y][x], laid out
 * [hi | lo][N][C/8][H][W][8]: p3d_act_image_bytes(N, C, H, W) = N*C*H*W*4 bytes, 16-byte aligned, C % 8 == 0.  An up-sampling layer
 * writes it from its FIR + bias_act pass when p3d_conv_args.y_img / y_img_styles (the NEXT layer's styles) are set and y is null; a
 * plain 3x3 layer (up = 1) given y_img writes it NEXT TO y (from the convolution's own epilogue where the pipelined kernel runs
 * unsplit, by one more pass otherwise) — the up-sampling layer of the next block reads it, ToRGB reads y.  Consumers (two-term
 * operands, demod_coefs given, x and styles unused; P3D_E_RANGE otherwise), both staging with buffer_load ... lds alone:
 *   a plain 3x3 layer (up = 1)   with I % 16 == 0 and W >= 32;
 *   an up-sampling layer (up = 2) with I % 16 == 0, O % 32 == 0 and W >= P3D_UP3_MIN_W (4; the environment variable of that name
 *                                 overrides it, P3D_NO_UP3 in the environment disables the kernel — both read once per process).
 * p3d_conv_takes_image(I, O, W, up) returns 1 exactly when the library accepts x_img for that layer: bind THAT, not a copy of the rule.  The pieces are bit for bit what the layer
 * computes itself from the fp32 tensor, so results do not change.  p3d_act_to_image_f32 builds one from an fp32 tensor (styles may
 * be null = 1).  |16 s x| > 65504: clamped, *saturated |= 1, as in p3d_modconv2d_f16x2mma_f32. */
int p3d_conv_takes_image(int I, int O, int W, int up);
size_t p3d_act_image_bytes(int N, int C, int H, int W);
int p3d_act_to_image_f32(const float* x, const float* styles, int N, int C, int H, int W, void* img, uint32_t* saturated, void* stream);
/* (ABI 9) the same fused with the in-place conditioning add that PAniC-3D's SynthesisNetwork.forward applies to x between two blocks
 * (networks_stylegan2.py:554-560 the resnet "chonk" on the first channels of the 8x8 map, :600-622 `add_4` / `add_shuffle2_4` on the last
 * quarter): x[:, c0 : c0 + Ca] += add (add [Na][Ca][H][W], Na = 1 or N; c0, Ca multiples of 8), written back to x, and the image of the
 * updated x — one launch instead of an elementwise add and a conversion pass; the same bits as the two. */
int p3d_act_to_image_add_f32(float* x, const float* styles, int N, int C, int H, int W, const float* add, int c0, int Ca, int Na, void* img,
                             uint32_t* saturated, void* stream);

/* ToRGBLayer.forward (networks_stylegan2.py:366-380: 1x1 modulated convolution without demodulation + bias [+ clamp]) fused with
 * the skip connection of SynthesisBlock.forward (:476-478: img = upsample2d(img) + y) — ONE launch that reads the activation
 * once: a [O x I] x [I x H*W] GEMM on v_mfma_f32_32x32x2_f32 (exact fp32) whose B operand goes from global memory straight into
 * MFMA registers.  x [N][I][H][W]; w_t = the layer's weights [O][I] transposed and padded by p3d_torgb_weights_f32 (once per
 * layer: [I][32] for O <= 32, [I][96] for O <= 96; O > 96: P3D_E_RANGE, use p3d_modconv2d_f32); styles [N][I] already
 * multiplied by the layer's weight_gain; bias [O] or null; clamp < 0: none; skip [N][O][H/2][W/2] + skip_fir (the 4x4 resample
 * filter flipped and multiplied by 4, upfirdn2d.py:341-350) or both null; y [N][O][H][W].  Without skip the result equals
 * p3d_modconv2d_f32(ks = 1, demodulate = 0) bit for bit wherever that runs without split-K; the skip term uses
 * p3d_upfirdn2d_f32's summation order (bit-identical to the three-launch form). */
int p3d_torgb_weights_f32(const float* w, int O, int I, float* w_t, void* stream);
int p3d_torgb_f32(const float* x, int N, int I, int H, int W, const float* w_t, int O, const float* styles, const float* bias, float clamp,
                  const float* skip, const float* skip_fir, float* y, void* stream);

/* upfirdn2d (torch_utils/ops/upfirdn2d.py:120-167; plugin signature upfirdn2d.cpp:20): zero-insert by `up`, pad/crop,
 * correlate with f [fh][fw] (pass the filter already flipped for convolution and multiplied by the gain), decimate by
 * `down`.  x [NC][H][W] -> y [NC][(H*up+pady0+pady1-fh)/down+1][(W*up+padx0+padx1-fw)/down+1]. */
int p3d_upfirdn2d_f32(const float* x, int64_t NC, int H, int W, const float* f, int fh, int fw, int up, int down, int padx0,
                      int padx1, int pady0, int pady1, float* y, void* stream);

/* upsample2d of the skip image fused with the ToRGB accumulation of SynthesisBlock.forward (networks_stylegan2.py:476-478:
 * `img = upsample2d(img, resample_filter); img = img.add_(y)`): y = upfirdn2d(x, up = 2, pad [2,1,2,1], f4x4) + add, the FIR in
 * polyphase form with the generic operator's summation order (bit-identical to p3d_upfirdn2d_f32).  x [NC][H][W];
 * f4x4: the 4x4 filter already flipped and multiplied by up^2; add [NC][2H][2W] or NULL; y [NC][2H][2W].  Needs 2W % 4 == 0 and
 * 16-byte aligned add / y (P3D_E_RANGE otherwise: use p3d_upfirdn2d_f32). */
int p3d_upsample2d_add_f32(const float* x, int64_t NC, int H, int W, const float* f4x4, const float* add, float* y, void* stream);

/* bias_act (torch_utils/ops/bias_act.py:54-88; plugin signature bias_act.cpp:36): y = clamp(act(x + b[c]) * gain), x viewed
 * as [outer][C][inner]; act 0 linear / 1 lrelu(alpha); b may be NULL; clamp < 0: none. */
int p3d_bias_act_f32(const float* x, const float* b, int64_t outer, int C, int64_t inner, int act, float alpha, float gain,
                     float clamp, float* y, void* stream);

/* sigma -> density of get_eg3d_volume with its two masks, one pass (_util/eg3d_metrics3d.py:65-69,153-163):
 * d = 1 - exp(-softplus(sigma - 1)); cropmask[m] != 0 -> -1000; cull_thresh >= 0: the reference's cull mask is evaluated on the
 * DENSITIES (sic): 1 - exp(-softplus(d - 1)) < cull_thresh -> -1000.  cropmask may be NULL; cull_thresh < 0: no cull.
 * May run in place (out_density == sigma). */
int p3d_sigma2density_f32(const float* sigma, const unsigned char* cropmask, int64_t M, float cull_thresh, float* out_density,
                          void* stream);

/* ---- front-view paste --------------------------------------------------------------------------------- */

/* paste_front (training/triplane.py:607-691, mode 'default', front_weight_erosion = 0, force_image = None — what
 * _scripts/eval/generate.py:55-66 uses) in ONE launch: the four masks (bilinear / nearest F.interpolate to the illustration's
 * size, kornia.filters.sobel restated, thresholds), sample_orthofront's grid_sample of the illustration (:555-564) and the final
 * torch.lerp.  Render-resolution inputs [N][.][r][r]: weights (image_weights), xyz (image_xyz), occ (image_weights of the
 * front-occlusion pass, :565-578), rays_o / rays_d (x['force_rays']); front [N or 1][3][S][S] (cond['image_ortho_front'], in [0,1];
 * front_shared != 0: one illustration for all N views), image [N][3][S][S] (the super-resolved image).  Outputs at [N][.][S][S]:
 * out_image [3] (= torch.lerp(image, paste, mask)), out_paste [3], out_mask, out_mask_weights, out_mask_edges, out_mask_occ,
 * out_mask_dxyz [1].  out_image may alias image. */
typedef struct {
    const float *weights, *xyz, *occ, *rays_o, *rays_d, *front, *image;
    float *out_image, *out_paste, *out_mask, *out_mask_weights, *out_mask_edges, *out_mask_occ, *out_mask_dxyz;
    int32_t N, r, S, front_shared, normalize_images;
    float thresh_weight, thresh_edges, thresh_occ, thresh_dxyz, box_warp;
} p3d_paste_args;
int p3d_paste_front_f32(const p3d_paste_args* args, void* stream);

/* ---- iso-surface of the density grid on the device ---------------------------------------------------- */

/* Replaces skimage.measure.marching_cubes(vol, level, spacing=(1,1,1), gradient_direction='descent', method='lewiner') as
 * called by _util/eg3d_metrics3d.py:186-210 (generate.py:98-103).  Specification: oracle/p3d_oracle_mc.c (triangulation of
 * ambiguous cubes, vertex / face order and degenerate handling differ from Lewiner's; see that header), case table
 * include/p3d_mc_table.h.  vol [n][n][n] f32 in (axis0, axis1, axis2) order; flip0 != 0 reads axis 0 reversed, i.e. takes
 * the flat `densities` of p3d_grid_density_f32 directly (the reference flips that axis, eg3d_metrics3d.py:166-168).
 * Two calls because the caller owns the output buffers:
 *   p3d_mc_count_f32  classifies + scans; out_counts (DEVICE uint64[2]) = {#vertices, #triangles}
 *   p3d_mc_emit_f32   with the SAME vol / n / flip0 / level / workspace: out_verts [V][3] (index space), out_normals [V][3]
 *                     (unit, towards lower values), out_values [V] (max of the edge's end points), out_faces [F][3] int32.
 * workspace: 16-byte aligned, p3d_mc_workspace_bytes(n) bytes (4 B per grid point + block sums); 2 <= n <= 1024. */
size_t p3d_mc_workspace_bytes(int n);
int p3d_mc_count_f32(const float* vol, int n, int flip0, float level, void* workspace, size_t workspace_bytes,
                     uint64_t* out_counts, void* stream);
int p3d_mc_emit_f32(const float* vol, int n, int flip0, float level, void* workspace, size_t workspace_bytes, int64_t nverts,
                    int64_t ntris, float* out_verts, float* out_normals, float* out_values, int32_t* out_faces, void* stream);

/* Library / build identification ("gfx950"), and the ABI version this library was built with (P3D_ABI_VERSION). */
const char* p3d_build_info(void);
int p3d_abi_version(void);

/* The layout of the four POD structs above AS THIS LIBRARY WAS COMPILED, for bindings that restate them in another language
 * (ctypes, cgo, JNA ...): which = P3D_STRUCT_OPTS / _DUMPS / _PASTE_ARGS / _CONV_ARGS; out[0] = sizeof, out[1 + i] = offsetof
 * field i in declaration order; returns the number of entries (1 + #fields) or P3D_E_ARG / P3D_E_RANGE (cap too small, unknown
 * struct).  A binding compares them with its own mirror once at load time (the in-tree one does: _lib.py) — the plugin seam of the
 * reference validates its arguments with TORCH_CHECK (torch_utils/ops/bias_act.cpp:36-101); a struct passed through a foreign
 * FFI has no such check unless the library offers one. */
#define P3D_STRUCT_OPTS 0
#define P3D_STRUCT_DUMPS 1
#define P3D_STRUCT_PASTE_ARGS 2
#define P3D_STRUCT_CONV_ARGS 3
int p3d_struct_layout(int which, size_t* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
