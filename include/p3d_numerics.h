/* p3d_numerics.h — the ARITHMETIC CONTRACT of libpanic3d_hip.
 *
 * Every floating-point result the library returns is defined as a fixed sequence of IEEE-754
 * binary32 operations (+, -, *, /, fma, floor, rint; round-to-nearest-even; subnormals kept; no
 * contraction, no reassociation, no hardware transcendental approximations) plus a few binary64
 * accumulations where the reference's own CPU kernels accumulate in double.  The sequence is
 * written out below.  The HIP kernels (panic3d-anime-reconstruction_amd/csrc) implement it on
 * gfx950; the CPU oracle (oracle/p3d_oracle.c) restates it independently in plain C.  Because both
 * sides follow the same contract the parity tests can demand BIT-EXACT agreement — including the
 * inverse-CDF bin indices and the depth-sort permutation ("ray hit indices") — instead of a
 * tolerance.  Agreement of the contract with the reference's PyTorch code is a tolerance question
 * (different summation orders / libm) and is pinned by tests/golden/.
 *
 * This header holds only constants and documentation (no code is shared between the product and
 * its checker).  Constants are produced by tools/gen_numerics.py.
 *
 * Reference code the contract restates (all under /root/reference/_train/eg3dc/src/training):
 *   volumetric_rendering/renderer.py:52-81   project_onto_planes / sample_from_planes
 *   triplane.py:516-544                      OSGDecoder
 *   networks_stylegan2.py:102-133            FullyConnectedLayer
 *   volumetric_rendering/renderer.py:138-153 crop / cull masks
 *   volumetric_rendering/ray_marcher.py:25-57 MipRayMarcher2.run_forward
 *   volumetric_rendering/renderer.py:303-387 sample_stratified / sample_importance / sample_pdf
 *   volumetric_rendering/renderer.py:289-301 unify_samples
 */
#ifndef P3D_NUMERICS_H
#define P3D_NUMERICS_H

/* ---- p3d_exp(x) ---------------------------------------------------------------------------
 *   x <  P3D_EXP_LO   -> 0          (the ldexp below would give 0 anyway; stated so that (int)n never overflows)
 *   x >  P3D_EXP_HI   -> +inf
 *   n = rint(x * P3D_LOG2E);  r = fma(n, -P3D_LN2_HI, x);  r = fma(n, -P3D_LN2_LO, r);
 *   p = C6; p = fma(p,r,C5); ... p = fma(p,r,C0);   result = ldexp(p, (int)n)
 *   (ldexp = exact scaling by 2^n with IEEE round-to-nearest-even into the subnormal range: C ldexpf, v_ldexp_f32)
 *   NaN in -> NaN out.   max error ~1.2 ulp. */
#define P3D_EXP_LO   (-200.0f)
#define P3D_EXP_HI   (88.0f)
#define P3D_LOG2E    0x1.715476p+0f   /* 1.44269502 */
#define P3D_LN2_HI   0x1.63p-1f       /* 0.693359375 */
#define P3D_LN2_LO   (-0x1.bd0106p-13f) /* -2.12194442e-4 ; ln2 = LN2_HI + LN2_LO */
#define P3D_EXP_C0 0x1.000000p+0f   /* 1 */
#define P3D_EXP_C1 0x1.000000p+0f   /* 1 */
#define P3D_EXP_C2 0x1.000000p-1f   /* 0.5 */
#define P3D_EXP_C3 0x1.5554dcp-3f   /* 0.166665763 */
#define P3D_EXP_C4 0x1.5554e8p-5f   /* 0.041666463 */
#define P3D_EXP_C5 0x1.120b74p-7f   /* 0.00836318173 */
#define P3D_EXP_C6 0x1.6d4332p-10f  /* 0.00139336579 */

/* ---- p3d_log1p01(z), 0 <= z <= 1 -------------------------------------------------------------
 *   q = L8; q = fma(q,z,L7); ... q = fma(q,z,L0);  result = q * z          (abs error ~1e-7) */
#define P3D_L1P_C0 0x1.000000p+0f    /* 1 */
#define P3D_L1P_C1 (-0x1.fffeb2p-2f) /* -0.499995023 */
#define P3D_L1P_C2 0x1.553078p-2f    /* 0.333192706 */
#define P3D_L1P_C3 (-0x1.fcd00cp-3f) /* -0.248443693 */
#define P3D_L1P_C4 0x1.8766f0p-3f    /* 0.191114306 */
#define P3D_L1P_C5 (-0x1.180f2ep-3f) /* -0.136747703 */
#define P3D_L1P_C6 0x1.40f82ap-4f    /* 0.0783616677 */
#define P3D_L1P_C7 (-0x1.e4c732p-6f) /* -0.0295885075 */
#define P3D_L1P_C8 0x1.584a66p-8f    /* 0.00525345793 */

/* ---- p3d_exp_nonpos(x)  (arguments known to be <= 0: softplus / sigmoid / cull paths)
 *   p3d_exp_nonpos(x) = the p3d_exp recipe applied to max(x, P3D_EXP_LO)   (fmaxf; the ldexp underflows to exactly 0 there,
 *   so this equals p3d_exp(x) for every x <= 0 including -inf)
 * ---- p3d_softplus(x)  (torch Softplus beta=1 threshold=20: triplane.py:524, ray_marcher.py:33)
 *   z = p3d_exp_nonpos(-|x|);  result = max(x, 0) + p3d_log1p01(z)
 *   (torch returns x itself above the threshold 20; the expression above already equals x there in binary32, because
 *    log1p(exp(-20)) = 2.1e-9 is far below half an ulp of 20 — so no select is needed and none is specified)
 * ---- p3d_rcp12(d), 1 <= d <= 2 : reciprocal by a fixed Newton sequence (cheaper than IEEE division, error <= ~1 ulp)
 *   r = fma(d, -P3D_RCP_A, P3D_RCP_B);   then three times:  e = fma(-d, r, 1);  r = fma(r, e, r)
 * ---- p3d_sigmoid(x)   (triplane.py:540)
 *   z = p3d_exp_nonpos(-|x|);  r = p3d_rcp12(1 + z);  result = (x >= 0) ? r : z * r */
#define P3D_RCP_A 0x1.e1e1e2p-2f /* 8/17  */
#define P3D_RCP_B 0x1.696968p+0f /* 24/17 */
#define P3D_SOFTPLUS_THRESHOLD 20.0f

/* ---- triplane sample (renderer.py:68-81; grid_sample bilinear / zeros / align_corners=False) ----
 *   q = p * coord_scale                      coord_scale = (float)(2.0 / box_warp), host-computed
 *   plane 0: (gx,gy) = (q.x,q.y)   plane 1: (q.x,q.z)   plane 2: (q.y,q.z) if plane_mode==1 else (q.z,q.x)
 *   ix = (gx + 1) * (W/2) - 0.5 ;  iy = (gy + 1) * (H/2) - 0.5          (three separate roundings)
 *   if !(ix > -1 && ix < W && iy > -1 && iy < H) : plane feature = 0 for all channels
 *   x0 = floor(ix); y0 = floor(iy); wx1 = ix - x0; wx0 = 1 - wx1; wy1 = iy - y0; wy0 = 1 - wy1
 *   nw = wy0*wx0; ne = wy0*wx1; sw = wy1*wx0; se = wy1*wx1
 *   tap value v(y,x) = plane[c][y][x] if 0<=x<W && 0<=y<H else 0
 *   f = nw*v(y0,x0); f = fma(ne, v(y0,x0+1), f); f = fma(sw, v(y0+1,x0), f); f = fma(se, v(y0+1,x0+1), f)
 *   X[c] = ((f_plane0[c] + f_plane1[c]) + f_plane2[c]) * P3D_THIRD           (mean over planes)
 *
 * ---- decoder (triplane.py:528-544; weights pre-scaled by the host: w*weight_gain, b*bias_gain) ----
 * Accumulation orders are those of a v_mfma_f32_32x32x2_f32 chain with the operands swapped
 * (D = W * X^T, samples on the N axis), which is bitwise an fma chain in k order:
 *   hidden n (0..63):  a = b0[n]; for s in 0..15 { a = fma(w0[n][s], X[s], a); a = fma(w0[n][16+s], X[16+s], a); }
 *                      h[n] = p3d_softplus(a)
 *   with nlo(t,s) = 32*t + (s&3) + 8*(s>>2),  nhi = nlo + 4   (t in 0..1, s in 0..15):
 *   rgb o (1..32):     a = b1[o]; for t,s { a = fma(w1[o][nlo], h[nlo], a); a = fma(w1[o][nhi], h[nhi], a); }
 *                      rgb[o-1] = p3d_sigmoid(a)              (force_sigmoid)
 *                               = p3d_sigmoid(a)*1.002f - 0.001f   otherwise (two roundings)
 *   sigma (o = 0):     alo = b1[0]; ahi = 0; for t,s { alo = fma(w1[0][nlo], h[nlo], alo); ahi = fma(w1[0][nhi], h[nhi], ahi); }
 *                      sigma = alo + ahi
 *
 * ---- masks on raw sigma (renderer.py:138-153,187-198) ----
 *   crop:  |p.x| > crop_limit || |p.z| > crop_limit -> sigma = -1000      crop_limit = (float)(box_warp/2 - triplane_crop)
 *   a = 1 - p3d_exp(-p3d_softplus(sigma - 1));  binarize: sigma = (a < thr) ? -1000 : 1000 ; cull: a < thr -> sigma = -1000
 */
#define P3D_THIRD 0x1.555556p-2f /* (float)(1.0/3.0) */
#define P3D_SIGMA_MASKED (-1000.0f)
#define P3D_SIGMA_SOLID  (1000.0f)

/* ---- stratified depths (renderer.py:303-326, numeric ray_start/ray_end branch) ----
 *   step = (end - start) / (S-1)  in binary32;   torch.linspace (CPU) is symmetric and fused:
 *   lin[i] = fma(step, (float)i, start)  for i <  S/2 ;   lin[i] = fma(-step, (float)(S-1-i), end)  for i >= S/2
 *   delta = (float)((double)(end - start... as python floats) / (S-1))
 *   t[i] = lin[i] + jitter[i] * delta                     (two roundings)
 *
 * ---- device draws (p3d_render_rng_f32; opt-in replacement of the two torch.rand tensors) ----
 *   fmix32(h): h ^= h >> 16; h *= 0x85EBCA6B; h ^= h >> 13; h *= 0xC2B2AE35; h ^= h >> 16      (32-bit unsigned arithmetic)
 *   draw(seed, stream, ray, i):  ray = n * R + r (64-bit), stream 0 = jitter (renderer.py:324), 1 = u (:371), i = sample index
 *     h = fmix32((uint32)seed ^ ((uint32)ray * 0x9E3779B1))
 *     h = fmix32(h ^ (uint32)(seed >> 32) ^ ((uint32)(ray >> 32) * 0x7FEB352D) ^ (((uint32)i * 2 + stream) * 0x846CA68B))
 *     value = (float)(h >> 8) * 2^-24          (24 random bits, [0, 1), exact in binary32 — torch.rand's float32 grid)
 *
 * ---- compositing (ray_marcher.py:25-57) over S sorted samples, K colour channels ----
 *   for i in 0..S-2:
 *     dl = t[i+1] - t[i];  sm = (sg[i] + sg[i+1]) * 0.5f;  tm = (t[i] + t[i+1]) * 0.5f;  cm[k] = (c[i][k] + c[i+1][k]) * 0.5f
 *     rho = p3d_softplus(sm - 1);  alpha = 1 - p3d_exp(-(rho * dl))
 *     T = (float)Td;  w[i] = alpha * T;  Td = Td * (double)((1 - alpha) + 1e-10f)       (Td binary64, starts at 1.0:
 *                                                                                        torch.cumprod accumulates in double)
 *     C[k] = fma(w[i], cm[k], C[k]);  Wsum = Wsum + w[i];  Dsum = fma(w[i], tm, Dsum)
 *   depth = Dsum / Wsum; NaN -> +inf; clamp to [min t, max t] over ALL depths of the call
 *   white_back: C[k] = (C[k] + 1) - Wsum;   C[k] = C[k]*2 - 1 (fma(C,2,-1) is the same value: *2 is exact)
 *
 * ---- importance resampling (renderer.py:328-387), per ray, L = Sc-1 weights ----
 *   m[0] = w[0]; m[j] = max(w[j-1], w[j]) (1<=j<=L-1); m[L] = w[L-1]
 *   ws[j] = (m[j] + m[j+1]) * 0.5f + 0.01f  (j in 0..L-1);   bins b[j] = 0.5f * (t[j] + t[j+1])
 *   v[j] = ws[j+1] + 1e-5f  (j in 0..Ns-1, Ns = Sc-3);  sum = (float)(sum_j (double)v[j])  (in index order)
 *   pdf[j] = v[j] / sum;   cdf[0] = 0;  cdf[j+1] = (float)(acc += (double)pdf[j])    (torch.cumsum accumulates in double)
 *   k = #{ j in 0..Ns : cdf[j] <= u }          (searchsorted right=True)
 *   below = max(k-1,0); above = min(k,Ns); den = cdf[above]-cdf[below]; if den < 1e-5f: den = 1
 *   t_f = b[below] + ((u - cdf[below]) / den) * (b[above] - b[below])
 *
 * ---- merge (renderer.py:289-301) ----
 *   stable ascending sort by depth of [coarse(0..Sc-1) ++ fine(0..Sf-1)]; perm[j] = source index.
 */
#endif
