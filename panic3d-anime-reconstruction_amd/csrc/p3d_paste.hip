// p3d_paste.hip — paste_front (training/triplane.py:607-691) as ONE launch on gfx950.
//
// The reference composes the front-view paste from six tensor ops per mask (three bilinear `F.interpolate`s to the
// illustration's size, `kornia.filters.sobel`, a nearest `F.interpolate`, `F.grid_sample` of the illustration, `torch.lerp`),
// every one a pass over 512^2 x N pixels through HBM.  Here one thread produces one output pixel: it reads the render-resolution
// maps (weights, xyz, front-occlusion weights, rays: a few hundred KB, L2-resident), evaluates the up-sampled xyz at the pixel
// and its 8 neighbours for the Sobel, samples the illustration and writes image / paste / the five masks once.
// HBM-bound: ~11 floats written + 3 read per output pixel.
//
// Semantics restated (fp32, the same formulas torch's kernels evaluate):
//   F.interpolate(x, S, mode='bilinear', align_corners=False):  src = max((i + 0.5) * (r / S) - 0.5, 0), i0 = floor(src),
//       i1 = min(i0 + 1, r - 1), l = src - i0;  v = (1-ly) * ((1-lx) a00 + lx a01) + ly * ((1-lx) a10 + lx a11)
//   F.interpolate(x, S, mode='nearest'):  src = min(floor(i * (r / S)), r - 1)
//   kornia.filters.sobel (0.6.5, normalized=True, eps=1e-6): 3x3 Sobel kernels / 8 on the replicate-padded image,
//       sqrt(gx^2 + gy^2 + eps)  — restated, NOT pinned against kornia (it is not installable here)
//   F.grid_sample(bilinear, padding_mode='border', align_corners=False): ix = clamp(((gx + 1) * W - 1) / 2, 0, W - 1), taps at
//       floor / floor + 1 (the out-of-range tap has weight 0)
//   torch.lerp(a, b, w) = w < 0.5 ? a + w (b - a) : b - (b - a)(1 - w)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/panic3d_hip.h"

#define DEV __device__ __forceinline__

struct UpIdx { int i0, i1; float l; };
DEV UpIdx up_index(int i, float scale, int r) {
    float src = ((float)i + 0.5f) * scale - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    UpIdx u;
    u.i0 = (int)src;  // src >= 0: truncation = floor
    u.i0 = u.i0 < r - 1 ? u.i0 : r - 1;
    u.i1 = u.i0 + 1 < r ? u.i0 + 1 : r - 1;
    u.l = src - (float)u.i0;
    return u;
}
DEV float bilerp(const float* m, int r, const UpIdx& y, const UpIdx& x) {
    const float a00 = m[y.i0 * r + x.i0], a01 = m[y.i0 * r + x.i1], a10 = m[y.i1 * r + x.i0], a11 = m[y.i1 * r + x.i1];
    const float w0 = 1.0f - x.l, h0 = 1.0f - y.l;
    return h0 * (w0 * a00 + x.l * a01) + y.l * (w0 * a10 + x.l * a11);
}

__global__ __launch_bounds__(256) void k_paste_front(p3d_paste_args a) {
    const int S = a.S, r = a.r;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)a.N * S * S) return;
    const int X = (int)(idx % S), Y = (int)((idx / S) % S), n = (int)(idx / ((long long)S * S));
    const float scale = (float)r / (float)S;
    const size_t rr = (size_t)r * r;
    const float* wmap = a.weights + (size_t)n * rr;
    const float* xyz = a.xyz + (size_t)n * 3 * rr;
    const float* occ = a.occ + (size_t)n * rr;

    const UpIdx uy = up_index(Y, scale, r), ux = up_index(X, scale, r);
    // visible-weight mask: interpolate(image_weights) > thresh_weight
    const float wmask = bilerp(wmap, r, uy, ux) > a.thresh_weight ? 1.0f : 0.0f;
    // front-occlusion mask: interpolate((occ < thresh_occ).float())
    float fmask;
    {
        const float o00 = occ[uy.i0 * r + ux.i0] < a.thresh_occ ? 1.0f : 0.0f, o01 = occ[uy.i0 * r + ux.i1] < a.thresh_occ ? 1.0f : 0.0f;
        const float o10 = occ[uy.i1 * r + ux.i0] < a.thresh_occ ? 1.0f : 0.0f, o11 = occ[uy.i1 * r + ux.i1] < a.thresh_occ ? 1.0f : 0.0f;
        const float w0 = 1.0f - ux.l, h0 = 1.0f - uy.l;
        fmask = h0 * (w0 * o00 + ux.l * o01) + uy.l * (w0 * o10 + ux.l * o11);
    }
    // xyz-discrepancy mask: nearest-interpolated distance of the rendered xyz from its own ray (triplane.py:600-605) < thresh_dxyz
    float dmask;
    {
        int sy = (int)((float)Y * scale), sx = (int)((float)X * scale);
        sy = sy < r - 1 ? sy : r - 1;
        sx = sx < r - 1 ? sx : r - 1;
        const size_t q = (size_t)sy * r + sx;
        const float* ro = a.rays_o + (size_t)n * 3 * rr;
        const float* rd = a.rays_d + (size_t)n * 3 * rr;
        const float px = -xyz[q], py = xyz[rr + q], pz = -xyz[2 * rr + q];
        const float dx = px - ro[q], dy = py - ro[rr + q], dz = pz - ro[2 * rr + q];
        const float nx = rd[q], ny = rd[rr + q], nz = rd[2 * rr + q];
        const float dot = (dx * nx + dy * ny) + dz * nz;
        const float ex = dx - dot * nx, ey = dy - dot * ny, ez = dz - dot * nz;
        dmask = sqrtf((ex * ex + ey * ey) + ez * ez) < a.thresh_dxyz ? 1.0f : 0.0f;
    }
    // crevice mask: |sobel(interpolate(image_xyz))|_2 over the 3 channels < thresh_edges; replicate padding = clamped neighbours
    float smask, upx, upy;
    {
        UpIdx ys[3], xs[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int yy = Y + k - 1, xx = X + k - 1;
            yy = yy < 0 ? 0 : (yy > S - 1 ? S - 1 : yy);
            xx = xx < 0 ? 0 : (xx > S - 1 ? S - 1 : xx);
            ys[k] = up_index(yy, scale, r);
            xs[k] = up_index(xx, scale, r);
        }
        float sum = 0.0f, centre[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* m = xyz + (size_t)c * rr;
            float v[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) v[i][j] = bilerp(m, r, ys[i], xs[j]);
            centre[c] = v[1][1];
            const float gx = ((v[0][2] - v[0][0]) + 2.0f * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0])) * 0.125f;
            const float gy = ((v[2][0] - v[0][0]) + 2.0f * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2])) * 0.125f;
            const float mag = sqrtf(gx * gx + gy * gy + 1e-6f);
            sum += mag * mag;
        }
        smask = sqrtf(sum) < a.thresh_edges ? 1.0f : 0.0f;
        upx = centre[0];
        upy = centre[1];
    }
    const float mask = ((wmask * smask) * fmask) * dmask;  // (* mask_frontweight = 1: front_weight_erosion is not used by generate.py)
    // sample_orthofront (triplane.py:555-564): vij = 1 - (xyz[[1,0]] + bw/2) / bw; grid = vij * 2 - 1 on the TRANSPOSED illustration
    const float* front = a.front + (size_t)(a.front_shared ? 0 : n) * 3 * S * S;
    float paste[3];
    {
        const float v0 = 1.0f - (upy + a.box_warp * 0.5f) / a.box_warp, v1 = 1.0f - (upx + a.box_warp * 0.5f) / a.box_warp;
        const float gx = v0 * 2.0f - 1.0f, gy = v1 * 2.0f - 1.0f;  // grid x <- vij[0], grid y <- vij[1]
        float ix = ((gx + 1.0f) * (float)S - 1.0f) * 0.5f, iy = ((gy + 1.0f) * (float)S - 1.0f) * 0.5f;
        ix = fminf(fmaxf(ix, 0.0f), (float)(S - 1));
        iy = fminf(fmaxf(iy, 0.0f), (float)(S - 1));
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float tx = ix - fx0, ty = iy - fy0;
        const float wnw = (1.0f - tx) * (1.0f - ty), wne = tx * (1.0f - ty), wsw = (1.0f - tx) * ty, wse = tx * ty;
        const bool bx = x1 < S, by = y1 < S;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // transposed input: sample at (row y, column x) of front^T = front[c][x][y]
            const float* f = front + (size_t)c * S * S;
            auto at = [&](int yy, int xx) { float t = f[(size_t)xx * S + yy]; return a.normalize_images ? t * 2.0f - 1.0f : t; };
            float acc = at(y0, x0) * wnw;
            if (bx) acc += at(y0, x1) * wne;
            if (by) acc += at(y1, x0) * wsw;
            if (bx && by) acc += at(y1, x1) * wse;
            paste[c] = acc;
        }
    }
    const size_t pix = (size_t)Y * S + X, img = (size_t)n * 3 * S * S;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t o = img + (size_t)c * S * S + pix;
        const float s0 = a.image[o], e = paste[c];
        a.out_image[o] = mask < 0.5f ? s0 + mask * (e - s0) : e - (e - s0) * (1.0f - mask);
        a.out_paste[o] = e;
    }
    const size_t mo = (size_t)n * S * S + pix;
    a.out_mask[mo] = mask;
    a.out_mask_weights[mo] = wmask;
    a.out_mask_edges[mo] = smask;
    a.out_mask_occ[mo] = fmask;
    a.out_mask_dxyz[mo] = dmask;
}

// Library identification with the content hash of ALL kernel sources (the build passes -DP3D_SRC_HASH to this translation unit
// only, so that a change in one source recompiles that source and this small file, not the others).
#ifndef P3D_SRC_HASH
#define P3D_SRC_HASH "unknown"
#endif
extern "C" const char* p3d_build_info(void) { return "libpanic3d_hip gfx950 (MI355X) f32 contract v1 src=" P3D_SRC_HASH; }

extern "C" int p3d_paste_front_f32(const p3d_paste_args* args, void* stream) {
    if (!args) return P3D_E_ARG;
    const p3d_paste_args& a = *args;
    if (!a.weights || !a.xyz || !a.occ || !a.rays_o || !a.rays_d || !a.front || !a.image || !a.out_image || !a.out_paste || !a.out_mask ||
        !a.out_mask_weights || !a.out_mask_edges || !a.out_mask_occ || !a.out_mask_dxyz || a.N <= 0 || a.r <= 0 || a.S <= 0)
        return P3D_E_ARG;
    if (a.r > 4096 || a.S > 8192) return P3D_E_RANGE;
    const long long total = (long long)a.N * a.S * a.S;
    hipLaunchKernelGGL(k_paste_front, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? P3D_OK : (int)e;
}
