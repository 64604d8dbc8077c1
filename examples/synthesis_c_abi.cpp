// synthesis_c_abi.cpp — the synthesis entry points of libpanic3d_hip.so from a host with no PyTorch and no Python in it
// (include/panic3d_hip.h only): one SynthesisLayer (modulated 3x3 convolution + bias_act, networks_stylegan2.py:334-353) through the
// struct entry p3d_modconv2d_ex_f32 on two-term f16 operands, then the block's ToRGB + skip connection (:376-380, :476-478)
// through p3d_torgb_f32.  tests/test_c_abi_host.py runs it against the Python path on the same bytes.
//
// It also checks what a foreign-language binding has to: that ITS idea of the four POD structs (here: this compiler's sizeof /
// offsetof of the header) equals the layout the library was compiled with (p3d_struct_layout).
//
//   hipcc --offload-arch=gfx950 -O2 examples/synthesis_c_abi.cpp -I include -L panic3d-anime-reconstruction_amd \
//         -lpanic3d_hip -Wl,-rpath,$PWD/panic3d-anime-reconstruction_amd -o synthesis_c_abi
//   ./synthesis_c_abi <dir>   reads  <dir>/{meta.txt,x.bin,w.bin,styles.bin,noise.bin,bias.bin,fir.bin,wrgb.bin,srgb.bin,brgb.bin,skip.bin}
//                             writes <dir>/{y.bin,img.bin}
// meta.txt: N I O H W up ORGB          (ORGB <= 96 image channels; skip is [N][ORGB][H*up/2][W*up/2])
//   ./synthesis_c_abi <dir> ride     the ABI-9 path of a super-resolution block's tail (superresolution.py:277-293): the input as an activation
//                             image (p3d_act_to_image_f32), the weights in the layout the library asks for (p3d_conv_weight_layout +
//                             p3d_conv_weights_to_f16x2_layout), conv1 with the block's 3-channel ToRGB riding on its launch (p3d_conv_args.rgb_*,
//                             no fp32 activation written), p3d_torgb_combine_f32 with bias and skip image.  Reads <dir>/{meta.txt (N I O H W),
//                             x.bin, w.bin, styles.bin, dcoef.bin, noise.bin, bias.bin, fir.bin, wrgb.bin [3][O], srgb.bin, brgb.bin,
//                             skip.bin [N][3][H/2][W/2]}, writes <dir>/rgb.bin
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "panic3d_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_P3D(x) do { int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "%s failed: %d\n", #x, rc_); return 3; } } while (0)

static std::vector<float> read_f32(const std::string& fn, size_t n) {
    std::vector<float> v(n);
    FILE* f = std::fopen(fn.c_str(), "rb");
    if (!f || std::fread(v.data(), 4, n, f) != n) { std::fprintf(stderr, "cannot read %zu floats from %s\n", n, fn.c_str()); std::exit(1); }
    std::fclose(f);
    return v;
}
static void write_f32(const std::string& fn, const std::vector<float>& v) {
    FILE* f = std::fopen(fn.c_str(), "wb");
    if (!f || std::fwrite(v.data(), 4, v.size(), f) != v.size()) { std::fprintf(stderr, "cannot write %s\n", fn.c_str()); std::exit(1); }
    std::fclose(f);
}
static float* upload(const std::vector<float>& v) {
    float* d = nullptr;
    if (hipMalloc(&d, v.size() * 4 + 16) != hipSuccess || hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        std::fprintf(stderr, "upload failed\n"); std::exit(2);
    }
    return d;
}

// this translation unit's layout of a struct: {sizeof, offsetof(field)...} in declaration order
static int layouts_agree() {
    size_t lib[64];
    {
        const size_t mine[] = {sizeof(p3d_opts), offsetof(p3d_opts, coord_scale), offsetof(p3d_opts, ray_start), offsetof(p3d_opts, ray_end),
                               offsetof(p3d_opts, depth_delta), offsetof(p3d_opts, crop_limit), offsetof(p3d_opts, cull_thresh),
                               offsetof(p3d_opts, Sc), offsetof(p3d_opts, Sf), offsetof(p3d_opts, plane_mode), offsetof(p3d_opts, flags)};
        const int n = p3d_struct_layout(P3D_STRUCT_OPTS, lib, 64);
        if (n != (int)(sizeof(mine) / sizeof(mine[0]))) return 0;
        for (int i = 0; i < n; ++i) if (lib[i] != mine[i]) return 0;
    }
    {
        const size_t mine[] = {sizeof(p3d_conv_args), offsetof(p3d_conv_args, x), offsetof(p3d_conv_args, w), offsetof(p3d_conv_args, w_f16),
                               offsetof(p3d_conv_args, styles), offsetof(p3d_conv_args, demod_coefs), offsetof(p3d_conv_args, noise),
                               offsetof(p3d_conv_args, bias), offsetof(p3d_conv_args, fir), offsetof(p3d_conv_args, y),
                               offsetof(p3d_conv_args, workspace), offsetof(p3d_conv_args, saturated), offsetof(p3d_conv_args, x_img),
                               offsetof(p3d_conv_args, y_img), offsetof(p3d_conv_args, y_img_styles), offsetof(p3d_conv_args, rgb_w),
                               offsetof(p3d_conv_args, rgb_styles), offsetof(p3d_conv_args, rgb_partial), offsetof(p3d_conv_args, workspace_bytes),
                               offsetof(p3d_conv_args, N), offsetof(p3d_conv_args, I), offsetof(p3d_conv_args, H), offsetof(p3d_conv_args, W),
                               offsetof(p3d_conv_args, O), offsetof(p3d_conv_args, ks), offsetof(p3d_conv_args, up),
                               offsetof(p3d_conv_args, demodulate), offsetof(p3d_conv_args, noise_per_sample), offsetof(p3d_conv_args, act),
                               offsetof(p3d_conv_args, mma), offsetof(p3d_conv_args, alpha), offsetof(p3d_conv_args, gain),
                               offsetof(p3d_conv_args, clamp), offsetof(p3d_conv_args, rgb_channels), offsetof(p3d_conv_args, w_f16_layout)};
        const int n = p3d_struct_layout(P3D_STRUCT_CONV_ARGS, lib, 64);
        if (n != (int)(sizeof(mine) / sizeof(mine[0]))) return 0;
        for (int i = 0; i < n; ++i) if (lib[i] != mine[i]) return 0;
    }
    if (p3d_struct_layout(P3D_STRUCT_DUMPS, lib, 64) != 10 || lib[0] != sizeof(p3d_dumps)) return 0;
    if (p3d_struct_layout(P3D_STRUCT_PASTE_ARGS, lib, 64) != 25 || lib[0] != sizeof(p3d_paste_args)) return 0;
    if (p3d_struct_layout(99, lib, 64) != P3D_E_RANGE || p3d_struct_layout(P3D_STRUCT_OPTS, lib, 3) != P3D_E_RANGE) return 0;
    return 1;
}

// conv1 of a block with <= 4 image channels the way the super-resolution runs it (ABI 9)
static int ride(const std::string& dir) {
    int N, I, O, H, W;
    FILE* m = std::fopen((dir + "meta.txt").c_str(), "r");
    if (!m || std::fscanf(m, "%d %d %d %d %d", &N, &I, &O, &H, &W) != 5) { std::fprintf(stderr, "bad meta.txt\n"); return 1; }
    std::fclose(m);
    const int R = 3;
    if (!p3d_conv_takes_image(I, O, W, 1) || !p3d_conv_fuses_torgb(N, I, O, H, W, R)) { std::fprintf(stderr, "the library does not take the ToRGB along for this shape\n"); return 8; }
    float* x = upload(read_f32(dir + "x.bin", (size_t)N * I * H * W));
    float* w = upload(read_f32(dir + "w.bin", (size_t)O * I * 9));
    float* styles = upload(read_f32(dir + "styles.bin", (size_t)N * I));
    float* dcoef = upload(read_f32(dir + "dcoef.bin", (size_t)N * O));
    float* noise = upload(read_f32(dir + "noise.bin", (size_t)H * W));
    float* bias = upload(read_f32(dir + "bias.bin", O));
    float* fir = upload(read_f32(dir + "fir.bin", 16));
    float* wrgb = upload(read_f32(dir + "wrgb.bin", (size_t)R * O));
    float* srgb = upload(read_f32(dir + "srgb.bin", (size_t)N * O));
    float* brgb = upload(read_f32(dir + "brgb.bin", R));
    float* skip = upload(read_f32(dir + "skip.bin", (size_t)N * R * (H / 2) * (W / 2)));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    void *w16, *ximg, *ws; float *part, *rgb; uint32_t* sat;
    CHECK_HIP(hipMalloc(&w16, (size_t)2 * O * 9 * I * 2));
    CHECK_HIP(hipMalloc(&ximg, p3d_act_image_bytes(N, I, H, W)));
    CHECK_HIP(hipMalloc((void**)&part, p3d_torgb_partial_bytes(N, O, H, W, R)));
    CHECK_HIP(hipMalloc((void**)&rgb, (size_t)N * R * H * W * 4));
    CHECK_HIP(hipMalloc((void**)&sat, 4));
    CHECK_HIP(hipMemsetAsync(sat, 0, 4, st));
    const size_t wsb = p3d_modconv2d_workspace_bytes(N, I, O, H, W, 1);
    CHECK_HIP(hipMalloc(&ws, wsb));
    const int layout = p3d_conv_weight_layout(I, O, W, 1);  // the layout THIS layer's dispatch consumes (P3D_WLAYOUT_PLAIN here)
    CHECK_P3D(p3d_conv_weights_to_f16x2_layout(w, O, I, 3, layout, w16, st));
    CHECK_P3D(p3d_act_to_image_f32(x, styles, N, I, H, W, ximg, sat, st));  // (in a generator the previous layer writes it: y_img)
    p3d_conv_args a;
    a.x = nullptr; a.w = w; a.w_f16 = w16; a.styles = nullptr; a.demod_coefs = dcoef; a.noise = noise; a.bias = bias; a.fir = nullptr;
    a.y = nullptr;  // nobody reads the fp32 activation: ToRGB rides on this launch, the next block would take y_img
    a.workspace = ws; a.saturated = sat; a.x_img = ximg; a.y_img = nullptr; a.y_img_styles = nullptr;
    a.rgb_w = wrgb; a.rgb_styles = srgb; a.rgb_partial = part; a.rgb_channels = R; a.w_f16_layout = layout; a.workspace_bytes = wsb;
    a.N = N; a.I = I; a.H = H; a.W = W; a.O = O; a.ks = 3; a.up = 1; a.demodulate = 1; a.noise_per_sample = 0; a.act = 1; a.mma = P3D_CONV_MMA_F16X2;
    a.alpha = 0.2f; a.gain = 1.41421356237309515f; a.clamp = -1.0f;
    CHECK_P3D(p3d_modconv2d_ex_f32(&a, st));
    CHECK_P3D(p3d_torgb_combine_f32(part, O / 64, N, R, H, W, brgb, -1.0f, skip, fir, rgb, st));
    CHECK_HIP(hipStreamSynchronize(st));
    uint32_t hsat = 1;
    CHECK_HIP(hipMemcpy(&hsat, sat, 4, hipMemcpyDeviceToHost));
    std::vector<float> h((size_t)N * R * H * W);
    CHECK_HIP(hipMemcpy(h.data(), rgb, h.size() * 4, hipMemcpyDeviceToHost));
    write_f32(dir + "rgb.bin", h);
    std::printf("conv1 %d -> %d @%dx%d with ToRGB on its launch, weight layout %d; operand domain flag %u\n", I, O, H, W, layout, hsat);
    // a copy of the weights in another image layout than the layer's is refused
    a.w_f16_layout = P3D_WLAYOUT_UP;
    if (p3d_modconv2d_ex_f32(&a, st) != P3D_E_RANGE) return 4;
    return hsat == 0 ? 0 : 7;
}

int main(int argc, char** argv) {
    if (argc != 2 && argc != 3) { std::fprintf(stderr, "usage: %s <dir> [ride]\n", argv[0]); return 1; }
    if (p3d_abi_version() != P3D_ABI_VERSION) { std::fprintf(stderr, "header / library ABI mismatch\n"); return 5; }
    if (!layouts_agree()) { std::fprintf(stderr, "struct layouts of this host and of the library differ\n"); return 6; }
    std::printf("%s: struct layouts agree\n", p3d_build_info());
    const std::string dir = std::string(argv[1]) + "/";
    if (argc == 3) return ride(dir);
    int N, I, O, H, W, up, ORGB;
    FILE* m = std::fopen((dir + "meta.txt").c_str(), "r");
    if (!m || std::fscanf(m, "%d %d %d %d %d %d %d", &N, &I, &O, &H, &W, &up, &ORGB) != 7) { std::fprintf(stderr, "bad meta.txt\n"); return 1; }
    std::fclose(m);
    const int OH = H * up, OW = W * up;
    float* x = upload(read_f32(dir + "x.bin", (size_t)N * I * H * W));
    float* w = upload(read_f32(dir + "w.bin", (size_t)O * I * 9));
    float* styles = upload(read_f32(dir + "styles.bin", (size_t)N * I));
    float* noise = upload(read_f32(dir + "noise.bin", (size_t)OH * OW));
    float* bias = upload(read_f32(dir + "bias.bin", O));
    float* fir = upload(read_f32(dir + "fir.bin", 16));  // setup_filter([1,3,3,1]) flipped, times up^2 (p3d_modconv2d_f32)
    float* wrgb = upload(read_f32(dir + "wrgb.bin", (size_t)ORGB * O));
    float* srgb = upload(read_f32(dir + "srgb.bin", (size_t)N * O));  // ToRGB styles, weight_gain applied
    float* brgb = upload(read_f32(dir + "brgb.bin", ORGB));
    float* skip = upload(read_f32(dir + "skip.bin", (size_t)N * ORGB * (OH / 2) * (OW / 2)));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    // once per layer: the two-term f16 operand copy of the weights, the transposed ToRGB weights
    void* w16; float* wrgb_t; float *y, *img; void* ws; uint32_t* sat;
    CHECK_HIP(hipMalloc(&w16, (size_t)2 * O * 9 * I * 2));
    CHECK_HIP(hipMalloc(&wrgb_t, (size_t)O * 96 * 4));
    CHECK_HIP(hipMalloc(&y, (size_t)N * O * OH * OW * 4));
    CHECK_HIP(hipMalloc(&img, (size_t)N * ORGB * OH * OW * 4));
    CHECK_HIP(hipMalloc(&sat, 4));
    CHECK_HIP(hipMemsetAsync(sat, 0, 4, st));
    const size_t wsb = p3d_modconv2d_workspace_bytes(N, I, O, H, W, up);
    CHECK_HIP(hipMalloc(&ws, wsb));
    CHECK_P3D(p3d_conv_weights_to_f16x2(w, O, I, 3, w16, st));
    CHECK_P3D(p3d_torgb_weights_f32(wrgb, ORGB, O, wrgb_t, st));
    // SynthesisLayer.forward: modulate, 3x3 (transposed + FIR when up = 2), demodulate, + noise, + bias, lrelu * sqrt(2)
    p3d_conv_args a;
    a.x = x; a.w = w; a.w_f16 = w16; a.styles = styles; a.demod_coefs = nullptr; a.noise = noise; a.bias = bias; a.fir = up == 2 ? fir : nullptr;
    a.y = y; a.workspace = ws; a.saturated = sat; a.x_img = nullptr; a.y_img = nullptr; a.y_img_styles = nullptr; a.workspace_bytes = wsb;
    a.N = N; a.I = I; a.H = H; a.W = W; a.O = O; a.ks = 3; a.up = up; a.demodulate = 1; a.noise_per_sample = 0; a.act = 1; a.mma = P3D_CONV_MMA_F16X2;
    a.alpha = 0.2f; a.gain = 1.41421356237309515f; a.clamp = -1.0f;
    a.rgb_w = nullptr; a.rgb_styles = nullptr; a.rgb_partial = nullptr; a.rgb_channels = 0;  // (ABI 9: no ToRGB riding on this launch)
    a.w_f16_layout = P3D_WLAYOUT_OIK;  // (the copy p3d_conv_weights_to_f16x2 made)
    CHECK_P3D(p3d_modconv2d_ex_f32(&a, st));
    // ToRGBLayer.forward + img = upsample2d(img) + y
    CHECK_P3D(p3d_torgb_f32(y, N, O, OH, OW, wrgb_t, ORGB, srgb, brgb, -1.0f, skip, fir, img, st));
    CHECK_HIP(hipStreamSynchronize(st));
    uint32_t hsat = 1;
    CHECK_HIP(hipMemcpy(&hsat, sat, 4, hipMemcpyDeviceToHost));
    std::vector<float> hy((size_t)N * O * OH * OW), hi((size_t)N * ORGB * OH * OW);
    CHECK_HIP(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hi.data(), img, hi.size() * 4, hipMemcpyDeviceToHost));
    write_f32(dir + "y.bin", hy); write_f32(dir + "img.bin", hi);
    std::printf("conv %dx%d -> %dx%d, %d -> %d channels; torgb -> %d; operand domain flag %u\n", H, W, OH, OW, I, O, ORGB, hsat);
    // argument errors come back as codes: a struct with no weights, an image input without two-term operands
    a.w = nullptr;
    if (p3d_modconv2d_ex_f32(&a, st) != P3D_E_ARG) return 4;
    a.w = w; a.mma = P3D_CONV_MMA_F32; a.x_img = w16;
    if (p3d_modconv2d_ex_f32(&a, st) != P3D_E_RANGE) return 4;
    return hsat == 0 ? 0 : 7;
}
