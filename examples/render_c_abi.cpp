// render_c_abi.cpp — a host that binds libpanic3d_hip.so through its C ABI ONLY (include/panic3d_hip.h): no PyTorch, no
// Python; device memory through the HIP runtime.  It is what a non-Python integrator of the reference's renderer would
// write, and tests/test_c_abi_host.py runs it against the Python path on the same bytes (bit-exact).
//
//   hipcc --offload-arch=gfx950 -O2 examples/render_c_abi.cpp -I include -L panic3d-anime-reconstruction_amd \
//         -lpanic3d_hip -Wl,-rpath,$PWD/panic3d-anime-reconstruction_amd -o render_c_abi
//   ./render_c_abi <dir>      reads  <dir>/{meta.txt,planes.bin,rays_o.bin,rays_d.bin,jitter.bin,u.bin,w0.bin,b0.bin,w1.bin,b1.bin}
//                             writes <dir>/{feat.bin,depth.bin,wsum.bin,xyz.bin}
// meta.txt: N H W R tile_w Sc Sf plane_mode flags coord_scale ray_start ray_end depth_delta crop_limit cull_thresh
// planes.bin is the REFERENCE layout [N][3][32][H][W] (training/triplane.py:200-206); the library transposes it.
#include <hip/hip_runtime.h>
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

int main(int argc, char** argv) {
    if (argc != 2) { std::fprintf(stderr, "usage: %s <dir>\n", argv[0]); return 1; }
    const std::string dir = std::string(argv[1]) + "/";
    int N, H, W, tile_w; long long R; p3d_opts o;
    FILE* m = std::fopen((dir + "meta.txt").c_str(), "r");
    if (!m || std::fscanf(m, "%d %d %d %lld %d %d %d %d %d %f %f %f %f %f %f", &N, &H, &W, &R, &tile_w, &o.Sc, &o.Sf, &o.plane_mode,
                          &o.flags, &o.coord_scale, &o.ray_start, &o.ray_end, &o.depth_delta, &o.crop_limit, &o.cull_thresh) != 15) {
        std::fprintf(stderr, "bad meta.txt\n"); return 1;
    }
    std::fclose(m);
    std::printf("%s\n", p3d_build_info());
    const size_t NR = (size_t)N * R;
    float* planes = upload(read_f32(dir + "planes.bin", (size_t)N * 3 * 32 * H * W));
    float* ro = upload(read_f32(dir + "rays_o.bin", NR * 3));
    float* rd = upload(read_f32(dir + "rays_d.bin", NR * 3));
    float* jit = upload(read_f32(dir + "jitter.bin", NR * o.Sc));
    float* u = o.Sf > 0 ? upload(read_f32(dir + "u.bin", NR * o.Sf)) : nullptr;
    float* w0 = upload(read_f32(dir + "w0.bin", 64 * 32));
    float* b0 = upload(read_f32(dir + "b0.bin", 64));
    float* w1 = upload(read_f32(dir + "w1.bin", 33 * 64));
    float* b1 = upload(read_f32(dir + "b1.bin", 33));
    float *nhwc, *feat, *depth, *wsum, *xyz; void* ws;
    CHECK_HIP(hipMalloc(&nhwc, (size_t)N * 3 * 32 * H * W * 4));
    CHECK_HIP(hipMalloc(&feat, NR * 32 * 4)); CHECK_HIP(hipMalloc(&depth, NR * 4));
    CHECK_HIP(hipMalloc(&wsum, NR * 4)); CHECK_HIP(hipMalloc(&xyz, NR * 3 * 4));
    const size_t wsb = p3d_render_workspace_bytes(N, R, o.Sc, o.Sf);
    CHECK_HIP(hipMalloc(&ws, wsb));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));  // stream-ordered: everything below is enqueued on the caller's stream
    CHECK_P3D(p3d_planes_to_nhwc_f32(planes, N * 3, 32, H, W, nhwc, st));
    CHECK_P3D(p3d_render_f32(nhwc, N, H, W, ro, rd, R, tile_w, jit, u, w0, b0, w1, b1, &o, feat, depth, wsum, xyz, ws, wsb, nullptr, st));
    CHECK_HIP(hipStreamSynchronize(st));
    std::vector<float> hf(NR * 32), hd(NR), hw(NR), hx(NR * 3);
    CHECK_HIP(hipMemcpy(hf.data(), feat, hf.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hd.data(), depth, hd.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hw.data(), wsum, hw.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hx.data(), xyz, hx.size() * 4, hipMemcpyDeviceToHost));
    write_f32(dir + "feat.bin", hf); write_f32(dir + "depth.bin", hd); write_f32(dir + "wsum.bin", hw); write_f32(dir + "xyz.bin", hx);
    double s = 0; for (float v : hw) s += v;
    std::printf("rendered %zu rays, mean accumulated weight %.6f\n", NR, s / (double)NR);
    // argument errors come back as codes, not crashes (include/panic3d_hip.h conventions)
    if (p3d_render_f32(nullptr, N, H, W, ro, rd, R, tile_w, jit, u, w0, b0, w1, b1, &o, feat, depth, wsum, xyz, ws, wsb, nullptr, st) != P3D_E_ARG) return 4;
    return 0;
}
