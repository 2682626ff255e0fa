// Standalone A/B harness for ds_stereo_warp (polylines / naive fills) through the C ABI: no Python, seconds of GPU time per run.
//
//     hipcc -O2 -std=c++17 tools/stereo_harness.cpp -o tools/stereo_harness -ldl
//     tools/stereo_harness <library.so> n h w fill noise reps out.bin
//
// fill: 0 none, 1 naive, 2 naive_interpolating, 3 polylines_soft, 4 polylines_sharp (FILL_IDS of src/_native.py).  The depth is
// SURVEY.md's integer pattern (ramp + 8-pixel checker steps + two occluders) plus `noise` (0..65535) of per-pixel pseudo-random
// amplitude: 0 = the smooth synthetic depth (few general pixels), a few hundred = what a network's prediction looks like, more =
// the regime that sends pixels to the general pass and rows to the exact sweep.  Both eyes of a side-by-side pair (divergence
// 2.5 %) are rendered reps times between HIP events; the pair is written to out.bin, so two builds of the library compare with
// `cmp` (the product build is bit-exact against the reference: an experiments build that reproduces its file is, too).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

struct ds_ctx;
struct ds_eye { double divergence_px, separation_px; uint8_t *out; int64_t out_row_stride, out_img_stride; };
typedef int (*ctx_create_t)(ds_ctx **, int);
typedef int (*ctx_destroy_t)(ds_ctx *);
typedef const char *(*last_error_t)(void);
typedef int (*warp_t)(ds_ctx *, const uint8_t *, const void *, int, int, int, int, int, double, const double *, int, const ds_eye *, int, void *);
typedef int (*stats_t)(ds_ctx *, int64_t *, void *);
typedef int (*prof_enable_t)(ds_ctx *, int);
typedef int (*prof_last_t)(ds_ctx *, float *, float *);

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }

int main(int argc, char **argv)
{
    if (argc < 9) { fprintf(stderr, "usage: %s <library.so> n h w fill noise reps out.bin\n", argv[0]); return 1; }
    const int n = atoi(argv[2]), h = atoi(argv[3]), w = atoi(argv[4]), fill = atoi(argv[5]), noise = atoi(argv[6]), reps = atoi(argv[7]);
    void *lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    auto ctx_create = (ctx_create_t)dlsym(lib, "ds_ctx_create");
    auto ctx_destroy = (ctx_destroy_t)dlsym(lib, "ds_ctx_destroy");
    auto last_error = (last_error_t)dlsym(lib, "ds_last_error");
    auto warp = (warp_t)dlsym(lib, "ds_stereo_warp");
    auto stats = (stats_t)dlsym(lib, "ds_stereo_last_stats");
    auto prof_enable = (prof_enable_t)dlsym(lib, "ds_profile_enable");
    auto prof_last = (prof_last_t)dlsym(lib, "ds_profile_last_ms");
    if (!ctx_create || !ctx_destroy || !last_error || !warp || !stats || !prof_enable || !prof_last) { fprintf(stderr, "missing symbol\n"); return 1; }
    const size_t px = (size_t)n * h * w;
    std::vector<uint8_t> img(px * 3);
    std::vector<uint16_t> depth(px);
    uint32_t seed = 777u;
    for (auto &v : img) v = (uint8_t)(lcg(seed) >> 24);
    for (int i = 0; i < n; i++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int d = (int)(((long long)x * 30000) / (w - 1)) + (((x / 8 + y / 8) % 2) * 8000);
                if (y >= h / 4 && y < h / 2 && x >= w / 3 && x < 2 * w / 3) d = 60000;
                if (y >= 3 * h / 4 && x < w / 5) d = 1000;
                if (noise > 0) d += (int)(lcg(seed) >> 16) % (noise + 1) - noise / 2;
                depth[((size_t)i * h + y) * w + x] = (uint16_t)(d < 0 ? 0 : (d > 65535 ? 65535 : d));
            }
    uint8_t *d_img, *d_out;
    uint16_t *d_depth;
    const size_t out_bytes = px * 2 * 3;                                    // side by side: rows of 2 w pixels
    CHECK(hipMalloc(&d_img, px * 3));
    CHECK(hipMalloc(&d_depth, px * 2));
    CHECK(hipMalloc(&d_out, out_bytes));
    CHECK(hipMemcpy(d_img, img.data(), px * 3, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_depth, depth.data(), px * 2, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_out, 0xAB, out_bytes));
    ds_ctx *ctx = nullptr;
    if (ctx_create(&ctx, 0) != 0) { fprintf(stderr, "ds_ctx_create: %s\n", last_error()); return 2; }
    prof_enable(ctx, 1);
    const double div_px = 2.5 / 100.0 * w;                                  // create_stereoimages, balance 0: +div/2 and -div/2
    ds_eye eyes[2] = { { div_px * 0.5, 0.0, d_out, (int64_t)2 * w * 3, (int64_t)h * 2 * w * 3 },
                       { -div_px * 0.5, 0.0, d_out + (size_t)w * 3, (int64_t)2 * w * 3, (int64_t)h * 2 * w * 3 } };
    for (int i = 0; i < 2; i++)
        if (warp(ctx, d_img, d_depth, 0 /* DS_DEPTH_U16 */, n, h, w, 3, 1.0, nullptr, fill, eyes, 2, nullptr) != 0) {
            fprintf(stderr, "ds_stereo_warp: %s\n", last_error()); return 2;
        }
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; i++) warp(ctx, d_img, d_depth, 0, n, h, w, 3, 1.0, nullptr, fill, eyes, 2, nullptr);
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f, render_ms = 0.f, exact_ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= (float)reps;
    prof_last(ctx, &render_ms, &exact_ms);
    int64_t st[2] = { 0, 0 };
    stats(ctx, st, nullptr);
    std::vector<uint8_t> out(out_bytes);
    CHECK(hipMemcpy(out.data(), d_out, out_bytes, hipMemcpyDeviceToHost));
    unsigned long long sum = 0;
    for (auto v : out) sum = sum * 1099511628211ull + v;                    // FNV-style running hash of the pair
    FILE *f = fopen(argv[8], "wb");
    if (f) { fwrite(out.data(), 1, out_bytes, f); fclose(f); }
    const double algo = 11.0 * h * (double)w * n;                           // 11 bytes per pixel and pair
    printf("stereo fill %d n %d %dx%d noise %d: %.4f ms per pair batch (%.1f GB/s algorithmic), render kernel %.4f ms, exact sweep %.4f ms, "
           "exact rows %lld, general-pixel chunks %lld, hash %016llx\n", fill, n, w, h, noise, ms, algo / (ms * 1e-3) / 1e9, render_ms, exact_ms,
           (long long)st[0], (long long)st[1], sum);
    ctx_destroy(ctx);
    return 0;
}
