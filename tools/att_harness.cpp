// Standalone A/B harness for ds_attention_fwd (no Python, no torch: a run costs seconds of GPU time, not a minute).
//
//     hipcc -O2 -std=c++17 tools/att_harness.cpp -o tools/att_harness -ldl
//     tools/att_harness <library.so> B H n Np bias reps out.bin
//
// Loads the given build of the library through its C ABI (include/depthstereo.h), fills qk [B, Np, 2, H, 64], vt [B, H*64, Np]
// and (bias = 1) a [H, n, n] float32 table with a fixed pseudo-random sequence, packs the bias, runs the kernel reps times
// between HIP events and writes the output tensor to out.bin.  Two builds (or two kernel generations of the experiments build:
// DS_ATT_GEN is read once per process) are compared with `cmp a.bin b.bin`: generation 3 restates generation 2's arithmetic
// in another order of independent operations, so the files must be identical.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

struct ds_ctx;
typedef int (*ctx_create_t)(ds_ctx **, int);
typedef int (*ctx_destroy_t)(ds_ctx *);
typedef const char *(*last_error_t)(void);
typedef int (*bias_pack_t)(ds_ctx *, const float *, int, int, int, int, void *, void *);
typedef int (*attn_t)(ds_ctx *, const void *, const void *, const void *, void *, int, int, int, int, float, int, void *);
typedef int (*prof_t)(unsigned long long *, int);      // experiments library only: the phase clock of generation 2 (DS_ATT_PROF=1)

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }
static float unif(uint32_t &s) { return (float)(lcg(s) >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f; }      // [-1, 1)

int main(int argc, char **argv)
{
    if (argc < 9) { fprintf(stderr, "usage: %s <library.so> B H n Np bias reps out.bin\n", argv[0]); return 1; }
    const int B = atoi(argv[2]), H = atoi(argv[3]), n = atoi(argv[4]), Np = atoi(argv[5]), with_bias = atoi(argv[6]), reps = atoi(argv[7]);
    void *lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    auto ctx_create = (ctx_create_t)dlsym(lib, "ds_ctx_create");
    auto ctx_destroy = (ctx_destroy_t)dlsym(lib, "ds_ctx_destroy");
    auto last_error = (last_error_t)dlsym(lib, "ds_last_error");
    auto bias_pack = (bias_pack_t)dlsym(lib, "ds_attention_bias_pack");
    auto attn = (attn_t)dlsym(lib, "ds_attention_fwd");
    if (!ctx_create || !ctx_destroy || !last_error || !bias_pack || !attn) { fprintf(stderr, "missing symbol\n"); return 1; }
    const size_t n_qk = (size_t)B * Np * 2 * H * 64, n_vt = (size_t)B * H * 64 * Np, n_out = (size_t)B * Np * H * 64;
    const int Np64 = (Np + 63) / 64 * 64;
    std::vector<_Float16> h_qk(n_qk), h_vt(n_vt);
    uint32_t seed = 12345u;
    for (auto &v : h_qk) v = (_Float16)unif(seed);
    for (auto &v : h_vt) v = (_Float16)unif(seed);
    void *d_qk, *d_vt, *d_out, *d_bias = nullptr, *d_table = nullptr;
    CHECK(hipMalloc(&d_qk, n_qk * 2));
    CHECK(hipMalloc(&d_vt, n_vt * 2));
    CHECK(hipMalloc(&d_out, n_out * 2));
    CHECK(hipMemcpy(d_qk, h_qk.data(), n_qk * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_vt, h_vt.data(), n_vt * 2, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_out, 0xff, n_out * 2));
    ds_ctx *ctx = nullptr;
    if (ctx_create(&ctx, 0) != 0) { fprintf(stderr, "ds_ctx_create: %s\n", last_error()); return 2; }
    if (with_bias) {
        std::vector<float> table((size_t)H * n * n);
        for (auto &v : table) v = 2.0f * unif(seed);
        CHECK(hipMalloc(&d_table, table.size() * 4));
        CHECK(hipMemcpy(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMalloc(&d_bias, (size_t)H * Np64 * Np64 * 2));
        if (bias_pack(ctx, (const float *)d_table, H, n, Np64, 1 /* DS_DTYPE_F16 */, d_bias, nullptr) != 0) {
            fprintf(stderr, "ds_attention_bias_pack: %s\n", last_error()); return 2;
        }
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++)
        if (attn(ctx, d_qk, d_vt, d_bias, d_out, B, Np, H, n, 0.125f, 1, nullptr) != 0) { fprintf(stderr, "ds_attention_fwd: %s\n", last_error()); return 2; }
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; i++) attn(ctx, d_qk, d_vt, d_bias, d_out, B, Np, H, n, 0.125f, 1, nullptr);
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= (float)reps;
    std::vector<_Float16> h_out(n_out);
    CHECK(hipMemcpy(h_out.data(), d_out, n_out * 2, hipMemcpyDeviceToHost));
    double sum = 0.0;
    size_t bad = 0;
    for (int b = 0; b < B; b++)
        for (int t = 0; t < n; t++)                       // valid query rows only (rows in [n, Np) are padding)
            for (int c = 0; c < H * 64; c++) {
                const float v = (float)h_out[((size_t)b * Np + t) * (H * 64) + c];
                if (!(v == v) || v > 1e4f || v < -1e4f) bad++;
                sum += v;
            }
    FILE *f = fopen(argv[8], "wb");
    if (f) { fwrite(h_out.data(), 2, n_out, f); fclose(f); }
    const double flops = 4.0 * n * (double)n * H * 64 * B;
    printf("attention B %d H %d n %d Np %d bias %d: %.4f ms  %.1f TF/s  checksum %.6f  non-finite or huge %zu\n", B, H, n, Np, with_bias, ms,
           flops / (ms * 1e-3) / 1e12, sum, bad);
    if (auto prof = (prof_t)dlsym(lib, "ds_experiments_attention_profile")) {
        unsigned long long acc[16];
        prof(acc, 1);                                     // drop what the timed launches accumulated ...
        attn(ctx, d_qk, d_vt, d_bias, d_out, B, Np, H, n, 0.125f, 1, nullptr);      // ... and clock ONE launch
        if (prof(acc, 1) == 0 && acc[7] + acc[8] > 0) {
            static const char *names[7] = { "loop top", "fetch issue", "S (fragment reads + MFMAs)", "mask + softmax + rescale", "P.V (reads + MFMAs)",
                                            "K / V^T landed + stash", "barrier" };
            unsigned long long tot = 0;
            for (int i = 0; i < 7; i++) tot += acc[i];
            printf("phase clock of one launch: %llu live + %llu idle waves, %.0f cycles per wave\n", acc[7], acc[8], (double)tot / (double)(acc[7] + acc[8]));
            for (int i = 0; i < 7; i++) printf("  %5.1f %%  %s\n", 100.0 * (double)acc[i] / (double)tot, names[i]);
        }
    }
    ctx_destroy(ctx);
    return bad ? 3 : 0;
}
