// Fused element-wise pieces of a transformer block (the parts of
// ddepth_anything_v2/depth_anything_v2/dinov2_layers/block.py:82-107 and dmidas/backbones/beit.py:94-107 that sit between
// the GEMMs):   x <- x + gamma * branch ;  h <- LayerNorm(x) * w + b     in ONE pass over the token matrix.
// The reference runs LayerScale multiply, residual add and LayerNorm as three kernels (five tensor passes); here a row
// is read once (x, branch), written once (x, h).  One wave per token row, float32 statistics on the ROUNDED residual
// stream (so h is exactly LayerNorm of the x that is stored), two-pass variance in registers.
#include "ds_common.h"

template <int BF16> struct eo_traits;
template <> struct eo_traits<0> { typedef _Float16 T; };
template <> struct eo_traits<1> { typedef __bf16 T; };

template <int BF16, int EPL, int W>
__global__ __launch_bounds__(256) void k_residual_layernorm(const void *x_, const void *o_, const void *gamma_, const void *lnw_,
                                                             const void *lnb_, void *xout_, void *hout_, int M, float eps)
{
    typedef typename eo_traits<BF16>::T T;
    constexpr int C = EPL * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T *x = (const T *)x_ + (size_t)row * C;
    const T *o = o_ ? (const T *)o_ + (size_t)row * C : nullptr;
    const T *gamma = (const T *)gamma_, *lnw = (const T *)lnw_, *lnb = (const T *)lnb_;
    T *xout = xout_ ? (T *)xout_ + (size_t)row * C : nullptr;
    T *hout = (T *)hout_ + (size_t)row * C;
    float v[EPL];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < EPL / W; k++) {
        const int idx = (k * 64 + lane) * W;
        T xv[W], ov[W], gv[W];
        __builtin_memcpy(xv, x + idx, sizeof(xv));
        if (o) {
            __builtin_memcpy(ov, o + idx, sizeof(ov));
            if (gamma) __builtin_memcpy(gv, gamma + idx, sizeof(gv));
        }
        T rv[W];
#pragma unroll
        for (int t = 0; t < W; t++) {
            float f = (float)xv[t];
            if (o) f += (gamma ? (float)gv[t] : 1.0f) * (float)ov[t];
            rv[t] = (T)f;
            v[k * W + t] = (float)rv[t];
            sum += v[k * W + t];
        }
        if (o && xout) __builtin_memcpy(xout + idx, rv, sizeof(rv));
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) sum += __shfl_xor(sum, s, 64);
    const float mean = sum * (1.0f / C);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; i++) { const float d = v[i] - mean; sq += d * d; }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) sq += __shfl_xor(sq, s, 64);
    const float rstd = rsqrtf(sq * (1.0f / C) + eps);
#pragma unroll
    for (int k = 0; k < EPL / W; k++) {
        const int idx = (k * 64 + lane) * W;
        T wv[W], bv[W], hv[W];
        __builtin_memcpy(wv, lnw + idx, sizeof(wv));
        __builtin_memcpy(bv, lnb + idx, sizeof(bv));
#pragma unroll
        for (int t = 0; t < W; t++) hv[t] = (T)((v[k * W + t] - mean) * rstd * (float)wv[t] + (float)bv[t]);
        __builtin_memcpy(hout + idx, hv, sizeof(hv));
    }
}

template <int BF16>
static int eo_launch(int C, const void *x, const void *o, const void *gamma, const void *lnw, const void *lnb, void *xout, void *hout,
                     int M, float eps, hipStream_t st)
{
    dim3 grid((M + 3) / 4), block(256);
    switch (C) {
    case 384: hipLaunchKernelGGL((k_residual_layernorm<BF16, 6, 2>), grid, block, 0, st, x, o, gamma, lnw, lnb, xout, hout, M, eps); break;
    case 768: hipLaunchKernelGGL((k_residual_layernorm<BF16, 12, 4>), grid, block, 0, st, x, o, gamma, lnw, lnb, xout, hout, M, eps); break;
    case 1024: hipLaunchKernelGGL((k_residual_layernorm<BF16, 16, 8>), grid, block, 0, st, x, o, gamma, lnw, lnb, xout, hout, M, eps); break;
    case 1536: hipLaunchKernelGGL((k_residual_layernorm<BF16, 24, 8>), grid, block, 0, st, x, o, gamma, lnw, lnb, xout, hout, M, eps); break;
    default: ds_set_error("ds_residual_layernorm: channel count %d not built (384, 768, 1024, 1536)", C); return DS_EUNSUPPORTED;
    }
    return DS_OK;
}

DS_API int ds_residual_layernorm(ds_ctx *ctx, const void *x, const void *branch, const void *gamma, const void *ln_weight,
                                 const void *ln_bias, void *x_out, void *h_out, int64_t rows, int channels, float eps, int dtype,
                                 void *stream)
{
    DS_REQUIRE(ctx && x && ln_weight && ln_bias && h_out, DS_EINVAL, "ds_residual_layernorm: null argument");
    DS_REQUIRE(rows > 0 && rows < (1ll << 31), DS_EINVAL, "ds_residual_layernorm: bad row count");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_residual_layernorm: dtype must be f16 or bf16");
    DS_REQUIRE(branch == nullptr || x_out != nullptr, DS_EINVAL, "ds_residual_layernorm: x_out is required with a branch");
    DS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)h_out & 15) == 0 && ((uintptr_t)branch & 15) == 0 && ((uintptr_t)x_out & 15) == 0,
               DS_EINVAL, "ds_residual_layernorm: operands must be 16-byte aligned");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = dtype == DS_DTYPE_F16
        ? eo_launch<0>(channels, x, branch, gamma, ln_weight, ln_bias, x_out, h_out, (int)rows, eps, (hipStream_t)stream)
        : eo_launch<1>(channels, x, branch, gamma, ln_weight, ln_bias, x_out, h_out, (int)rows, eps, (hipStream_t)stream);
    if (rc) return rc;
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}
