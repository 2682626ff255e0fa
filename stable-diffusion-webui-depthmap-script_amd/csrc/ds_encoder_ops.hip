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

// ------------------------------------------------------------------------------------------------------------------------
// ds_upsample_bilinear_nhwc: F.interpolate(x, size, mode="bilinear", align_corners=...) for channels_last activations of the
// DPT decoders (dmidas/blocks.py:429-431, ddepth_anything_v2/.../util/blocks.py:141-145, the heads' Interpolate).  The
// op is pure HBM streaming (the output is 4x the input); one lane produces 8 channels (16 bytes) of one output pixel from
// four 16-byte reads that hit L1/L2, consecutive lanes walk the channel axis, so every store instruction writes whole lines.
template <int BF16>
__global__ __launch_bounds__(256) void k_upsample_bilinear_nhwc(const void *in_, void *out_, int C8, int ih, int iw, int oh, int ow,
                                                                 float sy, float sx, int align_corners, long long total)
{
    typedef typename eo_traits<BF16>::T T;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    long long r = idx / C8;
    const int ox = (int)(r % ow); r /= ow;
    const int oy = (int)(r % oh);
    const int b = (int)(r / oh);
    float fy, fx;
    if (align_corners) { fy = sy * oy; fx = sx * ox; }
    else { fy = fmaxf(sy * (oy + 0.5f) - 0.5f, 0.f); fx = fmaxf(sx * (ox + 0.5f) - 0.5f, 0.f); }
    const int y0 = min((int)fy, ih - 1), x0 = min((int)fx, iw - 1);
    const int y1 = min(y0 + 1, ih - 1), x1 = min(x0 + 1, iw - 1);
    const float ty = fy - y0, tx = fx - x0;
    const T *in = (const T *)in_ + (size_t)b * ih * iw * C8 * 8 + (size_t)c8 * 8;
    T a[8], bq[8], c[8], d[8], o[8];
    __builtin_memcpy(a, in + ((size_t)y0 * iw + x0) * C8 * 8, 16);
    __builtin_memcpy(bq, in + ((size_t)y0 * iw + x1) * C8 * 8, 16);
    __builtin_memcpy(c, in + ((size_t)y1 * iw + x0) * C8 * 8, 16);
    __builtin_memcpy(d, in + ((size_t)y1 * iw + x1) * C8 * 8, 16);
    const float w00 = (1.f - ty) * (1.f - tx), w01 = (1.f - ty) * tx, w10 = ty * (1.f - tx), w11 = ty * tx;
#pragma unroll
    for (int k = 0; k < 8; k++)
        o[k] = (T)(w00 * (float)a[k] + w01 * (float)bq[k] + w10 * (float)c[k] + w11 * (float)d[k]);
    __builtin_memcpy((T *)out_ + (size_t)idx * 8, o, 16);
}

DS_API int ds_upsample_bilinear_nhwc(ds_ctx *ctx, const void *in, void *out, int batch, int channels, int in_h, int in_w,
                                     int out_h, int out_w, int align_corners, int dtype, void *stream)
{
    DS_REQUIRE(ctx && in && out, DS_EINVAL, "ds_upsample_bilinear_nhwc: null argument");
    DS_REQUIRE(batch > 0 && channels > 0 && (channels % 8) == 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0, DS_EINVAL,
               "ds_upsample_bilinear_nhwc: bad shape (channels must be a multiple of 8)");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_upsample_bilinear_nhwc: dtype must be f16 or bf16");
    DS_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, DS_EINVAL, "ds_upsample_bilinear_nhwc: 16-byte alignment");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    float sy, sx;
    if (align_corners) {
        sy = out_h > 1 ? (float)(in_h - 1) / (float)(out_h - 1) : 0.f;
        sx = out_w > 1 ? (float)(in_w - 1) / (float)(out_w - 1) : 0.f;
    } else {
        sy = (float)in_h / (float)out_h;
        sx = (float)in_w / (float)out_w;
    }
    const int C8 = channels / 8;
    const long long total = (long long)batch * out_h * out_w * C8;
    DS_REQUIRE((total + 255) / 256 < (1ll << 31), DS_EUNSUPPORTED, "ds_upsample_bilinear_nhwc: tensor too large");
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == DS_DTYPE_F16)
        hipLaunchKernelGGL((k_upsample_bilinear_nhwc<0>), grid, dim3(256), 0, (hipStream_t)stream, in, out, C8, in_h, in_w, out_h, out_w, sy, sx, align_corners, total);
    else
        hipLaunchKernelGGL((k_upsample_bilinear_nhwc<1>), grid, dim3(256), 0, (hipStream_t)stream, in, out, C8, in_h, in_w, out_h, out_w, sy, sx, align_corners, total);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}
