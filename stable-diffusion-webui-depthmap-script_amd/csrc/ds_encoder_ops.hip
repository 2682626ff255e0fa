// Fused element-wise pieces of a transformer block (the parts of
// ddepth_anything_v2/depth_anything_v2/dinov2_layers/block.py:82-107 and dmidas/backbones/beit.py:94-107 that sit between
// the GEMMs):   x <- x + gamma * branch ;  h <- LayerNorm(x) * w + b     in ONE pass over the token matrix.
// The reference runs LayerScale multiply, residual add and LayerNorm as three kernels (five tensor passes); here a row
// is read once (x, branch), written once (x, h).  One wave per token row, float32 statistics on the ROUNDED residual
// stream (so h is exactly LayerNorm of the x that is stored), two-pass variance in registers.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

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

// ds_row_stats: the statistics half of a LayerNorm -- per row {rstd, -mean * rstd} in float32 (mean, then the centred sum of
// squares, exactly as k_residual_layernorm computes them) -- for the GEMMs that fold the LayerNorm into their epilogue
// (ds_linear_ln, ds_linear_vt_ln in csrc/ds_linear.hip): reads x once, writes 8 bytes per row instead of a normalised copy.
template <int BF16, int EPL, int W>
__global__ __launch_bounds__(256) void k_row_stats(const void *x_, float2 *stats, int M, float eps)
{
    typedef typename eo_traits<BF16>::T T;
    constexpr int C = EPL * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T *x = (const T *)x_ + (size_t)row * C;
    float v[EPL];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < EPL / W; k++) {
        T xv[W];
        __builtin_memcpy(xv, x + (k * 64 + lane) * W, sizeof(xv));
#pragma unroll
        for (int t = 0; t < W; t++) { v[k * W + t] = (float)xv[t]; sum += v[k * W + t]; }
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
    if (lane == 0) stats[row] = make_float2(rstd, -mean * rstd);
}

DS_API int ds_row_stats(ds_ctx *ctx, const void *x, void *stats, int64_t rows, int channels, float eps, int dtype, void *stream)
{
    DS_REQUIRE(ctx && x && stats, DS_EINVAL, "ds_row_stats: null argument");
    DS_REQUIRE(rows > 0 && rows < (1ll << 31), DS_EINVAL, "ds_row_stats: bad row count");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_row_stats: dtype must be f16 or bf16");
    DS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)stats & 15) == 0, DS_EINVAL, "ds_row_stats: operands must be 16-byte aligned");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    float2 *out = (float2 *)stats;
    const int M = (int)rows;
#define RS_CASE(C_, EPL_, W_) case C_: if (dtype == DS_DTYPE_F16) hipLaunchKernelGGL((k_row_stats<0, EPL_, W_>), grid, block, 0, st, x, out, M, eps); \
                                       else hipLaunchKernelGGL((k_row_stats<1, EPL_, W_>), grid, block, 0, st, x, out, M, eps); break;
    switch (channels) {
    RS_CASE(384, 6, 2) RS_CASE(768, 12, 4) RS_CASE(1024, 16, 8) RS_CASE(1536, 24, 8)
    default: ds_set_error("ds_row_stats: channel count %d not built (384, 768, 1024, 1536)", channels); return DS_EUNSUPPORTED;
    }
#undef RS_CASE
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// ds_upsample_bilinear_nhwc: F.interpolate(x, size, mode="bilinear", align_corners=...) for channels_last activations of the
// DPT decoders (dmidas/blocks.py:429-431, ddepth_anything_v2/.../util/blocks.py:141-145, the heads' Interpolate).  The
// op is pure HBM streaming (the output is 4x the input); one lane produces 8 channels (16 bytes) of one output pixel from
// four 16-byte reads that hit L1/L2, consecutive lanes walk the channel axis, so every store instruction writes whole lines.
// Grid: x = 256-lane pieces of one output row (ow * C8 lanes), y = output row, z = image -- the row and the image are uniform
// (their interpolation weights and source rows live in scalar registers) and a lane's column is ONE 32-bit division; rounds 1-5
// derived (image, row, column, channel group) from a flat 64-bit index with three 64-bit divisions per lane, which cost more
// vector instructions than the interpolation itself and held the pass at ~3 TB/s.  Same arithmetic per element: same bits.
template <int BF16>
__global__ __launch_bounds__(256) void k_upsample_bilinear_nhwc(const void *in_, void *out_, int C8, int ih, int iw, int oh, int ow,
                                                                 float sy, float sx, int align_corners)
{
    typedef typename eo_traits<BF16>::T T;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (unsigned)ow * (unsigned)C8) return;
    const int ox = (int)(i / (unsigned)C8), c8 = (int)(i - (unsigned)ox * (unsigned)C8);
    const int b = blockIdx.z;
    float fx;
    if (align_corners) fx = sx * ox;
    else fx = fmaxf(sx * (ox + 0.5f) - 0.5f, 0.f);
    const int x0 = min((int)fx, iw - 1), x1 = min(x0 + 1, iw - 1);
    const float tx = fx - x0;
    const T *in = (const T *)in_ + (size_t)b * ih * iw * C8 * 8 + (size_t)c8 * 8;
    // a workgroup renders TWO output rows (2 * blockIdx.y and the next one): when upsampling they mostly interpolate between the same
    // two source rows, whose four pieces are then loaded once -- the pass is bound by the vector cache's request rate (four 16-byte
    // requests per 16 bytes written), not by HBM.  Which rows are shared is uniform (scalar branches).
    int py0 = -1, py1 = -1;
    T a[8], bq[8], c[8], d[8], o[8];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int oy = 2 * (int)blockIdx.y + r;
        if (oy >= oh) break;
        float fy;
        if (align_corners) fy = sy * oy;
        else fy = fmaxf(sy * (oy + 0.5f) - 0.5f, 0.f);
        const int y0 = min((int)fy, ih - 1), y1 = min(y0 + 1, ih - 1);
        const float ty = fy - y0;
        if (y0 == py1 && y0 != py0) {                         // the previous bottom row is this top row
#pragma unroll
            for (int k = 0; k < 8; k++) { a[k] = c[k]; bq[k] = d[k]; }
        } else if (y0 != py0) {
            __builtin_memcpy(a, in + ((size_t)y0 * iw + x0) * C8 * 8, 16);
            __builtin_memcpy(bq, in + ((size_t)y0 * iw + x1) * C8 * 8, 16);
        }
        if (y1 != py1 || r == 0) {
            if (y1 == y0) {
#pragma unroll
                for (int k = 0; k < 8; k++) { c[k] = a[k]; d[k] = bq[k]; }
            } else {
                __builtin_memcpy(c, in + ((size_t)y1 * iw + x0) * C8 * 8, 16);
                __builtin_memcpy(d, in + ((size_t)y1 * iw + x1) * C8 * 8, 16);
            }
        }
        py0 = y0; py1 = y1;
        const float w00 = (1.f - ty) * (1.f - tx), w01 = (1.f - ty) * tx, w10 = ty * (1.f - tx), w11 = ty * tx;
#pragma unroll
        for (int k = 0; k < 8; k++)
            o[k] = (T)(w00 * (float)a[k] + w01 * (float)bq[k] + w10 * (float)c[k] + w11 * (float)d[k]);
        __builtin_memcpy((T *)out_ + ((((size_t)b * oh + oy) * ow) * C8 + i) * 8, o, 16);
    }
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
    DS_REQUIRE((long long)out_w * C8 < (1ll << 31) && out_h <= 65535 && batch <= 65535, DS_EUNSUPPORTED, "ds_upsample_bilinear_nhwc: tensor too large");
    dim3 grid((unsigned)(((long long)out_w * C8 + 255) / 256), (unsigned)((out_h + 1) / 2), (unsigned)batch);
    if (dtype == DS_DTYPE_F16)
        hipLaunchKernelGGL((k_upsample_bilinear_nhwc<0>), grid, dim3(256), 0, (hipStream_t)stream, in, out, C8, in_h, in_w, out_h, out_w, sy, sx, align_corners);
    else
        hipLaunchKernelGGL((k_upsample_bilinear_nhwc<1>), grid, dim3(256), 0, (hipStream_t)stream, in, out, C8, in_h, in_w, out_h, out_w, sy, sx, align_corners);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// ds_preprocess_bicubic: the image -> network-input chain of estimatemidas / estimatedepthanything_v2 in ONE pass
// (src/depthmap_generation.py:381 `cvtColor(BGR2RGB) / 255`, :457-476 Resize(INTER_CUBIC) -> NormalizeImage -> PrepareForNet;
// ddepth_anything_v2/depth_anything_v2/dpt.py:196-221): uint8 [B, H, W, 3] in, [B, 3, h, w] in channels_last memory out,
//     out[b][c][y][x] = (bicubic(in[b][:, :, flip ? 2 - c : c] / 255)(y, x) - mean[c]) / std[c].
// The host used to run it as six torch kernels over the float32 image (flip, permute + convert, / 255, bicubic, normalise,
// cast: 0.9 ms per 32 x 1024^2, a third of it the float32 resize); here every image byte is read once.  The resampling is
// torch's upsample_bicubic2d (align_corners=False, no antialias: the cubic convolution kernel with A = -0.75 on half-pixel
// centres, border replicated, float32 accumulation) -- the stand-in the product has used for cv2.INTER_CUBIC since round 1
// (same kernel and centres; OpenCV's own arithmetic is unpinned), so the network input is the same to float32 rounding.
// One lane per output pixel (three channels): 16 pixel reads that hit L1 / L2 (neighbouring lanes share them).
struct PreParams {
    const uint8_t *in;
    void *out;
    int B, ih, iw, oh, ow, flip;
    float sy, sx;                // in / out
    float mean[3], istd[3];      // per OUTPUT channel
    long long total;
};

__device__ __forceinline__ void pre_cubic(float t, float *c)
{
    const float A = -0.75f;
    const float x0 = t + 1.0f, x3 = 2.0f - t, x2 = 1.0f - t;
    c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    c[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
    c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

template <int DT>      // 0 f16, 1 bf16, 2 f32
__global__ __launch_bounds__(256) void k_preprocess_bicubic(PreParams P)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= P.total) return;
    const int ox = (int)(idx % P.ow);
    long long r = idx / P.ow;
    const int oy = (int)(r % P.oh);
    const int b = (int)(r / P.oh);
    const float fy = P.sy * ((float)oy + 0.5f) - 0.5f, fx = P.sx * ((float)ox + 0.5f) - 0.5f;
    const float fy0 = floorf(fy), fx0 = floorf(fx);
    float cy[4], cx[4];
    pre_cubic(fy - fy0, cy);
    pre_cubic(fx - fx0, cx);
    const int iy = (int)fy0, ix = (int)fx0;
    const uint8_t *img = P.in + (size_t)b * P.ih * P.iw * 3;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 4; ky++) {
        const int y = min(max(iy - 1 + ky, 0), P.ih - 1);
        const uint8_t *row = img + (size_t)y * P.iw * 3;
        float h[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int kx = 0; kx < 4; kx++) {
            const int x = min(max(ix - 1 + kx, 0), P.iw - 1);
            const uint8_t *px = row + (size_t)x * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) h[c] += cx[kx] * ((float)px[c] / 255.0f);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) acc[c] += cy[ky] * h[c];
    }
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) o[c] = (acc[P.flip ? 2 - c : c] - P.mean[c]) * P.istd[c];
    if (DT == 2) {
        float *q = (float *)P.out + (size_t)idx * 3;
        q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
    } else {
        typedef typename eo_traits<DT == 1 ? 1 : 0>::T T;
        T *q = (T *)P.out + (size_t)idx * 3;
        q[0] = (T)o[0]; q[1] = (T)o[1]; q[2] = (T)o[2];
    }
}

DS_API int ds_preprocess_bicubic(ds_ctx *ctx, const void *images, void *out, int batch, int in_h, int in_w, int out_h, int out_w,
                                 int flip_channels, const float *mean, const float *std, int dtype, void *stream)
{
    DS_REQUIRE(ctx && images && out && mean && std, DS_EINVAL, "ds_preprocess_bicubic: null argument");
    DS_REQUIRE(batch > 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0, DS_EINVAL, "ds_preprocess_bicubic: bad shape");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16 || dtype == DS_DTYPE_F32, DS_EINVAL, "ds_preprocess_bicubic: dtype must be f16, bf16 or f32");
    DS_REQUIRE(std[0] != 0.f && std[1] != 0.f && std[2] != 0.f, DS_EINVAL, "ds_preprocess_bicubic: std must be non-zero");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    PreParams P;
    P.in = (const uint8_t *)images; P.out = out;
    P.B = batch; P.ih = in_h; P.iw = in_w; P.oh = out_h; P.ow = out_w; P.flip = flip_channels ? 1 : 0;
    P.sy = (float)in_h / (float)out_h; P.sx = (float)in_w / (float)out_w;
    for (int c = 0; c < 3; c++) { P.mean[c] = mean[c]; P.istd[c] = 1.0f / std[c]; }
    P.total = (long long)batch * out_h * out_w;
    DS_REQUIRE((P.total + 255) / 256 < (1ll << 31), DS_EUNSUPPORTED, "ds_preprocess_bicubic: tensor too large");
    dim3 grid((unsigned)((P.total + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DS_DTYPE_F16) hipLaunchKernelGGL((k_preprocess_bicubic<0>), grid, dim3(256), 0, st, P);
    else if (dtype == DS_DTYPE_BF16) hipLaunchKernelGGL((k_preprocess_bicubic<1>), grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL((k_preprocess_bicubic<2>), grid, dim3(256), 0, st, P);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// ds_dpt_head_tail: the tail of the DPT depth heads, fused:
//     bilinear upsample (align_corners=True) -> conv3x3 128->32 (pad 1) -> ReLU -> conv1x1 32->1 -> ReLU
// (dmidas/dpt_depth.py:150-158: scratch.output_conv[1:6]; ddepth_anything_v2/.../dpt.py:146-147 + output_conv2, :105-111).
// The reference materialises the upsampled 128-channel tensor (2.1 GB at batch 32, 512^2) and the 32-channel tensor; the
// 3x3 convolution on it is the least efficient library call of the whole forward (MIOpen: 3.9 ms, 158 TF/s).  Here a
// workgroup builds the upsampled activations of an 8 x 32 output tile (+1 halo) in LDS, runs the convolution as an
// implicit GEMM on the matrix cores with the roles swapped (A = weights: 32 output channels, B = activations: 32 pixels
// of one row), so that a lane ends up holding all 32 channels of ONE pixel (16 registers + the lane^32 partner) and the
// ReLU + 1x1 convolution + ReLU is an in-lane dot product.  Nothing but the final 1-channel map is written.
#define HT_TW 32
#define HT_PW (HT_TW + 2)

typedef _Float16 ht_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ht_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ht_f32x16 __attribute__((ext_vector_type(16)));

struct HeadTailParams {
    const void *x;            // [B, ih, iw, 128]
    const void *wfrag;        // conv3x3 weights as MFMA fragments: [tap 9][s 8][hi 2][co 32][8]
    const float *b2, *w3;     // [32] conv3x3 bias, [32] conv1x1 weight
    void *out;                // [B, oh, ow]
    float b3;
    int B, ih, iw, oh, ow, relu_out;
    float sy, sx;
};

template <int BF16, int RPW>      // RPW = output rows per wave; tile = 4*RPW rows x 32 columns
__global__ __launch_bounds__(256) void k_dpt_head_tail(HeadTailParams P)
{
    typedef typename eo_traits<BF16>::T T;
    constexpr int HT_TH = 4 * RPW;
    constexpr int HT_NPIX = (HT_TH + 2) * HT_PW;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_act[];      // [HT_NPIX][16 chunks of 16 B], chunk ^ (pix & 15)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, ty0 = blockIdx.y * HT_TH, tx0 = blockIdx.x * HT_TW;
    const T *x = (const T *)P.x + (size_t)b * P.ih * P.iw * 128;

    // ---- upsampled activations of the tile + halo; outside the image = the convolution's zero padding -------------------
    for (int i0 = tid; i0 < HT_NPIX * 16; i0 += 256 * 4) {
        T a[4][8], bq[4][8], cq[4][8], d[4][8];
        float w00[4], w01[4], w10[4], w11[4];
        bool inside[4], live[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {                       // issue all 16 loads of four items before using any
            const int i = i0 + u * 256;
            live[u] = i < HT_NPIX * 16;
            const int pix = min(i, HT_NPIX * 16 - 1) >> 4, chunk = i & 15;
            const int r = pix / HT_PW, c = pix - r * HT_PW;
            const int oy = ty0 - 1 + r, ox = tx0 - 1 + c;
            inside[u] = live[u] && oy >= 0 && oy < P.oh && ox >= 0 && ox < P.ow;
            const float fy = P.sy * max(oy, 0), fx = P.sx * max(ox, 0);
            const int y0 = min((int)fy, P.ih - 1), x0 = min((int)fx, P.iw - 1);
            const int y1 = min(y0 + 1, P.ih - 1), x1 = min(x0 + 1, P.iw - 1);
            const float ty = fy - y0, tx = fx - x0;
            w00[u] = (1.f - ty) * (1.f - tx); w01[u] = (1.f - ty) * tx; w10[u] = ty * (1.f - tx); w11[u] = ty * tx;
            __builtin_memcpy(a[u], x + ((size_t)y0 * P.iw + x0) * 128 + chunk * 8, 16);
            __builtin_memcpy(bq[u], x + ((size_t)y0 * P.iw + x1) * 128 + chunk * 8, 16);
            __builtin_memcpy(cq[u], x + ((size_t)y1 * P.iw + x0) * 128 + chunk * 8, 16);
            __builtin_memcpy(d[u], x + ((size_t)y1 * P.iw + x1) * 128 + chunk * 8, 16);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!live[u]) continue;
            const int i = i0 + u * 256;
            const int pix = i >> 4, chunk = i & 15;
            T o[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                o[k] = inside[u] ? (T)(w00[u] * (float)a[u][k] + w01[u] * (float)bq[u][k] + w10[u] * (float)cq[u][k] + w11[u] * (float)d[u][k])
                                 : (T)0.f;
            __builtin_memcpy(s_act + pix * 256 + ((chunk ^ (pix & 15)) << 4), o, 16);
        }
    }
    __syncthreads();

    // ---- implicit GEMM: D[co][pixel] += W[co][tap, ci] * act[pixel + tap][ci]; this wave: output rows RPW*wave .. +RPW-1 ----
    ht_f32x16 acc[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; rr++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[rr][r] = 0.f;
    const T *wf = (const T *)P.wfrag + ((size_t)hi * 32 + l31) * 8;
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            uint4 wraw = *reinterpret_cast<const uint4 *>(wf + (size_t)(tap * 8 + s) * 2 * 32 * 8);
#pragma unroll
            for (int rr = 0; rr < RPW; rr++) {
                const int pix = (RPW * wave + rr + dy) * HT_PW + l31 + dx;
                const uint4 araw = *reinterpret_cast<const uint4 *>(s_act + pix * 256 + (((2 * s + hi) ^ (pix & 15)) << 4));
                if (BF16) {
                    union { uint4 u; ht_bf16x8 v; } wa, ab; wa.u = wraw; ab.u = araw;
                    acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa.v, ab.v, acc[rr], 0, 0, 0);
                } else {
                    union { uint4 u; ht_f16x8 v; } wa, ab; wa.u = wraw; ab.u = araw;
                    acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa.v, ab.v, acc[rr], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue: lane holds channels crow(r, hi) of pixel l31: + bias, ReLU, dot with the 1x1 weights, + bias, ReLU ------
#pragma unroll
    for (int rr = 0; rr < RPW; rr++) {
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ch = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float v = fmaxf(acc[rr][r] + P.b2[ch], 0.f);
            part += P.w3[ch] * v;
        }
        part += __shfl_xor(part, 32, 64);
        float res = part + P.b3;
        if (P.relu_out) res = fmaxf(res, 0.f);
        const int oy = ty0 + RPW * wave + rr, ox = tx0 + l31;
        if (hi == 0 && oy < P.oh && ox < P.ow) ((T *)P.out)[((size_t)b * P.oh + oy) * P.ow + ox] = (T)res;
    }
}

// Persistent variant (DS_HEAD_PERSIST=1; prepared at the end of round 2, NOT yet measured -- the default stays the kernel above).
// The kernel above is bound by L2 traffic, not by its MFMAs: per 4 x 32 tile every wave re-reads all 72 weight fragments
// (288 KB per workgroup, 18.9 GB per launch at batch 32) and the 6 x 34 halo tile is gathered with four 16-byte loads per
// upsampled pixel and channel chunk (209 KB per workgroup).  Here ONE workgroup of 8 waves per CU keeps the 73.7 KB of
// weight fragments in LDS for its whole life (lane-linear, conflict-free reads), walks 8 x 32 tiles (10 x 34 halo tile =
// 87 KB: 160.8 KB of the 160 KB = 163 840 B of LDS) in an XCD-aware order, one output row per wave: weight traffic from L2
// drops to 73.7 KB per CU, the halo overhead of the gather from 1.59x to 1.33x.
#define HTP_TH 8
#define HTP_NPIX ((HTP_TH + 2) * HT_PW)
#define HTP_W_BYTES (9 * 8 * 2 * 32 * 8 * 2)

template <int BF16>
__global__ __launch_bounds__(512) void k_dpt_head_tail_p(HeadTailParams P, int tiles_x, int tiles_y, int ntiles, int chunk)
{
    typedef typename eo_traits<BF16>::T T;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_all[];
    unsigned char *s_w = s_all, *s_act = s_all + HTP_W_BYTES;   // weights | [HTP_NPIX][16 chunks of 16 B], chunk ^ (pix & 15)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    for (int i = tid; i < HTP_W_BYTES / 16; i += 512)
        reinterpret_cast<uint4 *>(s_w)[i] = reinterpret_cast<const uint4 *>(P.wfrag)[i];
    for (int orig = blockIdx.x; orig < 8 * chunk; orig += gridDim.x) {
        const int tile = (orig & 7) * chunk + (orig >> 3);         // XCD x walks tiles [x * chunk, (x + 1) * chunk)
        if (tile >= ntiles) continue;                               // (uniform per workgroup)
        const int txi = tile % tiles_x, tyi = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int ty0 = tyi * HTP_TH, tx0 = txi * HT_TW;
        const T *x = (const T *)P.x + (size_t)b * P.ih * P.iw * 128;
        __syncthreads();                                            // the previous tile's fragment reads are done (first pass: the weights are in)
        // ---- upsampled activations of the tile + halo; outside the image = the convolution's zero padding ------------------
        for (int i0 = tid; i0 < HTP_NPIX * 16; i0 += 512 * 4) {
            T a[4][8], bq[4][8], cq[4][8], d[4][8];
            float w00[4], w01[4], w10[4], w11[4];
            bool inside[4], live[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * 512;
                live[u] = i < HTP_NPIX * 16;
                const int pix = min(i, HTP_NPIX * 16 - 1) >> 4, chunk16 = i & 15;
                const int r = pix / HT_PW, c = pix - r * HT_PW;
                const int oy = ty0 - 1 + r, ox = tx0 - 1 + c;
                inside[u] = live[u] && oy >= 0 && oy < P.oh && ox >= 0 && ox < P.ow;
                const float fy = P.sy * max(oy, 0), fx = P.sx * max(ox, 0);
                const int y0 = min((int)fy, P.ih - 1), x0 = min((int)fx, P.iw - 1);
                const int y1 = min(y0 + 1, P.ih - 1), x1 = min(x0 + 1, P.iw - 1);
                const float ty = fy - y0, tx = fx - x0;
                w00[u] = (1.f - ty) * (1.f - tx); w01[u] = (1.f - ty) * tx; w10[u] = ty * (1.f - tx); w11[u] = ty * tx;
                __builtin_memcpy(a[u], x + ((size_t)y0 * P.iw + x0) * 128 + chunk16 * 8, 16);
                __builtin_memcpy(bq[u], x + ((size_t)y0 * P.iw + x1) * 128 + chunk16 * 8, 16);
                __builtin_memcpy(cq[u], x + ((size_t)y1 * P.iw + x0) * 128 + chunk16 * 8, 16);
                __builtin_memcpy(d[u], x + ((size_t)y1 * P.iw + x1) * 128 + chunk16 * 8, 16);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!live[u]) continue;
                const int i = i0 + u * 512;
                const int pix = i >> 4, chunk16 = i & 15;
                T o[8];
#pragma unroll
                for (int k = 0; k < 8; k++)
                    o[k] = inside[u] ? (T)(w00[u] * (float)a[u][k] + w01[u] * (float)bq[u][k] + w10[u] * (float)cq[u][k] + w11[u] * (float)d[u][k])
                                     : (T)0.f;
                __builtin_memcpy(s_act + pix * 256 + ((chunk16 ^ (pix & 15)) << 4), o, 16);
            }
        }
        __syncthreads();
        // ---- implicit GEMM: D[co][pixel] += W[co][tap, ci] * act[pixel + tap][ci]; this wave: output row `wave` ----------------
        ht_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        const unsigned char *wf = s_w + ((size_t)hi * 32 + l31) * 16;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const int pix = (wave + dy) * HT_PW + l31 + dx;
#pragma unroll
            for (int s8 = 0; s8 < 8; s8++) {
                const uint4 wraw = *reinterpret_cast<const uint4 *>(wf + (size_t)(tap * 8 + s8) * 2 * 32 * 16);
                const uint4 araw = *reinterpret_cast<const uint4 *>(s_act + pix * 256 + (((2 * s8 + hi) ^ (pix & 15)) << 4));
                if (BF16) {
                    union { uint4 u; ht_bf16x8 v; } wa, ab; wa.u = wraw; ab.u = araw;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa.v, ab.v, acc, 0, 0, 0);
                } else {
                    union { uint4 u; ht_f16x8 v; } wa, ab; wa.u = wraw; ab.u = araw;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa.v, ab.v, acc, 0, 0, 0);
                }
            }
        }
        // ---- epilogue: lane holds channels crow(r, hi) of pixel l31: + bias, ReLU, dot with the 1x1 weights, + bias, ReLU --------
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ch = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float v = fmaxf(acc[r] + P.b2[ch], 0.f);
            part += P.w3[ch] * v;
        }
        part += __shfl_xor(part, 32, 64);
        float res = part + P.b3;
        if (P.relu_out) res = fmaxf(res, 0.f);
        const int oy = ty0 + wave, ox = tx0 + l31;
        if (hi == 0 && oy < P.oh && ox < P.ow) ((T *)P.out)[((size_t)b * P.oh + oy) * P.ow + ox] = (T)res;
    }
}

// Streaming variant (DS_HEAD_MODE=stream).  The persistent kernel above still spends 2 LDS fragment reads per MFMA (every wave
// re-reads the weights for its single output row: 256 B per cycle and CU wanted, 128 available) and runs the gather of the
// upsampled tile and the convolution one after the other.  Here a workgroup owns a 32-pixel-wide column strip and walks DOWN it:
//   * waves 4-7 are PRODUCERS: four new upsampled rows (34 pixels x 128 channels) per step into a ring of 10 rows in LDS.
//     Producer wave w owns ~9 of the 34 columns: it copies the 4 x 7 source pixels under them into a private staging tile in
//     LDS ONCE (7 global loads per lane, issued one step ahead) and interpolates out of that tile -- the first version loaded
//     the four neighbours of every upsampled pixel straight from memory, 36 16-byte loads per lane and step = 144 KB per step
//     through a 64 B / clock vector L1: 0.81 ms per launch, the L1 port alone cost as much as all the MFMAs;
//   * waves 0-3 are CONSUMERS: wave (kh, rp) = half of the input channels x two output rows.  Its 9 x 4 weight fragments live
//     in REGISTERS for the whole launch (144 VGPRs), an activation fragment read serves up to three taps (the three rows it
//     is a neighbour of): 48 fragment reads per 72 MFMAs, every one of them  row base + immediate.  The two channel halves
//     exchange one accumulator each through LDS (16 KB per step) and each finishes ONE row (bias, ReLU, 1x1 convolution,
//     ReLU) at the start of the next step;
//   * ONE barrier per step of four output rows; both roles run between the same two barriers on different rows of the ring
//     (step t reads rows 4t-1 .. 4t+4 while rows 4t+5 .. 4t+8 are produced), so the vector work of the upsample sits beside
//     the matrix work instead of in front of it, and the halo in y disappears.
// Needs an upsample by at least ~1.6 (3 sy < 2, 8 sx < 5: the source pixels under 4 rows x 9 columns fit the 4 x 7 staging tile);
// anything else takes the persistent kernel.
#define HTS_RING 10
#define HTS_PIXB 272              // bytes per pixel in the ring: 256 + 16, so that consecutive pixels start one 16-byte slot apart and
                                  // the fragment reads (32 consecutive pixels, one chunk) are conflict-free WITHOUT an address swizzle:
                                  // every read of a step is  row base + immediate(dx * 272 + chunk * 16)
#define HTS_ROWB (HT_PW * HTS_PIXB)
#define HTS_ACT_BYTES (HTS_RING * HTS_ROWB)
#define HTS_PART_BYTES (4 * 4 * 64 * 16)
#define HTS_SRC_ROWS 4
#define HTS_SRC_COLS 7
#define HTS_SRC_LOADS (HTS_SRC_ROWS * HTS_SRC_COLS * 16 / 64)          // 16-byte loads per lane for one staging tile
#define HTS_STAGE_BYTES (HTS_SRC_ROWS * HTS_SRC_COLS * 256)
#define HTS_LDS_BYTES (HTS_ACT_BYTES + 2 * HTS_PART_BYTES + 256 + 4 * HTS_STAGE_BYTES)
#define HTS_NIT 9                 // columns per producer wave (9, 9, 8, 8)

__device__ __forceinline__ void hts_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// what a producer wave knows about its columns of the strip (constant while the workgroup walks down the strip; wave-uniform)
struct HtsCols {
    int cx0[HTS_NIT], cx1[HTS_NIT];             // byte offsets (256 x column) of the two source columns inside the staging tile
    float wx0[HTS_NIT], wx1[HTS_NIT];           // their weights; both 0 outside the image (the convolution's zero padding)
    int xlo;                                    // first source column of the staging tile
};

// out = w00 a + w01 b + w10 c + w11 d for the 8 halves of four 16-byte vectors, float32 accumulation, ONE rounding (what
// torch's upsample does on half tensors).  f16: v_fma_mix_f32 / v_fma_mixlo_f16 / v_fma_mixhi_f16 read the halves in place.
template <int BF16>
__device__ __forceinline__ uint4 hts_lerp4(const uint4 &a, const uint4 &b, const uint4 &c, const uint4 &d, float w00, float w01, float w10, float w11)
{
    uint4 o;
    if (BF16) {
        typedef typename eo_traits<BF16>::T T;
        union { uint4 q; T h[8]; } A, B, C, D, O;
        A.q = a; B.q = b; C.q = c; D.q = d;
#pragma unroll
        for (int k = 0; k < 8; k++)
            O.h[k] = (T)__builtin_fmaf(w11, (float)D.h[k], __builtin_fmaf(w10, (float)C.h[k], __builtin_fmaf(w01, (float)B.h[k], w00 * (float)A.h[k])));
        o = O.q;
    } else {
        const uint32_t av[4] = { a.x, a.y, a.z, a.w }, bv[4] = { b.x, b.y, b.z, b.w }, cv[4] = { c.x, c.y, c.z, c.w }, dv[4] = { d.x, d.y, d.z, d.w };
        uint32_t ov[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float lo, hi2;
            uint32_t r;
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(av[k]), "v"(w00));
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi2) : "v"(av[k]), "v"(w00));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(bv[k]), "v"(w01));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi2) : "v"(bv[k]), "v"(w01));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(cv[k]), "v"(w10));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi2) : "v"(cv[k]), "v"(w10));
            asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(dv[k]), "v"(w11), "v"(lo));      // (the high half is written next)
            asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(dv[k]), "v"(w11), "v"(hi2));
            ov[k] = r;
        }
        o = make_uint4(ov[0], ov[1], ov[2], ov[3]);
    }
    return o;
}

// first source row under the gather whose first output row is jfirst (relative to the segment's first output row)
__device__ __forceinline__ int hts_ylo(const HeadTailParams &P, int y0seg, int jfirst)
{
    return min((int)(P.sy * max(y0seg + jfirst, 0)), P.ih - 1);
}

// the 4 x 7 source pixels under a producer wave's part of a gather -> registers: load m of a lane is 16-byte slot lane + 64 m of
// the staging tile [row][column][16 chunks].  (Seven named registers, not an array: carried across the step loop and its
// barriers an array stays in scratch memory.)
struct HtsSrc { uint4 r0, r1, r2, r3, r4, r5, r6; };
__device__ __forceinline__ uint4 hts_src_load1(const HeadTailParams &P, const unsigned char *xb, int lane, int ylo, int xlo, int m)
{
    const int id = lane + 64 * m, chunk = id & 15, pixel = id >> 4;
    const int r = pixel / HTS_SRC_COLS, cx = pixel - r * HTS_SRC_COLS;
    const int y = min(ylo + r, P.ih - 1), x = min(xlo + cx, P.iw - 1);
    return *reinterpret_cast<const uint4 *>(xb + ((uint32_t)(y * P.iw + x) * 256u + (uint32_t)chunk * 16u));
}
__device__ __forceinline__ HtsSrc hts_src_load(const HeadTailParams &P, const unsigned char *xb, int lane, int ylo, int xlo)
{
    HtsSrc R;
    R.r0 = hts_src_load1(P, xb, lane, ylo, xlo, 0); R.r1 = hts_src_load1(P, xb, lane, ylo, xlo, 1);
    R.r2 = hts_src_load1(P, xb, lane, ylo, xlo, 2); R.r3 = hts_src_load1(P, xb, lane, ylo, xlo, 3);
    R.r4 = hts_src_load1(P, xb, lane, ylo, xlo, 4); R.r5 = hts_src_load1(P, xb, lane, ylo, xlo, 5);
    R.r6 = hts_src_load1(P, xb, lane, ylo, xlo, 6);
    return R;
}

// producer wave: rows jfirst .. jfirst + nrows - 1 (nrows = 4, or 2 for the segment's first two rows; -1 = the halo row above the
// segment), columns cb .. cb + ncols - 1 of the strip's upsampled activations -> ring slots (j + 1) % HTS_RING.  R holds the
// source tile of THIS gather (hts_src_load, issued a step ago); it goes to the wave's staging tile, the loads of the NEXT
// gather (first row jnext) are issued into R, and the 16 lanes x 4 rows interpolate out of LDS:
// lane = 16 q + chunk owns row jfirst + q and the 16-byte channel chunk.  LDS operations of one wave execute in order, so the
// tile needs no barrier: only this wave touches it.
template <int BF16>
__device__ __forceinline__ void hts_produce(const HeadTailParams &P, const unsigned char *xb, unsigned char *s_act, unsigned char *s_stage,
                                            const HtsCols &X, int lane, int y0seg, int jfirst, int nrows, int cb, int ncols,
                                            HtsSrc &R, int jnext)
{
    static_assert(HTS_SRC_LOADS == 7, "HtsSrc holds seven loads");
    uint4 *st = reinterpret_cast<uint4 *>(s_stage) + lane;
    st[0] = R.r0; st[64] = R.r1; st[128] = R.r2; st[192] = R.r3; st[256] = R.r4; st[320] = R.r5; st[384] = R.r6;
    const int ylo = hts_ylo(P, y0seg, jfirst);
    R = hts_src_load(P, xb, lane, hts_ylo(P, y0seg, jnext), X.xlo);         // unconditional: the addresses are clamped
    const int chunk = lane & 15, q = lane >> 4;
    const int oy = y0seg + jfirst + q;
    const bool inside_y = oy >= 0 && oy < P.oh;
    const float fy = P.sy * max(oy, 0);
    const int y0 = min((int)fy, P.ih - 1), y1 = min(y0 + 1, P.ih - 1);
    const float ty = fy - y0;
    const float wy0 = inside_y ? 1.f - ty : 0.f, wy1 = inside_y ? ty : 0.f;
    const unsigned char *s0 = s_stage + (y0 - ylo) * (HTS_SRC_COLS * 256) + chunk * 16;
    const unsigned char *s1 = s_stage + (y1 - ylo) * (HTS_SRC_COLS * 256) + chunk * 16;
    const int slot = (jfirst + 1 + q) % HTS_RING;
    unsigned char *dst = s_act + (slot * HT_PW + cb) * HTS_PIXB + chunk * 16;
    if (q < nrows) {
        // the four source reads of item k + 3 are issued before item k is interpolated: a producer wave is alone on its SIMD
        // for vector work, so nothing else hides the LDS latency (the consumers keep the LDS busy)
        constexpr int DEPTH = 3;
        uint4 ra[DEPTH], rb[DEPTH], rc[DEPTH], rd[DEPTH];
#define HTS_RD(k) do { ra[(k) % DEPTH] = *reinterpret_cast<const uint4 *>(s0 + X.cx0[k]); rb[(k) % DEPTH] = *reinterpret_cast<const uint4 *>(s0 + X.cx1[k]);   \
                       rc[(k) % DEPTH] = *reinterpret_cast<const uint4 *>(s1 + X.cx0[k]); rd[(k) % DEPTH] = *reinterpret_cast<const uint4 *>(s1 + X.cx1[k]); } while (0)
#pragma unroll
        for (int k = 0; k < DEPTH; k++) HTS_RD(k);
#pragma unroll
        for (int k = 0; k < HTS_NIT; k++) {
            const uint4 o = hts_lerp4<BF16>(ra[k % DEPTH], rb[k % DEPTH], rc[k % DEPTH], rd[k % DEPTH],
                                            wy0 * X.wx0[k], wy0 * X.wx1[k], wy1 * X.wx0[k], wy1 * X.wx1[k]);
            if (k + DEPTH < HTS_NIT) HTS_RD(k + DEPTH);
            if (k < ncols) *reinterpret_cast<uint4 *>(dst + k * HTS_PIXB) = o;
        }
#undef HTS_RD
    }
}

// VARIANT 1: the previous step's row is finished between the MFMA groups (own accumulator copied out: 16 registers), fragment reads one
// group ahead, the order pinned with sched_barrier.  VARIANT 0: the same stream without the pins (the compiler's own order).
template <int BF16, int VARIANT>
__global__ __launch_bounds__(512) void k_dpt_head_tail_s(HeadTailParams P, int strips_x, int nseg, int seg_rows, int nitems, int dbg)
{
    typedef typename eo_traits<BF16>::T T;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_all[];
    unsigned char *s_act = s_all;                                        // ring: [HTS_RING rows][34 pixels][HTS_PIXB bytes: 16 chunks of 16 B + 16 B of padding]
    unsigned char *s_part = s_all + HTS_ACT_BYTES;                       // [2][wave 4][quad 4][lane 64] 16 B: the accumulator a wave hands to its partner
    float *s_cst = reinterpret_cast<float *>(s_all + HTS_ACT_BYTES + 2 * HTS_PART_BYTES);   // b2[32], w3[32]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    if (tid < 32) { s_cst[tid] = P.b2[tid]; s_cst[32 + tid] = P.w3[tid]; }           // visible after the first barrier
    const int per_img = strips_x * nseg;

    if (wave >= 4) {                                                     // ---------------- producers ----------------
        const int w4 = wave - 4;
        const int cb = w4 == 0 ? 0 : (w4 == 1 ? 9 : (w4 == 2 ? 18 : 26)), ncols = w4 < 2 ? 9 : 8;
        unsigned char *s_stage = s_all + HTS_ACT_BYTES + 2 * HTS_PART_BYTES + 256 + w4 * HTS_STAGE_BYTES;
        HtsSrc R;
        for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
            const int b = item / per_img, rem = item - b * per_img, sxi = rem / nseg, seg = rem - sxi * nseg;
            const int y0seg = seg * seg_rows, tx0 = sxi * HT_TW;
            const int nsteps = (min(seg_rows, P.oh - y0seg) + 3) >> 2;
            const unsigned char *xb = (const unsigned char *)P.x + (size_t)b * P.ih * P.iw * 256;
            HtsCols X;
            X.xlo = min((int)(P.sx * max(tx0 - 1 + cb, 0)), P.iw - 1);
#pragma unroll
            for (int k = 0; k < HTS_NIT; k++) {
                const int ox = tx0 - 1 + cb + k;
                const bool inside_x = k < ncols && ox >= 0 && ox < P.ow;
                const float fx = P.sx * max(ox, 0);
                const int x0 = min((int)fx, P.iw - 1), x1 = min(x0 + 1, P.iw - 1);
                const float tx = fx - x0;
                X.cx0[k] = k < ncols ? (x0 - X.xlo) * 256 : 0;           // (item 8 of the 8-column waves is read, then dropped)
                X.cx1[k] = k < ncols ? (x1 - X.xlo) * 256 : 0;
                X.wx0[k] = inside_x ? 1.f - tx : 0.f;
                X.wx1[k] = inside_x ? tx : 0.f;
            }
            // step t reads rows 4t - 1 .. 4t + 4; rows -1, 0 and 1 .. 4 before the first step, rows 4t + 5 .. 4t + 8 during step t
            R = hts_src_load(P, xb, lane, hts_ylo(P, y0seg, -1), X.xlo);
            hts_produce<BF16>(P, xb, s_act, s_stage, X, lane, y0seg, -1, 2, cb, ncols, R, 1);
            hts_produce<BF16>(P, xb, s_act, s_stage, X, lane, y0seg, 1, 4, cb, ncols, R, 5);
            hts_barrier();
            for (int t = 0; t < nsteps; t++) {
                if (t + 1 < nsteps && !(dbg & 1)) hts_produce<BF16>(P, xb, s_act, s_stage, X, lane, y0seg, 4 * t + 5, 4, cb, ncols, R, 4 * t + 9);
                hts_barrier();
            }
        }
        return;
    }

    // ---------------- consumers: wave (kh, rp): input channels [64 kh, 64 kh + 64), output rows 2 rp, 2 rp + 1 of a step ----------------
    const int kh = wave & 1, rp = wave >> 1;
    uint4 wreg[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int s = 0; s < 4; s++)
            wreg[tap][s] = reinterpret_cast<const uint4 *>(P.wfrag)[((size_t)(tap * 8 + 4 * kh + s) * 2 + hi) * 32 + l31];
    ht_f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int b = item / per_img, rem = item - b * per_img, sxi = rem / nseg, seg = rem - sxi * nseg;
        const int y0seg = seg * seg_rows, tx0 = sxi * HT_TW;
        const int yend = min(y0seg + seg_rows, P.oh);
        const int nsteps = (yend - y0seg + 3) >> 2;
        hts_barrier();                                                   // the producers' rows -1 .. 4 are in the ring
        const int ox = tx0 + l31;
#define HTS_MFMA(ACC, TAP, S, BQ) do {                                                                                  \
            if (BF16) { union { uint4 u; ht_bf16x8 v; } wa, ab; wa.u = wreg[TAP][S]; ab.u = BQ;                          \
                        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa.v, ab.v, ACC, 0, 0, 0); }                        \
            else { union { uint4 u; ht_f16x8 v; } wa, ab; wa.u = wreg[TAP][S]; ab.u = BQ;                                 \
                   ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa.v, ab.v, ACC, 0, 0, 0); }                              \
        } while (0)
#define HTS_FIN_QUAD(ACC, q) do {               /* channels (r & 3) + 8 (r >> 2) + 4 hi; the bias went in with the accumulator's initial value */ \
            const float4 o = *reinterpret_cast<const float4 *>(pp + (q) * 1024);                                            \
            const float4 ww = *reinterpret_cast<const float4 *>(s_cst + 32 + 8 * (q) + 4 * hi);                             \
            part = __builtin_fmaf(ww.x, fmaxf(ACC[4 * (q) + 0] + o.x, 0.f), part);                                         \
            part = __builtin_fmaf(ww.y, fmaxf(ACC[4 * (q) + 1] + o.y, 0.f), part);                                         \
            part = __builtin_fmaf(ww.z, fmaxf(ACC[4 * (q) + 2] + o.z, 0.f), part);                                         \
            part = __builtin_fmaf(ww.w, fmaxf(ACC[4 * (q) + 3] + o.w, 0.f), part);                                         \
        } while (0)
        for (int t = 0; t < nsteps; t++) {
            // Row 2 rp + kh of step t - 1: own accumulator + the partner's (the other channel half, written before the last
            // barrier).  At t = 0 the same code runs on stale values and its store is switched off: no branch in the stream.
            const unsigned char *pp = s_part + ((t + 1) & 1) * HTS_PART_BYTES + (size_t)((wave ^ 1) * 4) * 1024 + lane * 16;   // (t - 1) & 1
            float part = 0.f;
            // activation rows 4t - 1 + 2 rp + i, i = 0 .. 3, live in ring slots (4t + 2 rp + i) % HTS_RING
            const int sb = (4 * t + 2 * rp) % HTS_RING;
            const unsigned char *rowp[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int sl = sb + i; sl = sl >= HTS_RING ? sl - HTS_RING : sl;
                rowp[i] = s_act + (sl * HT_PW + l31) * HTS_PIXB + (8 * kh + hi) * 16;       // chunk 2 (4 kh + s) + hi = (8 kh + hi) + 2 s
            }
            // 12 groups g = (dx, s) of 4 fragment reads + 6 MFMAs; byte offset of group g: (g >> 2) * HTS_PIXB + (g & 3) * 32
            constexpr int NBUF = 2;              // (three buffers = reads two groups ahead does not fit: the weights start to spill)
            uint4 bq[NBUF][4];
            ht_f32x16 fin;
            fin = kh ? acc1 : acc0;
#pragma unroll
            for (int i = 0; i < 4; i++) bq[0][i] = *reinterpret_cast<const uint4 *>(rowp[i]);
            // the accumulator of the row this wave finishes starts at the 3x3 convolution's bias, the other one at zero
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 bv = *reinterpret_cast<const float4 *>(s_cst + 8 * q + 4 * hi);
                acc0[4 * q + 0] = kh ? 0.f : bv.x; acc0[4 * q + 1] = kh ? 0.f : bv.y; acc0[4 * q + 2] = kh ? 0.f : bv.z; acc0[4 * q + 3] = kh ? 0.f : bv.w;
                acc1[4 * q + 0] = kh ? bv.x : 0.f; acc1[4 * q + 1] = kh ? bv.y : 0.f; acc1[4 * q + 2] = kh ? bv.z : 0.f; acc1[4 * q + 3] = kh ? bv.w : 0.f;
            }
#pragma unroll
            for (int g = 0; g < 12; g++) {
                if (dbg & 2) break;
                const int dx = g >> 2, s4 = g & 3, cur = g % NBUF, gn = g + NBUF - 1;
                if (gn < 12) {
#pragma unroll
                    for (int i = 0; i < 4; i++) bq[gn % NBUF][i] = *reinterpret_cast<const uint4 *>(rowp[i] + (gn >> 2) * HTS_PIXB + (gn & 3) * 32);
                }
                if (VARIANT == 1) __builtin_amdgcn_sched_barrier(0);
                HTS_MFMA(acc0, 0 * 3 + dx, s4, bq[cur][0]);          // output row 2 rp:     taps dy = 0, 1, 2 on activation rows i = 0, 1, 2
                HTS_MFMA(acc1, 0 * 3 + dx, s4, bq[cur][1]);          // output row 2 rp + 1: taps dy = 0, 1, 2 on activation rows i = 1, 2, 3
                HTS_MFMA(acc0, 1 * 3 + dx, s4, bq[cur][1]);
                HTS_MFMA(acc1, 1 * 3 + dx, s4, bq[cur][2]);
                HTS_MFMA(acc0, 2 * 3 + dx, s4, bq[cur][2]);
                HTS_MFMA(acc1, 2 * 3 + dx, s4, bq[cur][3]);
                if ((g & 1) && g < 8) HTS_FIN_QUAD(fin, g >> 1);
                if (VARIANT == 1) __builtin_amdgcn_sched_barrier(0);
            }
            {
                part += __shfl_xor(part, 32, 64);
                float res = part + P.b3;
                if (P.relu_out) res = fmaxf(res, 0.f);
                const int oy = y0seg + 4 * (t - 1) + 2 * rp + kh;
                if (t > 0 && hi == 0 && oy < yend && ox < P.ow) ((T *)P.out)[((size_t)b * P.oh + oy) * P.ow + ox] = (T)res;
            }
            // hand the accumulator of the row this wave does NOT finish to its partner
            {
                unsigned char *pw = s_part + (t & 1) * HTS_PART_BYTES + (size_t)(wave * 4) * 1024 + lane * 16;
                const ht_f32x16 give = kh ? acc0 : acc1;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    *reinterpret_cast<float4 *>(pw + q * 1024) = make_float4(give[4 * q], give[4 * q + 1], give[4 * q + 2], give[4 * q + 3]);
            }
            hts_barrier();
        }
        {   // the last step's row
            const ht_f32x16 fin = kh ? acc1 : acc0;
            const unsigned char *pp = s_part + ((nsteps - 1) & 1) * HTS_PART_BYTES + (size_t)((wave ^ 1) * 4) * 1024 + lane * 16;
            float part = 0.f;
            HTS_FIN_QUAD(fin, 0); HTS_FIN_QUAD(fin, 1); HTS_FIN_QUAD(fin, 2); HTS_FIN_QUAD(fin, 3);
            part += __shfl_xor(part, 32, 64);
            float res = part + P.b3;
            if (P.relu_out) res = fmaxf(res, 0.f);
            const int oy = y0seg + 4 * (nsteps - 1) + 2 * rp + kh;
            if (hi == 0 && oy < yend && ox < P.ow) ((T *)P.out)[((size_t)b * P.oh + oy) * P.ow + ox] = (T)res;
        }
#undef HTS_MFMA
#undef HTS_FIN_QUAD
    }
}

DS_API int ds_dpt_head_tail(ds_ctx *ctx, const void *x, int batch, int in_h, int in_w, int out_h, int out_w,
                            const void *conv3_wfrag, const float *conv3_bias, const float *conv1_weight, float conv1_bias,
                            int relu_out, void *out, int dtype, void *stream)
{
    DS_REQUIRE(ctx && x && conv3_wfrag && conv3_bias && conv1_weight && out, DS_EINVAL, "ds_dpt_head_tail: null argument");
    DS_REQUIRE(batch > 0 && batch <= 65535 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0, DS_EINVAL, "ds_dpt_head_tail: bad shape");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_dpt_head_tail: dtype must be f16 or bf16");
    DS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)conv3_wfrag & 15) == 0, DS_EINVAL, "ds_dpt_head_tail: 16-byte alignment");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    HeadTailParams P;
    P.x = x; P.wfrag = conv3_wfrag; P.b2 = conv3_bias; P.w3 = conv1_weight; P.b3 = conv1_bias; P.out = out;
    P.B = batch; P.ih = in_h; P.iw = in_w; P.oh = out_h; P.ow = out_w; P.relu_out = relu_out;
    P.sy = out_h > 1 ? (float)(in_h - 1) / (float)(out_h - 1) : 0.f;
    P.sx = out_w > 1 ? (float)(in_w - 1) / (float)(out_w - 1) : 0.f;
    // The streaming kernel (column strips, producer / consumer waves) is the default since round 3: 1.69 -> 0.72 ms at 32 x 256^2 ->
    // 512^2 (profiles/round3_head_tail_ab.txt).  DS_HEAD_MODE=persist / tile select the older kernels (A/B runs; persist is also what
    // an upsample by less than ~1.6 gets); DS_HEAD_PERSIST=0 is the round-2 spelling of tile.  The switches are read per call.
    int mode = 2;
    { const char *e = getenv("DS_HEAD_PERSIST"); if (e && atoi(e) == 0) mode = 0; }
    { const char *e = getenv("DS_HEAD_MODE"); if (e) mode = !strcmp(e, "stream") ? 2 : (!strcmp(e, "tile") ? 0 : (!strcmp(e, "persist") ? 1 : mode)); }
    if (mode == 2 && !(3.f * P.sy < 1.99f && 8.f * P.sx < 4.99f)) mode = 1;      // the staging tile of the streaming kernel: 4 x 7 source pixels
    if (mode == 2) {
        int ncu = 0, dev = 0;
        DS_HIP_CHECK(hipGetDevice(&dev));
        DS_HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        const int strips_x = (out_w + HT_TW - 1) / HT_TW;
        // segments of a strip: as long as possible (1.5 producer steps of lead-in each), but at least ~4 work items per CU
        int seg_rows = (out_h + 3) & ~3;
        while ((long long)strips_x * batch * ((out_h + seg_rows - 1) / seg_rows) < 4ll * ncu && seg_rows > 16) seg_rows = ((seg_rows / 2) + 3) & ~3;
        { const char *e = getenv("DS_HEAD_SEG"); if (e && atoi(e) >= 4) seg_rows = (atoi(e) + 3) & ~3; }
        const int nseg = (out_h + seg_rows - 1) / seg_rows;
        const long long nitems = (long long)strips_x * nseg * batch;
        DS_REQUIRE(nitems < (1ll << 30), DS_EUNSUPPORTED, "ds_dpt_head_tail: too many work items");
        int dbg = 0;                                        // timing ablations (WRONG results): -DDS_EXPERIMENTS builds only
#ifdef DS_EXPERIMENTS
        { const char *e = getenv("DS_HEAD_ABLATE"); if (e) dbg = atoi(e); }        // 1: no steady-state producer work, 2: no MFMAs
#endif
        const int grid = (int)std::min<long long>(ncu, nitems);
        int variant = 1;                                    // pinned consumer order: consumers alone 0.51 vs 0.58 ms, the whole kernel 0.725 vs 0.732
        { const char *e = getenv("DS_HEAD_VARIANT"); if (e) variant = atoi(e) == 0 ? 0 : 1; }        // A/B switch, read per call
#define HTS_LAUNCH(BF, V) do {                                                                                                                              \
            DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dpt_head_tail_s<BF, V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HTS_LDS_BYTES)); \
            hipLaunchKernelGGL((k_dpt_head_tail_s<BF, V>), dim3(grid), dim3(512), HTS_LDS_BYTES, (hipStream_t)stream, P, strips_x, nseg, seg_rows, (int)nitems, dbg);    \
        } while (0)
        if (dtype == DS_DTYPE_F16) { if (variant) HTS_LAUNCH(0, 1); else HTS_LAUNCH(0, 0); }
        else { if (variant) HTS_LAUNCH(1, 1); else HTS_LAUNCH(1, 0); }
#undef HTS_LAUNCH
        DS_HIP_CHECK(hipGetLastError());
        return DS_OK;
    }
    if (mode == 1) {
        const int tiles_x = (out_w + HT_TW - 1) / HT_TW, tiles_y = (out_h + HTP_TH - 1) / HTP_TH;
        const long long nt = (long long)tiles_x * tiles_y * batch;
        DS_REQUIRE(nt < (1ll << 30), DS_EUNSUPPORTED, "ds_dpt_head_tail: too many tiles");
        int ncu = 0, dev = 0;
        DS_HIP_CHECK(hipGetDevice(&dev));
        DS_HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        ncu = ncu >= 8 ? ncu / 8 * 8 : 8;
        const int chunk = (int)((nt + 7) / 8);
        const int grid = (int)std::min<long long>(ncu, 8ll * chunk);
        const size_t ldsp = (size_t)HTP_W_BYTES + (size_t)HTP_NPIX * 256;
        if (dtype == DS_DTYPE_F16) {
            DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dpt_head_tail_p<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp));
            hipLaunchKernelGGL((k_dpt_head_tail_p<0>), dim3(grid), dim3(512), ldsp, (hipStream_t)stream, P, tiles_x, tiles_y, (int)nt, chunk);
        } else {
            DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dpt_head_tail_p<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp));
            hipLaunchKernelGGL((k_dpt_head_tail_p<1>), dim3(grid), dim3(512), ldsp, (hipStream_t)stream, P, tiles_x, tiles_y, (int)nt, chunk);
        }
        DS_HIP_CHECK(hipGetLastError());
        return DS_OK;
    }
    static int s_rpw = 0;                                    // DS_HEAD_RPW: output rows per wave (tile height = 4x), 1 or 2
    if (s_rpw == 0) { const char *e = getenv("DS_HEAD_RPW"); s_rpw = (e && atoi(e) == 2) ? 2 : 1; }
    const int th = 4 * s_rpw;
    const size_t lds = (size_t)(th + 2) * HT_PW * 256;
    dim3 grid((out_w + HT_TW - 1) / HT_TW, (out_h + th - 1) / th, batch);
    DS_REQUIRE(grid.y <= 65535, DS_EUNSUPPORTED, "ds_dpt_head_tail: image too tall");
#define HT_LAUNCH(BF, RP) do {                                                                                              \
        DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dpt_head_tail<BF, RP>),                           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
        hipLaunchKernelGGL((k_dpt_head_tail<BF, RP>), grid, dim3(256), lds, (hipStream_t)stream, P);                        \
    } while (0)
    if (dtype == DS_DTYPE_F16) { if (s_rpw == 2) HT_LAUNCH(0, 2); else HT_LAUNCH(0, 1); }
    else { if (s_rpw == 2) HT_LAUNCH(1, 2); else HT_LAUNCH(1, 1); }
#undef HT_LAUNCH
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}


// ---- ds_reassemble_readout: the read-out of the DPT reassemble stage (dmidas/backbones/utils.py:28-39) --------------------
// The reference builds cat(tokens[:, 1:], cls expanded) [B, N-1, 2C], runs Linear(2C -> C) and GELU.  The linear map of a
// concatenation splits:  W.[tok ; cls] + b = W_tok.tok + (W_cls.cls + b), and the second term is ONE vector per image.  The
// host runs the token GEMM on the tap as it is (K = C instead of 2C, no concatenated copy) and the tiny cls GEMM; this
// kernel is the epilogue: out[b, n-1, :] = gelu_erf(proj[b, n, :] + clsvec[b, :]) for n = 1 .. N-1, written densely as the
// [B, N-1, C] = [B, h, w, C] (NHWC) operand of the 1x1 convolution that follows -- bias add, exact (erf) GELU, the drop of
// the cls row and the token -> image reshape in one pass at HBM speed.
template <int BF16>
__global__ __launch_bounds__(256) void k_reassemble_readout(const void *proj_, const void *cls_, void *out_, int N, int C8, long long total8)
{
    typedef typename eo_traits<BF16>::T T;
    const T *proj = (const T *)proj_, *cls = (const T *)cls_;
    T *out = (T *)out_;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        const long long row = i / C8;                      // output row = b * (N - 1) + (n - 1)
        const int b = (int)(row / (N - 1)), n1 = (int)(row - (long long)b * (N - 1));
        T pv[8], cv[8], ov[8];
        __builtin_memcpy(pv, proj + (((size_t)b * N + n1 + 1) * C8 + c8) * 8, 16);
        __builtin_memcpy(cv, cls + ((size_t)b * C8 + c8) * 8, 16);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const float x = (float)(T)((float)pv[t] + (float)cv[t]);      // the Linear's output rounded to the storage type, like torch
            ov[t] = (T)(0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)));
        }
        __builtin_memcpy(out + (size_t)i * 8, ov, 16);
    }
}

DS_API int ds_reassemble_readout(ds_ctx *ctx, const void *proj, const void *clsvec, void *out, int batch, int tokens, int channels,
                                 int dtype, void *stream)
{
    DS_REQUIRE(ctx && proj && clsvec && out, DS_EINVAL, "ds_reassemble_readout: null argument");
    DS_REQUIRE(batch > 0 && tokens > 1 && channels > 0 && (channels % 8) == 0, DS_EINVAL,
               "ds_reassemble_readout: need batch > 0, tokens > 1, channels a multiple of 8 (got %d, %d, %d)", batch, tokens, channels);
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_reassemble_readout: dtype must be f16 or bf16");
    DS_REQUIRE((((uintptr_t)proj | (uintptr_t)clsvec | (uintptr_t)out) & 15) == 0, DS_EINVAL, "ds_reassemble_readout: operands must be 16-byte aligned");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    const long long total8 = (long long)batch * (tokens - 1) * (channels / 8);
    const int blocks = (int)std::min<long long>((total8 + 255) / 256, 256 * 32);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DS_DTYPE_F16) hipLaunchKernelGGL((k_reassemble_readout<0>), dim3(blocks), dim3(256), 0, st, proj, clsvec, out, tokens, channels / 8, total8);
    else hipLaunchKernelGGL((k_reassemble_readout<1>), dim3(blocks), dim3(256), 0, st, proj, clsvec, out, tokens, channels / 8, total8);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}


// ---- ds_bias_act_nhwc: the element-wise tail of a decoder convolution ------------------------------------------------------
// out = [relu]( conv_out + bias[c] [+ res1] [+ res2] ) over a channels-last activation, one pass.  The residual convolution
// units of the DPT decoders (dmidas/blocks.py:352-377, ddepth_anything_v2/.../util/blocks.py:56-85, and the skip add of
// the fusion blocks :427 / :135) are  conv -> +bias -> ReLU -> conv -> +bias -> +x [-> +skip]: the library convolution is
// called WITHOUT its bias and this kernel does the rest, so a unit makes 3 element-wise passes instead of 5-6.
template <int BF16>
__global__ __launch_bounds__(256) void k_bias_act_nhwc(const void *x_, const void *bias_, const void *r1_, const void *r2_, void *out_,
                                                       long long total8, int C8, int relu)
{
    typedef typename eo_traits<BF16>::T T;
    const T *x = (const T *)x_, *bias = (const T *)bias_, *r1 = (const T *)r1_, *r2 = (const T *)r2_;
    T *out = (T *)out_;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        T xv[8], bv[8], av[8], cv[8], ov[8];
        __builtin_memcpy(xv, x + (size_t)i * 8, 16);
        __builtin_memcpy(bv, bias + (size_t)c8 * 8, 16);
        if (r1) __builtin_memcpy(av, r1 + (size_t)i * 8, 16);
        if (r2) __builtin_memcpy(cv, r2 + (size_t)i * 8, 16);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            float f = (float)xv[t] + (float)bv[t];
            if (r1) f += (float)av[t];
            if (r2) f += (float)cv[t];
            if (relu) f = f > 0.f ? f : 0.f;
            ov[t] = (T)f;
        }
        __builtin_memcpy(out + (size_t)i * 8, ov, 16);
    }
}

DS_API int ds_bias_act_nhwc(ds_ctx *ctx, const void *x, const void *bias, const void *res1, const void *res2, void *out,
                            int64_t elements, int channels, int relu, int dtype, void *stream)
{
    DS_REQUIRE(ctx && x && bias && out, DS_EINVAL, "ds_bias_act_nhwc: null argument");
    DS_REQUIRE(elements > 0 && channels > 0 && (channels % 8) == 0 && (elements % channels) == 0, DS_EINVAL,
               "ds_bias_act_nhwc: elements must be a positive multiple of channels, channels a multiple of 8");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_bias_act_nhwc: dtype must be f16 or bf16");
    DS_REQUIRE((((uintptr_t)x | (uintptr_t)bias | (uintptr_t)res1 | (uintptr_t)res2 | (uintptr_t)out) & 15) == 0, DS_EINVAL,
               "ds_bias_act_nhwc: operands must be 16-byte aligned");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    const long long total8 = elements / 8;
    const int blocks = (int)std::min<long long>((total8 + 255) / 256, 256 * 32);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DS_DTYPE_F16) hipLaunchKernelGGL((k_bias_act_nhwc<0>), dim3(blocks), dim3(256), 0, st, x, bias, res1, res2, out, total8, channels / 8, relu);
    else hipLaunchKernelGGL((k_bias_act_nhwc<1>), dim3(blocks), dim3(256), 0, st, x, bias, res1, res2, out, total8, channels / 8, relu);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

// ---- GroupNorm (+ residual) (+ ReLU) on an NCHW activation (round 6) ------------------------------------------------------------
// The ResNetV2-50 stem of dpt_hybrid_384 (timm's GroupNormAct, 32 groups: dmidas/backbones/vit.py of the reference's timm model) runs
// 52 GroupNorms per image, and at batch 1 -- BASELINE config 2, a latency line -- torch spends three launches on each: RowwiseMoments
// (one workgroup per group = 32 workgroups on 256 CUs: 15.5 us), ComputeFusedParams, the element-wise kernel, then a separate ReLU
// and, behind norm3, a separate add + ReLU: 1.3 ms of a 4.8 ms image.  Here: one launch for the moments -- every (image, group) split
// over up to 32 workgroups, float32 sum / sum of squares per slice -- and one that combines the slices in float64 (mean, var =
// E[x^2] - mean^2 with the population variance, rstd = 1 / sqrt(var + eps) as torch), and applies y = x * (rstd gamma_c) + (beta_c -
// mean rstd gamma_c) [+ res] [relu] in float32, rounded once to the activation's type.
#define GN_MAX_SPLIT 32
#define GN_MAX_NG 4096           // (images x groups) of one call: the moments block is allocated once at this size -- captured graphs hold its address
template <int BF16>
__global__ __launch_bounds__(256) void k_gn_moments(const void *x_, float2 *partial, int group_len, int splits)
{
    typedef typename eo_traits<BF16>::T T;
    const int sp = blockIdx.x, ng = blockIdx.y;              // slice of the group, (image * groups + group)
    const T *x = (const T *)x_ + (size_t)ng * group_len;
    const int vecs = group_len >> 3;
    const int per = (vecs + splits - 1) / splits;
    const int v0 = sp * per, v1 = min(vecs, v0 + per);
    float s = 0.f, q = 0.f;
    for (int v = v0 + threadIdx.x; v < v1; v += 256) {
        T xv[8];
        __builtin_memcpy(xv, x + (size_t)v * 8, 16);
#pragma unroll
        for (int t = 0; t < 8; t++) { const float f = (float)xv[t]; s += f; q = __builtin_fmaf(f, f, q); }
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) { s += __shfl_xor(s, sft, 64); q += __shfl_xor(q, sft, 64); }
    __shared__ float2 red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_float2(s, q);
    __syncthreads();
    if (threadIdx.x == 0) {
        float2 r = red[0];
        for (int k = 1; k < 4; k++) { r.x += red[k].x; r.y += red[k].y; }
        partial[(size_t)ng * GN_MAX_SPLIT + sp] = r;
    }
}

template <int BF16>
__global__ __launch_bounds__(256) void k_gn_apply(const void *x_, const float2 *partial, const void *gamma_, const void *beta_, const void *res_,
                                                  void *out_, int group_len, int splits, int hw, int cpg, int groups, float eps, int relu, int chunks)
{
    typedef typename eo_traits<BF16>::T T;
    const int ck = blockIdx.x, ng = blockIdx.y, g = ng % groups;
    __shared__ float s_stat[2];
    if (threadIdx.x < 64) {
        double s = 0.0, q = 0.0;
        if ((int)threadIdx.x < splits) { const float2 p = partial[(size_t)ng * GN_MAX_SPLIT + threadIdx.x]; s = p.x; q = p.y; }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) { s += __shfl_xor(s, sft, 64); q += __shfl_xor(q, sft, 64); }
        if (threadIdx.x == 0) {
            const double mean = s / (double)group_len;
            double var = q / (double)group_len - mean * mean;
            if (var < 0.0) var = 0.0;
            s_stat[0] = (float)mean;
            s_stat[1] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    const float mean = s_stat[0], rstd = s_stat[1];
    const T *x = (const T *)x_ + (size_t)ng * group_len, *res = res_ ? (const T *)res_ + (size_t)ng * group_len : nullptr;
    const T *gamma = (const T *)gamma_, *beta = (const T *)beta_;
    T *out = (T *)out_ + (size_t)ng * group_len;
    const int vecs = group_len >> 3;
    const int per = (vecs + chunks - 1) / chunks;
    const int v0 = ck * per, v1 = min(vecs, v0 + per);
    for (int v = v0 + threadIdx.x; v < v1; v += 256) {
        const int c = g * cpg + (v * 8) / hw;                // hw % 8 == 0: the 8 values belong to one channel
        const float a = rstd * (float)gamma[c], b = (float)beta[c] - mean * a;
        T xv[8], rv[8], ov[8];
        __builtin_memcpy(xv, x + (size_t)v * 8, 16);
        if (res) __builtin_memcpy(rv, res + (size_t)v * 8, 16);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            float f = __builtin_fmaf((float)xv[t], a, b);
            if (res) f += (float)rv[t];
            if (relu) f = f < 0.f ? 0.f : f;
            ov[t] = (T)f;
        }
        __builtin_memcpy(out + (size_t)v * 8, ov, 16);
    }
}

// One launch when a group fits the registers of one workgroup (1024 threads x up to 10 vectors of 8 values = 81.9 k values: every
// GroupNorm of the ResNetV2 stem at 384 x 384 -- at most 73.7 k): the values are read ONCE, kept packed while the workgroup reduces the
// moments, and written once.  Same arithmetic as the two-launch pair (float32 partial sums per thread and per wave, float64 combine).
#define GN_FUSED_THREADS 1024
#define GN_FUSED_VECS 10
template <int BF16>
__global__ __launch_bounds__(GN_FUSED_THREADS) void k_gn_fused(const void *x_, const void *gamma_, const void *beta_, const void *res_, void *out_,
                                                               int group_len, int hw, int cpg, int groups, float eps, int relu)
{
    typedef typename eo_traits<BF16>::T T;
    const int ng = blockIdx.x, g = ng % groups, tid = threadIdx.x;
    const T *x = (const T *)x_ + (size_t)ng * group_len, *res = res_ ? (const T *)res_ + (size_t)ng * group_len : nullptr;
    const T *gamma = (const T *)gamma_, *beta = (const T *)beta_;
    T *out = (T *)out_ + (size_t)ng * group_len;
    const int vecs = group_len >> 3;
    uint4 keep[GN_FUSED_VECS];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < GN_FUSED_VECS; k++) {
        const int v = tid + k * GN_FUSED_THREADS;
        if (v < vecs) {
            keep[k] = *(const uint4 *)(x + (size_t)v * 8);
            T xv[8];
            __builtin_memcpy(xv, &keep[k], 16);
#pragma unroll
            for (int t = 0; t < 8; t++) { const float f = (float)xv[t]; s += f; q = __builtin_fmaf(f, f, q); }
        }
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) { s += __shfl_xor(s, sft, 64); q += __shfl_xor(q, sft, 64); }
    __shared__ float2 red[GN_FUSED_THREADS / 64];
    __shared__ float s_stat[2];
    if ((tid & 63) == 0) red[tid >> 6] = make_float2(s, q);
    __syncthreads();
    if (tid < 64) {
        double ds = 0.0, dq = 0.0;
        if (tid < GN_FUSED_THREADS / 64) { ds = red[tid].x; dq = red[tid].y; }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) { ds += __shfl_xor(ds, sft, 64); dq += __shfl_xor(dq, sft, 64); }
        if (tid == 0) {
            const double mean = ds / (double)group_len;
            double var = dq / (double)group_len - mean * mean;
            if (var < 0.0) var = 0.0;
            s_stat[0] = (float)mean;
            s_stat[1] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    const float mean = s_stat[0], rstd = s_stat[1];
#pragma unroll
    for (int k = 0; k < GN_FUSED_VECS; k++) {
        const int v = tid + k * GN_FUSED_THREADS;
        if (v < vecs) {
            const int c = g * cpg + (v * 8) / hw;
            const float a = rstd * (float)gamma[c], b = (float)beta[c] - mean * a;
            T xv[8], rv[8], ov[8];
            __builtin_memcpy(xv, &keep[k], 16);
            if (res) __builtin_memcpy(rv, res + (size_t)v * 8, 16);
#pragma unroll
            for (int t = 0; t < 8; t++) {
                float f = __builtin_fmaf((float)xv[t], a, b);
                if (res) f += (float)rv[t];
                if (relu) f = f < 0.f ? 0.f : f;
                ov[t] = (T)f;
            }
            __builtin_memcpy(out + (size_t)v * 8, ov, 16);
        }
    }
}

DS_API int ds_group_norm_nchw(ds_ctx *ctx, const void *x, const void *gamma, const void *beta, const void *res, void *out, int n, int channels,
                              int hw, int groups, float eps, int relu, int dtype, void *stream)
{
    DS_REQUIRE(ctx && x && gamma && beta && out, DS_EINVAL, "ds_group_norm_nchw: null argument");
    DS_REQUIRE(n > 0 && channels > 0 && hw > 0 && groups > 0 && channels % groups == 0, DS_EINVAL, "ds_group_norm_nchw: bad shape n=%d c=%d hw=%d groups=%d", n, channels, hw, groups);
    DS_REQUIRE(hw % 8 == 0, DS_EUNSUPPORTED, "ds_group_norm_nchw: pixels per channel plane must be a multiple of 8 (got %d)", hw);
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_group_norm_nchw: dtype must be f16 or bf16");
    DS_REQUIRE((((uintptr_t)x | (uintptr_t)res | (uintptr_t)out) & 15) == 0, DS_EINVAL, "ds_group_norm_nchw: x, res and out must be 16-byte aligned");
    DS_REQUIRE((long long)n * groups <= GN_MAX_NG, DS_EUNSUPPORTED, "ds_group_norm_nchw: n * groups must be <= %d", GN_MAX_NG);
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    const int cpg = channels / groups;
    const long long group_len_ll = (long long)cpg * hw;
    DS_REQUIRE(group_len_ll < (1ll << 30), DS_EUNSUPPORTED, "ds_group_norm_nchw: a group of %lld values is too large", group_len_ll);
    const int group_len = (int)group_len_ll;
    int splits = group_len / 4096; if (splits < 1) splits = 1; if (splits > GN_MAX_SPLIT) splits = GN_MAX_SPLIT;
    int chunks = group_len / 8192; if (chunks < 1) chunks = 1; if (chunks > 64) chunks = 64;
    const int rc = ds_ctx_reserve(ctx, &ctx->gn_ws, &ctx->gn_ws_bytes, (size_t)GN_MAX_NG * GN_MAX_SPLIT * sizeof(float2));
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    float2 *partial = (float2 *)ctx->gn_ws;
    // few groups of a size one workgroup holds in registers (the batch-1 stem): ONE launch; many or larger groups: moments + apply,
    // every group spread over several workgroups.  DS_GN_FUSED=0: always the pair (A/B runs)
    static const int s_fused = getenv("DS_GN_FUSED") ? atoi(getenv("DS_GN_FUSED")) : 1;
    if (s_fused && (group_len >> 3) <= GN_FUSED_THREADS * GN_FUSED_VECS && n * groups <= 512) {
        if (dtype == DS_DTYPE_F16) hipLaunchKernelGGL((k_gn_fused<0>), dim3(n * groups), dim3(GN_FUSED_THREADS), 0, st, x, gamma, beta, res, out, group_len, hw, cpg, groups, eps, relu);
        else hipLaunchKernelGGL((k_gn_fused<1>), dim3(n * groups), dim3(GN_FUSED_THREADS), 0, st, x, gamma, beta, res, out, group_len, hw, cpg, groups, eps, relu);
        DS_HIP_CHECK(hipGetLastError());
        return DS_OK;
    }
    if (dtype == DS_DTYPE_F16) {
        hipLaunchKernelGGL((k_gn_moments<0>), dim3(splits, n * groups), dim3(256), 0, st, x, partial, group_len, splits);
        hipLaunchKernelGGL((k_gn_apply<0>), dim3(chunks, n * groups), dim3(256), 0, st, x, partial, gamma, beta, res, out, group_len, splits, hw, cpg, groups, eps, relu, chunks);
    } else {
        hipLaunchKernelGGL((k_gn_moments<1>), dim3(splits, n * groups), dim3(256), 0, st, x, partial, group_len, splits);
        hipLaunchKernelGGL((k_gn_apply<1>), dim3(chunks, n * groups), dim3(256), 0, st, x, partial, gamma, beta, res, out, group_len, splits, hw, cpg, groups, eps, relu, chunks);
    }
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}
