// ds_colorize_u16: the heat map of the funnel (reference: src/core.py:271-274 -> dzoedepth/utils/misc.py:97-150,
// colorize(img_output, cmap='inferno') with its defaults).
//
// The reference normalises the uint16 depth with its 2nd / 85th percentiles in float64 (:121-127), maps it through a
// matplotlib colormap (Colormap.__call__ on floats: index = trunc(value * N), values below 0 take the first entry, values
// >= 1 the last, bytes=True table) and returns RGBA uint8.  Here: one pass, 2 bytes in and 4 bytes out per pixel, the
// 256-entry table (1 KB) in LDS; the percentiles are order statistics found on the device by the caller
// (src/video_mode._global_percentiles) and handed over as {vmin, vmax} per image.
#include "ds_common.h"

__global__ __launch_bounds__(256) void k_colorize_u16(const uint16_t *__restrict__ depth, const double *__restrict__ vmin_vmax,
                                                      const uint32_t *__restrict__ lut, uint32_t *__restrict__ out,
                                                      long long hw, int lut_n)
{
    __shared__ uint32_t s_lut[256];
    for (int i = threadIdx.x; i < lut_n; i += blockDim.x) s_lut[i] = lut[i];
    __syncthreads();
    const int img = blockIdx.y;
    const double vmin = vmin_vmax[2 * img], vmax = vmin_vmax[2 * img + 1];
    const bool flat = vmin == vmax;                         // misc.py:124-127: value * 0. instead of a 0/0
    const double span = vmax - vmin, nd = (double)lut_n;
    const uint16_t *d = depth + (size_t)img * hw;
    uint32_t *o = out + (size_t)img * hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long long)gridDim.x * blockDim.x) {
        const double v = flat ? 0.0 : ((double)d[i] - vmin) / span;
        const double x = v * nd;                            // Colormap.__call__: xa *= N; under -> first, over (>= N) -> last
        const int idx = x < 0.0 ? 0 : (x >= nd ? lut_n - 1 : (int)x);
        o[i] = s_lut[idx];
    }
}

DS_API int ds_colorize_u16(ds_ctx *ctx, const uint16_t *depth, int n, int h, int w, const double *vmin_vmax,
                           const uint8_t *lut_rgba, int lut_n, uint8_t *out, void *stream)
{
    DS_REQUIRE(ctx && depth && vmin_vmax && lut_rgba && out, DS_EINVAL, "ds_colorize_u16: null argument");
    DS_REQUIRE(n > 0 && h > 0 && w > 0, DS_EINVAL, "ds_colorize_u16: empty batch");
    DS_REQUIRE(lut_n > 0 && lut_n <= 256, DS_EINVAL, "ds_colorize_u16: the table must have 1..256 entries (got %d)", lut_n);
    DS_REQUIRE(((uintptr_t)lut_rgba & 3) == 0 && ((uintptr_t)out & 3) == 0, DS_EINVAL, "ds_colorize_u16: RGBA buffers must be 4-byte aligned");
    DS_REQUIRE(n <= 65535, DS_EUNSUPPORTED, "ds_colorize_u16: batch too large for the grid");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    const long long hw = (long long)h * w;
    const int blocks = (int)std::min<long long>((hw + 1023) / 1024, 2048);
    hipLaunchKernelGGL(k_colorize_u16, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, depth, vmin_vmax,
                       (const uint32_t *)lut_rgba, (uint32_t *)out, hw, lut_n);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}
