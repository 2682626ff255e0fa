// ds_boost_blend: the patch-merge step of Boost (reference: src/depthmap_generation.py:915-937 inside estimateboost).
//
// Per selected patch the reference (i) maps the merge network's 1024x1024 output onto the base estimate's value range
// with a degree-1 polynomial (:915-916), (ii) cubic-resizes it to the patch rectangle (:918), (iii) bilinearly resizes a
// 3000x3000 Gaussian mask template to the rectangle (:929-930) and (iv) blends
//         dst[rect] = dst[rect] * (1 - mask) + merged * mask                                   (:936)
// into one running float32 image, patch after patch, largest patch first (:1098): the result depends on the ORDER.
// That is four full-size temporaries and a read-modify-write of dst per patch.
//
// Here ONE launch blends ALL patches: a lane owns one pixel of dst and walks the patch list in order; for every
// rectangle that contains the pixel it samples the network output (4x4 cubic taps, a = -0.75, half-pixel centres,
// replicated border: cv2.INTER_CUBIC's kernel) and the mask template (2x2 bilinear, cv2.INTER_LINEAR's rule) directly
// at the pixel's position, applies the polynomial and blends.  dst is read once and written once, nothing else is
// materialised; the per-pixel order of the blends is the reference's order.  The polynomial commutes with the cubic
// resize (affine map, taps sum to one), so it is applied after sampling.
// Arithmetic follows numpy's promotion at :936: (dst * (1 - mask)) in float32, (merged * mask) in float64, the sum in
// float64, stored as float32.  cv2 is not available in the build container: the two resampling rules are restated from
// OpenCV's documentation, parity unpinned.
#include "ds_common.h"

struct BoostPatch {
    int x0, y0, w, h;         // rectangle in dst (already scaled and rounded like ImageandPatchs.__getitem__, :595-602)
    double p0, p1;            // merged = p0 * mapped + p1   (np.polyfit deg 1, :915)
};

__device__ __forceinline__ void bb_cubic_weights(double t, double *w)
{
    const double A = -0.75;   // OpenCV's bicubic coefficient
    w[0] = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A;
    w[1] = ((A + 2) * t - (A + 3)) * t * t + 1;
    w[2] = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1;
    w[3] = 1.0 - w[0] - w[1] - w[2];
}

__global__ __launch_bounds__(256) void k_boost_blend(float *__restrict__ dst, int64_t dst_stride, int H, int W,
                                                     const BoostPatch *__restrict__ patches, int n_patches,
                                                     const float *__restrict__ preds, int ps,
                                                     const float *__restrict__ mask_tpl, int ms)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    float v = dst[(int64_t)y * dst_stride + x];
    bool touched = false;
    for (int k = 0; k < n_patches; k++) {
        const BoostPatch P = patches[k];
        const int lx = x - P.x0, ly = y - P.y0;
        if (lx < 0 || ly < 0 || lx >= P.w || ly >= P.h) continue;
        // ---- cubic sample of the ps x ps network output at the pixel centre (cv2.resize INTER_CUBIC) --------------
        const float *pred = preds + (size_t)k * ps * ps;
        const double fx = (lx + 0.5) * ((double)ps / P.w) - 0.5, fy = (ly + 0.5) * ((double)ps / P.h) - 0.5;
        const int ix = (int)floor(fx), iy = (int)floor(fy);
        double wx[4], wy[4];
        bb_cubic_weights(fx - ix, wx);
        bb_cubic_weights(fy - iy, wy);
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int yy = min(max(iy - 1 + j, 0), ps - 1);
            double row = 0.0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int xx = min(max(ix - 1 + i, 0), ps - 1);
                row += wx[i] * (double)pred[(size_t)yy * ps + xx];
            }
            acc += wy[j] * row;
        }
        const double merged = P.p0 * acc + P.p1;
        // ---- bilinear sample of the ms x ms mask template (cv2.resize INTER_LINEAR) ----------------------------------
        double mxf = (lx + 0.5) * ((double)ms / P.w) - 0.5, myf = (ly + 0.5) * ((double)ms / P.h) - 0.5;
        int mx0 = (int)floor(mxf), my0 = (int)floor(myf);
        double tx = mxf - mx0, ty = myf - my0;
        if (mx0 < 0) { mx0 = 0; tx = 0.0; }
        if (my0 < 0) { my0 = 0; ty = 0.0; }
        if (mx0 >= ms - 1) { mx0 = ms - 2; tx = 1.0; }
        if (my0 >= ms - 1) { my0 = ms - 2; ty = 1.0; }
        const float *mp = mask_tpl + (size_t)my0 * ms + mx0;
        const double m = (1.0 - ty) * ((1.0 - tx) * (double)mp[0] + tx * (double)mp[1])
                       + ty * ((1.0 - tx) * (double)mp[ms] + tx * (double)mp[ms + 1]);
        const float mask = (float)m;                          // the resized mask is a float32 array (:930)
        // ---- blend (:936) ------------------------------------------------------------------------------------------------
        const float t1 = v * (1.0f - mask);
        v = (float)((double)t1 + merged * (double)mask);
        touched = true;
    }
    if (touched) dst[(int64_t)y * dst_stride + x] = v;
}

DS_API int ds_boost_blend(ds_ctx *ctx, float *dst, int64_t dst_row_stride, int height, int width, const void *patches,
                          int n_patches, const float *preds, int pred_size, const float *mask_template, int mask_size,
                          void *stream)
{
    DS_REQUIRE(ctx && dst && patches && preds && mask_template, DS_EINVAL, "ds_boost_blend: null argument");
    DS_REQUIRE(height > 0 && width > 0 && n_patches >= 0 && pred_size >= 2 && mask_size >= 2, DS_EINVAL, "ds_boost_blend: bad shape");
    DS_REQUIRE(height <= 4 * 65535, DS_EUNSUPPORTED, "ds_boost_blend: image too tall");
    if (n_patches == 0) return DS_OK;
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    dim3 grid((width + 63) / 64, (height + 3) / 4);
    hipLaunchKernelGGL(k_boost_blend, grid, dim3(256), 0, (hipStream_t)stream, dst, dst_row_stride, height, width,
                       (const BoostPatch *)patches, n_patches, preds, pred_size, mask_template, mask_size);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}
