// ds_stereo_warp: apply_stereo_divergence (reference: src/stereoimage_generation.py:77-92) and the
// naive scatter kernel with its three fills (apply_stereo_divergence_naive, :95-159) on gfx950.
// The polylines kernels live in ds_stereo_polylines.hip.
#include "ds_common.h"

int ds_minmax_launch(ds_ctx *ctx, const void *depth, int depth_dtype, int n, int64_t per_image, double *minmax_out, hipStream_t st);
int ds_polylines_launch(ds_ctx *ctx, const uint8_t *image, const void *depth, int depth_dtype, const double *minmax,
                        const double *lut, int n, int h, int w, int c, int sharp, const ds_eye *eyes, int n_eyes,
                        hipStream_t st);

#define NV_BLOCK 256

struct NaiveParams {
    const uint8_t *img;
    const void *depth;
    const double *minmax;
    const double *lut;
    int n, h, w, c;
    int n_eyes;
    int fill;
    double div_px[2], sep_px[2];
    uint8_t *out[2];
    int64_t ors[2], ois[2];
};

// One workgroup per (image, eye, row); the whole row lives in LDS:
//   s_win[w]  int   source column that ends up in each destination pixel (-1 = unfilled)
//   s_der[w*c] u8   derived_image row
// Painter's order (:105-111): columns are visited ascending when divergence_px < 0, else descending,
// and later writes overwrite earlier ones -> the surviving source column of a destination is the
// LARGEST (ascending) or SMALLEST (descending) column that maps to it: an LDS atomicMax / atomicMin.
template <int DT>
__global__ __launch_bounds__(NV_BLOCK) void k_naive(NaiveParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int eye = blockIdx.y % P.n_eyes, img = blockIdx.y / P.n_eyes, row = blockIdx.x;
    const int w = P.w, c = P.c;
    int *s_win = reinterpret_cast<int *>(smem);
    uint8_t *s_der = reinterpret_cast<uint8_t *>(s_win + w);

    const double div_px = P.div_px[eye], sep_px = P.sep_px[eye];
    const double mn = P.minmax[img * 2], mx = P.minmax[img * 2 + 1];
    const uint8_t *src = P.img + ((size_t)img * P.h + row) * (size_t)w * c;
    typedef typename ds_depth_traits<DT>::T DTy;
    const DTy *depth_row = (const DTy *)P.depth + ((size_t)img * P.h + row) * (size_t)w;
    uint8_t *dst = P.out[eye] + (int64_t)img * P.ois[eye] + (int64_t)row * P.ors[eye];

    const bool ascending = div_px < 0;                                   // :107
    const int empty = ascending ? -1 : 0x7fffffff;
    for (int i = tid; i < w; i += NV_BLOCK) s_win[i] = empty;
    __syncthreads();
    for (int col = tid; col < w; col += NV_BLOCK) {
        const DTy dv = depth_row[col];
        double nd;
        if (DT == DS_DEPTH_U16 && P.lut != nullptr) nd = P.lut[(size_t)img * 65536 + (unsigned)dv];
        else nd = ds_depth_traits<DT>::norm(dv, mn, mx);
        const double v = nd * div_px + sep_px;                           // :108
        // int(): truncate toward zero; NaN/inf scatter nothing (numba: fptosi -> INT64_MIN)
        if (v == v && v < 4.0e9 && v > -4.0e9) {
            const long long col_d = (long long)col + (long long)v;
            if (col_d >= 0 && col_d < w) {                               // :109
                if (ascending) atomicMax(&s_win[col_d], col); else atomicMin(&s_win[col_d], col);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < w; i += NV_BLOCK) {
        int wn = s_win[i];
        if (wn == empty) wn = -1;
        s_win[i] = wn;
        for (int k = 0; k < c; k++) s_der[(size_t)i * c + k] = wn >= 0 ? src[(size_t)wn * c + k] : (uint8_t)0;   // :101,:110
    }
    __syncthreads();

    if (P.fill == DS_FILL_NAIVE) {                                       // :142-157
        double adv = fabs(div_px);
        long long span = (adv < 4.0e9 ? (long long)adv : 0) + 2;         // range(1, abs(int(divergence_px)) + 2)
        for (int col = tid; col < w; col += NV_BLOCK) {
            if (s_win[col] >= 0) {
                for (int k = 0; k < c; k++) dst[(size_t)col * c + k] = s_der[(size_t)col * c + k];
                continue;
            }
            int from = -1;
            for (long long off = 1; off < span; off++) {
                const long long ro = col + off, lo = col - off;
                if (ro < w && s_win[ro] >= 0) { from = (int)ro; break; }                  // :151 right first
                if (lo >= 0 && s_win[lo] >= 0) { from = (int)lo; break; }                 // :154
                if (ro >= w && lo < 0) break;
            }
            for (int k = 0; k < c; k++) dst[(size_t)col * c + k] = from >= 0 ? s_der[(size_t)from * c + k] : (uint8_t)0;
        }
        return;
    }

    if (P.fill == DS_FILL_NAIVE_INTERPOLATING) {                         // :114-141
        // A gap run starts at the first unfilled pixel after an anchor (filled and non-black) or after the
        // row start, and never writes at or beyond the next anchor; runs are therefore independent and
        // each is replayed literally (reference statements) by one lane.
        for (int p = tid; p < w; p += NV_BLOCK) {
            if (s_win[p] >= 0) continue;
            int q = p - 1;
            bool start;
            for (;;) {
                if (q < 0) { start = true; break; }
                if (s_win[q] < 0) { start = false; break; }              // an earlier gap owns this stretch
                int sum = 0;
                for (int k = 0; k < c; k++) sum += s_der[(size_t)q * c + k];
                if (sum != 0) { start = true; break; }                   // anchor
                q--;
            }
            if (!start) continue;
            // end of the stretch: the next anchor of the ORIGINAL state (or w)
            int r_end = p + 1;
            while (r_end < w) {
                int sum = 0;
                for (int k = 0; k < c; k++) sum += s_der[(size_t)r_end * c + k];
                if (sum != 0 && s_win[r_end] >= 0) break;
                r_end++;
            }
            for (int l = p; l < r_end; l++) {                            // :116
                int sum = 0;
                for (int k = 0; k < c; k++) sum += s_der[(size_t)l * c + k];
                if (sum != 0 || s_win[l] >= 0) continue;                 // :118
                uint8_t lb[4] = { 0, 0, 0, 0 }, rb[4] = { 0, 0, 0, 0 };
                int ls = 0, rs = 0;
                if (l > 0) for (int k = 0; k < c; k++) { lb[k] = s_der[(size_t)(l - 1) * c + k]; ls += lb[k]; }   // :120
                int r = l + 1;                                           // :122
                while (r < w) {                                          // :123
                    int t = 0;
                    for (int k = 0; k < c; k++) t += s_der[(size_t)r * c + k];
                    if (t != 0 && s_win[r] >= 0) {                       // :124
                        for (int k = 0; k < c; k++) rb[k] = s_der[(size_t)r * c + k];
                        rs = t;
                        break;
                    }
                    r++;
                }
                if (ls == 0) { for (int k = 0; k < c; k++) lb[k] = rb[k]; }              // :128
                else if (rs == 0) { for (int k = 0; k < c; k++) rb[k] = lb[k]; }         // :130
                const int total_steps = 1 + r - l;                       // :137
                double step[4];
                for (int k = 0; k < c; k++) step[k] = ((double)rb[k] - (double)lb[k]) / (double)total_steps;      // :138
                for (int col = l; col < r; col++) {                      // :139
                    const double m = (double)(col - l + 1);
                    for (int k = 0; k < c; k++)
                        s_der[(size_t)col * c + k] = (uint8_t)(lb[k] + ds_f64_to_u8(step[k] * m));               // :140
                }
            }
        }
        __syncthreads();
    }

    for (int i = tid; i < w * c; i += NV_BLOCK) dst[i] = s_der[i];
}

template <int DT>
static void launch_naive(const NaiveParams &P, dim3 grid, size_t lds, hipStream_t st)
{
    hipLaunchKernelGGL((k_naive<DT>), grid, dim3(NV_BLOCK), lds, st, P);
}

DS_API int ds_stereo_warp(ds_ctx *ctx, const uint8_t *image, const void *depth, int depth_dtype,
                          int n, int h, int w, int c, double exponent, const double *pow_lut,
                          int fill, const ds_eye *eyes, int n_eyes, void *stream)
{
    DS_REQUIRE(ctx && image && depth && eyes, DS_EINVAL, "ds_stereo_warp: null argument");
    DS_REQUIRE(n > 0 && h > 0 && w > 0, DS_EINVAL, "ds_stereo_warp: bad shape n=%d h=%d w=%d", n, h, w);
    DS_REQUIRE(c >= 1 && c <= 4, DS_EUNSUPPORTED, "ds_stereo_warp: channels must be 1..4 (got %d)", c);
    DS_REQUIRE(n_eyes == 1 || n_eyes == 2, DS_EINVAL, "ds_stereo_warp: n_eyes must be 1 or 2");
    DS_REQUIRE(depth_dtype == DS_DEPTH_U16 || depth_dtype == DS_DEPTH_F32 || depth_dtype == DS_DEPTH_F64, DS_EINVAL,
               "ds_stereo_warp: unknown depth dtype %d", depth_dtype);
    DS_REQUIRE(fill >= DS_FILL_NONE && fill <= DS_FILL_POLYLINES_SHARP, DS_EINVAL, "ds_stereo_warp: unknown fill %d", fill);
    for (int e = 0; e < n_eyes; e++) DS_REQUIRE(eyes[e].out != nullptr, DS_EINVAL, "ds_stereo_warp: eye %d has no output", e);
    if (exponent != 1.0) {
        DS_REQUIRE(pow_lut != nullptr && depth_dtype == DS_DEPTH_U16, DS_EUNSUPPORTED,
                   "ds_stereo_warp: stereo_offset_exponent != 1 needs a host-built pow_lut over uint16 depth");
    } else {
        pow_lut = nullptr;
    }
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;

    int rc = ds_ctx_reserve(ctx, &ctx->minmax, &ctx->minmax_bytes, (size_t)n * 2 * sizeof(double));
    if (rc) return rc;
    rc = ds_minmax_launch(ctx, depth, depth_dtype, n, (int64_t)h * w, (double *)ctx->minmax, st);      // :79-80
    if (rc) return rc;
    const double *minmax = (const double *)ctx->minmax;

    if (fill == DS_FILL_POLYLINES_SOFT || fill == DS_FILL_POLYLINES_SHARP)
        return ds_polylines_launch(ctx, image, depth, depth_dtype, minmax, pow_lut, n, h, w, c,
                                   fill == DS_FILL_POLYLINES_SHARP, eyes, n_eyes, st);

    NaiveParams P;
    memset(&P, 0, sizeof(P));
    P.img = image; P.depth = depth; P.minmax = minmax; P.lut = pow_lut;
    P.n = n; P.h = h; P.w = w; P.c = c; P.n_eyes = n_eyes; P.fill = fill;
    for (int e = 0; e < n_eyes; e++) {
        DS_REQUIRE(eyes[e].divergence_px == eyes[e].divergence_px && eyes[e].separation_px == eyes[e].separation_px,
                   DS_EINVAL, "ds_stereo_warp: divergence/separation is NaN");
        P.div_px[e] = eyes[e].divergence_px; P.sep_px[e] = eyes[e].separation_px;
        P.out[e] = eyes[e].out; P.ors[e] = eyes[e].out_row_stride; P.ois[e] = eyes[e].out_img_stride;
    }
    const size_t lds = (size_t)w * (sizeof(int) + c);
    DS_REQUIRE(lds <= 160 * 1024, DS_EUNSUPPORTED, "ds_stereo_warp: row of %d pixels does not fit the LDS row buffer", w);
    DS_REQUIRE((int64_t)n * n_eyes <= 65535, DS_EUNSUPPORTED, "ds_stereo_warp: n*n_eyes must be <= 65535");
    dim3 grid(h, n * n_eyes);
    if (ctx->profile) (void)hipEventRecord(ctx->ev[0], st);
    switch (depth_dtype) {
    case DS_DEPTH_U16: launch_naive<DS_DEPTH_U16>(P, grid, lds, st); break;
    case DS_DEPTH_F32: launch_naive<DS_DEPTH_F32>(P, grid, lds, st); break;
    default: launch_naive<DS_DEPTH_F64>(P, grid, lds, st); break;
    }
    if (ctx->profile) {
        (void)hipEventRecord(ctx->ev[1], st); (void)hipEventRecord(ctx->ev[2], st); (void)hipEventRecord(ctx->ev[3], st);
        ctx->ev_recorded = 1;
    }
    DS_HIP_CHECK(hipGetLastError());
    ctx->last_exact_rows_valid = 0;
    return DS_OK;
}

DS_API int ds_stereo_last_exact_rows(ds_ctx *ctx, int64_t *rows_out, void *stream)
{
    DS_REQUIRE(ctx && rows_out, DS_EINVAL, "ds_stereo_last_exact_rows: null argument");
    *rows_out = 0;
    if (ctx->last_exact_rows_valid <= 0 || !ctx->row_flags) return DS_OK;
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    int v = 0;
    DS_HIP_CHECK(hipMemcpyAsync(&v, (int *)ctx->row_flags + ctx->last_exact_rows_valid, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    DS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    *rows_out = v;
    return DS_OK;
}

DS_API int ds_stereo_last_stats(ds_ctx *ctx, int64_t *stats_out, void *stream)
{
    DS_REQUIRE(ctx && stats_out, DS_EINVAL, "ds_stereo_last_stats: null argument");
    stats_out[0] = stats_out[1] = 0;
    if (ctx->last_exact_rows_valid <= 0 || !ctx->row_flags) return DS_OK;
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    int v[2] = { 0, 0 };
    DS_HIP_CHECK(hipMemcpyAsync(v, (int *)ctx->row_flags + ctx->last_exact_rows_valid, 2 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    DS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    stats_out[0] = v[0]; stats_out[1] = v[1];
    return DS_OK;
}
