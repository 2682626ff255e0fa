// libdepthstereo_hip.so -- context, error plumbing, and the small streaming kernels:
// depth min/max, depth -> uint16 (core.py:44-50,189-206), view copies and the red/cyan packer
// (stereoimage_generation.py:53-71,286-307).  gfx950 only.
#include <stdarg.h>

#include "ds_common.h"

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void ds_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

DS_API int ds_version(void) { return DS_VERSION; }
DS_API const char *ds_last_error(void) { return g_err; }

DS_API int ds_ctx_create(ds_ctx **out, int device)
{
    DS_REQUIRE(out != nullptr, DS_EINVAL, "ds_ctx_create: out is NULL");
    int ndev = 0;
    DS_HIP_CHECK(hipGetDeviceCount(&ndev));
    DS_REQUIRE(device >= 0 && device < ndev, DS_EINVAL, "ds_ctx_create: device %d out of range (%d devices)", device, ndev);
    ds_ctx *c = new (std::nothrow) ds_ctx();
    DS_REQUIRE(c != nullptr, DS_ENOMEM, "ds_ctx_create: out of host memory");
    memset(c, 0, sizeof(*c));
    c->device = device;
    *out = c;
    return DS_OK;
}

DS_API int ds_ctx_destroy(ds_ctx *ctx)
{
    if (!ctx) return DS_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(ctx->device);
    void *blocks[] = { ctx->minmax, ctx->partials, ctx->row_flags, ctx->row_list, ctx->exact_ws, ctx->tmp_a, ctx->tmp_b, ctx->zero_line, ctx->lin_ws, ctx->gn_ws };
    for (void *b : blocks) if (b) (void)hipFree(b);
    if (ctx->ev_created) for (int i = 0; i < 4; i++) (void)hipEventDestroy(ctx->ev[i]);
    for (int k = 0; k < DS_KT_KINDS; k++)
        if (ctx->kt_ev[k]) {
            for (int i = 0; i < DS_KT_RING; i++) { (void)hipEventDestroy(ctx->kt_ev[k][i][0]); (void)hipEventDestroy(ctx->kt_ev[k][i][1]); }
            delete[] ctx->kt_ev[k];
        }
    (void)hipSetDevice(prev);
    delete ctx;
    return DS_OK;
}

DS_API int ds_profile_enable(ds_ctx *ctx, int enable)
{
    DS_REQUIRE(ctx != nullptr, DS_EINVAL, "ds_profile_enable: ctx is NULL");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    if (enable && !ctx->ev_created) {
        for (int i = 0; i < 4; i++) DS_HIP_CHECK(hipEventCreate(&ctx->ev[i]));
        ctx->ev_created = 1;
    }
    ctx->profile = enable ? 1 : 0;
    ctx->ev_recorded = 0;
    return DS_OK;
}

DS_API int ds_profile_last_ms(ds_ctx *ctx, float *render_ms, float *exact_ms)
{
    DS_REQUIRE(ctx && render_ms && exact_ms, DS_EINVAL, "ds_profile_last_ms: null argument");
    DS_REQUIRE(ctx->profile && ctx->ev_recorded, DS_EINVAL, "ds_profile_last_ms: profiling is off or nothing was recorded");
    DS_HIP_CHECK(hipEventSynchronize(ctx->ev[3]));
    DS_HIP_CHECK(hipEventElapsedTime(render_ms, ctx->ev[0], ctx->ev[1]));
    DS_HIP_CHECK(hipEventElapsedTime(exact_ms, ctx->ev[2], ctx->ev[3]));
    return DS_OK;
}

// ---- in-step kernel timers ---------------------------------------------------------------------------------------------------
// bench.py's `roofline` wants the average duration of a kernel INSIDE the timed step, on the stream it is launched on: the entry
// points bracket their launch with an event pair out of a per-kind ring (no synchronisation; an event record is a marker packet
// on the stream), ds_kernel_timer_read synchronises once and adds the pairs up.  SINGLE-THREADED by contract: the slot counters are
// plain ints (bench.py's instrumented repeat drives one context from one thread; the funnel, which drives a context from several
// threads, never enables the timers).  Launches beyond the ring are not timed: ds_kernel_timer_read's count is the TIMED launches.
int ds_kt_begin(ds_ctx *ctx, int kind, hipStream_t st)
{
    if (!ctx->ktimer || kind < 0 || kind >= DS_KT_KINDS || ctx->kt_n[kind] >= DS_KT_RING) return -1;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return -1; }
    if (!ctx->kt_ev[kind]) {
        hipEvent_t (*ring)[2] = new (std::nothrow) hipEvent_t[DS_KT_RING][2];
        if (!ring) { ctx->ktimer = 0; return -1; }
        int made = 0;                                        // events created so far (two per slot)
        bool ok = true;
        for (int i = 0; i < DS_KT_RING && ok; i++)
            for (int j = 0; j < 2 && ok; j++) {
                if (hipEventCreate(&ring[i][j]) == hipSuccess) made++;
                else { (void)hipGetLastError(); ok = false; }
            }
        if (!ok) {                                           // nothing leaks, and the timers switch themselves off instead of retrying
            for (int e = 0; e < made; e++) (void)hipEventDestroy(ring[e >> 1][e & 1]);      // the whole allocation on every launch
            delete[] ring;
            ctx->ktimer = 0;
            return -1;
        }
        ctx->kt_ev[kind] = ring;
    }
    const int slot = ctx->kt_n[kind];
    if (hipEventRecord(ctx->kt_ev[kind][slot][0], st) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return slot;
}

void ds_kt_end(ds_ctx *ctx, int kind, int slot, hipStream_t st)
{
    if (slot < 0) return;
    if (hipEventRecord(ctx->kt_ev[kind][slot][1], st) == hipSuccess) ctx->kt_n[kind] = slot + 1;
    else (void)hipGetLastError();
}

DS_API int ds_kernel_timer_enable(ds_ctx *ctx, int enable)
{
    DS_REQUIRE(ctx != nullptr, DS_EINVAL, "ds_kernel_timer_enable: ctx is NULL");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    ctx->ktimer = enable ? 1 : 0;
    for (int k = 0; k < DS_KT_KINDS; k++) ctx->kt_n[k] = 0;
    return DS_OK;
}

DS_API int ds_kernel_timer_read(ds_ctx *ctx, int kind, int64_t *launches, double *total_ms)
{
    DS_REQUIRE(ctx && launches && total_ms, DS_EINVAL, "ds_kernel_timer_read: null argument");
    DS_REQUIRE(kind >= 0 && kind < DS_KT_KINDS, DS_EINVAL, "ds_kernel_timer_read: kind %d outside 0..%d", kind, DS_KT_KINDS - 1);
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    const int n = ctx->kt_n[kind];
    double sum = 0.0;
    if (n > 0) DS_HIP_CHECK(hipEventSynchronize(ctx->kt_ev[kind][n - 1][1]));
    for (int i = 0; i < n; i++) {
        float ms = 0.f;
        DS_HIP_CHECK(hipEventElapsedTime(&ms, ctx->kt_ev[kind][i][0], ctx->kt_ev[kind][i][1]));
        sum += (double)ms;
    }
    *launches = n;
    *total_ms = sum;
    return DS_OK;
}

DS_API int ds_kernel_timer_read_each(ds_ctx *ctx, int kind, float *ms_out, int64_t capacity, int64_t *launches)
{
    DS_REQUIRE(ctx && launches && (ms_out || capacity == 0), DS_EINVAL, "ds_kernel_timer_read_each: null argument");
    DS_REQUIRE(kind >= 0 && kind < DS_KT_KINDS, DS_EINVAL, "ds_kernel_timer_read_each: kind %d outside 0..%d", kind, DS_KT_KINDS - 1);
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    const int n = ctx->kt_n[kind];
    if (n > 0) DS_HIP_CHECK(hipEventSynchronize(ctx->kt_ev[kind][n - 1][1]));
    for (int i = 0; i < n && i < capacity; i++) DS_HIP_CHECK(hipEventElapsedTime(&ms_out[i], ctx->kt_ev[kind][i][0], ctx->kt_ev[kind][i][1]));
    *launches = n;
    return DS_OK;
}

int ds_ctx_reserve(ds_ctx *ctx, void **slot, size_t *cur, size_t need)
{
    if (*cur >= need && *slot) return DS_OK;
    // growing a scratch block: make sure nothing still in flight uses the old one
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    if (*slot) {
        DS_HIP_CHECK(hipDeviceSynchronize());
        DS_HIP_CHECK(hipFree(*slot));
        *slot = nullptr; *cur = 0;
    }
    size_t sz = need + need / 4 + 4096;
    hipError_t e = hipMalloc(slot, sz);
    if (e != hipSuccess) { ds_set_error("hipMalloc(%zu) failed: %s", sz, hipGetErrorString(e)); *slot = nullptr; return DS_ENOMEM; }
    *cur = sz;
    return DS_OK;
}

// ------------------------------------------------------------------------------------------------
// per-image min/max.  Stage 1: every block reduces a slice of one image to {min,max} doubles
// (uint16/float32/float64 all convert to double exactly and order-preservingly).  Stage 2: one
// wave per image reduces the block partials.
#define MM_BLOCK 256
#define MM_BLOCKS_PER_IMAGE 64

template <typename T>
__global__ __launch_bounds__(MM_BLOCK) void k_minmax_stage1(const T *__restrict__ depth, int64_t per_image, double *__restrict__ partials)
{
    const int img = blockIdx.y;
    const T *p = depth + (int64_t)img * per_image;
    double mn = __builtin_inf(), mx = -__builtin_inf();
    for (int64_t i = (int64_t)blockIdx.x * MM_BLOCK + threadIdx.x; i < per_image; i += (int64_t)gridDim.x * MM_BLOCK) {
        double v = (double)p[i];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    mn = ds_wave_min(mn); mx = ds_wave_max(mx);
    __shared__ double s[2][MM_BLOCK / 64];
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    if (ln == 0) { s[0][wv] = mn; s[1][wv] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < MM_BLOCK / 64; k++) { mn = s[0][k] < mn ? s[0][k] : mn; mx = s[1][k] > mx ? s[1][k] : mx; }
        double *o = partials + ((int64_t)img * gridDim.x + blockIdx.x) * 2;
        o[0] = mn; o[1] = mx;
    }
}

// 16-byte vector loads for the uint16 case (the hot one): 8 elements per lane per load
__global__ __launch_bounds__(MM_BLOCK) void k_minmax_stage1_u16x8(const uint16_t *__restrict__ depth, int64_t per_image, double *__restrict__ partials)
{
    const int img = blockIdx.y;
    const uint16_t *p = depth + (int64_t)img * per_image;
    unsigned mn = 0xFFFFu, mx = 0u;
    const int64_t nvec = per_image >> 3;
    const uint4 *pv = reinterpret_cast<const uint4 *>(p);
    for (int64_t i = (int64_t)blockIdx.x * MM_BLOCK + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * MM_BLOCK) {
        uint4 q = pv[i];
        unsigned wds[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            unsigned lo = wds[k] & 0xFFFFu, hi = wds[k] >> 16;
            mn = min(mn, min(lo, hi));
            mx = max(mx, max(lo, hi));
        }
    }
    if (blockIdx.x == 0) {
        for (int64_t i = (nvec << 3) + threadIdx.x; i < per_image; i += MM_BLOCK) { unsigned v = p[i]; mn = min(mn, v); mx = max(mx, v); }
    }
    double dmn = ds_wave_min((double)mn), dmx = ds_wave_max((double)mx);
    __shared__ double s[2][MM_BLOCK / 64];
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    if (ln == 0) { s[0][wv] = dmn; s[1][wv] = dmx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < MM_BLOCK / 64; k++) { dmn = s[0][k] < dmn ? s[0][k] : dmn; dmx = s[1][k] > dmx ? s[1][k] : dmx; }
        double *o = partials + ((int64_t)img * gridDim.x + blockIdx.x) * 2;
        o[0] = dmn; o[1] = dmx;
    }
}

__global__ __launch_bounds__(64) void k_minmax_stage2(const double *__restrict__ partials, int nparts, double *__restrict__ minmax)
{
    const int img = blockIdx.x;
    double mn = __builtin_inf(), mx = -__builtin_inf();
    for (int i = threadIdx.x; i < nparts; i += 64) {
        const double *q = partials + ((int64_t)img * nparts + i) * 2;
        mn = q[0] < mn ? q[0] : mn;
        mx = q[1] > mx ? q[1] : mx;
    }
    mn = ds_wave_min(mn); mx = ds_wave_max(mx);
    if (threadIdx.x == 0) { minmax[img * 2] = mn; minmax[img * 2 + 1] = mx; }
}

int ds_minmax_launch(ds_ctx *ctx, const void *depth, int depth_dtype, int n, int64_t per_image, double *minmax_out, hipStream_t st)
{
    int nb = MM_BLOCKS_PER_IMAGE;
    int64_t max_useful = (per_image + MM_BLOCK * 8 - 1) / (MM_BLOCK * 8);
    if (nb > max_useful) nb = (int)(max_useful < 1 ? 1 : max_useful);
    int rc = ds_ctx_reserve(ctx, &ctx->partials, &ctx->partials_bytes, (size_t)n * nb * 2 * sizeof(double));
    if (rc) return rc;
    dim3 grid(nb, n);
    double *parts = (double *)ctx->partials;
    switch (depth_dtype) {
    case DS_DEPTH_U16:
        if ((per_image & 7) == 0 && ((uintptr_t)depth & 15) == 0)
            hipLaunchKernelGGL(k_minmax_stage1_u16x8, grid, dim3(MM_BLOCK), 0, st, (const uint16_t *)depth, per_image, parts);
        else
            hipLaunchKernelGGL(k_minmax_stage1<uint16_t>, grid, dim3(MM_BLOCK), 0, st, (const uint16_t *)depth, per_image, parts);
        break;
    case DS_DEPTH_F32:
        hipLaunchKernelGGL(k_minmax_stage1<float>, grid, dim3(MM_BLOCK), 0, st, (const float *)depth, per_image, parts);
        break;
    case DS_DEPTH_F64:
        hipLaunchKernelGGL(k_minmax_stage1<double>, grid, dim3(MM_BLOCK), 0, st, (const double *)depth, per_image, parts);
        break;
    default:
        ds_set_error("unknown depth dtype %d", depth_dtype);
        return DS_EINVAL;
    }
    hipLaunchKernelGGL(k_minmax_stage2, dim3(n), dim3(64), 0, st, parts, nb, minmax_out);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

DS_API int ds_depth_minmax(ds_ctx *ctx, const void *depth, int depth_dtype, int n, int h, int w, double *minmax_out, void *stream)
{
    DS_REQUIRE(ctx && depth && minmax_out, DS_EINVAL, "ds_depth_minmax: null argument");
    DS_REQUIRE(n > 0 && h > 0 && w > 0, DS_EINVAL, "ds_depth_minmax: bad shape n=%d h=%d w=%d", n, h, w);
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    return ds_minmax_launch(ctx, depth, depth_dtype, n, (int64_t)h * w, minmax_out, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// core.py:189-206 (no clip) + convert_to_i16 (core.py:44-50), float32 predictions.
//   out = copy(pred); if invert: out *= -1; out = (out - out.min()) / (out.max() - out.min())   [float32]
//   i16 = clip(out*65536 + 0.0001, 0, 65535.9).astype(uint16)                                    [float32, NEP 50]
__global__ __launch_bounds__(256) void k_depth_to_u16(const float *__restrict__ pred, int64_t per_image, int invert,
                                                      const double *__restrict__ minmax, uint16_t *__restrict__ out,
                                                      float *__restrict__ norm_out)
{
    const int img = blockIdx.y;
    const float pmn = (float)minmax[img * 2], pmx = (float)minmax[img * 2 + 1];
    // abs(max - min) > np.finfo("float").eps, evaluated on the float32 difference (core.py:189)
    const bool ok = fabs((double)(float)(pmx - pmn)) > 2.220446049250313e-16;
    const float omn = invert ? -pmx : pmn, omx = invert ? -pmn : pmx;
    const float den = omx - omn;
    const float hi = (float)(65536.0 - 0.1);
    const float *p = pred + (int64_t)img * per_image;
    uint16_t *o = out + (int64_t)img * per_image;
    float *no = norm_out ? norm_out + (int64_t)img * per_image : nullptr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_image; i += (int64_t)gridDim.x * 256) {
        float v = 0.0f;
        if (ok) {
            float x = invert ? -p[i] : p[i];
            v = (x - omn) / den;
        }
        if (no) no[i] = v;
        float q = v * 65536.0f;
        q = q + 0.0001f;
        q = q < 0.0f ? 0.0f : q;
        q = q > hi ? hi : q;
        o[i] = (uint16_t)(int)q;
    }
}

DS_API int ds_depth_to_u16(ds_ctx *ctx, const float *pred, int n, int h, int w, int invert, uint16_t *out, float *norm_out, void *stream)
{
    DS_REQUIRE(ctx && pred && out, DS_EINVAL, "ds_depth_to_u16: null argument");
    DS_REQUIRE(n > 0 && h > 0 && w > 0, DS_EINVAL, "ds_depth_to_u16: bad shape n=%d h=%d w=%d", n, h, w);
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    int rc = ds_ctx_reserve(ctx, &ctx->minmax, &ctx->minmax_bytes, (size_t)n * 2 * sizeof(double));
    if (rc) return rc;
    const int64_t per_image = (int64_t)h * w;
    rc = ds_minmax_launch(ctx, pred, DS_DEPTH_F32, n, per_image, (double *)ctx->minmax, st);
    if (rc) return rc;
    int nb = (int)((per_image + 256 * 4 - 1) / (256 * 4));
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_depth_to_u16, dim3(nb, n), dim3(256), 0, st, pred, per_image, invert ? 1 : 0,
                       (const double *)ctx->minmax, out, norm_out);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void k_convert_to_i16(const T *__restrict__ arr, int64_t count, uint16_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        T q = arr[i] * (T)65536.0;
        q = q + (T)0.0001;
        const T hi = (T)(65536.0 - 0.1);
        q = q < (T)0 ? (T)0 : q;
        q = q > hi ? hi : q;
        out[i] = (uint16_t)(long long)q;
    }
}

DS_API int ds_convert_to_i16(ds_ctx *ctx, const void *arr, int is_f64, int64_t count, uint16_t *out, void *stream)
{
    DS_REQUIRE(ctx && arr && out, DS_EINVAL, "ds_convert_to_i16: null argument");
    DS_REQUIRE(count > 0, DS_EINVAL, "ds_convert_to_i16: count must be positive");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    int64_t nb = (count + 1023) / 1024;
    if (nb > 4096) nb = 4096;
    if (is_f64) hipLaunchKernelGGL(k_convert_to_i16<double>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const double *)arr, count, out);
    else hipLaunchKernelGGL(k_convert_to_i16<float>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const float *)arr, count, out);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

// ------------------------------------------------------------------------------------------------
// strided row copy: one block walks rows; 16-byte lanes when everything is 16-byte aligned.
__global__ __launch_bounds__(256) void k_copy_view(const uint8_t *__restrict__ src, int64_t srs, int64_t sis,
                                                   uint8_t *__restrict__ dst, int64_t drs, int64_t dis,
                                                   int h, int64_t row_bytes, int vec16)
{
    const int img = blockIdx.y;
    for (int row = blockIdx.x; row < h; row += gridDim.x) {
        const uint8_t *s = src + (int64_t)img * sis + (int64_t)row * srs;
        uint8_t *d = dst + (int64_t)img * dis + (int64_t)row * drs;
        if (vec16) {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(s);
            uint4 *d4 = reinterpret_cast<uint4 *>(d);
            for (int64_t i = threadIdx.x; i < (row_bytes >> 4); i += 256) d4[i] = s4[i];
        } else {
            for (int64_t i = threadIdx.x; i < row_bytes; i += 256) d[i] = s[i];
        }
    }
}

DS_API int ds_copy_view(ds_ctx *ctx, const uint8_t *src, int64_t src_row_stride, int64_t src_img_stride,
                        uint8_t *dst, int64_t dst_row_stride, int64_t dst_img_stride,
                        int n, int h, int64_t row_bytes, void *stream)
{
    DS_REQUIRE(ctx && src && dst, DS_EINVAL, "ds_copy_view: null argument");
    DS_REQUIRE(n > 0 && h > 0 && row_bytes > 0, DS_EINVAL, "ds_copy_view: bad shape");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    int vec16 = ((row_bytes | src_row_stride | src_img_stride | dst_row_stride | dst_img_stride) & 15) == 0 &&
                (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    int nb = h < 2048 ? h : 2048;
    hipLaunchKernelGGL(k_copy_view, dim3(nb, n), dim3(256), 0, (hipStream_t)stream, src, src_row_stride, src_img_stride,
                       dst, dst_row_stride, dst_img_stride, h, row_bytes, vec16);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

// ------------------------------------------------------------------------------------------------
// stereoimage_generation.py:286-307
__global__ __launch_bounds__(256) void k_overlap_red_cyan(const uint8_t *__restrict__ im1, int64_t r1, int64_t i1,
                                                          const uint8_t *__restrict__ im2, int64_t r2, int64_t i2,
                                                          int h, int w, int c, uint8_t *__restrict__ out)
{
    const int img = blockIdx.z, row = blockIdx.y;
    const uint8_t *a = im1 + (int64_t)img * i1 + (int64_t)row * r1;
    const uint8_t *b = im2 + (int64_t)img * i2 + (int64_t)row * r2;
    uint8_t *o = out + ((int64_t)img * h + row) * (int64_t)w * 3;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < w; x += gridDim.x * 256) {
        o[x * 3 + 0] = a[(int64_t)x * c + 0];
        o[x * 3 + 1] = b[(int64_t)x * c + 1];
        o[x * 3 + 2] = b[(int64_t)x * c + 2];
    }
}

DS_API int ds_overlap_red_cyan(ds_ctx *ctx, const uint8_t *im1, int64_t im1_row_stride, int64_t im1_img_stride,
                               const uint8_t *im2, int64_t im2_row_stride, int64_t im2_img_stride,
                               int n, int h, int w, int c, uint8_t *out, void *stream)
{
    DS_REQUIRE(ctx && im1 && im2 && out, DS_EINVAL, "ds_overlap_red_cyan: null argument");
    DS_REQUIRE(n > 0 && h > 0 && w > 0, DS_EINVAL, "ds_overlap_red_cyan: bad shape");
    DS_REQUIRE(c >= 3 && c <= 4, DS_EINVAL, "ds_overlap_red_cyan: needs >= 3 channels (got %d)", c);
    DS_REQUIRE(h <= 65535 && n <= 65535, DS_EUNSUPPORTED, "ds_overlap_red_cyan: h and n must be <= 65535");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_overlap_red_cyan, dim3((w + 255) / 256, h, n), dim3(256), 0, (hipStream_t)stream,
                       im1, im1_row_stride, im1_img_stride, im2, im2_row_stride, im2_img_stride, h, w, c, out);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}
