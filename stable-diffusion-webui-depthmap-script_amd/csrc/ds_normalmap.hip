// ds_normalmap: create_normalmap (reference: src/normalmap_generation.py:5-56) on gfx950.
//
// Fused path (no blur): one lane per output pixel reads its 3x3 (Sobel) or 4-neighbour (np.gradient)
// uint16 neighbourhood, does the float64 arithmetic of :20-54 and writes 3 bytes.  With inputs that
// are multiples of 2^-8 every sum is exact, so only sqrt and the three divisions round -- and those are
// correctly rounded IEEE operations on the device, hence bit-exact results.
//
// General path (Gaussian pre/post blur, other Sobel sizes): separable float64 passes over planes in
// the context's scratch, rows first then columns, taps accumulated in order (BORDER_REFLECT_101).
#include <math.h>
#include <stdlib.h>

#include "ds_common.h"

__device__ __forceinline__ int nm_reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; }
    return i;
}

// normal = (zx, -zy, 1)/|.| ; (+1)/2*256 clipped to [0, 255.9] ; truncate   (:34-39, :51-54)
// The kernels are bound by float64 VALU issue, not by memory (round 3: ~160 vector instructions per pixel), so the arithmetic is
// kept to what the result needs:
//   * ONE division, y = 1 / n: it is the z component, and the correctly rounded reciprocal turns the other two quotients into
//     q0 = a y;  r = fma(-q0, n, a);  q = fma(r, y, q0)  -- Markstein's sequence, equal to the correctly rounded a / n for every
//     operand of this kernel (the significand of n is never all ones; enumerated by check_normal_division.c of the test suite);
//   * (v + 1) / 2 * 256 = (v + 1) * 128 bit for bit (scaling by a power of two never rounds);
//   * after the clip the value lies in [0, 255.9]: the float64 -> uint8 cast is one v_cvt_i32_f64, not the general numpy
//     emulation (ds_f64_to_u8: int64 conversion + wrap) the unclipped paths of the stereo kernels need.
// (n >= 1: no NaN, no zero divisor.  a = -0.0 gives +0.0 where the division gives -0.0: v + 1 is 1.0 either way.)
__device__ __forceinline__ void nm_store(double zx, double zy, uint8_t *o)
{
    const double nx = zx, ny = -zy;
    const double n = sqrt(nx * nx + ny * ny + 1.0);
    const double y = 1.0 / n;
    const double qx0 = nx * y, qy0 = ny * y;
    const double v[3] = { __builtin_fma(__builtin_fma(-qx0, n, nx), y, qx0), __builtin_fma(__builtin_fma(-qy0, n, ny), y, qy0), y };
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double t = (v[k] + 1.0) * 128.0;
        t = t < 0.0 ? 0.0 : t;
        t = t > 256.0 - 0.1 ? 256.0 - 0.1 : t;
        o[k] = (uint8_t)(int)t;
    }
}

#define NM_BX 64
#define NM_BY 4

template <int SOBEL3>
__global__ __launch_bounds__(NM_BX * NM_BY) void k_normalmap_fused(const uint16_t *__restrict__ depth, int h, int w, int invert,
                                                                   uint8_t *__restrict__ out)
{
    const int img = blockIdx.z;
    const int x = blockIdx.x * NM_BX + threadIdx.x;
    const int y = blockIdx.y * NM_BY + threadIdx.y;
    if (x >= w || y >= h) return;
    const uint16_t *d = depth + (size_t)img * h * w;
    const double sgn = invert ? 1.0 : -1.0;
#define NMV(yy, xx) (((double)d[(size_t)(yy) * w + (xx)] * sgn) / 256.0)        /* :20-21 */
    double zx, zy;
    if (SOBEL3) {
        const int xm = nm_reflect101(x - 1, w), xp = nm_reflect101(x + 1, w);
        const int ym = nm_reflect101(y - 1, h), yp = nm_reflect101(y + 1, h);
        const double p00 = NMV(ym, xm), p01 = NMV(ym, x), p02 = NMV(ym, xp);
        const double p10 = NMV(y, xm), p12 = NMV(y, xp);
        const double p20 = NMV(yp, xm), p21 = NMV(yp, x), p22 = NMV(yp, xp);
        zx = (p02 - p00) + 2.0 * (p12 - p10) + (p22 - p20);                     /* cv2.Sobel dx, ksize 3 */
        zy = (p20 - p00) + 2.0 * (p21 - p01) + (p22 - p02);                     /* cv2.Sobel dy, ksize 3 */
    } else {                                                                    /* np.gradient, :31 */
        if (w == 1) zx = 0.0;
        else if (x == 0) zx = NMV(y, 1) - NMV(y, 0);
        else if (x == w - 1) zx = NMV(y, w - 1) - NMV(y, w - 2);
        else zx = (NMV(y, x + 1) - NMV(y, x - 1)) / 2.0;
        if (h == 1) zy = 0.0;
        else if (y == 0) zy = NMV(1, x) - NMV(0, x);
        else if (y == h - 1) zy = NMV(h - 1, x) - NMV(h - 2, x);
        else zy = (NMV(y + 1, x) - NMV(y - 1, x)) / 2.0;
    }
#undef NMV
    nm_store(zx, zy, out + ((size_t)img * h * w + (size_t)y * w + x) * 3);
}

// ---- round 6: the fused kernel at ~60 instead of ~140 vector instructions per pixel ----------------------------------------------
// The kernel is bound by float64 VALU issue (round 4: 62.8 M vector instructions per 32 x 1024^2 launch, 121-140 us against 35-40 us
// of memory time).  What a pixel needs, and what the generic code spent on it:
//   * the Sobel / np.gradient sums in INTEGERS.  The operands are uint16 / 256: every sum of the reference is exact in float64, so
//     the same real number comes out of  Zx = (e02 - e00) + 2 (e12 - e10) + (e22 - e20)  in int32 and ONE conversion + ONE multiply
//     by +-2^-8 (+-2^-9 for np.gradient's halved interior differences) -- instead of 18 conversions, 36 scalings and 10 float64
//     adds per four pixels;
//   * n^2 = fma(nx, nx, fma(ny, ny, 1)): exact (a multiple of 2^-18 below 2^22), so the two fused operations round nothing;
//   * sqrt and 1 / n WITHOUT the range scaling and the special-case fix-ups of the generic expansions: n^2 lies in [1, 2^22], n in
//     [1, 2^11] -- nm_sqrt / nm_rcp below are the compiler's own Newton / Goldschmidt sequences (AMDGPU lowering of llvm.sqrt.f64 and
//     of the float64 division) minus v_div_scale / v_ldexp / v_div_fixup and their compares and selects: the same instructions on the
//     same operands, bit for bit (tests/test_gpu_parity.py compares them with sqrt() and 1.0 / n on 2^27 operands of the domain);
//   * (v + 1) * 128 as fma(v, 128, 128) (scaling by a power of two commutes with the rounding of v + 1), the upper clip as one
//     v_min_f64 instead of compare + two 32-bit selects, and NO lower clip: |q| <= 1 because n >= |nx| (sqrt is monotonic and
//     correctly rounded, |nx| is representable), so the value is never negative.
__device__ __forceinline__ double nm_sqrt(double x)                 // x in [1, 2^22]: correctly rounded, as sqrt(x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

__device__ __forceinline__ double nm_rcp(double n)                  // n in [1, 2^11]: correctly rounded, as 1.0 / n
{
    double y = __builtin_amdgcn_rcp(n);
    y = __builtin_fma(y, __builtin_fma(-n, y, 1.0), y);
    y = __builtin_fma(y, __builtin_fma(-n, y, 1.0), y);
    return __builtin_fma(__builtin_fma(-n, y, 1.0), y, y);
}

// one pixel: Zx, Zy = the integer gradient sums, kx, ky = their scales (sign of the inversion included; ky also carries the minus
// sign of (zx, -zy, 1)); three bytes out
__device__ __forceinline__ uint32_t nm_pixel(int Zx, int Zy, double kx, double ky)
{
    const double nx = (double)Zx * kx, ny = (double)Zy * ky;
    const double n = nm_sqrt(__builtin_fma(nx, nx, __builtin_fma(ny, ny, 1.0)));
    const double y = nm_rcp(n);
    const double qx0 = nx * y, qy0 = ny * y;
    const double v[3] = { __builtin_fma(__builtin_fma(-qx0, n, nx), y, qx0), __builtin_fma(__builtin_fma(-qy0, n, ny), y, qy0), y };
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) o |= (uint32_t)(int)__builtin_fmin(__builtin_fma(v[k], 128.0, 128.0), 256.0 - 0.1) << (8 * k);
    return o;
}

// ds_normalmap_selfcheck: nm_sqrt / nm_rcp against the generic operations over a range of the operand domain, n^2 = K / 2^18 for
// K = k0 + stride * i (every n^2 the kernels can form is such a value with K in [2^18, 2^40))
__global__ void k_nm_check_sqrt_rcp(unsigned long long k0, unsigned long long stride, unsigned long long count, unsigned long long *bad)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * blockDim.x) {
        const double x = (double)(k0 + stride * i) * (1.0 / 262144.0);
        const double a = nm_sqrt(x), b = sqrt(x);
        if (a != b || nm_rcp(b) != 1.0 / b) atomicAdd(bad, 1ull);
    }
}

// Four pixels per lane (w % 4 == 0): the three input rows arrive as one aligned 8-byte load each plus the two edge columns,
// the 12 output bytes leave as three aligned 32-bit stores (round 1: one pixel per lane, three single-byte stores -- 0.20 ms
// per 32 x 1024^2 = 10 % of the HBM roofline).
template <int SOBEL3>
__global__ __launch_bounds__(NM_BX * NM_BY) void k_normalmap_fused4(const uint16_t *__restrict__ depth, int h, int w, int invert,
                                                                    uint8_t *__restrict__ out)
{
    const int img = blockIdx.z;
    const int x0 = (blockIdx.x * NM_BX + threadIdx.x) * 4;
    const int y = blockIdx.y * NM_BY + threadIdx.y;
    if (x0 >= w || y >= h) return;
    const uint16_t *d = depth + (size_t)img * h * w;
    const double sgn = invert ? 1.0 : -1.0;                                                 /* :20: depthmap * (-1.0) unless inverted */
    // rows y-1, y, y+1 and columns x0-1 .. x0+4; outside the image: BORDER_REFLECT_101 for Sobel, clamped (unused) for gradient
    int ry[3], cl, cr;
    if (SOBEL3) {
        ry[0] = nm_reflect101(y - 1, h); ry[2] = nm_reflect101(y + 1, h);
        cl = nm_reflect101(x0 - 1, w); cr = nm_reflect101(x0 + 4, w);
    } else {
        ry[0] = y > 0 ? y - 1 : 0; ry[2] = y < h - 1 ? y + 1 : h - 1;
        cl = x0 > 0 ? x0 - 1 : 0; cr = x0 + 4 < w ? x0 + 4 : w - 1;
    }
    ry[1] = y;
    int e[3][6];                                                                            /* depth codes: value = sgn * e / 256 (:20-21) */
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const uint16_t *row = d + (size_t)ry[r] * w;
        const uint2 q = *reinterpret_cast<const uint2 *>(row + x0);
        e[r][0] = row[cl]; e[r][1] = (int)(q.x & 0xffffu); e[r][2] = (int)(q.x >> 16);
        e[r][3] = (int)(q.y & 0xffffu); e[r][4] = (int)(q.y >> 16); e[r][5] = row[cr];
    }
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = x0 + i;
        int Zx, Zy;
        double kx = sgn * (1.0 / 256.0), ky = -kx;
        if (SOBEL3) {
            Zx = (e[0][i + 2] - e[0][i]) + 2 * (e[1][i + 2] - e[1][i]) + (e[2][i + 2] - e[2][i]);        /* cv2.Sobel dx, ksize 3 */
            Zy = (e[2][i] - e[0][i]) + 2 * (e[2][i + 1] - e[0][i + 1]) + (e[2][i + 2] - e[0][i + 2]);    /* cv2.Sobel dy, ksize 3 */
        } else {                                                                                         /* np.gradient, :31 */
            if (x == 0) Zx = e[1][i + 2] - e[1][i + 1];
            else if (x == w - 1) Zx = e[1][i + 1] - e[1][i];
            else { Zx = e[1][i + 2] - e[1][i]; kx *= 0.5; }
            if (y == 0) Zy = e[2][i + 1] - e[1][i + 1];
            else if (y == h - 1) Zy = e[1][i + 1] - e[0][i + 1];
            else { Zy = e[2][i + 1] - e[0][i + 1]; ky *= 0.5; }
        }
        o[i] = nm_pixel(Zx, Zy, kx, ky);
    }
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + ((size_t)img * h * w + (size_t)y * w + x0) * 3);
    dst[0] = o[0] | (o[1] << 24);
    dst[1] = (o[1] >> 8) | (o[2] << 16);
    dst[2] = (o[2] >> 16) | (o[3] << 8);
}

// ---- general path ---------------------------------------------------------------------------------
#define NM_MAXK 63
struct NmKernel { double cf[NM_MAXK]; int n; };

template <typename T>
__global__ __launch_bounds__(256) void k_nm_load(const T *__restrict__ depth, int64_t count, int invert, double *__restrict__ plane)
{
    const double sgn = invert ? 1.0 : -1.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
        plane[i] = ((double)depth[i] * sgn) / 256.0;
}

// one separable pass along x (AXIS 0) or y (AXIS 1): out = sum_k cf[k] * in[reflect101(pos + k - r)]
template <int AXIS>
__global__ __launch_bounds__(256) void k_nm_sep(const double *__restrict__ in, double *__restrict__ out, int h, int w, NmKernel K)
{
    const int img = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const double *p = in + (size_t)img * h * w;
    const int r = K.n / 2;
    double acc = 0.0;
    for (int k = 0; k < K.n; k++) {
        double v;
        if (AXIS == 0) v = p[(size_t)y * w + nm_reflect101(x + k - r, w)];
        else v = p[(size_t)nm_reflect101(y + k - r, h) * w + x];
        acc = acc + K.cf[k] * v;
    }
    out[(size_t)img * h * w + (size_t)y * w + x] = acc;
}

__global__ __launch_bounds__(256) void k_nm_gradient(const double *__restrict__ in, double *__restrict__ zx, double *__restrict__ zy, int h, int w)
{
    const int img = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const double *p = in + (size_t)img * h * w;
#define PV(yy, xx) p[(size_t)(yy) * w + (xx)]
    double gx, gy;
    if (w == 1) gx = 0.0;
    else if (x == 0) gx = PV(y, 1) - PV(y, 0);
    else if (x == w - 1) gx = PV(y, w - 1) - PV(y, w - 2);
    else gx = (PV(y, x + 1) - PV(y, x - 1)) / 2.0;
    if (h == 1) gy = 0.0;
    else if (y == 0) gy = PV(1, x) - PV(0, x);
    else if (y == h - 1) gy = PV(h - 1, x) - PV(h - 2, x);
    else gy = (PV(y + 1, x) - PV(y - 1, x)) / 2.0;
#undef PV
    const size_t o = (size_t)img * h * w + (size_t)y * w + x;
    zx[o] = gx; zy[o] = gy;
}

// planes n0 = zx, n1 = zy (in) -> unit normal components in n0,n1,n2  (:34-39)
__global__ __launch_bounds__(256) void k_nm_normalize_first(double *__restrict__ n0, double *__restrict__ n1, double *__restrict__ n2, int64_t count)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const double a = n0[i], b = -n1[i], c = 1.0;
        const double n = sqrt(a * a + b * b + c * c);
        n0[i] = a / n; n1[i] = b / n; n2[i] = c / n;
    }
}

// optional renormalise (:45-48) then quantise (:51-54)
__global__ __launch_bounds__(256) void k_nm_finish(const double *__restrict__ n0, const double *__restrict__ n1, const double *__restrict__ n2,
                                                   int64_t count, int renorm, uint8_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        double v[3] = { n0[i], n1[i], n2[i] };
        if (renorm) {
            const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            v[0] = v[0] / n; v[1] = v[1] / n; v[2] = v[2] / n;
        }
        for (int k = 0; k < 3; k++) {
            double t = v[k] + 1.0;
            t = t / 2.0;
            t = t * 256.0;
            t = t < 0.0 ? 0.0 : t;
            t = t > 256.0 - 0.1 ? 256.0 - 0.1 : t;
            out[i * 3 + k] = ds_f64_to_u8(t);
        }
    }
}

// cv2.getGaussianKernel(ksize, sigma > 0, CV_64F): exp(-x^2/(2 sigma^2)) normalised, sequential sum
static void nm_gaussian(int ksize, double sigma, NmKernel *K)
{
    K->n = ksize;
    const double scale2x = -0.5 / (sigma * sigma);
    double sum = 0.0;
    for (int i = 0; i < ksize; i++) {
        const double x = (double)i - (double)(ksize - 1) * 0.5;
        K->cf[i] = exp(scale2x * x * x);
        sum += K->cf[i];
    }
    const double inv = 1.0 / sum;
    for (int i = 0; i < ksize; i++) K->cf[i] = K->cf[i] * inv;
}

// cv2.getDerivKernels / getSobelKernels for one axis (order 0 = smoothing, 1 = derivative)
static int nm_sobel(int ksize, int order, NmKernel *K)
{
    if (ksize == 1) {
        if (order == 0) { K->n = 1; K->cf[0] = 1.0; }
        else { K->n = 3; K->cf[0] = -1.0; K->cf[1] = 0.0; K->cf[2] = 1.0; }
        return 0;
    }
    if (ksize == 3) {
        K->n = 3;
        if (order == 0) { K->cf[0] = 1.0; K->cf[1] = 2.0; K->cf[2] = 1.0; }
        else { K->cf[0] = -1.0; K->cf[1] = 0.0; K->cf[2] = 1.0; }
        return 0;
    }
    if (ksize + 1 > NM_MAXK) return -1;
    double ker[NM_MAXK + 1];
    for (int i = 0; i <= ksize; i++) ker[i] = 0.0;
    ker[0] = 1.0;
    for (int i = 0; i < ksize - order - 1; i++) {
        double oldv = ker[0];
        for (int j = 1; j <= ksize; j++) { const double nv = ker[j] + ker[j - 1]; ker[j - 1] = oldv; oldv = nv; }
    }
    for (int i = 0; i < order; i++) {
        double oldv = -ker[0];
        for (int j = 1; j <= ksize; j++) { const double nv = ker[j - 1] - ker[j]; ker[j - 1] = oldv; oldv = nv; }
    }
    K->n = ksize;
    for (int i = 0; i < ksize; i++) K->cf[i] = ker[i];
    return 0;
}

static int nm_run(ds_ctx *ctx, const void *depth_any, int is_f64, int n, int h, int w, int pre_blur,
                  int sobel_ksize, int post_blur, int invert, uint8_t *out, void *stream)
{
    const uint16_t *depth = (const uint16_t *)depth_any;
    DS_REQUIRE(ctx && depth && out, DS_EINVAL, "ds_normalmap: null argument");
    DS_REQUIRE(n > 0 && h > 0 && w > 0, DS_EINVAL, "ds_normalmap: bad shape n=%d h=%d w=%d", n, h, w);
    DS_REQUIRE(n <= 65535, DS_EUNSUPPORTED, "ds_normalmap: n must be <= 65535");
    if (pre_blur < 0) pre_blur = 0;
    if (post_blur < 0) post_blur = 0;
    if (sobel_ksize < 0) sobel_ksize = 0;
    DS_REQUIRE((pre_blur == 0 || (pre_blur & 1)) && (post_blur == 0 || (post_blur & 1)), DS_EINVAL,
               "ds_normalmap: Gaussian kernel sizes must be odd (cv2.GaussianBlur asserts)");
    DS_REQUIRE(sobel_ksize == 0 || (sobel_ksize & 1), DS_EINVAL, "ds_normalmap: Sobel kernel size must be odd");
    DS_REQUIRE(pre_blur <= NM_MAXK && post_blur <= NM_MAXK && sobel_ksize < NM_MAXK, DS_EUNSUPPORTED,
               "ds_normalmap: kernel sizes above %d are not supported", NM_MAXK);
    if (sobel_ksize == 0) DS_REQUIRE(h >= 2 && w >= 2, DS_EINVAL, "ds_normalmap: np.gradient needs at least 2 samples per axis");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;

    if (!is_f64 && pre_blur == 0 && post_blur == 0 && (sobel_ksize == 3 || sobel_ksize == 0)) {
        dim3 grid((w + NM_BX - 1) / NM_BX, (h + NM_BY - 1) / NM_BY, n), block(NM_BX, NM_BY);
        DS_REQUIRE(grid.y <= 65535, DS_EUNSUPPORTED, "ds_normalmap: image too tall");
        if ((w & 3) == 0 && w >= 4 && h >= 2 && (((uintptr_t)depth) & 7) == 0 && (((uintptr_t)out) & 3) == 0 && !getenv("DS_NM_SCALAR")) {
            dim3 grid4((w / 4 + NM_BX - 1) / NM_BX, grid.y, n);
            const int kt = ds_kt_begin(ctx, DS_KT_NORMALMAP, st);
            if (sobel_ksize == 3) hipLaunchKernelGGL(k_normalmap_fused4<1>, grid4, block, 0, st, depth, h, w, invert ? 1 : 0, out);
            else hipLaunchKernelGGL(k_normalmap_fused4<0>, grid4, block, 0, st, depth, h, w, invert ? 1 : 0, out);
            ds_kt_end(ctx, DS_KT_NORMALMAP, kt, st);
            DS_HIP_CHECK(hipGetLastError());
            return DS_OK;
        }
        if (sobel_ksize == 3) hipLaunchKernelGGL(k_normalmap_fused<1>, grid, block, 0, st, depth, h, w, invert ? 1 : 0, out);
        else hipLaunchKernelGGL(k_normalmap_fused<0>, grid, block, 0, st, depth, h, w, invert ? 1 : 0, out);
        DS_HIP_CHECK(hipGetLastError());
        return DS_OK;
    }

    // general path: 5 float64 planes in scratch
    const int64_t count = (int64_t)n * h * w;
    int rc = ds_ctx_reserve(ctx, &ctx->tmp_a, &ctx->tmp_a_bytes, (size_t)count * sizeof(double) * 5);
    if (rc) return rc;
    double *A = (double *)ctx->tmp_a, *B = A + count, *C = B + count, *D = C + count, *E = D + count;
    int nb = (int)((count + 1023) / 1024); if (nb > 4096) nb = 4096;
    dim3 g2((w + 63) / 64, (h + 3) / 4, n);
    DS_REQUIRE(g2.y <= 65535, DS_EUNSUPPORTED, "ds_normalmap: image too tall");
    if (is_f64) hipLaunchKernelGGL(k_nm_load<double>, dim3(nb), dim3(256), 0, st, (const double *)depth_any, count, invert ? 1 : 0, A);
    else hipLaunchKernelGGL(k_nm_load<uint16_t>, dim3(nb), dim3(256), 0, st, depth, count, invert ? 1 : 0, A);
    NmKernel G, KD, KS;
    if (pre_blur > 0) {                                            // :23-24
        nm_gaussian(pre_blur, (double)pre_blur, &G);
        hipLaunchKernelGGL(k_nm_sep<0>, g2, dim3(256), 0, st, A, B, h, w, G);
        hipLaunchKernelGGL(k_nm_sep<1>, g2, dim3(256), 0, st, B, A, h, w, G);
    }
    // gradients: zx -> C, zy -> D
    if (sobel_ksize > 0) {                                         // :27-29
        DS_REQUIRE(nm_sobel(sobel_ksize, 1, &KD) == 0 && nm_sobel(sobel_ksize, 0, &KS) == 0, DS_EUNSUPPORTED, "ds_normalmap: Sobel size");
        hipLaunchKernelGGL(k_nm_sep<0>, g2, dim3(256), 0, st, A, B, h, w, KD);
        hipLaunchKernelGGL(k_nm_sep<1>, g2, dim3(256), 0, st, B, C, h, w, KS);
        hipLaunchKernelGGL(k_nm_sep<0>, g2, dim3(256), 0, st, A, B, h, w, KS);
        hipLaunchKernelGGL(k_nm_sep<1>, g2, dim3(256), 0, st, B, D, h, w, KD);
    } else {
        hipLaunchKernelGGL(k_nm_gradient, g2, dim3(256), 0, st, A, C, D, h, w);
    }
    hipLaunchKernelGGL(k_nm_normalize_first, dim3(nb), dim3(256), 0, st, C, D, E, count);
    if (post_blur > 0) {                                           // :42-48
        nm_gaussian(post_blur, (double)post_blur, &G);
        double *planes[3] = { C, D, E };
        for (int k = 0; k < 3; k++) {
            hipLaunchKernelGGL(k_nm_sep<0>, g2, dim3(256), 0, st, planes[k], B, h, w, G);
            hipLaunchKernelGGL(k_nm_sep<1>, g2, dim3(256), 0, st, B, planes[k], h, w, G);
        }
    }
    hipLaunchKernelGGL(k_nm_finish, dim3(nb), dim3(256), 0, st, C, D, E, count, post_blur > 0 ? 1 : 0, out);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

DS_API int ds_normalmap(ds_ctx *ctx, const uint16_t *depth, int n, int h, int w, int pre_blur,
                        int sobel_ksize, int post_blur, int invert, uint8_t *out, void *stream)
{
    return nm_run(ctx, depth, 0, n, h, w, pre_blur, sobel_ksize, post_blur, invert, out, stream);
}

// float32 depth with sobel_gradient None / <= 0 (src/normalmap_generation.py:31): the one combination the reference runs in
// FLOAT32 from end to end -- `depthmap * (-1.0) / 256.0` keeps float32 (:20-21), np.gradient, np.dstack, np.linalg.norm
// (x * x element-wise, the three squares added left to right, sqrt), the three divisions and `+= 1; /= 2; * 256; clip` are all
// float32 operations, each correctly rounded (IEEE binary32; hipcc's default float32 division and sqrt are the correctly
// rounded ones, and the library is built with -ffp-contract=off): one fused pass, bit-identical to numpy.
__global__ __launch_bounds__(256) void k_nm_gradient_f32(const float *__restrict__ depth, int h, int w, int invert, uint8_t *__restrict__ out)
{
    const int img = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float *p = depth + (size_t)img * h * w;
    const float sgn = invert ? 1.0f : -1.0f;
#define PV(yy, xx) ((p[(size_t)(yy) * w + (xx)] * sgn) / 256.0f)
    float gx, gy;
    if (x == 0) gx = (PV(y, 1) - PV(y, 0)) / 1.0f;
    else if (x == w - 1) gx = (PV(y, w - 1) - PV(y, w - 2)) / 1.0f;
    else gx = (PV(y, x + 1) - PV(y, x - 1)) / 2.0f;
    if (y == 0) gy = (PV(1, x) - PV(0, x)) / 1.0f;
    else if (y == h - 1) gy = (PV(h - 1, x) - PV(h - 2, x)) / 1.0f;
    else gy = (PV(y + 1, x) - PV(y - 1, x)) / 2.0f;
#undef PV
    const float a = gx, b = -gy, c = 1.0f;
    float s = a * a;
    s = s + b * b;
    s = s + c * c;
    const float n = __fsqrt_rn(s);
    const float v[3] = { __fdiv_rn(a, n), __fdiv_rn(b, n), __fdiv_rn(c, n) };
    uint8_t *o = out + ((size_t)img * h * w + (size_t)y * w + x) * 3;
    for (int k = 0; k < 3; k++) {
        float t = v[k] + 1.0f;
        t = t / 2.0f;
        t = t * 256.0f;
        t = t < 0.0f ? 0.0f : t;                             // np.clip(., 0, 256 - 0.1): the bounds become float32 (NEP 50)
        t = t > (float)(256 - 0.1) ? (float)(256 - 0.1) : t;
        o[k] = (t == t) ? (uint8_t)(int)t : (uint8_t)0;
    }
}

DS_API int ds_normalmap_gradient_f32(ds_ctx *ctx, const float *depth, int n, int h, int w, int invert, uint8_t *out, void *stream)
{
    DS_REQUIRE(ctx && depth && out, DS_EINVAL, "ds_normalmap_gradient_f32: null argument");
    DS_REQUIRE(n > 0 && n <= 65535 && h >= 2 && w >= 2, DS_EINVAL, "ds_normalmap_gradient_f32: np.gradient needs at least 2 samples per axis (n=%d h=%d w=%d)", n, h, w);
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    dim3 g2((w + 63) / 64, (h + 3) / 4, n);
    DS_REQUIRE(g2.y <= 65535, DS_EUNSUPPORTED, "ds_normalmap_gradient_f32: image too tall");
    hipLaunchKernelGGL(k_nm_gradient_f32, g2, dim3(256), 0, (hipStream_t)stream, depth, h, w, invert ? 1 : 0, out);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

// float16 depth with sobel_gradient None / <= 0 and no blur (round 6): numpy keeps float16 from end to end (:20-21 `* (-1.0)` and
// `/ 256.0` with Python scalars do not promote; np.gradient allocates its output in the input's inexact dtype; np.dstack,
// np.linalg.norm, the in-place divisions and the quantisation stay float16).  numpy's float16 loops convert the operands to
// float32, operate, and round the float32 result to half: that IS the correctly rounded half operation for + - * / sqrt
// (24 >= 2 * 11 + 2 bits), and it is what this kernel does, operation by operation.  np.linalg.norm's add.reduce over the three
// squares is the one multi-operand step: HALF_add's reduce loop keeps a FLOAT32 accumulator, adds the squares left to right and
// rounds to half once (established on numpy 2.2 with operand triples on which (s0 + s1) + s2, s0 + (s1 + s2) and the
// half-rounded chain differ: a CPU test holds numpy to it).
__device__ __forceinline__ float nm_r16(float x) { return (float)(_Float16)x; }

__global__ __launch_bounds__(256) void k_nm_gradient_f16(const _Float16 *__restrict__ depth, int h, int w, int invert, uint8_t *__restrict__ out)
{
    const int img = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const _Float16 *p = depth + (size_t)img * h * w;
    const float sgn = invert ? 1.0f : -1.0f;
#define PV(yy, xx) nm_r16(nm_r16((float)p[(size_t)(yy) * w + (xx)] * sgn) / 256.0f)
    float gx, gy;
    if (x == 0) gx = nm_r16(nm_r16(PV(y, 1) - PV(y, 0)) / 1.0f);
    else if (x == w - 1) gx = nm_r16(nm_r16(PV(y, w - 1) - PV(y, w - 2)) / 1.0f);
    else gx = nm_r16(nm_r16(PV(y, x + 1) - PV(y, x - 1)) / 2.0f);
    if (y == 0) gy = nm_r16(nm_r16(PV(1, x) - PV(0, x)) / 1.0f);
    else if (y == h - 1) gy = nm_r16(nm_r16(PV(h - 1, x) - PV(h - 2, x)) / 1.0f);
    else gy = nm_r16(nm_r16(PV(y + 1, x) - PV(y - 1, x)) / 2.0f);
#undef PV
    const float a = gx, b = -gy, c = 1.0f;
    const float sa = nm_r16(a * a), sb = nm_r16(b * b), sc = nm_r16(c * c);
    float sum = sa + sb;                                             // float32 accumulator, left to right, ONE rounding to half
    sum = sum + sc;
    const float n = nm_r16(__fsqrt_rn(nm_r16(sum)));
    const float v[3] = { nm_r16(__fdiv_rn(a, n)), nm_r16(__fdiv_rn(b, n)), nm_r16(__fdiv_rn(c, n)) };
    uint8_t *o = out + ((size_t)img * h * w + (size_t)y * w + x) * 3;
    const float hi = (float)(_Float16)(256 - 0.1);                  // np.clip's bounds become float16 (NEP 50): 255.875
    for (int k = 0; k < 3; k++) {
        float t = nm_r16(v[k] + 1.0f);
        t = nm_r16(t / 2.0f);
        t = nm_r16(t * 256.0f);
        t = t < 0.0f ? 0.0f : t;
        t = t > hi ? hi : t;
        o[k] = (t == t) ? (uint8_t)(int)t : (uint8_t)0;
    }
}

DS_API int ds_normalmap_gradient_f16(ds_ctx *ctx, const void *depth, int n, int h, int w, int invert, uint8_t *out, void *stream)
{
    DS_REQUIRE(ctx && depth && out, DS_EINVAL, "ds_normalmap_gradient_f16: null argument");
    DS_REQUIRE(n > 0 && n <= 65535 && h >= 2 && w >= 2, DS_EINVAL, "ds_normalmap_gradient_f16: np.gradient needs at least 2 samples per axis (n=%d h=%d w=%d)", n, h, w);
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    dim3 g2((w + 63) / 64, (h + 3) / 4, n);
    DS_REQUIRE(g2.y <= 65535, DS_EUNSUPPORTED, "ds_normalmap_gradient_f16: image too tall");
    hipLaunchKernelGGL(k_nm_gradient_f16, g2, dim3(256), 0, (hipStream_t)stream, (const _Float16 *)depth, h, w, invert ? 1 : 0, out);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

// float32 depth with np.gradient AND a Gaussian blur in front of and / or behind it (round 6): the reference hands float32 arrays to
// cv2.GaussianBlur (:24 the depth plane, :43 the three-channel normal), which filters CV_32F data with CV_32F coefficients
// (createGaussianKernels takes max(depth, CV_32F); getGaussianKernel rounds exp() to float, sums the floats in double, scales by
// the double reciprocal and rounds again) and float32 accumulators; everything around the blurs is numpy float32 as in
// k_nm_gradient_f32.  OpenCV's own summation order inside the separable filter is not reproducible without the library (cv2 is
// absent from the image: the blur arithmetic is a stand-in, tests hold it to one LSB); the structure is the reference's.
struct NmKernelF { float cf[NM_MAXK]; int n; };

__global__ __launch_bounds__(256) void k_nm_load_f32(const float *__restrict__ depth, int64_t count, int invert, float *__restrict__ plane)
{
    const float sgn = invert ? 1.0f : -1.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
        plane[i] = (depth[i] * sgn) / 256.0f;
}

template <int AXIS>
__global__ __launch_bounds__(256) void k_nm_sep_f32(const float *__restrict__ in, float *__restrict__ out, int h, int w, NmKernelF K)
{
    const int img = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float *p = in + (size_t)img * h * w;
    const int r = K.n / 2;
    float acc = 0.0f;
    for (int k = 0; k < K.n; k++) {
        float v;
        if (AXIS == 0) v = p[(size_t)y * w + nm_reflect101(x + k - r, w)];
        else v = p[(size_t)nm_reflect101(y + k - r, h) * w + x];
        acc = acc + K.cf[k] * v;
    }
    out[(size_t)img * h * w + (size_t)y * w + x] = acc;
}

// np.gradient (:31) of the (blurred) plane and the first normalisation (:33-39), float32 like numpy: planes n0, n1, n2 out
__global__ __launch_bounds__(256) void k_nm_gradnorm_f32(const float *__restrict__ in, float *__restrict__ n0, float *__restrict__ n1,
                                                         float *__restrict__ n2, int h, int w)
{
    const int img = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float *p = in + (size_t)img * h * w;
#define PV(yy, xx) p[(size_t)(yy) * w + (xx)]
    float gx, gy;
    if (x == 0) gx = (PV(y, 1) - PV(y, 0)) / 1.0f;
    else if (x == w - 1) gx = (PV(y, w - 1) - PV(y, w - 2)) / 1.0f;
    else gx = (PV(y, x + 1) - PV(y, x - 1)) / 2.0f;
    if (y == 0) gy = (PV(1, x) - PV(0, x)) / 1.0f;
    else if (y == h - 1) gy = (PV(h - 1, x) - PV(h - 2, x)) / 1.0f;
    else gy = (PV(y + 1, x) - PV(y - 1, x)) / 2.0f;
#undef PV
    const float a = gx, b = -gy, c = 1.0f;
    float s = a * a;
    s = s + b * b;
    s = s + c * c;
    const float n = __fsqrt_rn(s);
    const size_t o = (size_t)img * h * w + (size_t)y * w + x;
    n0[o] = __fdiv_rn(a, n); n1[o] = __fdiv_rn(b, n); n2[o] = __fdiv_rn(c, n);
}

__global__ __launch_bounds__(256) void k_nm_finish_f32(const float *__restrict__ n0, const float *__restrict__ n1, const float *__restrict__ n2,
                                                       int64_t count, int renorm, uint8_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        float v[3] = { n0[i], n1[i], n2[i] };
        if (renorm) {                                               // :45-48
            float s = v[0] * v[0];
            s = s + v[1] * v[1];
            s = s + v[2] * v[2];
            const float n = __fsqrt_rn(s);
            v[0] = __fdiv_rn(v[0], n); v[1] = __fdiv_rn(v[1], n); v[2] = __fdiv_rn(v[2], n);
        }
        for (int k = 0; k < 3; k++) {                               // :51-54
            float t = v[k] + 1.0f;
            t = t / 2.0f;
            t = t * 256.0f;
            t = t < 0.0f ? 0.0f : t;
            t = t > (float)(256 - 0.1) ? (float)(256 - 0.1) : t;
            out[i * 3 + k] = (t == t) ? (uint8_t)(int)t : (uint8_t)0;
        }
    }
}

static void nm_gaussian_f32(int ksize, double sigma, NmKernelF *K)   // getGaussianKernel(ksize, sigma > 0, CV_32F)
{
    K->n = ksize;
    const double scale2x = -0.5 / (sigma * sigma);
    double sum = 0.0;
    for (int i = 0; i < ksize; i++) {
        const double x = (double)i - (double)(ksize - 1) * 0.5;
        K->cf[i] = (float)exp(scale2x * x * x);
        sum += K->cf[i];
    }
    const double inv = 1.0 / sum;
    for (int i = 0; i < ksize; i++) K->cf[i] = (float)(K->cf[i] * inv);
}

DS_API int ds_normalmap_gradient_blur_f32(ds_ctx *ctx, const float *depth, int n, int h, int w, int pre_blur, int post_blur, int invert,
                                          uint8_t *out, void *stream)
{
    DS_REQUIRE(ctx && depth && out, DS_EINVAL, "ds_normalmap_gradient_blur_f32: null argument");
    DS_REQUIRE(n > 0 && n <= 65535 && h >= 2 && w >= 2, DS_EINVAL, "ds_normalmap_gradient_blur_f32: np.gradient needs at least 2 samples per axis (n=%d h=%d w=%d)", n, h, w);
    if (pre_blur < 0) pre_blur = 0;
    if (post_blur < 0) post_blur = 0;
    if (pre_blur == 0 && post_blur == 0) return ds_normalmap_gradient_f32(ctx, depth, n, h, w, invert, out, stream);
    DS_REQUIRE((pre_blur == 0 || (pre_blur & 1)) && (post_blur == 0 || (post_blur & 1)), DS_EINVAL,
               "ds_normalmap_gradient_blur_f32: Gaussian kernel sizes must be odd (cv2.GaussianBlur asserts)");
    DS_REQUIRE(pre_blur <= NM_MAXK && post_blur <= NM_MAXK, DS_EUNSUPPORTED, "ds_normalmap_gradient_blur_f32: kernel sizes above %d are not supported", NM_MAXK);
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t count = (int64_t)n * h * w;
    int rc = ds_ctx_reserve(ctx, &ctx->tmp_a, &ctx->tmp_a_bytes, (size_t)count * sizeof(float) * 5);
    if (rc) return rc;
    float *A = (float *)ctx->tmp_a, *B = A + count, *C = B + count, *D = C + count, *E = D + count;
    int nb = (int)((count + 1023) / 1024); if (nb > 4096) nb = 4096;
    dim3 g2((w + 63) / 64, (h + 3) / 4, n);
    DS_REQUIRE(g2.y <= 65535, DS_EUNSUPPORTED, "ds_normalmap_gradient_blur_f32: image too tall");
    hipLaunchKernelGGL(k_nm_load_f32, dim3(nb), dim3(256), 0, st, depth, count, invert ? 1 : 0, A);
    NmKernelF G;
    if (pre_blur > 0) {                                            // :23-24
        nm_gaussian_f32(pre_blur, (double)pre_blur, &G);
        hipLaunchKernelGGL(k_nm_sep_f32<0>, g2, dim3(256), 0, st, A, B, h, w, G);
        hipLaunchKernelGGL(k_nm_sep_f32<1>, g2, dim3(256), 0, st, B, A, h, w, G);
    }
    hipLaunchKernelGGL(k_nm_gradnorm_f32, g2, dim3(256), 0, st, A, C, D, E, h, w);
    if (post_blur > 0) {                                           // :42-48
        nm_gaussian_f32(post_blur, (double)post_blur, &G);
        float *planes[3] = { C, D, E };
        for (int k = 0; k < 3; k++) {
            hipLaunchKernelGGL(k_nm_sep_f32<0>, g2, dim3(256), 0, st, planes[k], B, h, w, G);
            hipLaunchKernelGGL(k_nm_sep_f32<1>, g2, dim3(256), 0, st, B, planes[k], h, w, G);
        }
    }
    hipLaunchKernelGGL(k_nm_finish_f32, dim3(nb), dim3(256), 0, st, C, D, E, count, post_blur > 0 ? 1 : 0, out);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

// Any other real dtype of the reference's `depthmap` argument (:20-21 promote it to float64; the host casts): the separable
// float64 passes for every kernel size (general float64 data has no exact 3x3 shortcut).
DS_API int ds_normalmap_f64(ds_ctx *ctx, const double *depth, int n, int h, int w, int pre_blur,
                            int sobel_ksize, int post_blur, int invert, uint8_t *out, void *stream)
{
    return nm_run(ctx, depth, 1, n, h, w, pre_blur, sobel_ksize, post_blur, invert, out, stream);
}

DS_API int ds_normalmap_selfcheck(ds_ctx *ctx, unsigned long long k0, unsigned long long stride, unsigned long long count,
                                  unsigned long long *mismatches, void *stream)
{
    DS_REQUIRE(ctx && mismatches, DS_EINVAL, "ds_normalmap_selfcheck: null argument");
    DS_REQUIRE(k0 >= (1ull << 18) && stride >= 1 && count >= 1 && k0 + stride * (count - 1) < (1ull << 40), DS_EINVAL,
               "ds_normalmap_selfcheck: K = k0 + stride * i must stay in [2^18, 2^40)");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *d_bad = nullptr;
    DS_HIP_CHECK(hipMalloc(&d_bad, sizeof(*d_bad)));
    DS_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(*d_bad), st));
    hipLaunchKernelGGL(k_nm_check_sqrt_rcp, dim3(4096), dim3(256), 0, st, k0, stride, count, d_bad);
    hipError_t e = hipMemcpyAsync(mismatches, d_bad, sizeof(*d_bad), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_bad);
    DS_HIP_CHECK(e);
    return DS_OK;
}
