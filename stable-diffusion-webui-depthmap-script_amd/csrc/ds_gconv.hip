// ds_gconv3x3_nhwc_f32: the grouped 3 x 3 convolutions of the ResNeXt-101 32x8d encoder of LeReS in float32 -- `conv2` of every
// Bottleneck (lib/Resnext_torch.py:96-118 of the reference: conv1 1x1 -> bn -> relu -> conv2 3x3, groups = 32 -> bn -> relu -> conv3
// 1x1 -> bn, + identity, relu), the base estimator of Boost (src/depthmap_generation.py:406-440, :774-941; Boost never runs it in
// half: :271).  With 32 groups a group is 8 / 16 / 32 / 64 channels wide (layers 1 .. 4): per group a GEMM with N = 8 .. 64 and
// K = 72 .. 576, which the library's implicit-GEMM convolutions run at 5.5 TF/s (layer 1: 2.7 ms per 8 x 224^2 x 256, 0.3 TB/s of
// input + output) to 37 TF/s -- 40 % of the encoder's convolution time for 7 % of its flops (profiles/round5_c4_conv_probe.txt).
//
// This kernel is a direct convolution on the float32 vector pipe (float32 MFMA has the vector rate on gfx950, and N = 8 fills no
// MFMA tile): a 256-thread workgroup owns an 8 x 32 tile of output pixels and a block of 32 channels (4 / 2 / 1 groups); the
// (8 + 2) x (32 + 2) input pixels of those 32 channels are staged ONCE in LDS (128-byte pieces of NHWC rows: coalesced; pixel
// stride padded to 36 words, so the 16-byte reads of 16 consecutive pixels fall into 16 different bank quads), one thread = one output
// pixel.  The weights of a (group, tap, input channel) are the SAME for every lane of a wave: they are read through the scalar path
// (uniform addresses -> s_load into SGPRs, the FMA takes the SGPR as an operand) -- through LDS they would be broadcast reads that
// keep the LDS pipe busier than the four SIMDs' FMAs.  Bias (the folded BatchNorm) and ReLU in the epilogue.
// Bound: float32 vector FMA issue (64 flop per clock and SIMD, 157 TF/s); HBM traffic = read the input once per 32-channel block
// (+ the halo), write the output once.
#include "ds_common.h"

#define GC_TH 8
#define GC_TW 32
#define GC_CB 32                  // channels per workgroup
#define GC_PS 36                  // LDS words per staged pixel (32 + 4: the conflict-free stride for 16-byte reads)
#define GC_IH (GC_TH + 2)
#define GC_IW (GC_TW + 2)

// ReLU as torch computes it: a NaN activation stays NaN (fmaxf(v, 0) would turn it into 0 and hide it from the output)
__device__ __forceinline__ float gc_relu(float v) { return v < 0.f ? 0.f : v; }

// x [batch, H, W, C] float32 NHWC; wt [C / CPG groups][9 taps][CPG in][CPG out]; bias [C] or null; y [batch, H, W, C]
// GC_NT threads: two sets of 256 share one staged tile -- set `part` renders every other (group, half) unit of the 32-channel block for
// the same 256 pixels.  The tile's 49 KB of LDS allow three workgroups per CU; with 256 threads that was 3 waves per SIMD for a kernel
// whose inner loop waits on scalar weight loads and LDS reads, with 512 it is 6 (26-33 VGPRs: registers are no limit).
#define GC_NT 512
template <int CPG>
__global__ __launch_bounds__(GC_NT) void k_gconv3x3_nhwc_f32(const float *__restrict__ x, const float *__restrict__ wt, const float *__restrict__ bias,
                                                             float *__restrict__ y, int H, int W, int C, int tiles_x, int tiles_y, int relu)
{
    __shared__ __attribute__((aligned(16))) float tile[GC_IH * GC_IW * GC_PS];
    constexpr int COT = CPG < 16 ? CPG : 16;                 // output channels per pass (the accumulators of one thread)
    constexpr int HALVES = CPG / COT, UNITS = (GC_CB / CPG) * HALVES;      // 4 / 2 / 2 units for 8 / 16 / 32 channels per group
    const int tid = threadIdx.x & 255;
    const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));      // uniform per wave: the weights stay scalar loads
    int t = blockIdx.x;
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int b = t / tiles_y;
    const int c0 = blockIdx.y * GC_CB;                       // first channel of this workgroup's block
    const int oy0 = by * GC_TH, ox0 = bx * GC_TW;
    const float *xb = x + (size_t)b * H * W * C + c0;

    // ---- stage the input tile: (pixel, 16-byte piece) pairs, 8 pieces per pixel; out-of-image pixels are the zero padding ----
    for (int i = threadIdx.x; i < GC_IH * GC_IW * (GC_CB / 4); i += GC_NT) {
        const int piece = i & (GC_CB / 4 - 1), pix = i / (GC_CB / 4);
        const int iy = pix / GC_IW, ix = pix - iy * GC_IW;
        const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *(const float4 *)(xb + ((size_t)gy * W + gx) * C + piece * 4);
        *(float4 *)(tile + pix * GC_PS + piece * 4) = v;
    }
    __syncthreads();

    const int ty = tid >> 5, tx = tid & 31;
    const int oy = oy0 + ty, ox = ox0 + tx;
    const bool live = oy < H && ox < W;
    float *yp = y + (((size_t)b * H + oy) * W + ox) * C + c0;
#pragma unroll 1
    for (int unit = part; unit < UNITS; unit += GC_NT / 256) {
        const int g = unit / HALVES, half = unit - g * HALVES;
        const int grp = (c0 + g * CPG) / CPG;                // uniform: the weights below are scalar loads
        const float *wg = wt + (size_t)grp * 9 * CPG * CPG;
        {
            float acc[COT];
#pragma unroll
            for (int co = 0; co < COT; ++co) acc[co] = bias ? bias[c0 + g * CPG + half * COT + co] : 0.f;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - 3 * dy;
                const float *src = tile + ((ty + dy) * GC_IW + tx + dx) * GC_PS + g * CPG;
                const float *wtap = wg + (size_t)tap * CPG * CPG + half * COT;
#pragma unroll 2
                for (int ci4 = 0; ci4 < CPG / 4; ++ci4) {
                    const float4 xin = *(const float4 *)(src + ci4 * 4);
                    const float xv[4] = {xin.x, xin.y, xin.z, xin.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float *wrow = wtap + (size_t)(ci4 * 4 + j) * CPG;          // [ci][co]: COT consecutive weights
#pragma unroll
                        for (int co = 0; co < COT; ++co) acc[co] = __builtin_fmaf(xv[j], wrow[co], acc[co]);
                    }
                }
            }
            if (live) {
#pragma unroll
                for (int co = 0; co < COT; co += 4) {
                    float4 o = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
                    if (relu) { o.x = gc_relu(o.x); o.y = gc_relu(o.y); o.z = gc_relu(o.z); o.w = gc_relu(o.w); }
                    *(float4 *)(yp + g * CPG + half * COT + co) = o;
                }
            }
        }
    }
}

// Replaces `F.relu(F.conv2d(x, w, b, stride=1, padding=1, groups=C / cpg))` for channels_last float32 tensors: conv2 + folded bn2 + relu
// of the ResNeXt bottlenecks (lib/Resnext_torch.py:104-110; the product folds the BatchNorm into weight and bias once per module).
DS_API int ds_gconv3x3_nhwc_f32(ds_ctx *ctx, const float *x, const float *w_gtio, const float *bias, float *y, int batch, int height, int width,
                                int channels, int channels_per_group, int relu, void *stream)
{
    DS_REQUIRE(ctx && x && w_gtio && y, DS_EINVAL, "ds_gconv3x3_nhwc_f32: null argument");
    DS_REQUIRE(batch > 0 && height > 0 && width > 0 && channels > 0, DS_EINVAL, "ds_gconv3x3_nhwc_f32: bad shape");
    DS_REQUIRE(channels_per_group == 8 || channels_per_group == 16 || channels_per_group == 32, DS_EUNSUPPORTED,
               "ds_gconv3x3_nhwc_f32: channels per group must be 8, 16 or 32 (got %d)", channels_per_group);
    DS_REQUIRE(channels % GC_CB == 0, DS_EUNSUPPORTED, "ds_gconv3x3_nhwc_f32: channels must be a multiple of %d", GC_CB);
    DS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)w_gtio & 15) == 0 && x != y, DS_EINVAL,
               "ds_gconv3x3_nhwc_f32: x, w and y must be 16-byte aligned, y must not alias x");
    const int tiles_x = (width + GC_TW - 1) / GC_TW, tiles_y = (height + GC_TH - 1) / GC_TH;
    DS_REQUIRE((long long)tiles_x * tiles_y * batch < (1ll << 31) && channels / GC_CB <= 65535, DS_EUNSUPPORTED, "ds_gconv3x3_nhwc_f32: grid too large");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    dim3 grid((unsigned)(tiles_x * tiles_y * batch), (unsigned)(channels / GC_CB)), block(GC_NT);
    hipStream_t st = (hipStream_t)stream;
    switch (channels_per_group) {
    case 8: hipLaunchKernelGGL(k_gconv3x3_nhwc_f32<8>, grid, block, 0, st, x, w_gtio, bias, y, height, width, channels, tiles_x, tiles_y, relu ? 1 : 0); break;
    case 16: hipLaunchKernelGGL(k_gconv3x3_nhwc_f32<16>, grid, block, 0, st, x, w_gtio, bias, y, height, width, channels, tiles_x, tiles_y, relu ? 1 : 0); break;
    default: hipLaunchKernelGGL(k_gconv3x3_nhwc_f32<32>, grid, block, 0, st, x, w_gtio, bias, y, height, width, channels, tiles_x, tiles_y, relu ? 1 : 0); break;
    }
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

// y = relu(a + b) on float32 arrays: the tail of every ResNeXt bottleneck, `self.relu(out + identity)` (lib/Resnext_torch.py:115-118),
// which torch runs as an add pass and a clamp pass over the stage's widest activation.
__global__ __launch_bounds__(256) void k_add_relu_f32(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ y, long long n4)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 p = a[i], q = b[i];
        y[i] = make_float4(gc_relu(p.x + q.x), gc_relu(p.y + q.y), gc_relu(p.z + q.z), gc_relu(p.w + q.w));
    }
}

DS_API int ds_add_relu_f32(ds_ctx *ctx, const float *a, const float *b, float *y, int64_t count, void *stream)
{
    DS_REQUIRE(ctx && a && b && y, DS_EINVAL, "ds_add_relu_f32: null argument");
    DS_REQUIRE(count > 0 && count % 4 == 0, DS_EINVAL, "ds_add_relu_f32: count must be a positive multiple of 4");
    DS_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && ((uintptr_t)y & 15) == 0, DS_EINVAL, "ds_add_relu_f32: 16-byte alignment");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    const long long n4 = count / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_add_relu_f32, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)a, (const float4 *)b, (float4 *)y, n4);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

// y = [relu]((x + bias[c]) [+ res]) on float32 NHWC rows (in place when y == x): what follows a LIBRARY convolution of LeReS in the
// reference's order -- the folded BatchNorm's bias (torch adds a convolution's bias in a pass of its own on ROCm), the ReLU
// (lib/Resnext_torch.py:100-102: conv1 -> bn1 -> relu), and for conv3 the shortcut add with its ReLU (:112-118); the decoder's
// conv -> bias -> ReLU / + x pairs (lib/network_auxi.py:116-121).  One pass instead of two or three.  c4 = channels / 4.
template <int RELU, int RES>
__global__ __launch_bounds__(256) void k_bias_act_f32(const float4 *__restrict__ x, const float4 *__restrict__ bias, const float4 *__restrict__ res,
                                                       float4 *__restrict__ y, long long n4, int c4)
{
    // (the channel group of element i is carried along the grid stride: no 64-bit division per element)
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const int step = (int)(((long long)gridDim.x * 256) % c4);
    int c = (int)(i0 % c4);
    for (long long i = i0; i < n4; i += (long long)gridDim.x * 256, c = c + step >= c4 ? c + step - c4 : c + step) {
        const float4 p = x[i], b = bias[c];
        float4 v = make_float4(p.x + b.x, p.y + b.y, p.z + b.z, p.w + b.w);
        if (RES) {
            const float4 q = res[i];
            v = make_float4(v.x + q.x, v.y + q.y, v.z + q.z, v.w + q.w);
        }
        if (RELU) v = make_float4(gc_relu(v.x), gc_relu(v.y), gc_relu(v.z), gc_relu(v.w));
        y[i] = v;
    }
}

DS_API int ds_bias_act_f32(ds_ctx *ctx, const float *x, const float *bias, const float *res, float *y, int64_t pixels, int channels, int relu,
                           void *stream)
{
    DS_REQUIRE(ctx && x && bias && y, DS_EINVAL, "ds_bias_act_f32: null argument");
    DS_REQUIRE(pixels > 0 && channels > 0 && channels % 4 == 0, DS_EINVAL, "ds_bias_act_f32: channels must be a positive multiple of 4");
    DS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)res & 15) == 0 && ((uintptr_t)y & 15) == 0, DS_EINVAL,
               "ds_bias_act_f32: 16-byte alignment");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    const long long n4 = (long long)pixels * (channels / 4);
    long long blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const dim3 g((unsigned)blocks), b(256);
    hipStream_t st = (hipStream_t)stream;
    const float4 *x4 = (const float4 *)x, *b4 = (const float4 *)bias, *r4 = (const float4 *)res;
    if (relu && res) hipLaunchKernelGGL((k_bias_act_f32<1, 1>), g, b, 0, st, x4, b4, r4, (float4 *)y, n4, channels / 4);
    else if (relu) hipLaunchKernelGGL((k_bias_act_f32<1, 0>), g, b, 0, st, x4, b4, r4, (float4 *)y, n4, channels / 4);
    else if (res) hipLaunchKernelGGL((k_bias_act_f32<0, 1>), g, b, 0, st, x4, b4, r4, (float4 *)y, n4, channels / 4);
    else hipLaunchKernelGGL((k_bias_act_f32<0, 0>), g, b, 0, st, x4, b4, r4, (float4 *)y, n4, channels / 4);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

// y[p] = relu(cat(a[p], b[p])) on float32 NHWC rows: the skip concatenation of the pix2pix U-Net's up path and the ReLU the next
// level applies to it (pix2pix/models/networks.py:545-550 `torch.cat([x, self.model(x)], 1)` followed by the parent's `uprelu`, :519):
// the concatenated tensor is read by nothing else, so it is written rectified -- one pass instead of a copy pass and a clamp pass.
__global__ __launch_bounds__(256) void k_relu_cat_f32(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ y, long long n4,
                                                       int ca4, int cb4)
{
    const int ct4 = ca4 + cb4;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    const long long pstep = stride / ct4;
    const int cstep = (int)(stride - pstep * ct4);
    long long p = i0 / ct4;
    int c = (int)(i0 - p * ct4);
    for (long long i = i0; i < n4; i += stride) {
        const float4 v = c < ca4 ? a[p * ca4 + c] : b[p * cb4 + (c - ca4)];
        p += pstep; c += cstep;
        if (c >= ct4) { c -= ct4; ++p; }
        y[i] = make_float4(gc_relu(v.x), gc_relu(v.y), gc_relu(v.z), gc_relu(v.w));
    }
}

// The same with the result in NCHW (what the U-Net's up path runs in: MIOpen computes the float32 transposed convolutions as GEMM +
// col2im on NCHW tensors, and torch.cat of a channels_last skip and an NCHW up-convolution is an NCHW tensor): b is NCHW, a is NHWC
// (A_NHWC: the skip, produced by the down path's NHWC convolutions -- transposed through LDS, 64 pixels x 32 channels per workgroup)
// or NCHW.  Grid: (pixel tiles of 64, channel tiles of 32 over a then b, batch).
template <int A_NHWC>
__global__ __launch_bounds__(256) void k_relu_cat_nchw_f32(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ y,
                                                            long long plane, int ca, int cb)
{
    __shared__ float tile[64][33];
    const int t = threadIdx.x, n = blockIdx.z;
    const long long p0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32, ct = ca + cb;
    float *yo = y + ((long long)n * ct + c0) * plane;
    if (A_NHWC && c0 < ca) {
        const int ci = t & 31;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int pi = (t >> 5) + 8 * k;
            if (p0 + pi < plane) tile[pi][ci] = a[((long long)n * plane + p0 + pi) * ca + c0 + ci];
        }
        __syncthreads();
        const int po = t & 63;
        if (p0 + po < plane) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int co = (t >> 6) + 4 * k;
                yo[(long long)co * plane + p0 + po] = gc_relu(tile[po][co]);
            }
        }
        return;
    }
    const float *src = c0 < ca ? a + ((long long)n * ca + c0) * plane : b + ((long long)n * cb + (c0 - ca)) * plane;
    const int po = t & 63;
    if (p0 + po < plane) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int co = (t >> 6) + 4 * k;
            yo[(long long)co * plane + p0 + po] = gc_relu(src[(long long)co * plane + p0 + po]);
        }
    }
}

DS_API int ds_relu_cat_f32(ds_ctx *ctx, const float *a, const float *b, float *y, int batch, int64_t plane, int channels_a, int channels_b,
                           int layout, void *stream)
{
    DS_REQUIRE(ctx && a && b && y, DS_EINVAL, "ds_relu_cat_f32: null argument");
    DS_REQUIRE(batch > 0 && plane > 0 && channels_a > 0 && channels_b > 0, DS_EINVAL, "ds_relu_cat_f32: bad shape");
    DS_REQUIRE(layout >= 0 && layout <= 2, DS_EINVAL, "ds_relu_cat_f32: layout must be 0 (all NHWC), 1 (a NHWC; b, y NCHW) or 2 (all NCHW)");
    DS_REQUIRE(y != a && y != b, DS_EINVAL, "ds_relu_cat_f32: y must not alias an input");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    if (layout == 0) {
        DS_REQUIRE(channels_a % 4 == 0 && channels_b % 4 == 0, DS_EINVAL, "ds_relu_cat_f32: NHWC channel counts must be multiples of 4");
        DS_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && ((uintptr_t)y & 15) == 0, DS_EINVAL, "ds_relu_cat_f32: 16-byte alignment");
        const long long n4 = (long long)batch * plane * ((channels_a + channels_b) / 4);
        long long blocks = (n4 + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(k_relu_cat_f32, dim3((unsigned)blocks), dim3(256), 0, st, (const float4 *)a, (const float4 *)b, (float4 *)y, n4, channels_a / 4,
                           channels_b / 4);
    } else {
        DS_REQUIRE(channels_a % 32 == 0 && channels_b % 32 == 0, DS_EUNSUPPORTED, "ds_relu_cat_f32: NCHW output needs channel counts that are multiples of 32");
        DS_REQUIRE((channels_a + channels_b) / 32 <= 65535 && batch <= 65535 && (plane + 63) / 64 < (1ll << 31), DS_EUNSUPPORTED, "ds_relu_cat_f32: grid too large");
        const dim3 grid((unsigned)((plane + 63) / 64), (unsigned)((channels_a + channels_b) / 32), (unsigned)batch);
        if (layout == 1) hipLaunchKernelGGL((k_relu_cat_nchw_f32<1>), grid, dim3(256), 0, st, a, b, y, (long long)plane, channels_a, channels_b);
        else hipLaunchKernelGGL((k_relu_cat_nchw_f32<0>), grid, dim3(256), 0, st, a, b, y, (long long)plane, channels_a, channels_b);
    }
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}
