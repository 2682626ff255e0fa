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
template <int CPG>
__global__ __launch_bounds__(256) void k_gconv3x3_nhwc_f32(const float *__restrict__ x, const float *__restrict__ wt, const float *__restrict__ bias,
                                                           float *__restrict__ y, int H, int W, int C, int tiles_x, int tiles_y, int relu)
{
    __shared__ __attribute__((aligned(16))) float tile[GC_IH * GC_IW * GC_PS];
    constexpr int COT = CPG < 16 ? CPG : 16;                 // output channels per pass (the accumulators of one thread)
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int b = t / tiles_y;
    const int c0 = blockIdx.y * GC_CB;                       // first channel of this workgroup's block
    const int oy0 = by * GC_TH, ox0 = bx * GC_TW;
    const float *xb = x + (size_t)b * H * W * C + c0;

    // ---- stage the input tile: (pixel, 16-byte piece) pairs, 8 pieces per pixel; out-of-image pixels are the zero padding ----
    for (int i = tid; i < GC_IH * GC_IW * (GC_CB / 4); i += 256) {
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
    for (int g = 0; g < GC_CB / CPG; ++g) {
        const int grp = (c0 + g * CPG) / CPG;                // uniform: the weights below are scalar loads
        const float *wg = wt + (size_t)grp * 9 * CPG * CPG;
#pragma unroll 1
        for (int half = 0; half < CPG / COT; ++half) {
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
    dim3 grid((unsigned)(tiles_x * tiles_y * batch), (unsigned)(channels / GC_CB)), block(256);
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
