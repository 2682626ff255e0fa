/*
 * depthstereo.h -- C ABI of libdepthstereo_hip.so: the MI355X (gfx950) replacement for the per-pixel
 * hot path of thygate/stable-diffusion-webui-depthmap-script v0.4.8.
 *
 * The reference has no FFI layer: the path is plain Python/numba.  Each entry point
 * below names the reference code it replaces (file:line under the reference tree); INTEGRATION.md
 * shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative DS_E* code on failure; nothing throws across
 *     the ABI; ds_last_error() returns a thread-local message for the last failure.
 *   - all image/depth/output pointers are DEVICE pointers (HBM) owned by the caller; the callee
 *     never frees or retains them.  Buffers are batched: n images of h x w pixels, c channels,
 *     pixel-interleaved (HWC), densely packed unless a stride argument says otherwise.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     asynchronous with respect to the host unless stated otherwise.
 *   - a ds_ctx owns scratch memory (row flags, work lists, min/max slots, LUTs).  One ctx per
 *     host thread/stream; calls on the same ctx must be issued on one stream at a time.
 *   - results are bit-identical to the reference (uint8/uint16 outputs); float64 arithmetic is
 *     done in IEEE-754 binary64 without FMA contraction, in the reference's operation order.
 */
#ifndef DEPTHSTEREO_H
#define DEPTHSTEREO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS_VERSION 100

/* error codes */
#define DS_OK            0
#define DS_EINVAL       -1   /* bad argument (null pointer, bad shape, unknown enum) */
#define DS_EUNSUPPORTED -2   /* legal in the reference but outside what the kernels cover */
#define DS_EHIP         -3   /* a HIP runtime call failed (message has the hipError string) */
#define DS_ENOMEM       -4

/* depth element types accepted by the stereo entry points */
#define DS_DEPTH_U16 0       /* what core_generation_funnel passes (core.py:211,253) */
#define DS_DEPTH_F32 1
#define DS_DEPTH_F64 2

/* fill techniques, src/stereoimage_generation.py:85-92 */
#define DS_FILL_NONE                0
#define DS_FILL_NAIVE               1
#define DS_FILL_NAIVE_INTERPOLATING 2
#define DS_FILL_POLYLINES_SOFT      3
#define DS_FILL_POLYLINES_SHARP     4

typedef struct ds_ctx ds_ctx;

/* library / context ----------------------------------------------------------------------------- */
int ds_version(void);
const char *ds_last_error(void);
/* device = HIP device ordinal the context allocates its scratch on */
int ds_ctx_create(ds_ctx **out, int device);
int ds_ctx_destroy(ds_ctx *ctx);

/* One eye of a stereo pair: where it is written and how far pixels move. */
typedef struct ds_eye {
    double divergence_px;    /* (divergence/100)*w, signed; stereoimage_generation.py:82 */
    double separation_px;    /* (separation/100)*w, signed; stereoimage_generation.py:83 */
    uint8_t *out;            /* device pointer of pixel (image 0, row 0, col 0) of this eye */
    int64_t out_row_stride;  /* bytes between rows     (w*c for a lone eye, 2*w*c inside a side-by-side) */
    int64_t out_img_stride;  /* bytes between images */
} ds_eye;

/*
 * ds_stereo_warp -- replaces apply_stereo_divergence (src/stereoimage_generation.py:77-92) and the
 * numba kernels it dispatches to: apply_stereo_divergence_naive (:95-159) and
 * apply_stereo_divergence_polylines (:162-283).  One call renders n_eyes (1 or 2) views of each of
 * the n images; min/max normalisation of the depth (:79-81) is done per image on the device.
 *   image   n*h*w*c uint8, c in 1..4
 *   depth   n*h*w elements of depth_dtype
 *   exponent  stereo_offset_exponent.  1.0 is computed directly; any other value needs a host-built
 *           table (pow must match the host libm): pass pow_lut = device pointer to n*65536 doubles,
 *           pow_lut[i*65536+v] = ((v-min_i)/(max_i-min_i))**exponent, depth_dtype must be U16.
 *           pow_lut == NULL with exponent != 1.0 -> DS_EUNSUPPORTED.
 * Undefined corners of the reference are defined in DESIGN.md ("Defined corners").
 */
int ds_stereo_warp(ds_ctx *ctx, const uint8_t *image, const void *depth, int depth_dtype,
                   int n, int h, int w, int c, double exponent, const double *pow_lut,
                   int fill, const ds_eye *eyes, int n_eyes, void *stream);

/* Per-image depth min/max as doubles {min,max} (device, n*2 doubles) -- the reduction of
 * stereoimage_generation.py:79-80 exposed so the host can build pow_lut.  Asynchronous. */
int ds_depth_minmax(ds_ctx *ctx, const void *depth, int depth_dtype, int n, int h, int w,
                    double *minmax_out, void *stream);

/* Kernel timing for roofline accounting.  When enabled, ds_stereo_warp brackets its dominant kernel (the naive or
 * polylines render kernel) and the exact-fallback kernel with HIP events recorded ON THE CALLER'S STREAM;
 * ds_profile_last_ms synchronises those events and returns the two durations of the most recent call. */
int ds_profile_enable(ds_ctx *ctx, int enable);
int ds_profile_last_ms(ds_ctx *ctx, float *render_ms, float *exact_ms);

/* In-step kernel timers (no counterpart in the reference; bench.py's `roofline` is the average duration of a kernel INSIDE the
 * timed step, measured on the stream it is launched on).  While enabled, the entry points below bracket their kernel launch with
 * a pair of HIP events out of a ring of 1024 pairs per kind (launches beyond the ring, and launches made while the stream is
 * being captured into a graph, are not timed); nothing synchronises until ds_kernel_timer_read, which waits for the last pair of
 * the kind and returns how many launches were timed and the sum of their durations.  Enabling (or disabling) resets every ring.
 * One thread at a time may launch on a context while its timers are on (the slot counters are not atomic).
 * A GEMM kind that hands its last tiles to the ragged round has a second timer, kind + DS_KT_RAGGED, around that launch. */
#define DS_KT_LINEAR_GELU      0   /* ds_linear, act = 1 (fc1 + GELU of an encoder block) */
#define DS_KT_ATTENTION        1   /* ds_attention_fwd */
#define DS_KT_LINEAR_RESIDUAL  2   /* ds_linear_residual (projection / fc2 + LayerScale + residual) */
#define DS_KT_LINEAR           3   /* ds_linear, act = 0 or 2 */
#define DS_KT_LINEAR_VT        4   /* ds_linear_vt */
#define DS_KT_CONV3X3          5   /* ds_conv3x3_nhwc */
#define DS_KT_LINEAR_READOUT   6   /* ds_linear_readout */
#define DS_KT_LINEAR_SHUFFLE   7   /* ds_linear_shuffle */
#define DS_KT_NORMALMAP        8   /* ds_normalmap (uint16, the fused kernel) */
#define DS_KT_RAGGED          12   /* + kind: the ragged round (k_linear_ragged) of that GEMM kind */
int ds_kernel_timer_enable(ds_ctx *ctx, int enable);
int ds_kernel_timer_read(ds_ctx *ctx, int kind, int64_t *launches, double *total_ms);
/* the same, launch by launch in launch order: the first min(*launches, capacity) durations (ms) go to ms_out (bench.py tells the
 * projection launches of ds_linear_residual from the fc2 launches: they alternate) */
int ds_kernel_timer_read_each(ds_ctx *ctx, int kind, float *ms_out, int64_t capacity, int64_t *launches);

/* Number of image rows the last ds_stereo_warp on this ctx re-rendered with the exact sequential
 * sweep (polylines only; see DESIGN.md "exact fallback").  Synchronises the stream. */
int ds_stereo_last_exact_rows(ds_ctx *ctx, int64_t *rows_out, void *stream);
/* Same, plus the number of queue chunks (128 pixels each, partly filled) the polylines kernel handed to its
 * general-pixel kernel: stats_out[0] = exact rows, stats_out[1] = queue chunks.  Synchronises the stream. */
int ds_stereo_last_stats(ds_ctx *ctx, int64_t *stats_out, void *stream);

/*
 * ds_copy_view -- the "eye is the untouched original" branch of create_stereoimages
 * (stereoimage_generation.py:46,49) plus the np.hstack/np.vstack copies (:56-62): strided copy of
 * n images h x (w*c bytes) into a packed output.
 */
int ds_copy_view(ds_ctx *ctx, const uint8_t *src, int64_t src_row_stride, int64_t src_img_stride,
                 uint8_t *dst, int64_t dst_row_stride, int64_t dst_img_stride,
                 int n, int h, int64_t row_bytes, void *stream);

/*
 * ds_overlap_red_cyan -- replaces overlap_red_cyan (src/stereoimage_generation.py:286-307):
 * out[...,0] = im1[...,0]; out[...,1:3] = im2[...,1:3]; out is n*h*w*3 dense.
 * im1/im2 are addressed with (row, image) byte strides so they can live inside a side-by-side.
 */
int ds_overlap_red_cyan(ds_ctx *ctx, const uint8_t *im1, int64_t im1_row_stride, int64_t im1_img_stride,
                        const uint8_t *im2, int64_t im2_row_stride, int64_t im2_img_stride,
                        int n, int h, int w, int c, uint8_t *out, void *stream);

/*
 * ds_normalmap -- replaces create_normalmap (src/normalmap_generation.py:5-56) for uint16 depth.
 *   pre_blur / post_blur: Gaussian kernel size (odd, >0) or 0 to disable (:23-24, :42-48)
 *   sobel_ksize: 3 (default) / 1 / 5 / 7 ... odd, or 0 for np.gradient (:27-31)
 *   out: n*h*w*3 uint8
 */
int ds_normalmap(ds_ctx *ctx, const uint16_t *depth, int n, int h, int w, int pre_blur,
                 int sobel_ksize, int post_blur, int invert, uint8_t *out, void *stream);

/*
 * ds_normalmap_f64 -- the same for float64 depth: create_normalmap accepts any real array
 * (src/normalmap_generation.py:20-21 promote `depthmap * (-1.0) / 256.0` to float64; integer and float32
 * inputs are cast to float64 by the host, which is exact).  Always the separable float64 passes.
 */
int ds_normalmap_f64(ds_ctx *ctx, const double *depth, int n, int h, int w, int pre_blur,
                     int sobel_ksize, int post_blur, int invert, uint8_t *out, void *stream);

/*
 * ds_normalmap_gradient_f32 -- create_normalmap(float32 depth, sobel_gradient=None) without blurs: the one combination the
 * reference evaluates in FLOAT32 from end to end (src/normalmap_generation.py:20-21 keep float32, :31 np.gradient, :34-39
 * np.linalg.norm + three divisions, :51-54 quantisation; cv2.Sobel would be fed np.float64(...) at :28-29, np.gradient is
 * not).  One fused pass, every operation the correctly rounded binary32 one in numpy's order: bit-identical.
 * depth n*h*w float32, h, w >= 2; out n*h*w*3 uint8.
 */
int ds_normalmap_gradient_f32(ds_ctx *ctx, const float *depth, int n, int h, int w, int invert, uint8_t *out, void *stream);

/*
 * ds_normalmap_gradient_f16 -- create_normalmap(float16 depth, sobel_gradient=None) without blurs: numpy keeps float16 from end to
 * end (src/normalmap_generation.py:20-21 do not promote, :31 np.gradient returns the input's inexact dtype, :34-39 and :51-54 are
 * float16 ufunc loops: each operation is computed in float32 and rounded to half; np.linalg.norm's add.reduce sums the squares in
 * float32 and rounds once).  One fused pass with exactly those roundings: bit-identical.  depth n*h*w IEEE binary16 (2 bytes each),
 * h, w >= 2; out n*h*w*3 uint8.
 */
int ds_normalmap_gradient_f16(ds_ctx *ctx, const void *depth, int n, int h, int w, int invert, uint8_t *out, void *stream);

/*
 * ds_normalmap_gradient_blur_f32 -- create_normalmap(float32 depth, sobel_gradient=None, pre_blur and / or post_blur): the reference
 * runs cv2.GaussianBlur on float32 data (src/normalmap_generation.py:23-24 the depth plane, :42-48 the three-channel normal followed
 * by the second normalisation) between numpy float32 steps.  CV_32F coefficients (getGaussianKernel's float rounding), float32
 * accumulation, BORDER_REFLECT_101; OpenCV's summation order is unpinned (cv2 is absent), everything else is numpy's float32
 * arithmetic.  Blur sizes odd, <= 63; 0 / negative = off (both off forwards to ds_normalmap_gradient_f32).
 */
int ds_normalmap_gradient_blur_f32(ds_ctx *ctx, const float *depth, int n, int h, int w, int pre_blur, int post_blur, int invert,
                                   uint8_t *out, void *stream);

/*
 * ds_normalmap_selfcheck -- device self-test of the fused normal-map kernels' arithmetic (tests): their square root and reciprocal
 * are the generic float64 expansions WITHOUT range scaling and special-case fix-ups (dead for n^2 = zx^2 + zy^2 + 1 in [1, 2^22],
 * src/normalmap_generation.py:34-39); the kernel compares them with sqrt() and 1.0 / n on n^2 = K / 2^18, K = k0 + stride * i,
 * i < count, K in [2^18, 2^40), and returns the number of operands on which they differ (must be 0).  Synchronises the stream.
 */
int ds_normalmap_selfcheck(ds_ctx *ctx, unsigned long long k0, unsigned long long stride, unsigned long long count,
                           unsigned long long *mismatches, void *stream);

/*
 * ds_depth_to_u16 -- replaces the depth post-processing of core_generation_funnel
 * (src/core.py:189-206, no-clip branch) followed by convert_to_i16 (src/core.py:44-50) for a batch
 * of float32 predictions: per image min/max, optional negate (models whose raw output is
 * near-is-dark), normalise to [0,1] in float32, then clip(x*65536+1e-4, 0, 65535.9) -> uint16.
 * A flat prediction (max-min <= float64 eps) gives zeros (:204-206).
 *   pred n*h*w float32; out n*h*w uint16; norm_out (optional, may be NULL) n*h*w float32 = the
 *   normalised map before quantisation.
 */
int ds_depth_to_u16(ds_ctx *ctx, const float *pred, int n, int h, int w, int invert,
                    uint16_t *out, float *norm_out, void *stream);

/*
 * ds_colorize_u16 -- the heat map output of core_generation_funnel (src/core.py:271-274), i.e.
 * dzoedepth/utils/misc.py:97-150 colorize(depth, cmap=...) with its default arguments: per image
 *     value = (depth - vmin) / (vmax - vmin)   in float64 (zeros when vmin == vmax, :124-127)
 *     index = trunc(value * lut_n), below 0 -> entry 0, >= lut_n -> the last entry (matplotlib Colormap.__call__)
 *     out   = lut_rgba[index]
 *   depth      n*h*w uint16;  vmin_vmax  n*2 doubles (device): the reference's np.percentile(depth, 2 / 85);
 *   lut_rgba   lut_n*4 bytes (device), the colormap's bytes=True table;  out  n*h*w*4 bytes (RGBA).
 */
int ds_colorize_u16(ds_ctx *ctx, const uint16_t *depth, int n, int h, int w, const double *vmin_vmax,
                    const uint8_t *lut_rgba, int lut_n, uint8_t *out, void *stream);

/* convert_to_i16 alone (src/core.py:44-50): float32 or float64 input already in [0,1]. */
int ds_convert_to_i16(ds_ctx *ctx, const void *arr, int is_f64, int64_t count, uint16_t *out, void *stream);

/* element types of the tensor-core entry points */
#define DS_DTYPE_F16  1
#define DS_DTYPE_BF16 2
#define DS_DTYPE_F32  3       /* ds_preprocess_bicubic only */

/*
 * ds_attention_fwd -- fused attention forward of one transformer block; replaces the q@k^T -> (+bias) -> softmax -> @v
 * sequence of ddepth_anything_v2/depth_anything_v2/dinov2_layers/attention.py:49-62 (MemEffAttention falls back to it,
 * :65-82) and of dmidas/backbones/beit.py:65-91 (attention_forward with relative position bias); head_dim = 64.
 *   qk      [B, Np, 2, H, 64]  Q (index 0) and K (index 1), token major, as the projection GEMM writes them
 *   vt      [B, H*64, Np]      V transposed (key index contiguous)
 *   bias_packed  NULL, or the operand made by ds_attention_bias_pack for this H and Np ROUNDED UP to a multiple of 64, same dtype
 *   out     [B, Np, H*64]
 * Np is the token stride of the operands, a multiple of 8 (the padded token count: a tight pad keeps the token GEMMs around
 * the kernel close to whole rounds of tiles -- 32 x 1032 rows are 129 row panels, 32 x 1088 are 136); the kernel walks whole
 * 64-key tiles, rows in [Np, roundup(Np, 64)) read as pad keys and are not stored.  Keys >= n_valid are masked; query rows >=
 * n_valid are computed like any other row (they stay finite) and are never read as keys.  scale multiplies q.k before the bias
 * is added.
 */
int ds_attention_fwd(ds_ctx *ctx, const void *qk, const void *vt, const void *bias_packed, void *out,
                     int B, int Np, int H, int n_valid, float scale, int dtype, void *stream);

/*
 * ds_attention_bias_pack -- prepares the additive logits bias of ds_attention_fwd (the relative position bias of
 * dmidas/backbones/beit.py:29-62,84-88: bias[h][query][key], one table per block, constant per input resolution).
 *   bias    [H, n, n] float32, natural units (what the reference adds to q.k*scale)
 *   packed  H*Np*Np elements of dtype, OPAQUE to the caller -- only ds_attention_fwd of the same library build reads it.
 *           Layout (informative): the bias enters the logits through the matrix pipe (S^T = K.Q^T + Bias^T.I), so the operand
 *           is stored as the MFMA A fragments of that product, in units of 1/scale (head_dim 64: the values times 8, exact in
 *           f16/bf16 up to the rounding of the value itself), zero padded to Np x Np:
 *           [H][Np/32 query blocks][Np/64 key tiles][4 chunks c = 2*kb + s][64 lanes][8 values t], where lane (hi = lane >> 5,
 *           l = lane & 31) of chunk c holds bias[query 32*qblock + 16*s + 8*hi + t][key 64*ktile + 32*kb + l] / scale.
 *           A wave reads its 32 x 64 bias tile with four coalesced 16-byte loads.  ds_attention_fwd therefore requires
 *           scale == 0.125 when a packed bias is given (DS_EUNSUPPORTED otherwise).
 * Run once per (block, resolution); the packed operand is reused by every forward.
 */
int ds_attention_bias_pack(ds_ctx *ctx, const float *bias, int H, int n, int Np, int dtype, void *packed, void *stream);

/*
 * ds_attention_reload_env -- ds_attention_fwd reads its A/B switches (DS_ATT_GEN = 2 / 4: kernel generation -- generation 4 runs
 * one wave per SIMD with two query sub-blocks skewed inside the wave, generation 2, the default, two or three waves per SIMD;
 * DS_ATT_TAIL = 0 / 1: query blocks with at most four live rows as GEMVs instead of tiles; DS_ATT_NQB, DS_ATT_LATE, DS_ATT_ORDER)
 * once per process; this call reads them again.  Every setting computes the same function (dmidas/backbones/beit.py:65-91);
 * generations 2 and 4 are bit-identical with DS_ATT_TAIL = 0.  Tests and A/B runs only.
 */
int ds_attention_reload_env(void);

/*
 * ds_preprocess_bicubic -- image -> network input in one pass: the chain `cv2.cvtColor(.., COLOR_BGR2RGB) / 255.0`
 * (src/depthmap_generation.py:381) -> Resize(.., INTER_CUBIC) -> NormalizeImage -> PrepareForNet (:457-476, dmidas/transforms.py;
 * ddepth_anything_v2/depth_anything_v2/dpt.py:196-221) that estimatemidas / estimatedepthanything_v2 run on the host.
 *   images  uint8 [batch, in_h, in_w, 3]
 *   out     [batch, 3, out_h, out_w] in channels_last memory (physically [batch, out_h, out_w, 3]) of `dtype` (f16 / bf16 / f32):
 *           out[b][c][y][x] = (bicubic(images[b][:, :, flip_channels ? 2 - c : c] / 255)(y, x) - mean[c]) / std[c]
 * Resampling: the cubic convolution kernel (A = -0.75) on half-pixel centres, replicated border, no antialiasing -- cv2's
 * INTER_CUBIC definition as torch's upsample_bicubic2d evaluates it (float32).  mean / std: 3 host floats each.
 */
int ds_preprocess_bicubic(ds_ctx *ctx, const void *images, void *out, int batch, int in_h, int in_w, int out_h, int out_w,
                          int flip_channels, const float *mean, const float *std, int dtype, void *stream);

/*
 * ds_residual_layernorm -- the element-wise part of a transformer block between two GEMMs, fused:
 *     x_out = x + gamma * branch           (LayerScale + residual; dinov2_layers/block.py:100-107, beit.py:101-106)
 *     h_out = LayerNorm(x_out) * ln_weight + ln_bias
 * x, branch, x_out, h_out: [rows, channels]; gamma, ln_weight, ln_bias: [channels]; all of `dtype` (f16/bf16).
 * branch == NULL: plain LayerNorm of x (x_out unused).  gamma == NULL: gamma = 1.  x_out may alias x.
 * channels in {384, 768, 1024, 1536}.
 */
int ds_residual_layernorm(ds_ctx *ctx, const void *x, const void *branch, const void *gamma, const void *ln_weight,
                          const void *ln_bias, void *x_out, void *h_out, int64_t rows, int channels, float eps, int dtype,
                          void *stream);

/*
 * ds_reassemble_readout -- epilogue of the DPT reassemble read-out (ProjectReadout, dmidas/backbones/utils.py:28-39;
 * forward_adapted_unflatten :83-124): the reference concatenates every token with the cls token, applies Linear(2C -> C)
 * and GELU.  The host splits the linear map (W.[tok ; cls] + b = W_tok.tok + (W_cls.cls + b)) into the token GEMM and one
 * vector per image; this entry point does the rest in one pass:
 *     out[b, n-1, :] = gelu_erf(proj[b, n, :] + clsvec[b, :]),  n = 1 .. tokens-1
 * proj [batch, tokens, channels] (the token GEMM's output, cls row included), clsvec [batch, channels],
 * out [batch, tokens-1, channels] = the NHWC operand of the 1x1 convolution that follows.  f16/bf16, channels % 8 == 0.
 */
int ds_reassemble_readout(ds_ctx *ctx, const void *proj, const void *clsvec, void *out, int batch, int tokens, int channels,
                          int dtype, void *stream);

/*
 * ds_bias_act_nhwc -- element-wise tail of a decoder convolution on a channels-last activation, one pass:
 *     out = [relu]( x + bias[channel] [+ res1] [+ res2] )
 * (ResidualConvUnit_custom, dmidas/blocks.py:352-377; ResidualConvUnit, ddepth_anything_v2/.../util/blocks.py:56-85; the
 * skip add of the fusion blocks).  x, res1, res2, out: `elements` values, channel fastest; res1 / res2 may be NULL; out may
 * alias x.  f16/bf16, channels % 8 == 0.
 */
int ds_bias_act_nhwc(ds_ctx *ctx, const void *x, const void *bias, const void *res1, const void *res2, void *out,
                     int64_t elements, int channels, int relu, int dtype, void *stream);

/*
 * ds_group_norm_nchw -- GroupNorm [+ residual] [+ ReLU] of an NCHW activation in two launches (no counterpart function in the
 * reference: it is timm's GroupNormAct inside the ResNetV2-50 stem of dpt_hybrid_384, which the reference obtains through
 * timm.create_model in dmidas/backbones/vit.py:_make_pretrained_vitb_rn50_384; torch spends three launches + a ReLU + an add on it):
 *     out = [relu]( (x - mean_g) * rstd_g * gamma[c] + beta[c] [+ res] ),   mean / biased variance over the group's channels x pixels
 * x, res, out [n, channels, hw] contiguous (res may be NULL, out may alias x), gamma / beta [channels] in the activation's type;
 * f16 / bf16, hw % 8 == 0, n * groups <= 4096.  Moments in float32 per slice, combined in float64; the affine map in float32.
 */
int ds_group_norm_nchw(ds_ctx *ctx, const void *x, const void *gamma, const void *beta, const void *res, void *out, int n, int channels,
                       int hw, int groups, float eps, int relu, int dtype, void *stream);

/*
 * ds_linear -- y = act(x . W^T + bias), the token GEMMs of the ViT encoders with the epilogue fused (csrc/ds_linear.hip):
 * `fc1 -> nn.GELU` of the encoder MLP (timm Mlp as run by dmidas/backbones/beit.py:93-107; ddepth_anything_v2/
 * depth_anything_v2/dinov2_layers/mlp.py:33-39) is ONE kernel (act = 1: erf-GELU evaluated on the fp32 accumulator, see
 * ln_gelu for the error bound), act = 0 is a plain Linear (fc2, proj, qkv), act = 2 applies ReLU.
 * x [rows, in_features], W [out_features, in_features] (torch.nn.Linear layout), bias [out_features] or NULL,
 * y [rows, ldy] (ldy >= out_features, in elements).  f16/bf16, fp32 accumulation.  out_features % 256 == 0,
 * in_features % 128 == 0, rows >= 256 (the last 256-row panel is shifted up to end at the last row).  y must not alias x.
 * 256 x 256 tiles on the MFMA units, walked by one persistent workgroup per CU; when the last round of that walk would be
 * at most a quarter full, its tiles are rendered as 128 x 64 pieces by a second kernel on the same stream (the "ragged
 * round": 544 tiles on 256 CUs cost 2 rounds + the pieces instead of 3 rounds).  Long contractions (>= 2048) of a small ragged
 * round are shared by up to 8 workgroups per piece through a per-context workspace (8 MB, allocated at the first such launch:
 * one more reason why calls on one ctx go to one stream at a time); results stay bit-reproducible run to run.
 */
int ds_linear(ds_ctx *ctx, const void *x, const void *w, const void *bias, void *y, int64_t rows, int64_t out_features,
              int64_t in_features, int64_t ldy, int act, int dtype, void *stream);

/*
 * ds_linear_residual -- y = res + [gamma *] (x . W^T + bias): ds_linear with the residual add (and, when gamma is not NULL,
 * the LayerScale factor per output feature) in the epilogue: `x + gamma_1 * attn.proj(...)` of the encoder blocks
 * (dmidas/backbones/beit.py:99-103; dinov2_layers/block.py:88-96).  res and y [rows, out_features]; y must not alias res.
 * Shapes as ds_linear.  Used for the attention output projection and for fc2 of every encoder block: the LayerNorm pass that
 * follows then reads one operand instead of two.
 */
int ds_linear_residual(ds_ctx *ctx, const void *x, const void *w, const void *bias, const void *gamma, const void *res, void *y,
                       int64_t rows, int64_t out_features, int64_t in_features, int dtype, void *stream);

/*
 * ds_linear_vt -- V TRANSPOSED straight out of the projection GEMM: vt[b][c][n] = sum_k w_v[c][k] * h[b][n][k].
 * The reference computes qkv = Linear(h), reshapes and permutes (dmidas/backbones/beit.py:71-74; dinov2_layers/
 * attention.py:52-55); ds_attention_fwd wants V with the key index contiguous ([B, H*64, Np]), so the GEMM runs with W_v as
 * its row operand and all B * Np tokens as columns, and the epilogue scatters every group of 8 columns to its batch element.
 *   w_v [channels, in_features]   h [batch * tokens, in_features]   vt [batch, channels, tokens]
 * tokens % 64 == 0, (batch * tokens) % 256 == 0, channels >= 256, in_features % 128 == 0.  No bias: the V bias commutes with
 * the attention (softmax rows sum to one) and is folded into the output projection's bias by the host.
 */
int ds_linear_vt(ds_ctx *ctx, const void *w_v, const void *h, void *vt, int64_t channels, int64_t batch, int64_t tokens,
                 int64_t in_features, int dtype, void *stream);

/*
 * ds_linear_shuffle -- ConvTranspose2d with kernel_size == stride as ONE GEMM with the pixel shuffle in the store address: the
 * 4x4-stride-4 and 2x2-stride-2 transposed convolutions of the reassemble stage (dmidas/backbones/utils.py:196-205 and :215-224,
 * `nn.ConvTranspose2d(features[i], features[i], kernel_size=4 / 2, stride=4 / 2, padding=0)`; ddepth_anything_v2/
 * depth_anything_v2/dpt.py:57-71).  With kernel == stride no two input pixels write the same output pixel:
 *     y[b, yy*s + ky, xx*s + kx, co] = bias[(ky*s + kx)*C + co] + sum_ci x[b, yy, xx, ci] * w[(ky*s + kx)*C + co, ci]
 *   x     [pixels, in_features]   the NHWC input, pixels = batch * h * width (image-major rows)
 *   w     [stride*stride*out_channels, in_features]: torch's [in, out, kH, kW] weight permuted to (kH, kW, out, in) by the host
 *   bias  [stride*stride*out_channels] (the module's bias repeated per tap) or NULL
 *   y     [batch, h*stride, width*stride, out_channels] NHWC
 * stride*stride*out_channels % 256 == 0, out_channels % 8 == 0, in_features % 128 == 0, pixels >= 256.  f16 / bf16, fp32 accumulation.
 */
int ds_linear_shuffle(ds_ctx *ctx, const void *x, const void *w, const void *bias, void *y, int64_t pixels, int64_t in_features,
                      int width, int stride, int out_channels, int dtype, void *stream);

/*
 * ds_linear_readout -- the read-out of the reassemble stage as one GEMM on the padded token sequence: ProjectReadout
 * (dmidas/backbones/utils.py:28-39: `features = cat(x[:, start_index:], readout.expand_as(...)); project(features)` with project =
 * Linear(2C -> C) + GELU) followed by the Transpose(1, 2) / Unflatten of :165-169.  The linear map of a concatenation splits,
 * W.[tok ; cls] + b = W_tok.tok + (W_cls.cls + b), and the second term is one vector per image:
 *     y[b*(tokens-1) + t - 1, :] = GELU( x[b*tokens_padded + t, :] . w_tok^T + cls_vec[b, :] ),   t = 1 .. tokens - 1
 *   x        [images * tokens_padded, in_features]   block output as the encoder leaves it (row 0 of an image = cls token, rows >=
 *            tokens = padding)
 *   w_tok    [out_features, in_features]   the token half of the projection weight;  cls_vec [images, out_features]
 *   y        [images * (tokens - 1) + 1, out_features]: token-major = NHWC of the [h, w] patch grid; the LAST row is a dummy that
 *            receives the cls / pad rows of every image (a constant store count per tile)
 * out_features % 256 == 0, in_features % 128 == 0, images * tokens_padded >= 256.  erf-GELU on the fp32 accumulator as ds_linear.
 */
int ds_linear_readout(ds_ctx *ctx, const void *x, const void *w_tok, const void *cls_vec, void *y, int64_t images,
                      int64_t tokens_padded, int64_t tokens, int64_t out_features, int64_t in_features, int dtype, void *stream);

/*
 * LayerNorm folded into the Linear behind it -- the norm1 -> qkv and norm2 -> fc1 pairs of every encoder block (timm's Block as run
 * by dmidas/backbones/beit.py:94-107: `x + gamma_1 * attn(norm1(x))`, `x + gamma_2 * mlp(norm2(x))`; ddepth_anything_v2/
 * depth_anything_v2/dinov2_layers/block.py:82-107).  With W' = W . diag(ln_weight), colsum[n] = sum_k W'[n][k] (of the rounded
 * weights) and b' = b + W . ln_bias, all prepared once per module by the host,
 *     LN(x) . W^T + b  =  rstd[m] * (x . W'^T)[m][n]  -  mean[m] * rstd[m] * colsum[n]  +  b'[n],
 * so the GEMM reads the residual stream itself and the LayerNorm pass (read x, write the normalised copy: 2 x 67 MB per call at
 * the benchmark's shape) becomes a statistics pass (read x, write 8 bytes per token).
 *   ds_row_stats     stats[m] = {rstd, -mean * rstd} (float32 pairs) of the rows of x [rows, channels]; channels in 384 / 768 /
 *                    1024 / 1536; mean and the centred sum of squares in float32 like ds_residual_layernorm
 *   ds_linear_ln     y = act(rstd (x . w_scaled^T) + (-mean rstd) colsum + bias); shapes and act (0 none, 1 erf-GELU) as ds_linear
 *   ds_linear_vt_ln  ds_linear_vt on the un-normalised x: vt[b][c][n] = rstd[b,n] (w_v_scaled . x[b,n]) + (-mean rstd)[b,n] colsum[c]
 *                    (the constant W_v . ln_bias commutes with the attention like the V bias: the host folds it into the projection bias)
 */
int ds_row_stats(ds_ctx *ctx, const void *x, void *stats, int64_t rows, int channels, float eps, int dtype, void *stream);
int ds_linear_ln(ds_ctx *ctx, const void *x, const void *w_scaled, const float *colsum, const void *bias, const void *stats, void *y,
                 int64_t rows, int64_t out_features, int64_t in_features, int64_t ldy, int act, int dtype, void *stream);
int ds_linear_vt_ln(ds_ctx *ctx, const void *w_v_scaled, const float *colsum, const void *x, const void *stats, void *vt,
                    int64_t channels, int64_t batch, int64_t tokens, int64_t in_features, int dtype, void *stream);

/*
 * ds_linear_reload_env -- the GEMM path (ds_linear, ds_linear_residual, ds_linear_vt, ds_conv3x3_nhwc) reads its A/B switches
 * (DS_LIN_EARLY, DS_LIN_GRID, DS_LIN_RAGGED, DS_LIN_RAGGED_DEN, DS_LIN_RAGGED_RING, DS_LIN_RAGGED_PIPE: none of these changes a
 * result; DS_LIN_RAGGED_KSPLIT, DS_LIN_RAGGED_KSPLIT_MIN, DS_LIN_RAGGED_KSPLIT_KEEP: the K split of the ragged round changes the
 * fp32 SUMMATION ORDER of the rows it touches -- results are equal within fp32 rounding, and bit-reproducible for a fixed
 * setting, but NOT bitwise equal between two settings) from
 * the environment ONCE per process, not per launch; this re-reads them (tests and A/B runs that flip a switch in-process).
 * No counterpart in the reference.
 */
int ds_linear_reload_env(void);

/*
 * ds_conv3x3_nhwc -- y = act(conv3x3(x, W) + bias [+ res1] [+ res2]), stride 1, zero padding 1, as the implicit GEMM of
 * csrc/ds_linear.hip (same 256 x 256 MFMA tiles; a K-tile is 64 channels of one tap, fetched by LDS-DMA from the shifted
 * pixel or from a zero line for the padding ring): the 3x3 convolutions of the DPT decoders with their element-wise tails
 * -- ResidualConvUnit_custom (dmidas/blocks.py:352-377: conv -> +bias -> ReLU, conv -> +bias -> +x [-> +skip]) and
 * scratch.layerN_rn (dmidas/blocks.py:64-80, no bias); ddepth_anything_v2/depth_anything_v2/util/blocks.py:56-85.
 * x [batch, height, width, in_channels] (channels_last), W [out_channels, 3, 3, in_channels] (the channels_last memory of
 * torch's [out, in, 3, 3] weight), bias [out_channels] or NULL, res1 / res2 / y [batch, height, width, out_channels].
 * act: 0 none, 2 ReLU (applied after the adds), 6 = 2 | 4: ReLU on the output AND on x itself -- conv(relu(x)), the first
 * convolution of a residual unit (dmidas/blocks.py:361-363: `out = self.activation(x); out = self.conv1(out)`), the maximum
 * taken on the MFMA fragments inside the K loop instead of in a pass over x (no residual operands, out_channels % 256 == 0;
 * a NaN in x becomes 0 there, as it does in the ReLU of the epilogue).  f16/bf16, fp32 accumulation.  in_channels % 128 == 0,
 * out_channels % 128 == 0 (a multiple of 256 runs 256 x 256 tiles; otherwise 256 x 128 tiles, without residual operands:
 * the head's 256 -> 128 convolution, dmidas/dpt_depth.py:150), batch * height * width >= 256.  y must not alias x, res1 or res2.
 */
int ds_conv3x3_nhwc(ds_ctx *ctx, const void *x, const void *w, const void *bias, const void *res1, const void *res2, void *y,
                    int batch, int height, int width, int in_channels, int out_channels, int act, int dtype, void *stream);

/*
 * ds_upsample_bilinear_nhwc -- torch.nn.functional.interpolate(x, size, mode="bilinear", align_corners) for channels_last
 * activations: the x2 upsamples of the DPT decoders (dmidas/blocks.py:429-431; ddepth_anything_v2/.../util/blocks.py:141-145;
 * the heads' Interpolate, dmidas/dpt_depth.py:151, dpt.py:146).  in [batch, in_h, in_w, channels], out [batch, out_h, out_w,
 * channels]; channels % 8 == 0; f16/bf16; float32 interpolation arithmetic like torch's.
 */
int ds_upsample_bilinear_nhwc(ds_ctx *ctx, const void *in, void *out, int batch, int channels, int in_h, int in_w,
                              int out_h, int out_w, int align_corners, int dtype, void *stream);

/*
 * ds_dpt_head_tail -- the tail of the DPT depth heads in one kernel: bilinear upsample (align_corners=True) ->
 * conv3x3 128->32 (padding 1) -> ReLU -> conv1x1 32->1 -> optional ReLU (dmidas/dpt_depth.py:150-158 output_conv[1:6];
 * ddepth_anything_v2/depth_anything_v2/dpt.py:146-147 with output_conv2 :105-111).
 *   x            [batch, in_h, in_w, 128] f16/bf16, channels_last: the output of the head's first convolution
 *   conv3_wfrag  the 3x3 weights [32, 128, 3, 3] rearranged as MFMA fragments [tap 9][k-slice 8][half 2][co 32][8 ci]
 *                (ci = 16*slice + 8*half + j), same dtype as x
 *   conv3_bias, conv1_weight  float32 [32];  conv1_bias scalar
 *   out          [batch, out_h, out_w] same dtype as x
 */
int ds_dpt_head_tail(ds_ctx *ctx, const void *x, int batch, int in_h, int in_w, int out_h, int out_w,
                     const void *conv3_wfrag, const float *conv3_bias, const float *conv1_weight, float conv1_bias,
                     int relu_out, void *out, int dtype, void *stream);

/*
 * ds_gconv3x3_nhwc_f32 -- the grouped 3x3 convolution of the ResNeXt-101 32x8d bottlenecks of LeReS, Boost's base estimator, in
 * float32 with the folded BatchNorm's bias and the ReLU in the epilogue: `conv2` -> `bn2` -> `relu` of lib/Resnext_torch.py:104-110
 * (groups = 32, stride 1, padding 1; the strided first block of a stage stays a library call).  Direct convolution on the float32
 * vector pipe, input tile staged once in LDS, weights through the scalar path (csrc/ds_gconv.hip).
 *   x, y    [batch, height, width, channels] float32, channels_last;  channels % 32 == 0
 *   w_gtio  [channels / cpg groups][9 taps (ky * 3 + kx)][cpg in][cpg out] float32: torch's [out, in / groups, 3, 3] weight
 *           (BatchNorm folded in) regrouped by the host;  bias [channels] or NULL
 *   channels_per_group  8, 16 or 32
 * ds_add_relu_f32 -- y = relu(a + b) over `count` float32 values (count % 4 == 0): `self.relu(out + identity)`, the tail of every
 * bottleneck (lib/Resnext_torch.py:115-118), one pass instead of torch's add + clamp.
 */
int ds_gconv3x3_nhwc_f32(ds_ctx *ctx, const float *x, const float *w_gtio, const float *bias, float *y, int batch, int height, int width,
                         int channels, int channels_per_group, int relu, void *stream);
int ds_add_relu_f32(ds_ctx *ctx, const float *a, const float *b, float *y, int64_t count, void *stream);

/*
 * ds_bias_act_f32 -- y = [relu]((x + bias[c]) [+ res]) on float32 rows of `channels` values (channels_last activations; y may be x):
 * the element-wise tail of a LIBRARY convolution of LeReS in the reference's own order -- the BatchNorm that follows the
 * convolution (folded into weight and bias here; torch adds a convolution's bias in a separate pass on ROCm), the ReLU
 * (lib/Resnext_torch.py:100-102,104-110), for conv3 the shortcut add and its ReLU (:112-118), and the decoder's conv -> bias ->
 * ReLU / conv -> bias -> + x -> ReLU pairs (lib/network_auxi.py:116-121).  channels % 4 == 0; res NULL or laid out like x.
 * ds_relu_cat_f32 -- y[p] = relu(cat(a[p], b[p])) on float32 rows: the skip concatenation of the pix2pix U-Net's up path together
 * with the ReLU the parent level applies to it (pix2pix/models/networks.py:545-550 and :519); the concatenated tensor has no other
 * reader.  layout 0: a, b, y channels_last ([batch, plane, channels] rows; channel counts % 4 == 0).  layout 1: a channels_last, b and y
 * NCHW -- what the up path runs in (MIOpen computes the float32 transposed convolutions on NCHW tensors, the down path's skips are
 * channels_last; torch.cat of the two is NCHW); layout 2: all NCHW.  Layouts 1 / 2: channel counts % 32 == 0.  y must not alias an input.
 */
int ds_bias_act_f32(ds_ctx *ctx, const float *x, const float *bias, const float *res, float *y, int64_t pixels, int channels, int relu,
                    void *stream);
int ds_relu_cat_f32(ds_ctx *ctx, const float *a, const float *b, float *y, int batch, int64_t plane, int channels_a, int channels_b, int layout,
                    void *stream);

/*
 * ds_boost_blend -- the patch-merge step of Boost, all patches in one launch; replaces, per patch, np.polyval (:916),
 * cv2.resize INTER_CUBIC of the merged patch (:918), cv2.resize INTER_LINEAR of the Gaussian mask (:930) and the blend
 * `dst[rect] = dst[rect]*(1-mask) + merged*mask` (:936) of src/depthmap_generation.py:estimateboost.
 *   dst            float32 [height, width] running estimate, updated in place (row stride in ELEMENTS)
 *   patches        n_patches records {int32 x0, y0, w, h; float64 p0, p1} (32 bytes each, device memory), in BLEND ORDER
 *   preds          float32 [n_patches, pred_size, pred_size]: (fake_B + 1)/2 of the merge network per patch
 *   mask_template  float32 [mask_size, mask_size]: generatemask((3000, 3000)) (:944-953)
 */
int ds_boost_blend(ds_ctx *ctx, float *dst, int64_t dst_row_stride, int height, int width, const void *patches,
                   int n_patches, const float *preds, int pred_size, const float *mask_template, int mask_size,
                   void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DEPTHSTEREO_H */
