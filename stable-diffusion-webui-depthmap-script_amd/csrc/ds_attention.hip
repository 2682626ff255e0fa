// ds_attention_fwd: fused (flash-style) attention forward for the ViT encoders of the depth models, on the gfx950
// matrix cores.  Replaces, per transformer block, the reference's
//     attn = (q * scale) @ k^T [+ relative_position_bias];  attn = softmax(attn);  x = attn @ v
// (ddepth_anything_v2/depth_anything_v2/dinov2_layers/attention.py:49-62, dmidas/backbones/beit.py:65-91), which
// materialises a B x H x N x N tensor in HBM, by one kernel that never leaves the chip between Q.K^T and P.V.
//
// Operands (head_dim is 64 for every encoder the reference ships):
//   qk   [B, Np, 2, H, 64]  f16/bf16   Q and K exactly as the projection GEMM writes them (token major)
//   vt   [B, H*64, Np]      f16/bf16   V TRANSPOSED (key index contiguous), produced in that layout by its own GEMM
//   bias optional: the operand made by ds_attention_bias_pack (this file) from the [H, n, n] table -- x log2(e), zero
//        padded, stored in the register order of the logits tile, so a wave loads its 32 x 64 tile straight into registers
//   out  [B, Np, H*64]      f16/bf16
// Np is a multiple of 64; keys >= n_valid are masked (pad rows of the padded token sequence).
//
// Mapping: workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows.  Per 64-key tile:
//   S^T = K . Q^T   "swapped" so that a lane holds logits of ONE query (column lane&31) for 32 of the 64 keys: the
//                   row max / row sum of the online softmax are in-lane reductions plus one exchange with lane^32;
//                   2 key blocks x 4 d-slices of v_mfma_f32_32x32x16 (A = K rows from LDS, B = Q rows in registers)
//   O^T += V^T . P^T  A = V^T rows (two ds_read_b64 of 4 consecutive keys each), B = P^T built in registers straight
//                   from the S^T accumulators: the accumulator's row order fixes which keys sit in which k-slot, and
//                   the V^T reads use the same order, so no cross-lane traffic is needed between the two GEMMs.
// K tiles sit in LDS XOR-swizzled by 16-byte chunk (conflict-free ds_read_b128 of 32 rows x 128 B); V^T rows are padded
// to 136 B (conflict-free ds_read_b64).  The next tile is fetched into registers (buffer loads through descriptors) while
// the current one is computed and stashed into the other of two LDS buffers: one barrier per tile.  Workgroups are
// ordered so that one XCD's L2 serves all query blocks of a (batch, head) and one head's bias (see the kernel).
#include "ds_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define AT_THREADS 256
#define AT_QW 32                 // query rows per wave
#define AT_QB (AT_QW * 4)        // query rows per workgroup
#define AT_KB 64                 // keys per tile
#define AT_D 64
#define AT_VROW 136              // bytes per V^T row in LDS (128 + 8 pad)

template <int BF16> struct at_traits;
template <> struct at_traits<0> {
    typedef _Float16 T; typedef f16x8 V8;
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ T from_f32(float x) { return (_Float16)x; }
    static __device__ __forceinline__ float to_f32(T x) { return (float)x; }
};
template <> struct at_traits<1> {
    typedef __bf16 T; typedef bf16x8 V8;
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ T from_f32(float x) { return (__bf16)x; }
    static __device__ __forceinline__ float to_f32(T x) { return (float)x; }
};

// row of the 32x32 accumulator held in register r of a lane with hi = lane >> 5 (cdna_hip_programming.md, 3. MFMA)
__device__ __forceinline__ int at_crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

struct AttnParams {
    const void *qk, *vt, *bias;
    void *out;
    int B, Np, H, n_valid;
    int flags;                   // bit 0: raise the wave priority around the MFMA clusters (experiment switch)
    int nq, total, chunk;        // query blocks per (b,h); B*H*nq; ceil(total / 8) (XCD-aware work order, see the kernel)
    float k_logit;               // scale*log2(e): with a bias, the factor of the raw accumulator in  x = s*k_logit + bias
    float c_exp;                 // factor inside the exponent, p = exp2((x - max x)*c_exp): scale*log2(e) without a bias,
                                 // 1 with one (the packed bias is in log2 units)
};

template <int BF16, int HAS_BIAS>
__global__ __launch_bounds__(AT_THREADS, 3) void k_attention_fwd(AttnParams P)
{
    typedef at_traits<BF16> TR;
    typedef typename TR::T T;
    typedef typename TR::V8 V8;
    // two tile buffers: tile t+1 is stashed while tile t is still being read -> ONE barrier per tile
    __shared__ __attribute__((aligned(16))) unsigned char s_kbuf[2][AT_KB * 128];    // [key][64 d], chunk-swizzled
    __shared__ __attribute__((aligned(16))) unsigned char s_vbuf[2][AT_D * AT_VROW]; // [d][64 keys], padded rows

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    // Work order.  Workgroups are dealt round-robin to the 8 XCDs (id & 7), each with its own 4 MB L2.  Work items are
    // numbered head-major, L = (h*B + b)*nq + qblock, and XCD j takes the contiguous range [j*chunk, (j+1)*chunk) in order:
    // the query blocks of one (b, h) -- which all stream the same K / V^T -- run on ONE L2 at the same time, and a head's
    // bias table (2.4 MB at 1088 tokens) is read by one XCD only.  With chunk == 0 the launch falls back to the plain order.
    int L = blockIdx.x;
    if (P.chunk > 0) {
        L = (int)(blockIdx.x & 7) * P.chunk + (int)(blockIdx.x >> 3);
        if (L >= P.total) return;                              // grid = 8*chunk >= total: the tail of the last XCD's range
    }
    int qblk, b, h;
    if (P.flags & 2) {                                          // batch fastest: concurrent workgroups share the bias tiles
        b = L % P.B; qblk = (L / P.B) % P.nq; h = L / (P.B * P.nq);
    } else {                                                    // query block fastest: they share K / V^T
        qblk = L % P.nq; b = (L / P.nq) % P.B; h = L / (P.nq * P.B);
    }
    const int q0 = qblk * AT_QB + wave * AT_QW;
    const int Np = P.Np, H = P.H;
    const size_t tok_stride = (size_t)2 * H * AT_D;                                   // elements between tokens in qk
    const T *qk = (const T *)P.qk + (size_t)b * Np * tok_stride;
    const T *q_base = qk + (size_t)h * AT_D;                                          // s = 0
    const T *k_base = qk + (size_t)(H + h) * AT_D;                                    // s = 1
    const T *vt = (const T *)P.vt + ((size_t)b * H + h) * AT_D * (size_t)Np;
    const bool wave_live = q0 < Np;                                                   // whole wave beyond the padded sequence?

    // Q fragments: B operand of S^T = K.Q^T: lane holds Q[q0 + l31][16 s + 8 hi .. +7]
    V8 qf[4];
    {
        const int qrow = min(q0 + l31, Np - 1);
        const T *qp = q_base + (size_t)qrow * tok_stride + 8 * hi;
#pragma unroll
        for (int s = 0; s < 4; s++) qf[s] = *reinterpret_cast<const V8 *>(qp + 16 * s);
    }

    f32x16 o_acc[2];
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) o_acc[d][r] = 0.f;
    float m_run = -__builtin_inff(), l_run = 0.f;

    // staging assignment: K tile = 64 rows x 8 chunks of 16 B; V^T tile = 64 rows x 8 chunks: 512 chunks each, 2 per thread
    const int st_row = tid >> 3, st_chunk = tid & 7;                                  // rows st_row and st_row + 32
    // Tile fetches go through buffer descriptors: the per-lane part of every address is ONE loop-invariant 32-bit byte
    // offset per operand, the tile part a scalar offset -- no 64-bit address arithmetic in the loop.
    u32x4 kreg0, kreg1, vreg0, vreg1;
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
        (void *)k_base, 0, (int)(((size_t)Np * tok_stride - (size_t)(H + h) * AT_D) * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void *)vt, 0, (int)((size_t)AT_D * Np * sizeof(T)), 0x00020000);
    const int vo_k = (int)((st_row * tok_stride + 8 * st_chunk) * sizeof(T));
    const int vo_v = (int)((st_row * Np + 8 * st_chunk) * sizeof(T));
    const int so_k32 = (int)(32 * tok_stride * sizeof(T)), so_r32 = (int)(32 * Np * sizeof(T));
#define AT_FETCH(kt_) do {                                                                                             \
        const int key0_ = (kt_) * AT_KB;                                                                                \
        const int sk_ = __builtin_amdgcn_readfirstlane(key0_ * (int)(tok_stride * sizeof(T)));  /* provably scalar */    \
        const int sv_ = __builtin_amdgcn_readfirstlane(key0_ * (int)sizeof(T));                                         \
        kreg0 = __builtin_amdgcn_raw_buffer_load_b128(rs_k, vo_k, sk_, 0);                                              \
        kreg1 = __builtin_amdgcn_raw_buffer_load_b128(rs_k, vo_k, sk_ + so_k32, 0);                                     \
        vreg0 = __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo_v, sv_, 0);                                              \
        vreg1 = __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo_v, sv_ + so_r32, 0);                                     \
    } while (0)
    // Bias: the packed operand (ds_attention_bias_pack) is laid out in the register order of the S^T accumulators,
    // [head][32-query block][64-key tile][4 chunks][64 lanes][8 values]: a wave's 32 x 64 tile is four fully coalesced
    // 16-byte loads per lane straight into registers -- no LDS round trip, nothing shared between waves.
    u32x4 breg[HAS_BIAS ? 4 : 1];
    const int n_kt = Np / AT_KB;
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(HAS_BIAS ? (const T *)P.bias + (size_t)h * Np * (size_t)Np : (const T *)P.qk), 0,
        (int)((size_t)Np * Np * sizeof(T)), 0x00020000);
    const int vo_b = (int)((((size_t)(q0 / AT_QW) * n_kt) * 2048 + (size_t)lane * 8) * sizeof(T));
#define AT_FETCH_BIAS(kt_) do {                                                                                        \
        const int sb_ = __builtin_amdgcn_readfirstlane((kt_) * (int)(2048 * sizeof(T)));                                \
        _Pragma("unroll") for (int c_i = 0; c_i < 4; c_i++)                                                             \
            breg[c_i] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, vo_b + c_i * 1024, sb_, 0);                         \
    } while (0)
#define AT_STASH1(buf_, row_, kr_, vr_) do {                                                                            \
        *reinterpret_cast<u32x4 *>(s_kbuf[buf_] + (row_) * 128 + ((st_chunk ^ ((row_) & 7)) << 4)) = kr_;               \
        uint2 *vd_ = reinterpret_cast<uint2 *>(s_vbuf[buf_] + (row_) * AT_VROW + 16 * st_chunk);                        \
        vd_[0] = make_uint2(vr_.x, vr_.y);                                                                              \
        vd_[1] = make_uint2(vr_.z, vr_.w);                                                                              \
    } while (0)

    const int ntiles = (P.n_valid + AT_KB - 1) / AT_KB;
    AT_FETCH(0);
    if (HAS_BIAS && wave_live) AT_FETCH_BIAS(0);
    AT_STASH1(0, st_row, kreg0, vreg0);
    AT_STASH1(0, st_row + 32, kreg1, vreg1);
    __syncthreads();
    for (int kt = 0; kt < ntiles; kt++) {
        const int cur = kt & 1;
        const unsigned char *s_k = s_kbuf[cur], *s_v = s_vbuf[cur];
        const bool more = kt + 1 < ntiles;
        if (more) AT_FETCH(kt + 1);                         // in flight while this tile is computed
        f32x16 s_acc[2];
        if (wave_live) {
        // ---- S^T = K . Q^T for the 64 keys of the tile: 2 key blocks of 32 -----------------------------------
        if (P.flags & 1) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
#pragma unroll
            for (int r = 0; r < 16; r++) s_acc[kb][r] = 0.f;
            const int row = kb * 32 + l31;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const V8 kf = *reinterpret_cast<const V8 *>(s_k + row * 128 + (((2 * s + hi) ^ (row & 7)) << 4));
                s_acc[kb] = TR::mfma(kf, qf[s], s_acc[kb]);
            }
        }
        if (P.flags & 1) __builtin_amdgcn_s_setprio(0);
        // ---- logits, bias, key mask, running max (exp2 domain) ---------------------------------------------------------
        // x = the value the max runs over; p = exp2(x*c - m*c) with m = max(x) (c > 0 commutes with max).
        //   no bias:  x = raw accumulator,                 c = scale*log2(e)
        //   bias:     x = s*(scale*log2e/bmul) + bias,      c = bmul  (1 when the bias is stored in log2 units, else log2 e)
        // so a logit costs one FMA for the bias (the f16 operand converted in the same instruction where the ISA has
        // v_fma_mix), half a packed FMA for the exponent's argument, one exp2, half a packed add, half a packed convert.
        const int key0 = kt * AT_KB;
        const float c_ = P.c_exp;
        if (HAS_BIAS) {
#pragma unroll
            for (int c = 0; c < 4; c++) {               // chunk c = accumulator registers 8(c&1) .. +7 of key block c>>1
                T b8[8];
                __builtin_memcpy(b8, &breg[c], 16);
#pragma unroll
                for (int t = 0; t < 8; t++)
                    s_acc[c >> 1][8 * (c & 1) + t] = __builtin_fmaf(s_acc[c >> 1][8 * (c & 1) + t], P.k_logit, TR::to_f32(b8[t]));
            }
            if (more) AT_FETCH_BIAS(kt + 1);                // the registers are free again: next tile's bias lands under the
        }                                                   // softmax / P.V of this one and the S^T of the next
        if (key0 + AT_KB > P.n_valid) {                     // wave-uniform: only the last tile can hold pad keys
            asm volatile("; pad-key mask (kept out of the steady-state tiles: not a candidate for if-conversion)" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (key0 + kb * 32 + at_crow(r, hi) >= P.n_valid) s_acc[kb][r] = -__builtin_inff();
        }
        float m_loc = -__builtin_inff();
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int r = 0; r < 16; r++) m_loc = fmaxf(m_loc, s_acc[kb][r]);
        m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 32, 64));    // the other half of this query's keys
        const float m_new = fmaxf(m_run, m_loc);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_);   // first tile: exp2(-inf) = 0
        m_run = m_new;
        const f32x2 c2 = {c_, c_};
        const f32x2 mc2 = {-m_new * c_, -m_new * c_};
        f32x2 l2 = {0.f, 0.f};
        // ---- P = exp2(x*c - m*c); P^T fragments for the four 16-key slices: registers 8j..8j+7 of key block kb -------
        V8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
#pragma unroll
                for (int t = 0; t < 8; t += 2) {
                    const f32x2 x2 = {s_acc[kb][8 * j + t], s_acc[kb][8 * j + t + 1]};
                    const f32x2 a2 = __builtin_elementwise_fma(x2, c2, mc2);
                    const f32x2 p2 = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
                    l2 += p2;
                    pf[kb][j][t] = TR::from_f32(p2[0]);
                    pf[kb][j][t + 1] = TR::from_f32(p2[1]);
                }
            }
        }
        const float l_loc = l2[0] + l2[1];
        l_run = l_run * alpha + l_loc;
        if (!__all(alpha == 1.0f)) {                         // the running max moved for some query of this wave
#pragma unroll
            for (int d = 0; d < 2; d++)
#pragma unroll
                for (int r = 0; r < 16; r++) o_acc[d][r] *= alpha;
        }
        // ---- O^T += V^T . P^T : A = V^T[d][key slots], slot t of half hi = key 16j + (t&3) + 8(t>>2) + 4hi -------
        if (P.flags & 1) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const unsigned char *vrow = s_v + (d * 32 + l31) * AT_VROW;
#pragma unroll
            for (int kb = 0; kb < 2; kb++) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int kofs = kb * 32 + 16 * j + 4 * hi;
                    const uint2 lo = *reinterpret_cast<const uint2 *>(vrow + 2 * kofs);
                    const uint2 hi8 = *reinterpret_cast<const uint2 *>(vrow + 2 * (kofs + 8));
                    union { uint4 u; V8 v; } cvt;
                    cvt.u = make_uint4(lo.x, lo.y, hi8.x, hi8.y);
                    o_acc[d] = TR::mfma(cvt.v, pf[kb][j], o_acc[d]);
                }
            }
        }
        if (P.flags & 1) __builtin_amdgcn_s_setprio(0);
        }   // wave_live
        if (more) {
            AT_STASH1(cur ^ 1, st_row, kreg0, vreg0);
            AT_STASH1(cur ^ 1, st_row + 32, kreg1, vreg1);
        }
        __syncthreads();                                    // tile kt is read, tile kt+1 is in place
    }
    if (!wave_live) return;

    // ---- epilogue: O / l, lane holds O[q0 + l31][32 d + crow(r, hi)] -------------------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + l31;
    if (qrow < Np) {
        T *op = (T *)P.out + ((size_t)b * Np + qrow) * (size_t)(H * AT_D) + (size_t)h * AT_D;
#pragma unroll
        for (int d = 0; d < 2; d++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                T v4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) v4[t] = TR::from_f32(o_acc[d][4 * g + t] * inv);
                *reinterpret_cast<uint2 *>(op + d * 32 + 8 * g + 4 * hi) = *reinterpret_cast<const uint2 *>(v4);
            }
        }
    }
}

// ---- bias operand: [H, n, n] float32 (natural units) -> packed register order, log2 units, zero padded to Np ----------
template <int BF16>
__global__ void k_attention_bias_pack(const float *__restrict__ bias, typename at_traits<BF16>::T *__restrict__ out,
                                      int H, int n, int Np, float mul)
{
    typedef at_traits<BF16> TR;
    const long long total = (long long)H * Np * Np;
    const int n_kt = Np / AT_KB, nq32 = Np / AT_QW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63), c = (int)((idx >> 9) & 3);
        const long long tile = idx >> 11;
        const int kt = (int)(tile % n_kt), qb = (int)((tile / n_kt) % nq32), h = (int)(tile / ((long long)n_kt * nq32));
        const int q = qb * AT_QW + (lane & 31);
        const int k = kt * AT_KB + (c >> 1) * 32 + at_crow(8 * (c & 1) + j, lane >> 5);
        const float v = (q < n && k < n) ? bias[((size_t)h * n + q) * n + k] * mul : 0.f;
        out[idx] = TR::from_f32(v);
    }
}

DS_API int ds_attention_bias_pack(ds_ctx *ctx, const float *bias, int H, int n, int Np, int dtype, void *packed, void *stream)
{
    DS_REQUIRE(ctx && bias && packed, DS_EINVAL, "ds_attention_bias_pack: null argument");
    DS_REQUIRE(H > 0 && n > 0 && Np >= n && (Np % 64) == 0, DS_EINVAL, "ds_attention_bias_pack: need 0 < n <= Np, Np a multiple of 64 (n %d, Np %d)", n, Np);
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_attention_bias_pack: dtype must be f16 or bf16");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    const long long total = (long long)H * Np * Np;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 65536);
    hipStream_t st = (hipStream_t)stream;
    const float log2e = 1.4426950408889634f;
    if (dtype == DS_DTYPE_F16) hipLaunchKernelGGL((k_attention_bias_pack<0>), dim3(blocks), dim3(256), 0, st, bias, (_Float16 *)packed, H, n, Np, log2e);
    else hipLaunchKernelGGL((k_attention_bias_pack<1>), dim3(blocks), dim3(256), 0, st, bias, (__bf16 *)packed, H, n, Np, log2e);
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

DS_API int ds_attention_fwd(ds_ctx *ctx, const void *qk, const void *vt, const void *bias_packed, void *out,
                            int B, int Np, int H, int n_valid, float scale, int dtype, void *stream)
{
    DS_REQUIRE(ctx && qk && vt && out, DS_EINVAL, "ds_attention_fwd: null argument");
    const void *bias = bias_packed;
    DS_REQUIRE(B > 0 && H > 0 && Np > 0 && (Np % 64) == 0, DS_EINVAL, "ds_attention_fwd: Np must be a positive multiple of 64 (got %d)", Np);
    DS_REQUIRE(n_valid > 0 && n_valid <= Np, DS_EINVAL, "ds_attention_fwd: n_valid %d outside 1..%d", n_valid, Np);
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_attention_fwd: dtype must be f16 or bf16");
    DS_REQUIRE((long long)B * H * ((Np + AT_QB - 1) / AT_QB) < (1ll << 30), DS_EUNSUPPORTED, "ds_attention_fwd: batch x heads too large for the grid");
    DS_REQUIRE(((uintptr_t)qk & 15) == 0 && ((uintptr_t)vt & 15) == 0 && ((uintptr_t)out & 7) == 0 && ((uintptr_t)bias & 15) == 0, DS_EINVAL,
               "ds_attention_fwd: operands must be 16-byte aligned");
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    AttnParams P;
    P.qk = qk; P.vt = vt; P.bias = bias; P.out = out;
    P.B = B; P.Np = Np; P.H = H; P.n_valid = n_valid;
    const float log2e = 1.4426950408889634f;
    P.c_exp = bias ? 1.0f : scale * log2e;                  // the packed bias is in log2 units
    P.k_logit = scale * log2e;
    static const int att_flags = getenv("DS_ATT_FLAGS") ? atoi(getenv("DS_ATT_FLAGS")) : 1;   // setprio around the MFMA clusters: +2.5 % with bias
    P.flags = att_flags;
    P.nq = (Np + AT_QB - 1) / AT_QB;
    P.total = P.nq * H * B;
    static const int plain_order = getenv("DS_ATT_PLAIN_ORDER") ? atoi(getenv("DS_ATT_PLAIN_ORDER")) : 0;   // A/B switch
    P.chunk = plain_order ? 0 : (P.total + 7) / 8;
    dim3 grid(plain_order ? P.total : 8 * P.chunk);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DS_DTYPE_F16) {
        if (bias) hipLaunchKernelGGL((k_attention_fwd<0, 1>), grid, dim3(AT_THREADS), 0, st, P);
        else hipLaunchKernelGGL((k_attention_fwd<0, 0>), grid, dim3(AT_THREADS), 0, st, P);
    } else {
        if (bias) hipLaunchKernelGGL((k_attention_fwd<1, 1>), grid, dim3(AT_THREADS), 0, st, P);
        else hipLaunchKernelGGL((k_attention_fwd<1, 0>), grid, dim3(AT_THREADS), 0, st, P);
    }
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}
