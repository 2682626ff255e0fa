// Shared definitions of the fused attention kernels (ds_attention.hip: generation 2 and the C ABI; ds_attention4.hip: generation 4).
#pragma once
#include <type_traits>

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
    // acc + a[i] + a[i+1]: v_dot2_f32_f16 against (1, 1)
    static __device__ __forceinline__ float add2(V8 a, int i, float acc) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 v = { a[i], a[i + 1] }, one = { (_Float16)1.0f, (_Float16)1.0f };
        return __builtin_amdgcn_fdot2(v, one, acc, false);
    }
};
template <> struct at_traits<1> {
    typedef __bf16 T; typedef bf16x8 V8;
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ T from_f32(float x) { return (__bf16)x; }
    static __device__ __forceinline__ float to_f32(T x) { return (float)x; }
    static __device__ __forceinline__ float add2(V8 a, int i, float acc) { return acc + ((float)a[i] + (float)a[i + 1]); }
};

// row of the 32x32 accumulator held in register r of a lane with hi = lane >> 5 (cdna_hip_programming.md, 3. MFMA)
__device__ __forceinline__ int at_crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

struct AttnParams {
    const void *qk, *vt, *bias;
    void *out;
    int B, Np, H, n_valid;
    int flags;                   // bit 0: wave priority around the MFMA clusters (generation 1), 2: batch-fastest work order, 4: tail blocks as GEMVs
    int nq, total, chunk;        // query blocks per (b,h); B*H*nq; ceil(total / 8) (XCD-aware work order, see the kernel)
    float k_logit;               // scale*log2(e): with a bias, the factor of the raw accumulator in  x = s*k_logit + bias
    float c_exp;                 // factor inside the exponent, p = exp2((x - max x)*c_exp): scale*log2(e) without a bias,
                                 // 1 with one (the packed bias is in log2 units)
};


// three-input maximum: IEEE-754 maximum (v_maximum3_f32 on gfx950) needs none of the canonicalising v_max instructions the
// compiler puts in front of fmaxf() on MFMA results, and stays an ordinary VALU instruction for the scheduler
__device__ __forceinline__ float at_max3(float a, float b, float c)
{
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}

#define AT2_ROW 144                     // bytes per LDS row: 64 halves + 16 pad
#define AT2_TILE (64 * AT2_ROW)         // one K or V^T tile
#define AT2_QW 64                       // query rows per wave
#define AT2_QB 256                      // query rows per workgroup
#define AT2_THR 6.0f                    // deferred-max threshold, log2 units


// ---- tail blocks (both kernel generations) -----------------------------------------------------------------------------------
// A query block with at most AT_TAIL_ROWS live rows (BEiT / ViT: 1 + 32 x 32 tokens leave ONE row in the last block of every
// (batch, head); the tiled path walks the whole key sequence for it -- at one workgroup per CU, generation 4, that is a whole
// round of the launch: 10 rounds instead of 8 at the metric's shape) is not worth the matrix pipe.  Its rows are computed one at
// a time as what they are -- a GEMV against K, a softmax over one row, a GEMV against V^T -- by the 256 threads of the
// workgroup: logits in the exp2 domain into LDS (four threads per key, 16 d each), block maximum, probabilities (rounded to the
// operand type, like P of the tiled path) and their sum, then O[d] = sum_k p[k] V^T[d][k] (four threads per d).  Loads are
// issued eight steps at a time (the loop is latency bound otherwise: ~1 us per dependent L2 round trip).  float32 accumulation.
// smem: >= 4 * Np64 + 32 bytes.
#define AT_TAIL_ROWS 4
#define AT_TAIL_MAXN 8192                   // logits of one row in LDS: 4 bytes x Np64 <= 32 KB

template <int BF16, int HAS_BIAS>
__device__ __forceinline__ void at_tail_rows(const AttnParams &P, unsigned char *smem, int b, int h, int row0, int nrows)
{
    typedef at_traits<BF16> TR;
    typedef typename TR::T T;
    typedef typename TR::V8 V8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Np = P.Np, H = P.H, n_valid = P.n_valid;
    const int Np64 = (Np + 63) & ~63, n_kt = Np64 / AT_KB;
    const size_t tok_stride = (size_t)2 * H * AT_D;
    const T *qk = (const T *)P.qk + (size_t)b * Np * tok_stride;
    const T *k_base = qk + (size_t)(H + h) * AT_D;
    const T *vt = (const T *)P.vt + ((size_t)b * H + h) * AT_D * (size_t)Np;
    T *out_base = (T *)P.out + (size_t)b * Np * (size_t)(H * AT_D) + (size_t)h * AT_D;
    const T *bias = HAS_BIAS ? (const T *)P.bias + (size_t)h * Np64 * (size_t)Np64 : nullptr;
    float *sbuf = reinterpret_cast<float *>(smem);                          // [Np64] logits, then probabilities
    float *red = sbuf + Np64;                                               // [8] cross-wave reductions
    const float c2 = P.c_exp;                                               // scale log2(e)
    const int part = tid & 3;
    for (int r = 0; r < nrows; r++) {
        const int q = row0 + r;
        // ---- logits: thread (key = tid >> 2, part) of every 64-key pass ----
        float qv[16];
        {
            const T *qp = qk + (size_t)q * tok_stride + (size_t)h * AT_D + 16 * part;
            const V8 q0 = *reinterpret_cast<const V8 *>(qp), q1 = *reinterpret_cast<const V8 *>(qp + 8);
#pragma unroll
            for (int t = 0; t < 8; t++) { qv[t] = TR::to_f32(q0[t]); qv[8 + t] = TR::to_f32(q1[t]); }
        }
        // packed bias of (q, key): [32-query block][64-key tile][chunk 2 kb + s][lane (hi, key & 31)][t], see ds_attention_bias_pack
        const int rq = q & 31, kk = tid >> 2;
        const size_t bq = (size_t)(q >> 5) * n_kt * 2048 + (size_t)(rq >> 4) * 512 + (size_t)((rq >> 3) & 1) * 256 + (size_t)(rq & 7)
                          + (size_t)(kk >> 5) * 1024 + (size_t)(kk & 31) * 8;
        float mloc = -__builtin_inff();
        for (int kt0 = 0; kt0 < n_kt; kt0 += 8) {
            V8 k0[8], k1[8];
            T bv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {                                   // eight passes' loads in flight together
                const int key = min((kt0 + i) * AT_KB + kk, n_valid - 1);   // clamped: masked below
                const T *kp = k_base + (size_t)key * tok_stride + 16 * part;
                k0[i] = *reinterpret_cast<const V8 *>(kp);
                k1[i] = *reinterpret_cast<const V8 *>(kp + 8);
                if (HAS_BIAS) bv[i] = bias[bq + (size_t)min(kt0 + i, n_kt - 1) * 2048];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int key = (kt0 + i) * AT_KB + kk;
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < 8; t++) acc = __builtin_fmaf(qv[t], TR::to_f32(k0[i][t]), acc);
#pragma unroll
                for (int t = 0; t < 8; t++) acc = __builtin_fmaf(qv[8 + t], TR::to_f32(k1[i][t]), acc);
                acc += __shfl_xor(acc, 1, 64);
                acc += __shfl_xor(acc, 2, 64);
                if (part == 0 && kt0 + i < n_kt) {
                    float x = -__builtin_inff();
                    if (key < n_valid) x = (HAS_BIAS ? acc + TR::to_f32(bv[i]) : acc) * c2;     // (q.k + bias / scale) scale log2(e)
                    sbuf[key] = x;
                    mloc = fmaxf(mloc, x);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mloc = fmaxf(mloc, __shfl_xor(mloc, o, 64));
        if (lane == 0) red[wave] = mloc;
        __syncthreads();
        const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        // ---- probabilities and their sum ----
        float lloc = 0.f;
        for (int k = tid; k < Np64; k += AT_THREADS) {
            const float pr = TR::to_f32(TR::from_f32(__builtin_amdgcn_exp2f(sbuf[k] - mx)));      // exp2(-inf) = 0 for pad keys
            sbuf[k] = pr;
            lloc += pr;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lloc += __shfl_xor(lloc, o, 64);
        if (lane == 0) red[4 + wave] = lloc;
        __syncthreads();
        const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
        // ---- O[d] = sum_k p[k] V^T[d][k]: thread (d = tid >> 2, part): eight keys per step, steps part, part + 4, ... ----
        const int d = tid >> 2;
        const T *vrow = vt + (size_t)d * Np;
        float o = 0.f;
        for (int k0 = 8 * part; k0 < Np; k0 += 32 * 8) {                    // Np is a multiple of 8: whole 16-byte groups
            V8 v8[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v8[i] = *reinterpret_cast<const V8 *>(vrow + min(k0 + 32 * i, Np - 8));
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int kq = k0 + 32 * i;
                if (kq < Np) {
                    const float4 p0 = *reinterpret_cast<const float4 *>(sbuf + kq), p1 = *reinterpret_cast<const float4 *>(sbuf + kq + 4);
                    const float pp[8] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w };
#pragma unroll
                    for (int t = 0; t < 8; t++) if (pp[t] != 0.f) o = __builtin_fmaf(pp[t], TR::to_f32(v8[i][t]), o);   // pad keys: junk V^T, p = 0
                }
            }
        }
        o += __shfl_xor(o, 1, 64);
        o += __shfl_xor(o, 2, 64);
        if (part == 0) out_base[(size_t)q * (H * AT_D) + d] = TR::from_f32(o * inv);
        __syncthreads();                                                    // sbuf / red are rewritten by the next row
    }
}

// generation 4 (ds_attention4.hip): one wave per SIMD, two query sub-blocks skewed by half a tile inside the wave
void at4_launch(const AttnParams &P, int bf16, int has_bias, dim3 grid, hipStream_t st);
