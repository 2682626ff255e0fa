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
// Np (the token stride) is a multiple of 8; keys >= n_valid are masked (pad rows of the padded token sequence).
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
#include "ds_attention.h"

#ifdef DS_EXPERIMENTS          // the first kernel generation: A/B runs only (DS_ATT_V1=1), not in the shipped library
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


#endif  // DS_EXPERIMENTS

// =====================================================================================================================
// Version 2 of the kernel (the only one in the shipped library; -DDS_EXPERIMENTS builds keep the first one behind DS_ATT_V1=1).
//
// (A version 3 was built and measured in round 2 and removed again -- git history, profiles/round2_attention_v3_experiment.txt:
// one 8-wave workgroup = two independent 4-wave tasks whose wave-rows alternate, barrier by barrier, between an MFMA part
// (P.V of tile t, S of tile t+1) and a vector part (fragment reads, LDS-DMA of K / V^T three tiles ahead, softmax), the
// structure of csrc/ds_linear.hip.  Correct at the benchmark shapes, but 15-35 % SLOWER than this kernel (N = 1025 + bias:
// 0.344 vs 0.298 ms): with exactly two waves per SIMD the vector part -- ~200 instructions from ONE wave -- is bound by
// that wave's issue rate, and the softmax of a 64-wide head is as long as its MFMAs; the GEMM's memory part is 16
// instructions.  What remains is instruction-level interleaving inside a wave, i.e. a hand-scheduled kernel.)
//
// What changed, and why (round-1 profile: MFMA busy 20-26 %, ~200 VALU instructions per 16 MFMA with the bias, 3 waves/SIMD):
//   * 64 query rows per wave (two 32-row blocks), 256 per workgroup: every K / V^T fragment read from LDS feeds TWO MFMAs,
//     and K / V^T / the staging traffic per query row halve.  Two fat waves per SIMD (<= 256 VGPRs).
//   * the relative-position bias enters through the MATRIX pipe:  S^T = K.Q^T + Bias^T.I  with a constant identity B operand
//     (2 extra MFMAs per 32 x 32 logits block, exact: an f16 value times 1.0 accumulated in float32).  The packed operand is
//     stored as MFMA A fragments, in units of 1/scale (x8: exact), so the logits need NO per-element VALU work before the
//     softmax: 64 v_fma_mix + unpacking per tile become 8 MFMAs on a pipe that was three quarters idle.
//   * deferred running maximum: a query's maximum is only raised (and O rescaled) when it grows by more than 2^AT2_THR;
//     probabilities are then bounded by 2^AT2_THR instead of 1, harmless in f16/bf16 with float32 accumulation.
//   * K and V^T tiles use padded 144-byte rows (conflict-free ds_read_b128 for both; the XOR swizzle of version 1 left the
//     K reads 2-way conflicted), and V^T is stored key-permuted so that a P.V fragment is ONE ds_read_b128: within every 16
//     keys the order is [0-3, 8-11, 4-7, 12-15] -- the k-slot order the S^T accumulator hands to the P^T operand.
// ABL bits below 256: the same kernel with parts compiled out (timing experiments, results are WRONG) -- instantiated only in
// -DDS_EXPERIMENTS builds (DS_ATT_ABLATE); the shipped library has no switch that selects them.  A bit mask:
//   1 softmax reduced to a conversion   2 no K / V^T / bias fetch, no stash after the first tile   4 no barrier in the loop
//   8 no P.V MFMAs   16 no S MFMAs   32 fragments are not read from LDS (a register stands in)
// Bits 256 and up are OPTIONS with correct results (DS_ATT_OPT, A/B experiments):
//   256 s_setprio around the MFMA clusters   512 the bias MFMAs interleaved over the four accumulators (no back-to-back
//   dependent pair)   1024 row sums from the ROUNDED probabilities, two per v_dot2 (f16)   2048 every other workgroup starts
//   half a tile late (two waves of one SIMD otherwise run the same phase at the same time)   4096 (NQB = 1) late fetches --
//   the bias of a tile is requested at the top of that tile, K / V^T of the next one after S -- to fit 128 VGPRs: 4 waves/SIMD
// NQB = 32-row query blocks per wave: 2 (64 rows, <= 256 VGPRs, two waves per SIMD) or 1 (32 rows, four waves per SIMD).
// (Round 4 measured software-pipelined fragment reads -- the K / V^T fragments of slice s + 2 requested while slice s is multiplied,
// the first four ahead of the bias MFMAs, the two key blocks' accumulators alternating -- against this loop, whose S and P.V
// phases compile to "two reads, wait, two dependent MFMAs" four times each: 0.289 vs 0.285 ms at N = 1025 + bias, 0.375 vs 0.386 at
// N = 1370, 0.250 vs 0.250 at N = 2443, 0.848 vs 0.819 at N = 4097 + bias.  Three waves per SIMD already cover those round trips;
// the variant was removed.)
// Phase clock of generation 2 (-DDS_EXPERIMENTS builds, DS_ATT_PROF=1; ABL bit 8192): every wave accumulates the cycles between
// seven points of a tile -- [0] loop top, [1] K / V^T fetch issued, [2] S complete (LDS fragment reads + MFMAs, drained by a read of
// the last accumulator), [3] mask + softmax + rescale, [4] P.V complete, [5] K / V^T landed and stashed (vmcnt(0) + lgkmcnt(0)),
// [6] barrier passed -- and adds them to g_att_prof at the end ([7] = live waves, [8] = all waves).  The ticks serialise the
// phases (each one waits for what the phase started): the numbers say where a wave's time goes, not how fast the kernel is.
#ifdef DS_EXPERIMENTS
__device__ unsigned long long g_att_prof[16];
#define A2_TICK(i_, wait_) do { if (PROF) { asm volatile(wait_ ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); \
                                            tacc[i_] += t_ - tl; tl = t_; } } while (0)
#define A2_DRAIN(v_) do { if (PROF) { float d_; asm volatile("v_mov_b32 %0, %1" : "=v"(d_) : "v"(v_)); asm volatile("" :: "v"(d_)); } } while (0)
#else
#define A2_TICK(i_, wait_) do { } while (0)
#define A2_DRAIN(v_) do { } while (0)
#endif
template <int BF16, int HAS_BIAS, int NQB, int ABL>
__global__ __launch_bounds__(AT_THREADS, NQB == 2 ? 2 : ((ABL & 4096) ? 4 : 3)) void k_attention_fwd2(AttnParams P)
{
    [[maybe_unused]] constexpr bool PROF = (ABL & 8192) != 0;
    [[maybe_unused]] unsigned long long tacc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tl = 0;
    typedef at_traits<BF16> TR;
    typedef typename TR::T T;
    typedef typename TR::V8 V8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * AT2_TILE];       // K[2], V^T[2]: 36,864 B
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    int L = blockIdx.x;
    if (P.chunk > 0) {                                          // XCD-aware order, as in version 1
        L = (int)(blockIdx.x & 7) * P.chunk + (int)(blockIdx.x >> 3);
        if (L >= P.total) return;
    }
    // Work order inside an XCD's range.  Default: query blocks fastest -- the workgroups in flight on one L2 share the K / V^T
    // of one (batch, head), and a head's bias (2.4 MB at 1088 tokens) is read by one XCD only.  P.flags & 2: BATCH fastest
    // -- for sequences whose bias no longer fits an L2 (4160 tokens: 34.6 MB per head) the workgroups in flight then read
    // the SAME bias tiles at the same time (B of them per tile: one HBM read serves the batch), at the price of B times as
    // many K / V^T streams through that L2 (1 MB each).
    int qblk, b, h;
    if (P.flags & 2) { b = L % P.B; qblk = (L / P.B) % P.nq; h = L / (P.B * P.nq); }
    else { qblk = L % P.nq; b = (L / P.nq) % P.B; h = L / (P.nq * P.B); }
    const int q0 = qblk * (128 * NQB) + wave * (32 * NQB);
    // Np = token STRIDE of the operands (rows that exist per batch element, a multiple of 8); the key tiles and the packed
    // bias run on Np64, the stride rounded up to whole 64-key tiles.  Rows in [Np, Np64) do not exist: K rows there read as
    // zeros (the buffer descriptor ends at the batch element), V^T columns there alias the next row -- both are pad keys
    // (>= n_valid), whose probabilities are exactly 0 -- and query rows there are neither loaded nor stored.
    const int Np = P.Np, H = P.H;
    const int Np64 = (Np + 63) & ~63;
    const size_t tok_stride = (size_t)2 * H * AT_D;
    const T *qk = (const T *)P.qk + (size_t)b * Np * tok_stride;
    const T *q_base = qk + (size_t)h * AT_D;
    const T *k_base = qk + (size_t)(H + h) * AT_D;
    const T *vt = (const T *)P.vt + ((size_t)b * H + h) * AT_D * (size_t)Np;
    T *out_base = (T *)P.out + (size_t)b * Np * (size_t)(H * AT_D) + (size_t)h * AT_D;
    // a block with a handful of live rows (round 6): one GEMV per row instead of the tiled path (ds_attention.h: at_tail_rows)
    if ((ABL & 0xff) == 0 && !(ABL & 8192)) {                   // (not in the timing ablations / the phase clock)
        const int rows_live = P.n_valid - qblk * (128 * NQB);
        if (rows_live > 0 && rows_live <= AT_TAIL_ROWS && Np64 <= AT_TAIL_MAXN && (P.flags & 4)) {
            for (int row = qblk * (128 * NQB) + rows_live + (tid >> 3); row < min(Np, (qblk + 1) * (128 * NQB)); row += AT_THREADS / 8)
                *reinterpret_cast<uint4 *>(out_base + (size_t)row * (H * AT_D) + 8 * (tid & 7)) = make_uint4(0, 0, 0, 0);
            at_tail_rows<BF16, HAS_BIAS>(P, smem, b, h, qblk * (128 * NQB), rows_live);
            return;
        }
    }
    // a wave whose rows are all padding only helps staging; its output rows are zeroed (they feed the next GEMM as ordinary
    // rows and must stay finite), rows >= Np do not exist
    const bool wave_live = q0 < P.n_valid;
    if (!wave_live && q0 < Np) {
        const int row = q0 + lane;
        if (row < Np && lane < 32 * NQB) {
            uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 8; c++) *reinterpret_cast<uint4 *>(out_base + (size_t)row * (H * AT_D) + 8 * c) = z;
        }
    }

    V8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < NQB; qb++) {
        const int qrow = min(q0 + 32 * qb + l31, Np - 1);
        const T *qp = q_base + (size_t)qrow * tok_stride + 8 * hi;
#pragma unroll
        for (int s = 0; s < 4; s++) qf[qb][s] = *reinterpret_cast<const V8 *>(qp + 16 * s);
    }
    // identity B operand of the bias MFMAs: element t of slice s is I[k = 16 s + 8 hi + t][column l31]
    V8 ident[2];
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int t = 0; t < 8; t++) ident[s][t] = TR::from_f32((16 * s + 8 * hi + t) == l31 ? 1.0f : 0.0f);

    f32x16 o_acc[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; qb++)
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int r = 0; r < 16; r++) o_acc[qb][d][r] = 0.f;
    float m_run[2] = { -__builtin_inff(), -__builtin_inff() }, l_run[2] = { 0.f, 0.f };

    const int st_row = tid >> 3, st_chunk = tid & 7;
    u32x4 kreg0, kreg1, vreg0, vreg1;
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
        (void *)k_base, 0, (int)(((size_t)Np * tok_stride - (size_t)(H + h) * AT_D) * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void *)vt, 0, (int)((size_t)AT_D * Np * sizeof(T)), 0x00020000);
    const int vo_k = (int)((st_row * tok_stride + 8 * st_chunk) * sizeof(T));
    const int vo_v = (int)((st_row * Np + 8 * st_chunk) * sizeof(T));
    const int so_k32 = (int)(32 * tok_stride * sizeof(T)), so_r32 = (int)(32 * Np * sizeof(T));
#define A2_FETCH(kt_) do {                                                                                             \
        const int key0_ = (kt_) * AT_KB;                                                                                \
        const int sk_ = __builtin_amdgcn_readfirstlane(key0_ * (int)(tok_stride * sizeof(T)));                          \
        const int sv_ = __builtin_amdgcn_readfirstlane(key0_ * (int)sizeof(T));                                         \
        kreg0 = __builtin_amdgcn_raw_buffer_load_b128(rs_k, vo_k, sk_, 0);                                              \
        kreg1 = __builtin_amdgcn_raw_buffer_load_b128(rs_k, vo_k, sk_ + so_k32, 0);                                     \
        vreg0 = __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo_v, sv_, 0);                                              \
        vreg1 = __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo_v, sv_ + so_r32, 0);                                     \
    } while (0)
    // K row r, chunk c -> r*144 + 16c.  V^T row d, chunk c (keys 8c .. 8c+7) -> the permuted order described above: the low
    // four keys go to chunk (c & ~1), the high four to chunk (c | 1), each at byte offset 8 (c & 1)
    const int vst_lo = ((st_chunk & ~1) << 4) + ((st_chunk & 1) << 3), vst_hi = vst_lo + 16;
#define A2_STASH1(buf_, row_, kr_, vr_) do {                                                                            \
        *reinterpret_cast<u32x4 *>(smem + (buf_) * AT2_TILE + (row_) * AT2_ROW + (st_chunk << 4)) = kr_;                \
        unsigned char *vd_ = smem + (2 + (buf_)) * AT2_TILE + (row_) * AT2_ROW;                                         \
        *reinterpret_cast<uint2 *>(vd_ + vst_lo) = make_uint2(vr_.x, vr_.y);                                            \
        *reinterpret_cast<uint2 *>(vd_ + vst_hi) = make_uint2(vr_.z, vr_.w);                                            \
    } while (0)
    // Bias: [head][32-query block][64-key tile][chunk c = 2 kb + s][64 lanes][8]: lane (hi, l31) of chunk c holds
    // bias[query 16 s + 8 hi + t][key 32 kb + l31] / scale, t = 0..7 -- the A fragment of the MFMA that adds it.
    u32x4 breg[2][4];                                           // (unused, and eliminated, without a bias)
    const int n_kt = Np64 / AT_KB;
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(HAS_BIAS ? (const T *)P.bias + (size_t)h * Np64 * (size_t)Np64 : (const T *)P.qk), 0,
        (int)((size_t)Np64 * Np64 * sizeof(T)), 0x00020000);
    const int vo_b = (int)((((size_t)(q0 / 32) * n_kt) * 2048 + (size_t)lane * 8) * sizeof(T));
    const int so_bq = (int)((size_t)n_kt * 2048 * sizeof(T));                               // next 32-query block
#define A2_FETCH_BIAS(kt_) do {                                                                                        \
        const int sb_ = __builtin_amdgcn_readfirstlane((kt_) * (int)(2048 * sizeof(T)));                                \
        _Pragma("unroll") for (int qb_ = 0; qb_ < NQB; qb_++)                                                           \
        _Pragma("unroll") for (int c_i = 0; c_i < 4; c_i++)                                                             \
            breg[qb_][c_i] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, vo_b + qb_ * so_bq + c_i * 1024, sb_, 0);     \
    } while (0)

    const int ntiles = (P.n_valid + AT_KB - 1) / AT_KB;
    const float c_ = P.c_exp;                                   // scale * log2(e): the bias is stored in units of 1/scale
    const float thr_x = AT2_THR / c_;
    A2_FETCH(0);
    if (HAS_BIAS && wave_live && !(ABL & 4096)) A2_FETCH_BIAS(0);
    A2_STASH1(0, st_row, kreg0, vreg0);
    A2_STASH1(0, st_row + 32, kreg1, vreg1);
    __syncthreads();
    // One 64-key tile.  MASKED is a compile-time flag: only the last tile can hold pad keys, so the steady-state body has no
    // mask code and no control flow besides the (rare) rescale.
    auto tile = [&](const int kt, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int cur = kt & 1;
        const unsigned char *s_k = smem + cur * AT2_TILE, *s_v = smem + (2 + cur) * AT2_TILE;
        const bool more = kt + 1 < ntiles;
        // next tile's K / V^T: in flight while this tile is computed.  The staggered variant requests them after its first
        // mixed region, where the register pressure peaks (the remaining three regions still cover an L2 round trip)
        constexpr int abl = ABL;
        A2_TICK(0, "");
        if (more && !(abl & 2) && (!(abl & 4096) || !wave_live)) A2_FETCH(kt + 1);
        if (HAS_BIAS && (abl & 4096) && wave_live) A2_FETCH_BIAS(kt);
        A2_TICK(1, "");
        if (wave_live) {
            const int key0 = kt * AT_KB;
            f32x16 s_acc[2][2];
            V8 pf[2][2][2];
            float alpha[2];
            bool grow[2];
            // S^T = Bias^T.I (+) K.Q^T of query block qb_, key block kb_: the two bias MFMAs start the accumulation chain
#define A2_S_BIAS(qb_, kb_) do {                                                                                       \
                if (HAS_BIAS) {                                                                                         \
                    f32x16 z_;                                                                                          \
                    _Pragma("unroll") for (int r_ = 0; r_ < 16; r_++) z_[r_] = 0.f;                                     \
                    union { u32x4 u; V8 v; } b0_, b1_;                                                                  \
                    b0_.u = breg[qb_][2 * (kb_)]; b1_.u = breg[qb_][2 * (kb_) + 1];                                     \
                    s_acc[qb_][kb_] = TR::mfma(b0_.v, ident[0], z_);                                                    \
                    s_acc[qb_][kb_] = TR::mfma(b1_.v, ident[1], s_acc[qb_][kb_]);                                       \
                } else {                                                                                                \
                    _Pragma("unroll") for (int r_ = 0; r_ < 16; r_++) s_acc[qb_][kb_][r_] = 0.f;                        \
                }                                                                                                       \
            } while (0)
#define A2_MASK(qb_) do {                                                                                              \
                if (MASKED) {                                                                                           \
                    _Pragma("unroll") for (int kb_ = 0; kb_ < 2; kb_++)                                                 \
                    _Pragma("unroll") for (int r_ = 0; r_ < 16; r_++)                                                   \
                        if (key0 + kb_ * 32 + at_crow(r_, hi) >= P.n_valid) s_acc[qb_][kb_][r_] = -__builtin_inff();    \
                }                                                                                                       \
            } while (0)
            // online softmax of one 32-query block in the exp2 domain with a deferred maximum: the running maximum is only
            // raised when it would grow by more than the threshold (first tile: from -inf); alpha = 1 when kept
#define A2_SOFTMAX(qb_) do {                                                                                           \
                float mx_ = at_max3(s_acc[qb_][0][0], s_acc[qb_][1][0], s_acc[qb_][0][1]);                              \
                mx_ = at_max3(mx_, s_acc[qb_][1][1], s_acc[qb_][0][2]);                                                 \
                mx_ = at_max3(mx_, s_acc[qb_][1][2], s_acc[qb_][0][3]);                                                 \
                _Pragma("unroll") for (int r_ = 3; r_ < 15; r_ += 2) {                                                  \
                    mx_ = at_max3(mx_, s_acc[qb_][1][r_], s_acc[qb_][0][r_ + 1]);                                       \
                    mx_ = at_max3(mx_, s_acc[qb_][1][r_ + 1], s_acc[qb_][0][r_ + 2]);                                   \
                }                                                                                                       \
                mx_ = at_max3(mx_, s_acc[qb_][1][15], mx_);                                                             \
                mx_ = at_max3(mx_, __shfl_xor(mx_, 32, 64), mx_);                                                       \
                grow[qb_] = mx_ > m_run[qb_] + thr_x;                                                                   \
                const float mn_ = grow[qb_] ? mx_ : m_run[qb_];                                                         \
                alpha[qb_] = __builtin_amdgcn_exp2f((m_run[qb_] - mn_) * c_);                                           \
                m_run[qb_] = mn_;                                                                                       \
                const float mc_ = -mn_ * c_;                                                                            \
                float l0_ = 0.f, l1_ = 0.f;                                                                             \
                _Pragma("unroll") for (int kb_ = 0; kb_ < 2; kb_++)                                                     \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; j_++)                                                        \
                _Pragma("unroll") for (int t_ = 0; t_ < 8; t_ += 2) {                                                   \
                    const float p0_ = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[qb_][kb_][8 * j_ + t_], c_, mc_));    \
                    const float p1_ = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[qb_][kb_][8 * j_ + t_ + 1], c_, mc_)); \
                    pf[qb_][kb_][j_][t_] = TR::from_f32(p0_);                                                           \
                    pf[qb_][kb_][j_][t_ + 1] = TR::from_f32(p1_);                                                       \
                    if (ABL & 1024) { if (t_ & 2) l1_ = TR::add2(pf[qb_][kb_][j_], t_, l1_); else l0_ = TR::add2(pf[qb_][kb_][j_], t_, l0_); } \
                    else { l0_ += p0_; l1_ += p1_; }                                                                    \
                }                                                                                                       \
                l_run[qb_] = l_run[qb_] * alpha[qb_] + (l0_ + l1_);                                                     \
            } while (0)
#define A2_RESCALE(qb_) do {                                                                                           \
                if (__any(grow[qb_])) {                         /* some query of this block moved its maximum */        \
                    _Pragma("unroll") for (int d_ = 0; d_ < 2; d_++)                                                    \
                    _Pragma("unroll") for (int r_ = 0; r_ < 16; r_++) o_acc[qb_][d_][r_] *= alpha[qb_];                 \
                }                                                                                                       \
            } while (0)
            {
                // ---- S^T for both query blocks (every K fragment feeds two MFMAs), softmax, P.V (every V^T fragment too) ----
#pragma unroll
                for (int qb = 0; qb < NQB; qb++)
#pragma unroll
                    for (int kb = 0; kb < 2; kb++) {
                        if (abl & 16) {
#pragma unroll
                            for (int r = 0; r < 16; r++) s_acc[qb][kb][r] = (float)(r + kt);
                        } else if (HAS_BIAS && (abl & 512)) {
                            f32x16 z_;
#pragma unroll
                            for (int r = 0; r < 16; r++) z_[r] = 0.f;
                            union { u32x4 u; V8 v; } b0_;
                            b0_.u = breg[qb][2 * kb];
                            s_acc[qb][kb] = TR::mfma(b0_.v, ident[0], z_);
                        } else A2_S_BIAS(qb, kb);
                    }
                if (HAS_BIAS && (abl & 512) && !(abl & 16)) {
#pragma unroll
                    for (int qb = 0; qb < NQB; qb++)
#pragma unroll
                        for (int kb = 0; kb < 2; kb++) {
                            union { u32x4 u; V8 v; } b1_;
                            b1_.u = breg[qb][2 * kb + 1];
                            s_acc[qb][kb] = TR::mfma(b1_.v, ident[1], s_acc[qb][kb]);
                        }
                }
                if (abl & 256) __builtin_amdgcn_s_setprio(1);
                if (!(ABL && (abl & 16))) {
#pragma unroll
                    for (int kb = 0; kb < 2; kb++) {
                        const unsigned char *krow = s_k + (kb * 32 + l31) * AT2_ROW + (hi << 4);
#pragma unroll
                        for (int s = 0; s < 4; s++) {
                            const V8 kf = (abl & 32) ? qf[NQB - 1][s] : *reinterpret_cast<const V8 *>(krow + (s << 5));
                            s_acc[0][kb] = TR::mfma(kf, qf[0][s], s_acc[0][kb]);
                            if (NQB == 2) s_acc[1][kb] = TR::mfma(kf, qf[1][s], s_acc[1][kb]);
                        }
                    }
                }
                if (abl & 256) __builtin_amdgcn_s_setprio(0);
                A2_DRAIN(s_acc[NQB - 1][1][15]);
                A2_TICK(2, "s_waitcnt lgkmcnt(0)");
                // the bias registers are free: next tile's fragments land under the softmax / P.V of this one
                if (HAS_BIAS && more && !(abl & 2) && !(abl & 4096)) A2_FETCH_BIAS(kt + 1);
                if ((abl & 4096) && more) A2_FETCH(kt + 1);
                A2_MASK(0);
                if (NQB == 2) A2_MASK(1);
                if (abl & 1) {
#pragma unroll
                    for (int qb = 0; qb < NQB; qb++) {
#pragma unroll
                        for (int kb = 0; kb < 2; kb++)
#pragma unroll
                            for (int j = 0; j < 2; j++)
#pragma unroll
                                for (int t = 0; t < 8; t++) pf[qb][kb][j][t] = TR::from_f32(s_acc[qb][kb][8 * j + t]);
                        l_run[qb] += 1.0f; grow[qb] = false; alpha[qb] = 1.0f;
                    }
                } else {
                    A2_SOFTMAX(0); A2_RESCALE(0);
                    if (NQB == 2) { A2_SOFTMAX(1); A2_RESCALE(1); }
                }
                A2_TICK(3, "");
                if (!(abl & 8)) {
                    if (abl & 256) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int d = 0; d < 2; d++) {
                        const unsigned char *vrow = s_v + (d * 32 + l31) * AT2_ROW + (hi << 4);
#pragma unroll
                        for (int kb = 0; kb < 2; kb++)
#pragma unroll
                            for (int j = 0; j < 2; j++) {
                                const V8 vf = (abl & 32) ? qf[0][kb * 2 + j] : *reinterpret_cast<const V8 *>(vrow + ((kb * 4 + j * 2) << 4));
                                o_acc[0][d] = TR::mfma(vf, pf[0][kb][j], o_acc[0][d]);
                                if (NQB == 2) o_acc[1][d] = TR::mfma(vf, pf[1][kb][j], o_acc[1][d]);
                            }
                    }
                    if (abl & 256) __builtin_amdgcn_s_setprio(0);
                } else {
#pragma unroll
                    for (int qb = 0; qb < NQB; qb++)
#pragma unroll
                        for (int kb = 0; kb < 2; kb++)
#pragma unroll
                            for (int j = 0; j < 2; j++) asm volatile("" :: "v"(pf[qb][kb][j]));
                }
            }
        }
        if (wave_live) A2_DRAIN(o_acc[NQB - 1][1][15]);
        A2_TICK(4, "s_waitcnt lgkmcnt(0)");
        if (more && !(abl & 2)) {
            A2_STASH1(cur ^ 1, st_row, kreg0, vreg0);
            A2_STASH1(cur ^ 1, st_row + 32, kreg1, vreg1);
        }
        A2_TICK(5, "s_waitcnt vmcnt(0) lgkmcnt(0)");
        if (!(abl & 4)) __syncthreads();
        A2_TICK(6, "");
    };
    const bool pad_keys = (P.n_valid & (AT_KB - 1)) != 0;
    if ((ABL & 2048) && ((blockIdx.x >> 3) & 1)) __builtin_amdgcn_s_sleep(48);
#ifdef DS_EXPERIMENTS
    if (PROF) tl = __builtin_readcyclecounter();
#endif
    for (int kt = 0; kt + 1 < ntiles; kt++) tile(kt, std::false_type());
    if (pad_keys) tile(ntiles - 1, std::true_type());
    else tile(ntiles - 1, std::false_type());
#ifdef DS_EXPERIMENTS
    if (PROF && lane == 0) {
#pragma unroll
        for (int i = 0; i < 7; i++) atomicAdd(&g_att_prof[i], tacc[i]);
        atomicAdd(&g_att_prof[wave_live ? 7 : 8], 1ull);
    }
#endif
    if (!wave_live) return;
#pragma unroll
    for (int qb = 0; qb < NQB; qb++) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        const int qrow = q0 + 32 * qb + l31;
        if (qrow < Np) {
            T *op = out_base + (size_t)qrow * (size_t)(H * AT_D);
#pragma unroll
            for (int d = 0; d < 2; d++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    T v4[4];
#pragma unroll
                    for (int t = 0; t < 4; t++) v4[t] = TR::from_f32(o_acc[qb][d][4 * g + t] * inv);
                    *reinterpret_cast<uint2 *>(op + d * 32 + 8 * g + 4 * hi) = *reinterpret_cast<const uint2 *>(v4);
                }
        }
    }
}

#ifdef DS_EXPERIMENTS
// ---- generation 3: built at the end of round 4, compiled only in -DDS_EXPERIMENTS builds and selected there by DS_ATT_GEN=3
// for the shapes generation 2 serves with 32-row waves.  MEASURED (profiles/round4_attention_gen3_ab.txt, tools/att_ab.sh):
// bit-identical to generation 2 at (32, 1025, 16 heads, bias) and NOT faster -- 0.3135 against 0.3120 ms: the overlap gained
// inside a wave is paid for with the third wave per SIMD, and what binds both generations is the chain through the memory path
// (DESIGN.md 7.1).  Kept as the correct starting point of the hand-scheduled kernel, not as a candidate.  Motivation: in
// generation 2 a wave's tile is one serial chain -- S MFMAs -> softmax -> P.V MFMAs -- and the counters show the vector pipe
// (53 % of the cycles) and the matrix pipe (38 %) taking turns.  Here the chain is skewed by one tile INSIDE the wave:
//     iteration t:  S(t + 1) MFMAs   beside   row maxima + first half of the exponentials of tile t      (phase A)
//                   P.V(t) MFMAs of key block 0   beside   the second half of the exponentials             (phase B)
//                   P.V(t) MFMAs of key block 1                                                             (phase C)
// so that the vector work of a tile runs in the shadow of MFMAs that do not depend on it.  Cost: a second set of S accumulators
// (two waves per SIMD instead of three) and K staged one tile ahead of V^T.  Operands, work order, LDS image, bias operand,
// deferred maximum and every arithmetic operation are those of generation 2 with NQB = 1: the results must be bit-identical.
// LDS: K[2] | V^T[2] as before.  At the top of iteration t the buffers hold K(t + 1) in K[(t + 1) & 1] and V^T(t) in V[t & 1];
// K(t + 2) and V^T(t + 1) are fetched into registers during the iteration and stashed at its end into K[t & 1] (last read for
// S(t), one iteration ago) and V[(t + 1) & 1] (last read for P.V(t - 1)): one barrier per tile, no hazard inside an iteration.
template <int BF16, int HAS_BIAS>
__global__ __launch_bounds__(AT_THREADS, 2) void k_attention_fwd3(AttnParams P)
{
    typedef at_traits<BF16> TR;
    typedef typename TR::T T;
    typedef typename TR::V8 V8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * AT2_TILE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    int L = blockIdx.x;
    if (P.chunk > 0) {
        L = (int)(blockIdx.x & 7) * P.chunk + (int)(blockIdx.x >> 3);
        if (L >= P.total) return;
    }
    int qblk, b, h;
    if (P.flags & 2) { b = L % P.B; qblk = (L / P.B) % P.nq; h = L / (P.B * P.nq); }
    else { qblk = L % P.nq; b = (L / P.nq) % P.B; h = L / (P.nq * P.B); }
    const int q0 = qblk * 128 + wave * 32;
    const int Np = P.Np, H = P.H;
    const int Np64 = (Np + 63) & ~63;
    const size_t tok_stride = (size_t)2 * H * AT_D;
    const T *qk = (const T *)P.qk + (size_t)b * Np * tok_stride;
    const T *q_base = qk + (size_t)h * AT_D;
    const T *k_base = qk + (size_t)(H + h) * AT_D;
    const T *vt = (const T *)P.vt + ((size_t)b * H + h) * AT_D * (size_t)Np;
    T *out_base = (T *)P.out + (size_t)b * Np * (size_t)(H * AT_D) + (size_t)h * AT_D;
    const bool wave_live = q0 < P.n_valid;
    if (!wave_live && q0 < Np) {
        const int row = q0 + lane;
        if (row < Np && lane < 32) {
            uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 8; c++) *reinterpret_cast<uint4 *>(out_base + (size_t)row * (H * AT_D) + 8 * c) = z;
        }
    }
    V8 qf[4];
    {
        const int qrow = min(q0 + l31, Np - 1);
        const T *qp = q_base + (size_t)qrow * tok_stride + 8 * hi;
#pragma unroll
        for (int s = 0; s < 4; s++) qf[s] = *reinterpret_cast<const V8 *>(qp + 16 * s);
    }
    V8 ident[2];
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int t = 0; t < 8; t++) ident[s][t] = TR::from_f32((16 * s + 8 * hi + t) == l31 ? 1.0f : 0.0f);
    f32x16 o_acc[2];
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) o_acc[d][r] = 0.f;
    float m_run = -__builtin_inff(), l_run = 0.f;

    const int st_row = tid >> 3, st_chunk = tid & 7;
    u32x4 kreg0, kreg1, vreg0, vreg1;
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
        (void *)k_base, 0, (int)(((size_t)Np * tok_stride - (size_t)(H + h) * AT_D) * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void *)vt, 0, (int)((size_t)AT_D * Np * sizeof(T)), 0x00020000);
    const int vo_k = (int)((st_row * tok_stride + 8 * st_chunk) * sizeof(T));
    const int vo_v = (int)((st_row * Np + 8 * st_chunk) * sizeof(T));
    const int so_k32 = (int)(32 * tok_stride * sizeof(T)), so_r32 = (int)(32 * Np * sizeof(T));
#define A3_FETCH_K(kt_) do {                                                                                           \
        const int sk_ = __builtin_amdgcn_readfirstlane((kt_) * AT_KB * (int)(tok_stride * sizeof(T)));                  \
        kreg0 = __builtin_amdgcn_raw_buffer_load_b128(rs_k, vo_k, sk_, 0);                                              \
        kreg1 = __builtin_amdgcn_raw_buffer_load_b128(rs_k, vo_k, sk_ + so_k32, 0);                                     \
    } while (0)
#define A3_FETCH_V(kt_) do {                                                                                           \
        const int sv_ = __builtin_amdgcn_readfirstlane((kt_) * AT_KB * (int)sizeof(T));                                 \
        vreg0 = __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo_v, sv_, 0);                                              \
        vreg1 = __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo_v, sv_ + so_r32, 0);                                     \
    } while (0)
    const int vst_lo = ((st_chunk & ~1) << 4) + ((st_chunk & 1) << 3), vst_hi = vst_lo + 16;
#define A3_STASH_K(buf_) do {                                                                                          \
        *reinterpret_cast<u32x4 *>(smem + (buf_) * AT2_TILE + st_row * AT2_ROW + (st_chunk << 4)) = kreg0;              \
        *reinterpret_cast<u32x4 *>(smem + (buf_) * AT2_TILE + (st_row + 32) * AT2_ROW + (st_chunk << 4)) = kreg1;       \
    } while (0)
#define A3_STASH_V(buf_) do {                                                                                          \
        unsigned char *vd0_ = smem + (2 + (buf_)) * AT2_TILE + st_row * AT2_ROW;                                        \
        unsigned char *vd1_ = vd0_ + 32 * AT2_ROW;                                                                      \
        *reinterpret_cast<uint2 *>(vd0_ + vst_lo) = make_uint2(vreg0.x, vreg0.y);                                       \
        *reinterpret_cast<uint2 *>(vd0_ + vst_hi) = make_uint2(vreg0.z, vreg0.w);                                       \
        *reinterpret_cast<uint2 *>(vd1_ + vst_lo) = make_uint2(vreg1.x, vreg1.y);                                       \
        *reinterpret_cast<uint2 *>(vd1_ + vst_hi) = make_uint2(vreg1.z, vreg1.w);                                       \
    } while (0)
    u32x4 breg[4];
    const int n_kt = Np64 / AT_KB;
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(HAS_BIAS ? (const T *)P.bias + (size_t)h * Np64 * (size_t)Np64 : (const T *)P.qk), 0,
        (int)((size_t)Np64 * Np64 * sizeof(T)), 0x00020000);
    const int vo_b = (int)((((size_t)(q0 / 32) * n_kt) * 2048 + (size_t)lane * 8) * sizeof(T));
#define A3_FETCH_BIAS(kt_) do {                                                                                        \
        const int sb_ = __builtin_amdgcn_readfirstlane((kt_) * (int)(2048 * sizeof(T)));                                \
        _Pragma("unroll") for (int c_i = 0; c_i < 4; c_i++)                                                             \
            breg[c_i] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, vo_b + c_i * 1024, sb_, 0);                         \
    } while (0)
    // S^T of one tile into sacc_[2]: the two key blocks' accumulation chains alternate (a dependent MFMA waits for its
    // predecessor: two chains keep the pipe fed); the bias of THAT tile is in breg
#define A3_S(sacc_, kbuf_) do {                                                                                        \
        const unsigned char *s_k_ = smem + (kbuf_) * AT2_TILE;                                                          \
        if (HAS_BIAS) {                                                                                                 \
            f32x16 z_;                                                                                                  \
            _Pragma("unroll") for (int r_ = 0; r_ < 16; r_++) z_[r_] = 0.f;                                             \
            union { u32x4 u; V8 v; } b00_, b01_, b10_, b11_;                                                            \
            b00_.u = breg[0]; b01_.u = breg[1]; b10_.u = breg[2]; b11_.u = breg[3];                                     \
            sacc_[0] = TR::mfma(b00_.v, ident[0], z_);                                                                  \
            sacc_[1] = TR::mfma(b10_.v, ident[0], z_);                                                                  \
            sacc_[0] = TR::mfma(b01_.v, ident[1], sacc_[0]);                                                            \
            sacc_[1] = TR::mfma(b11_.v, ident[1], sacc_[1]);                                                            \
        } else {                                                                                                        \
            _Pragma("unroll") for (int r_ = 0; r_ < 16; r_++) sacc_[0][r_] = sacc_[1][r_] = 0.f;                        \
        }                                                                                                               \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; s_++)                                                                \
        _Pragma("unroll") for (int kb_ = 0; kb_ < 2; kb_++) {                                                           \
            const V8 kf_ = *reinterpret_cast<const V8 *>(s_k_ + (kb_ * 32 + l31) * AT2_ROW + (hi << 4) + (s_ << 5));    \
            sacc_[kb_] = TR::mfma(kf_, qf[s_], sacc_[kb_]);                                                             \
        }                                                                                                               \
    } while (0)

    const int ntiles = (P.n_valid + AT_KB - 1) / AT_KB;
    const float c_ = P.c_exp;
    const float thr_x = AT2_THR / c_;
    f32x16 sA[2], sB[2];
    // prologue: K(0), V^T(0) [, K(1)] into LDS, S(0) into sA, the bias of tile 1 requested
    A3_FETCH_K(0);
    A3_FETCH_V(0);
    if (HAS_BIAS && wave_live) A3_FETCH_BIAS(0);
    A3_STASH_K(0);
    A3_STASH_V(0);
    if (ntiles > 1) {
        A3_FETCH_K(1);
        A3_STASH_K(1);
    }
    __syncthreads();
    if (wave_live) {
        A3_S(sA, 0);
        if (HAS_BIAS && ntiles > 1) A3_FETCH_BIAS(1);
    }
    // iteration 0 ends by stashing K(2) over K(0): every wave must be done with S(0) first (found by the tile-level model,
    // tools/emulate_attention_skew.py, which flags a buffer written in the barrier interval in which it is read)
    __syncthreads();
    // one iteration: tile t from scur_ (S(t), complete), S(t + 1) into snxt_ when NEXT
    auto iter = [&](f32x16 (&scur)[2], f32x16 (&snxt)[2], const int t, auto next_tag, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool NEXT = decltype(next_tag)::value, MASKED = decltype(masked_tag)::value;
        const bool more2 = t + 2 < ntiles;
        if (more2) A3_FETCH_K(t + 2);
        if (NEXT) A3_FETCH_V(t + 1);
        if (wave_live) {
            const int key0 = t * AT_KB;
            const unsigned char *s_v = smem + (2 + (t & 1)) * AT2_TILE;
            V8 pf[2][2];
            if (MASKED) {
#pragma unroll
                for (int kb = 0; kb < 2; kb++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        if (key0 + kb * 32 + at_crow(r, hi) >= P.n_valid) scur[kb][r] = -__builtin_inff();
            }
            // ---- phase A: S(t + 1) MFMAs beside the row maximum and the exponentials of key block 0 ----
            if (NEXT) A3_S(snxt, (t + 1) & 1);
            float mx = at_max3(scur[0][0], scur[1][0], scur[0][1]);
            mx = at_max3(mx, scur[1][1], scur[0][2]);
            mx = at_max3(mx, scur[1][2], scur[0][3]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) {
                mx = at_max3(mx, scur[1][r], scur[0][r + 1]);
                mx = at_max3(mx, scur[1][r + 1], scur[0][r + 2]);
            }
            mx = at_max3(mx, scur[1][15], mx);
            {   // the other 32 keys of the row sit in lane ^ 32: one v_permlane32_swap (no LDS round trip in the chain)
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const bool grow = mx > m_run + thr_x;
            const float mn = grow ? mx : m_run;
            const float alpha = __builtin_amdgcn_exp2f((m_run - mn) * c_);
            m_run = mn;
            const float mc = -mn * c_;
            float l0 = 0.f, l1 = 0.f;
#define A3_EXP(kb_) do {                                                                                               \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; j_++)                                                        \
                _Pragma("unroll") for (int t_ = 0; t_ < 8; t_ += 2) {                                                   \
                    const float p0_ = __builtin_amdgcn_exp2f(__builtin_fmaf(scur[kb_][8 * j_ + t_], c_, mc));           \
                    const float p1_ = __builtin_amdgcn_exp2f(__builtin_fmaf(scur[kb_][8 * j_ + t_ + 1], c_, mc));       \
                    pf[kb_][j_][t_] = TR::from_f32(p0_);                                                                \
                    pf[kb_][j_][t_ + 1] = TR::from_f32(p1_);                                                            \
                    l0 += p0_; l1 += p1_;                                                                               \
                }                                                                                                       \
            } while (0)
            A3_EXP(0);
            // (the probabilities of key block 0 are only consumed behind the branches below: without this use the compiler
            // sinks their whole computation past them, out of the MFMAs' shadow)
            asm volatile("" :: "v"(pf[0][0]), "v"(pf[0][1]), "v"(l0), "v"(l1), "v"(alpha));
            if (NEXT) {
                // 12 (8 without a bias) MFMAs, each followed by its share of the ~80 vector instructions above
#pragma unroll
                for (int g = 0; g < (HAS_BIAS ? 12 : 8); g++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // the K fragment of the next one
                    __builtin_amdgcn_sched_group_barrier(0x002, HAS_BIAS ? 7 : 10, 0);   // vector work
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // breg is free: the bias of tile t + 2 (past the last tile the descriptor returns zeros or the next query block's
            // first tile: never used -- no branch here, the phases stay in straight-line code)
            if (HAS_BIAS && NEXT) A3_FETCH_BIAS(t + 2);
            // the running maximum moved for some query of the wave: rescale (rare after the first tiles)
            if (__any(grow)) {
#pragma unroll
                for (int d = 0; d < 2; d++)
#pragma unroll
                    for (int r = 0; r < 16; r++) o_acc[d][r] *= alpha;
            }
            // ---- phase B: P.V of key block 0 beside the exponentials of key block 1; phase C: P.V of key block 1 ----
#define A3_PV(kb_) do {                                                                                                \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; j_++)                                                        \
                _Pragma("unroll") for (int d_ = 0; d_ < 2; d_++) {                                                      \
                    const V8 vf_ = *reinterpret_cast<const V8 *>(s_v + (d_ * 32 + l31) * AT2_ROW + (hi << 4) + (((kb_) * 4 + j_ * 2) << 4)); \
                    o_acc[d_] = TR::mfma(vf_, pf[kb_][j_], o_acc[d_]);                                                  \
                }                                                                                                       \
            } while (0)
            A3_PV(0);
            A3_EXP(1);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);           // the four V^T fragments of key block 0
#pragma unroll
            for (int g = 0; g < 4; g++) {
                __builtin_amdgcn_sched_group_barrier(0x002, 14, 1);      // a quarter of the exponentials of key block 1
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);       // one P.V MFMA of key block 0
            }
            __builtin_amdgcn_sched_barrier(0);
            A3_PV(1);
            l_run = l_run * alpha + (l0 + l1);
#undef A3_EXP
#undef A3_PV
        }
        if (more2) A3_STASH_K(t & 1);
        if (NEXT) A3_STASH_V((t + 1) & 1);
        __syncthreads();
    };
    const bool pad_keys = (P.n_valid & (AT_KB - 1)) != 0;
    const int ntl = ntiles - 1;
    int t = 0;
    for (; t + 1 < ntl; t += 2) {
        iter(sA, sB, t, std::true_type(), std::false_type());
        iter(sB, sA, t + 1, std::true_type(), std::false_type());
    }
    if (t < ntl) {
        iter(sA, sB, t, std::true_type(), std::false_type());
        if (pad_keys) iter(sB, sA, ntl, std::false_type(), std::true_type());
        else iter(sB, sA, ntl, std::false_type(), std::false_type());
    } else {
        if (pad_keys) iter(sA, sB, ntl, std::false_type(), std::true_type());
        else iter(sA, sB, ntl, std::false_type(), std::false_type());
    }
#undef A3_S
#undef A3_FETCH_BIAS
#undef A3_STASH_V
#undef A3_STASH_K
#undef A3_FETCH_V
#undef A3_FETCH_K
    if (!wave_live) return;
    {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        const int qrow = q0 + l31;
        if (qrow < Np) {
            T *op = out_base + (size_t)qrow * (size_t)(H * AT_D);
#pragma unroll
            for (int d = 0; d < 2; d++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    T v4[4];
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) v4[tt] = TR::from_f32(o_acc[d][4 * g + tt] * inv);
                    *reinterpret_cast<uint2 *>(op + d * 32 + 8 * g + 4 * hi) = *reinterpret_cast<const uint2 *>(v4);
                }
        }
    }
}
#endif

// bias operand of version 2: [H][Np/32][Np/64][4 chunks][64 lanes][8], values bias / scale (scale = 1/8: exact)
template <int BF16>
__global__ void k_attention_bias_pack2(const float *__restrict__ bias, typename at_traits<BF16>::T *__restrict__ out,
                                       int H, int n, int Np, float mul)
{
    typedef at_traits<BF16> TR;
    const long long total = (long long)H * Np * Np;
    const int n_kt = Np / AT_KB, nq32 = Np / 32;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx & 7), lane = (int)((idx >> 3) & 63), c = (int)((idx >> 9) & 3);
        const long long tile = idx >> 11;
        const int kt = (int)(tile % n_kt), qb = (int)((tile / n_kt) % nq32), h = (int)(tile / ((long long)n_kt * nq32));
        const int q = qb * 32 + 16 * (c & 1) + 8 * (lane >> 5) + t;
        const int k = kt * AT_KB + 32 * (c >> 1) + (lane & 31);
        const float v = (q < n && k < n) ? bias[((size_t)h * n + q) * n + k] * mul : 0.f;
        out[idx] = TR::from_f32(v);
    }
}

#ifdef DS_EXPERIMENTS
static int at_version()
{
    static const int v = (getenv("DS_ATT_V1") && atoi(getenv("DS_ATT_V1"))) ? 1 : 2;
    return v;
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

#else
static constexpr int at_version() { return 2; }
#endif

// A/B switches of ds_attention_fwd, read once per process and again by ds_attention_reload_env (tests, A/B runs):
//   DS_ATT_GEN    0 / unset: generation by shape; 2: generation 2 everywhere; 4: generation 4 everywhere
//   DS_ATT_NQB    generation 2: 32-row query blocks per wave (1 or 2; unset: by sequence length)
//   DS_ATT_LATE   generation 2, 32 rows per wave: late K / V^T fetch (0 / 1; unset: on without a bias)
//   DS_ATT_ORDER  batch-fastest work order (0 / 1; unset: when one head's packed bias exceeds an L2)
//   DS_ATT_TAIL   query blocks with <= 4 live rows as GEMVs (ds_attention.h: at_tail_rows; 0 / 1, default 1)
struct AtEnv { int gen, nqb, late, order, tail; };
static AtEnv g_at_env = { -1, 0, -1, -1, 1 };
static void at_env_read()
{
    auto geti = [](const char *k, int d) { const char *v = getenv(k); return v && *v ? atoi(v) : d; };
    AtEnv e;
    e.gen = geti("DS_ATT_GEN", 0); e.nqb = geti("DS_ATT_NQB", 0); e.late = geti("DS_ATT_LATE", -1); e.order = geti("DS_ATT_ORDER", -1);
    e.tail = geti("DS_ATT_TAIL", 1);
    if (e.gen != 2 && e.gen != 4) e.gen = 0;
    g_at_env = e;
}
static const AtEnv &at_env()
{
    if (g_at_env.gen < 0) at_env_read();
    return g_at_env;
}
DS_API int ds_attention_reload_env(void)
{
    at_env_read();
    return DS_OK;
}

// Where generation 4 is the default: nowhere.  Measured on the MI355X (profiles/round6_attention_gen4.txt, tools/att_ab.sh), f16, ms:
//   (32, 1025, 16, bias) 0.283 generation 2 / 0.325 generation 4;  (8, 2443, 16) 0.282 / 0.281-0.303;  (8, 4097, 16, bias) 0.889 / 0.940;
//   (32, 577, 12) 0.058 / 0.070;  (4, 1370, 16) 0.049 / 0.058.   DS_ATT_GEN=4 selects it (A/B runs, tests).
static bool at4_wanted(int B, int Np, int H, int n_valid, bool with_bias)
{
    (void)B; (void)Np; (void)H; (void)n_valid; (void)with_bias;
    return false;
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
    [[maybe_unused]] const float log2e = 1.4426950408889634f;
    if (at_version() == 2) {               // A fragments of the bias MFMA, in units of 1/scale (head_dim 64: x 8, exact)
        if (dtype == DS_DTYPE_F16) hipLaunchKernelGGL((k_attention_bias_pack2<0>), dim3(blocks), dim3(256), 0, st, bias, (_Float16 *)packed, H, n, Np, 8.0f);
        else hipLaunchKernelGGL((k_attention_bias_pack2<1>), dim3(blocks), dim3(256), 0, st, bias, (__bf16 *)packed, H, n, Np, 8.0f);
    }
#ifdef DS_EXPERIMENTS
    else if (dtype == DS_DTYPE_F16) hipLaunchKernelGGL((k_attention_bias_pack<0>), dim3(blocks), dim3(256), 0, st, bias, (_Float16 *)packed, H, n, Np, log2e);
    else hipLaunchKernelGGL((k_attention_bias_pack<1>), dim3(blocks), dim3(256), 0, st, bias, (__bf16 *)packed, H, n, Np, log2e);
#endif
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

#ifdef DS_EXPERIMENTS
// experiments library only (not declared in include/depthstereo.h): the phase clock of generation 2, see g_att_prof
DS_API int ds_experiments_attention_profile(unsigned long long *out16, int reset)
{
    DS_REQUIRE(out16 != nullptr, DS_EINVAL, "ds_experiments_attention_profile: null argument");
    DS_HIP_CHECK(hipDeviceSynchronize());
    DS_HIP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_att_prof), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[16] = { 0 };
        DS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_att_prof), z, sizeof(z)));
    }
    return DS_OK;
}
#endif

DS_API int ds_attention_fwd(ds_ctx *ctx, const void *qk, const void *vt, const void *bias_packed, void *out,
                            int B, int Np, int H, int n_valid, float scale, int dtype, void *stream)
{
    DS_REQUIRE(ctx && qk && vt && out, DS_EINVAL, "ds_attention_fwd: null argument");
    const void *bias = bias_packed;
    DS_REQUIRE(B > 0 && H > 0 && Np > 0 && (Np % 8) == 0, DS_EINVAL, "ds_attention_fwd: Np must be a positive multiple of 8 (got %d)", Np);
    DS_REQUIRE(at_version() == 2 || (Np % 64) == 0, DS_EINVAL, "ds_attention_fwd: the first kernel generation needs Np to be a multiple of 64");
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
    if (at_version() == 2) {
        DS_REQUIRE(!bias || scale == 0.125f, DS_EUNSUPPORTED, "ds_attention_fwd: the packed bias is stored in units of 1/scale for "
                   "head_dim 64 (scale 0.125); got scale %g", (double)scale);
        DS_REQUIRE(((uintptr_t)out & 15) == 0, DS_EINVAL, "ds_attention_fwd: out must be 16-byte aligned");
        P.c_exp = scale * log2e; P.k_logit = 1.0f; P.flags = 0;
        // rows per wave / 32.  Measured (f16, MI355X): 32 rows x 3 waves per SIMD wins on short sequences (N = 1025 + bias:
        // 0.286 vs 0.310 ms at batch 32; N = 577: 0.060 vs 0.076), 64 rows x 2 waves on long ones (N = 4097 + bias: 0.903 vs
        // 0.944; N = 2443: 0.255 vs 0.263); N = 1370 is a tie.  DS_ATT_NQB overrides (A/B runs).
        // Late fetch (option 4096: K / V^T of the next tile requested after S, 4 waves per SIMD) A/B on one box, 32 rows per
        // wave: N = 577: 0.064 -> 0.060, N = 1370: 0.404 -> 0.387, N = 2443: 0.272 -> 0.254 (vs 0.261 for 64 rows), no bias;
        // with bias it loses (N = 1025: 0.291 -> 0.300) and at N = 4097 64 rows per wave stay ahead (0.911 vs 0.946).
        const AtEnv &E = at_env();
        // Generation 4 (ds_attention4.hip: one wave per SIMD, two 32-query sub-blocks skewed inside the wave, 256 rows per
        // workgroup) where its geometry fills the chip; DS_ATT_GEN=2 / 4 overrides (ds_attention_reload_env: A/B runs and tests).
        // The two generations are bit-identical (same arithmetic per 32-query sub-block, another order of independent operations).
        const bool gen4 = E.gen == 4 || (E.gen == 0 && at4_wanted(B, Np, H, n_valid, bias != nullptr));
        const int nqb = gen4 ? 2 : ((E.nqb == 1 || E.nqb == 2) ? E.nqb : (Np <= (bias ? 1280 : 2560) ? 1 : 2));
        P.nq = (Np + 128 * nqb - 1) / (128 * nqb);
        P.total = P.nq * H * B;
        P.chunk = (P.total + 7) / 8;
        dim3 grid2(8 * P.chunk);
        hipStream_t st2 = (hipStream_t)stream;
        const int late = E.late >= 0 ? E.late : (bias ? 0 : 1);                 // A/B switch DS_ATT_LATE, see option 4096
        // batch-fastest work order when one head's packed bias exceeds an L2 (see the kernel); DS_ATT_ORDER=0/1 overrides
        const int batch_fastest = E.order >= 0 ? E.order : (bias && B > 1 && (size_t)Np * Np * 2 > (size_t)(3u << 20) ? 1 : 0);
#ifdef DS_EXPERIMENTS
        static const int ablate = (getenv("DS_ATT_ABLATE") ? atoi(getenv("DS_ATT_ABLATE")) : 0)    // timing experiments, wrong results
                                  | (getenv("DS_ATT_OPT") ? atoi(getenv("DS_ATT_OPT")) : 0);        // options, correct results
#define A2_ABL(BI_, M_) case M_: hipLaunchKernelGGL((k_attention_fwd2<0, BI_, 2, M_>), grid2, dim3(AT_THREADS), 0, st2, P); break;
#define A2_EXPERIMENT(BF_, BI_)                                                                                         \
            if (ablate && BF_ == 0 && nqb == 2) {                                                                       \
                switch (ablate) {                                                                                       \
                A2_ABL(BI_, 1) A2_ABL(BI_, 2) A2_ABL(BI_, 4) A2_ABL(BI_, 6) A2_ABL(BI_, 7) A2_ABL(BI_, 8) A2_ABL(BI_, 16) A2_ABL(BI_, 32) \
                A2_ABL(BI_, 38) A2_ABL(BI_, 39) A2_ABL(BI_, 512) A2_ABL(BI_, 1024) A2_ABL(BI_, 2048)                     \
                default: ds_set_error("ds_attention_fwd: DS_ATT_ABLATE/OPT=%d is not an instantiated mask", ablate); return DS_EINVAL; \
                }                                                                                                       \
            } else
#else
#define A2_EXPERIMENT(BF_, BI_)
#endif
#ifdef DS_EXPERIMENTS
        static const int gen_env = getenv("DS_ATT_GEN") ? atoi(getenv("DS_ATT_GEN")) : 2;
        static const int prof_env = getenv("DS_ATT_PROF") ? atoi(getenv("DS_ATT_PROF")) : 0;
#define A3_TRY(BF_, BI_) if (gen_env == 3 && nqb == 1) hipLaunchKernelGGL((k_attention_fwd3<BF_, BI_>), grid2, dim3(AT_THREADS), 0, st2, P); \
            else if (prof_env && nqb == 1 && !late && BF_ == 0) hipLaunchKernelGGL((k_attention_fwd2<0, BI_, 1, 256 + 8192>), grid2, dim3(AT_THREADS), 0, st2, P); else
#else
#define A3_TRY(BF_, BI_)
#endif
#define A2_LAUNCH(BF_, BI_) do {                                                                                       \
            A3_TRY(BF_, BI_)                                                                                            \
            A2_EXPERIMENT(BF_, BI_)                                                                                     \
            if (nqb == 1 && late) hipLaunchKernelGGL((k_attention_fwd2<BF_, BI_, 1, 256 + 4096>), grid2, dim3(AT_THREADS), 0, st2, P); \
            else if (nqb == 1) hipLaunchKernelGGL((k_attention_fwd2<BF_, BI_, 1, 256>), grid2, dim3(AT_THREADS), 0, st2, P); \
            else hipLaunchKernelGGL((k_attention_fwd2<BF_, BI_, 2, 256>), grid2, dim3(AT_THREADS), 0, st2, P);          \
        } while (0)
        if (batch_fastest) P.flags |= 2;
        if (E.tail) P.flags |= 4;
        const int kt = ds_kt_begin(ctx, DS_KT_ATTENTION, st2);
        if (gen4) at4_launch(P, dtype == DS_DTYPE_BF16, bias != nullptr, grid2, st2);
        else if (dtype == DS_DTYPE_F16) { if (bias) A2_LAUNCH(0, 1); else A2_LAUNCH(0, 0); }
        else { if (bias) A2_LAUNCH(1, 1); else A2_LAUNCH(1, 0); }
        ds_kt_end(ctx, DS_KT_ATTENTION, kt, st2);
        DS_HIP_CHECK(hipGetLastError());
        return DS_OK;
    }
#ifdef DS_EXPERIMENTS
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
#else
    ds_set_error("ds_attention_fwd: unreachable");
    return DS_EINVAL;
#endif
}
