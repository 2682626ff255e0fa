// Generation 4 of the fused attention forward (round 6): ONE wave per SIMD with the whole 512-register file, two 32-query
// sub-blocks per wave whose tiles are skewed by half a tile INSIDE the wave, so that the matrix pipe works on one sub-block while
// the vector pipe runs the online softmax of the other.  Operands, work order, LDS image, bias operand, deferred maximum and
// every arithmetic operation are those of generation 2 (ds_attention.hip: k_attention_fwd2): the two generations are
// BIT-IDENTICAL on every row the tiled path computes; what differs is the order of independent operations.
//
// Replaces q@k^T (+bias) -> softmax -> @v of dmidas/backbones/beit.py:65-91 and
// ddepth_anything_v2/depth_anything_v2/dinov2_layers/attention.py:49-62.
//
// STATUS: built, correct, measured -- and NOT the default (ds_attention.hip: at4_wanted; DS_ATT_GEN=4 selects it).  At the
// metric's shape (32, 1025, 16 heads, bias) it runs 0.325-0.341 ms against 0.283 for generation 2; profiles/round6_attention_gen4.txt
// holds the ablations that say why: the bare MFMA stream of this schedule (no softmax, no LDS reads, no memory traffic) already
// takes 0.168 ms -- the chip clocks down to ~1.4 GHz under it -- and with one wave per SIMD every instruction beyond ~5 per MFMA
// gap adds serially (softmax +0.077 ms, fragment reads +0.03, bias loads +0.03, stash / barrier +0.045).  A variant with half the
// vector work per logit (logits accumulated in the exp2 domain with -m as the accumulator init, row sums on the matrix pipe) was
// built too and is NOT faster (0.336 / 0.359 ms) while it rounds Q a second time: removed again.
//
// Structure.  A workgroup is 4 waves = 256 query rows with one wave per SIMD (launch bound 1: up to 512 registers, nothing
// spills).  A, B = the two 32-query sub-blocks of a wave; T = number of 64-key tiles; every wave runs, for t = 0 .. T,
//
//     iteration t, phase 1:   matrix  P.V_A(t-1), S_A(t)      beside   vector  softmax_B(t-1)
//                  (between)  rescale O_B if a maximum moved;  stash K(t+1), V^T(t);  request K(t+2), V^T(t+1)
//                  phase 2:   matrix  P.V_B(t-1), S_B(t)      beside   vector  softmax_A(t)
//                  (end)      rescale O_A if a maximum moved;  barrier
//
// In a phase the matrix instructions touch only the registers of one sub-block and the vector instructions only those of the
// other.  The interleave is written BY HAND: a phase is NM steps fenced by sched_barrier(0), step g = MFMA g, the LDS read of
// the fragment four MFMAs ahead, and a slice of the softmax, itself software-pipelined across the steps (a dependent vector
// instruction sits one whole step behind its producer: with one wave per SIMD nothing else covers its latency).  (First
// attempt: one basic block + sched_group_barrier -- the scheduler clustered the MFMAs in front of the vector work.)  An iteration
// reads K(t) and V^T(t-1) only: two LDS slots per operand, one barrier per tile.  The bias fragments of tile t+1 are requested
// right behind the bias MFMAs of tile t, K / V^T one iteration ahead of their stash.  Iterations 0 and T are fill and drain.
#include <utility>

#include "ds_attention.h"

// A4_ABL: timing ablations (WRONG results), only ever set by tools/att_variants.sh, which builds side libraries for the harness:
//   1 no softmax micro-steps   2 no MFMAs   4 fragments are not read from LDS   8 no stash / fetch / barrier in the loop
//   16 no bias loads in the loop
#ifndef A4_ABL
#define A4_ABL 0
#endif
template <int I> using IC = std::integral_constant<int, I>;
template <class F, int... Is>
__device__ __forceinline__ void at4_for(F &&f, std::integer_sequence<int, Is...>) { (f(IC<Is>()), ...); }

template <int BF16, int HAS_BIAS>
__global__ __launch_bounds__(AT_THREADS, 1) void k_attention_fwd4(AttnParams P)
{
    typedef at_traits<BF16> TR;
    typedef typename TR::T T;
    typedef typename TR::V8 V8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * AT2_TILE];       // K[2], V^T[2]: 36,864 B (tail blocks: one row's logits)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    int L = blockIdx.x;
    if (P.chunk > 0) {                                          // XCD-aware order, as in generation 2
        L = (int)(blockIdx.x & 7) * P.chunk + (int)(blockIdx.x >> 3);
        if (L >= P.total) return;
    }
    int qblk, b, h;
    if (P.flags & 2) { b = L % P.B; qblk = (L / P.B) % P.nq; h = L / (P.B * P.nq); }
    else { qblk = L % P.nq; b = (L / P.nq) % P.B; h = L / (P.nq * P.B); }
    const int q0 = qblk * AT2_QB + wave * AT2_QW;
    const int Np = P.Np, H = P.H;
    const int Np64 = (Np + 63) & ~63;
    const size_t tok_stride = (size_t)2 * H * AT_D;
    const T *qk = (const T *)P.qk + (size_t)b * Np * tok_stride;
    const T *q_base = qk + (size_t)h * AT_D;
    const T *k_base = qk + (size_t)(H + h) * AT_D;
    const T *vt = (const T *)P.vt + ((size_t)b * H + h) * AT_D * (size_t)Np;
    T *out_base = (T *)P.out + (size_t)b * Np * (size_t)(H * AT_D) + (size_t)h * AT_D;
    // a block with a handful of live rows: one GEMV per row instead of the tiled path (ds_attention.h: at_tail_rows); its pad rows are zeroed
    {
        const int rows_live = P.n_valid - qblk * AT2_QB;
        if (rows_live > 0 && rows_live <= AT_TAIL_ROWS && Np64 <= AT_TAIL_MAXN && (P.flags & 4)) {
            for (int row = qblk * AT2_QB + rows_live + (tid >> 3); row < min(Np, (qblk + 1) * AT2_QB); row += AT_THREADS / 8)
                *reinterpret_cast<uint4 *>(out_base + (size_t)row * (H * AT_D) + 8 * (tid & 7)) = make_uint4(0, 0, 0, 0);
            at_tail_rows<BF16, HAS_BIAS>(P, smem, b, h, qblk * AT2_QB, rows_live);
            return;
        }
    }
    // a wave whose rows are all padding only helps staging; its output rows are zeroed (generation 2 does the same)
    const bool wave_live = q0 < P.n_valid;
    if (!wave_live && q0 < Np) {
        const int row = q0 + lane;
        if (row < Np) {
            uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 8; c++) *reinterpret_cast<uint4 *>(out_base + (size_t)row * (H * AT_D) + 8 * c) = z;
        }
    }

    V8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
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

    f32x16 o_acc[2][2], s_acc[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; qb++)
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int r = 0; r < 16; r++) { o_acc[qb][d][r] = 0.f; s_acc[qb][d][r] = 0.f; }
    V8 pf[2][2][2];
    float m_run[2] = { -__builtin_inff(), -__builtin_inff() }, l_run[2] = { 0.f, 0.f };
    float alpha[2] = { 1.f, 1.f };
    bool grow[2] = { false, false };

    // ---- staging: every thread moves two 16-byte chunks of K and two of V^T per tile (global -> registers -> LDS) ----
    const int st_row = tid >> 3, st_chunk = tid & 7;
    u32x4 kreg0, kreg1, vreg0, vreg1;
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
        (void *)k_base, 0, (int)(((size_t)Np * tok_stride - (size_t)(H + h) * AT_D) * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void *)vt, 0, (int)((size_t)AT_D * Np * sizeof(T)), 0x00020000);
    const int vo_k = (int)((st_row * tok_stride + 8 * st_chunk) * sizeof(T));
    const int vo_v = (int)((st_row * Np + 8 * st_chunk) * sizeof(T));
    const int so_k32 = (int)(32 * tok_stride * sizeof(T)), so_r32 = (int)(32 * Np * sizeof(T));
#define A4_FETCH_K(kt_) do {                                                                                           \
        const int sk_ = __builtin_amdgcn_readfirstlane((kt_) * AT_KB * (int)(tok_stride * sizeof(T)));                  \
        kreg0 = __builtin_amdgcn_raw_buffer_load_b128(rs_k, vo_k, sk_, 0);                                              \
        kreg1 = __builtin_amdgcn_raw_buffer_load_b128(rs_k, vo_k, sk_ + so_k32, 0);                                     \
    } while (0)
#define A4_FETCH_V(kt_) do {                                                                                           \
        const int sv_ = __builtin_amdgcn_readfirstlane((kt_) * AT_KB * (int)sizeof(T));                                 \
        vreg0 = __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo_v, sv_, 0);                                              \
        vreg1 = __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo_v, sv_ + so_r32, 0);                                     \
    } while (0)
    // K row r, chunk c -> r*144 + 16c.  V^T row d, chunk c (keys 8c .. 8c+7) -> key-permuted as in generation 2: within every
    // 16 keys the order is [0-3, 8-11, 4-7, 12-15], the k-slot order the S^T accumulator hands to the P^T operand
    const int vst_lo = ((st_chunk & ~1) << 4) + ((st_chunk & 1) << 3), vst_hi = vst_lo + 16;
#define A4_STASH_K(buf_) do {                                                                                          \
        *reinterpret_cast<u32x4 *>(smem + (buf_) * AT2_TILE + st_row * AT2_ROW + (st_chunk << 4)) = kreg0;              \
        *reinterpret_cast<u32x4 *>(smem + (buf_) * AT2_TILE + (st_row + 32) * AT2_ROW + (st_chunk << 4)) = kreg1;       \
    } while (0)
#define A4_STASH_V(buf_) do {                                                                                          \
        unsigned char *vd0_ = smem + (2 + (buf_)) * AT2_TILE + st_row * AT2_ROW;                                        \
        unsigned char *vd1_ = vd0_ + 32 * AT2_ROW;                                                                      \
        *reinterpret_cast<uint2 *>(vd0_ + vst_lo) = make_uint2(vreg0.x, vreg0.y);                                       \
        *reinterpret_cast<uint2 *>(vd0_ + vst_hi) = make_uint2(vreg0.z, vreg0.w);                                       \
        *reinterpret_cast<uint2 *>(vd1_ + vst_lo) = make_uint2(vreg1.x, vreg1.y);                                       \
        *reinterpret_cast<uint2 *>(vd1_ + vst_hi) = make_uint2(vreg1.z, vreg1.w);                                       \
    } while (0)
    // Bias operand (ds_attention_bias_pack): [head][32-query block][64-key tile][chunk c = 2 kb + s][64 lanes][8] -- lane
    // (hi, l31) of chunk c holds bias[query 16 s + 8 hi + t][key 32 kb + l31] / scale: the A fragment of the MFMA that adds it.
    // Tiles past the last one read the next query block's first tile or zeros (the descriptor ends with the head): never used.
    u32x4 breg[2][4];
    const int n_kt = Np64 / AT_KB;
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(HAS_BIAS ? (const T *)P.bias + (size_t)h * Np64 * (size_t)Np64 : (const T *)P.qk), 0,
        (int)((size_t)Np64 * Np64 * sizeof(T)), 0x00020000);
    const int vo_b = (int)((((size_t)(q0 / 32) * n_kt) * 2048 + (size_t)lane * 8) * sizeof(T));
    const int so_bq = (int)((size_t)n_kt * 2048 * sizeof(T));                               // next 32-query block
#define A4_FETCH_BIAS(X_, kt_) do {                                                                                    \
        const int sb_ = __builtin_amdgcn_readfirstlane((kt_) * (int)(2048 * sizeof(T)));                                \
        _Pragma("unroll") for (int c_i = 0; c_i < 4; c_i++)                                                             \
            breg[X_][c_i] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, vo_b + (X_) * so_bq + c_i * 1024, sb_, 0);     \
    } while (0)

    const int ntiles = (P.n_valid + AT_KB - 1) / AT_KB;
    const float c_ = P.c_exp;                                   // scale * log2(e): the bias is stored in units of 1/scale
    const float thr_x = AT2_THR / c_;
    const bool pad_keys = (P.n_valid & (AT_KB - 1)) != 0;

    // ---- one phase: the MFMAs of sub-block X beside the softmax of sub-block Y = 1 - X, interleaved by hand ---------------------
    // MFMA order: the bias MFMAs of S_X(t) (no LDS operand: they cover the first fragment reads; the bias registers are free
    // behind them and the next tile's are requested), P.V_X(t-1) in generation 2's (kb, j) order per accumulator, S_X(t).
    auto phase = [&](auto x_tag, const int t, auto first_tag, auto last_tag, auto mask_tag) __attribute__((always_inline)) {
        constexpr int X = decltype(x_tag)::value, Y = 1 - X;
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value, MASK = decltype(mask_tag)::value;
        constexpr bool DO_PV = !FIRST, DO_S = !LAST, DO_SM = (X == 0) ? !FIRST : !LAST;
        constexpr int NB = (DO_S && HAS_BIAS) ? 4 : 0, NPV = DO_PV ? 8 : 0, NS = DO_S ? 8 : 0;
        constexpr int NM = NB + NPV + NS, NF = NPV + NS, PF = 4, NU = 22;
        const unsigned char *s_v = smem + (2 + ((t - 1) & 1)) * AT2_TILE + l31 * AT2_ROW + (hi << 4);
        const unsigned char *s_k = smem + (t & 1) * AT2_TILE + l31 * AT2_ROW + (hi << 4);
        V8 fr[PF];
        auto frag = [&](auto f_tag) __attribute__((always_inline)) {
            constexpr int f = decltype(f_tag)::value;
            if constexpr ((A4_ABL & 4) != 0) {
                if constexpr (f < NF) fr[f % PF] = qf[0][f & 3];
            } else if constexpr (f < NPV) {
                constexpr int kb = f >> 2, j = (f >> 1) & 1, d = f & 1;
                fr[f % PF] = *reinterpret_cast<const V8 *>(s_v + d * 32 * AT2_ROW + ((kb * 4 + j * 2) << 4));
            } else if constexpr (f < NF) {
                constexpr int q = f - NPV, sl = q >> 1, kb = q & 1;
                fr[f % PF] = *reinterpret_cast<const V8 *>(s_k + kb * 32 * AT2_ROW + (sl << 5));
            }
        };
        // The softmax of sub-block Y (generation 2's A2_SOFTMAX, operation for operation) as NU micro-steps:
        //   0-1  four independent chains of four v_maximum3 over eight logits each (the maximum is exact: any order)
        //   2    combine, exchange with lane ^ 32         3  deferred-maximum decision, alpha
        //   4 + e, 5 + e, 6 + e  pair e of probabilities: fma | exp2 | row-sum adds, convert   (the sums run in generation 2's order)
        float mx = 0.f, mc = 0.f, l0 = 0.f, l1 = 0.f, lra = 0.f, mch[4] = { 0.f, 0.f, 0.f, 0.f };
        float xa[16][2], pa[16][2];
        unsigned int sw0 = 0, sw1 = 0;
        auto micro = [&](auto u_tag) __attribute__((always_inline)) {
            constexpr int u = decltype(u_tag)::value;
            if constexpr (u < 2) {                               // chain c runs over s_acc[Y][c >> 1][8 (c & 1) .. + 7]
#pragma unroll
                for (int o = 2 * u; o < 2 * u + 2; o++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const f32x16 &v = s_acc[Y][c >> 1];
                        const int r0 = 8 * (c & 1);
                        if (o == 0) mch[c] = at_max3(v[r0], v[r0 + 1], v[r0 + 2]);
                        else if (o == 3) mch[c] = at_max3(mch[c], v[r0 + 7], mch[c]);
                        else mch[c] = at_max3(mch[c], v[r0 + 2 * o + 1], v[r0 + 2 * o + 2]);
                    }
            } else if constexpr (u == 2) {
                mx = at_max3(mch[0], mch[1], mch[2]);
                mx = at_max3(mx, mch[3], mx);
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                sw0 = sw[0]; sw1 = sw[1];                       // the other 32 keys of the row sit in lane ^ 32
            } else if constexpr (u == 3) {
                mx = at_max3(__uint_as_float(sw0), __uint_as_float(sw1), mx);
                grow[Y] = mx > m_run[Y] + thr_x;
                const float mn = grow[Y] ? mx : m_run[Y];
                alpha[Y] = __builtin_amdgcn_exp2f((m_run[Y] - mn) * c_);
                m_run[Y] = mn;
                mc = -mn * c_;
                lra = l_run[Y] * alpha[Y];
            } else {
                if constexpr (u - 4 >= 0 && u - 4 < 16) {
                    constexpr int e = u - 4, kb = e >> 3, j = (e >> 2) & 1, tt = 2 * (e & 3);
                    xa[e][0] = __builtin_fmaf(s_acc[Y][kb][8 * j + tt], c_, mc);
                    xa[e][1] = __builtin_fmaf(s_acc[Y][kb][8 * j + tt + 1], c_, mc);
                }
                if constexpr (u - 5 >= 0 && u - 5 < 16) {
                    constexpr int e = u - 5;
                    pa[e][0] = __builtin_amdgcn_exp2f(xa[e][0]);
                    pa[e][1] = __builtin_amdgcn_exp2f(xa[e][1]);
                }
                if constexpr (u - 6 >= 0 && u - 6 < 16) {
                    constexpr int e = u - 6, kb = e >> 3, j = (e >> 2) & 1, tt = 2 * (e & 3);
                    pf[Y][kb][j][tt] = TR::from_f32(pa[e][0]);
                    pf[Y][kb][j][tt + 1] = TR::from_f32(pa[e][1]);
                    l0 += pa[e][0]; l1 += pa[e][1];
                }
                if constexpr (u == NU - 1) l_run[Y] = lra + (l0 + l1);
            }
        };
        if (DO_SM && MASK && pad_keys) {
            const int key0 = (X == 0 ? t - 1 : t) * AT_KB;
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (key0 + kb * 32 + at_crow(r, hi) >= P.n_valid) s_acc[Y][kb][r] = -__builtin_inff();
        }
        __builtin_amdgcn_sched_barrier(0);
        at4_for([&](auto f_tag) __attribute__((always_inline)) { frag(f_tag); }, std::make_integer_sequence<int, (NF < PF ? NF : PF)>());
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; r++) z[r] = 0.f;
        at4_for([&](auto g_tag) __attribute__((always_inline)) {
            constexpr int g = decltype(g_tag)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((A4_ABL & 2) != 0) {                  // no MFMAs: the accumulators stay opaque values, the fragments are consumed
                if constexpr (g >= NB) { asm volatile("" :: "v"(fr[(g - NB) % PF])); frag(IC<g - NB + PF>()); }
                if constexpr (g == NM - 1) {
                    asm volatile("" : "+v"(s_acc[X][0]), "+v"(s_acc[X][1]));
                    asm volatile("" : "+v"(o_acc[X][0]), "+v"(o_acc[X][1]));
                }
            } else if constexpr (g < NB) {
                union { u32x4 u; V8 v; } bb;
                bb.u = breg[X][2 * (g & 1) + (g >> 1)];          // g = 0, 1: chunks 0, 2 (x ident[0]); g = 2, 3: chunks 1, 3 (x ident[1])
                s_acc[X][g & 1] = TR::mfma(bb.v, ident[g >> 1], g < 2 ? z : s_acc[X][g & 1]);
                if constexpr (g == NB - 1 && !(A4_ABL & 16)) A4_FETCH_BIAS(X, t + 1);
            } else {
                constexpr int f = g - NB;
                if constexpr (f < NPV) {
                    constexpr int kb = f >> 2, j = (f >> 1) & 1, d = f & 1;
                    o_acc[X][d] = TR::mfma(fr[f % PF], pf[X][kb][j], o_acc[X][d]);
                } else {
                    constexpr int q = f - NPV, sl = q >> 1, kb = q & 1;
                    s_acc[X][kb] = TR::mfma(fr[f % PF], qf[X][sl], (sl == 0 && !HAS_BIAS) ? z : s_acc[X][kb]);
                }
                frag(IC<f + PF>());
            }
            if constexpr (DO_SM && !(A4_ABL & 1)) {
                at4_for([&](auto k_tag) __attribute__((always_inline)) { micro(IC<g * NU / NM + decltype(k_tag)::value>()); },
                        std::make_integer_sequence<int, (g + 1) * NU / NM - g * NU / NM>());
            }
        }, std::make_integer_sequence<int, NM>());
        if constexpr (DO_SM && (A4_ABL & 1) != 0) {             // no softmax: the logits are consumed, the probabilities opaque
            asm volatile("" :: "v"(s_acc[Y][0]), "v"(s_acc[Y][1]));
            asm volatile("" : "+v"(pf[Y][0][0]), "+v"(pf[Y][0][1]), "+v"(pf[Y][1][0]), "+v"(pf[Y][1][1]));
        }
        if constexpr (DO_SM) {
            // the probabilities are consumed a phase later, behind a branch: without this use the compiler sinks their
            // computation past the branch, out of the MFMAs' shadow
            asm volatile("" :: "v"(pf[Y][0][0]), "v"(pf[Y][0][1]), "v"(pf[Y][1][0]), "v"(pf[Y][1][1]), "v"(l_run[Y]), "v"(alpha[Y]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DO_SM) {
            if (__any(grow[Y])) {                               // some query of this sub-block moved its maximum (rare after the first tiles)
#pragma unroll
                for (int d = 0; d < 2; d++)
#pragma unroll
                    for (int r = 0; r < 16; r++) o_acc[Y][d][r] *= alpha[Y];
            }
        }
    };

    // ---- prologue: K(0) into LDS; K(1), V^T(0) and the first bias fragments requested ----
    A4_FETCH_K(0);
    if (HAS_BIAS && wave_live) { A4_FETCH_BIAS(0, 0); A4_FETCH_BIAS(1, 0); }
    A4_STASH_K(0);
    A4_FETCH_K(1);
    A4_FETCH_V(0);
    __syncthreads();

    // one iteration; FIRST: no tile t-1 (fill), LAST: no tile t (drain), MASKA / MASKB: the tile whose softmax runs here may
    // hold pad keys (only the last tile can: the steady-state body has no mask code).  LIVE is a wave-uniform compile-time tag: a
    // wave whose rows are all padding runs a loop of its own that only stages -- with the branch inside one loop the waitcnt
    // pass has to merge both paths at every join and waits for the bias loads in front of the K / V^T stash (vmcnt(3) instead
    // of vmcnt(11)).
    auto iter = [&](const int t, auto live_tag, auto first_tag, auto last_tag, auto maska_tag, auto maskb_tag) __attribute__((always_inline)) {
        constexpr bool LIVE = decltype(live_tag)::value, LAST = decltype(last_tag)::value;
        // ---- phase 1: matrix pipe on sub-block A, vector pipe on the softmax of sub-block B's previous tile ----
        if constexpr (LIVE) phase(IC<0>(), t, first_tag, last_tag, maskb_tag);
        if (!LAST && !(A4_ABL & 8)) {
            // K(t+1) -> the slot K(t-1) left an iteration ago; V^T(t) -> the slot V^T(t-2) left (read for P.V(t-2) in t-1)
            A4_STASH_K((t + 1) & 1);
            A4_STASH_V(t & 1);
            if (t + 1 < ntiles) { A4_FETCH_K(t + 2); A4_FETCH_V(t + 1); }
        }
        // ---- phase 2: matrix pipe on sub-block B, vector pipe on the softmax of sub-block A's current tile ----
        if constexpr (LIVE) phase(IC<1>(), t, first_tag, last_tag, maska_tag);
        if (!LAST && !(A4_ABL & 8)) __syncthreads();            // K(t+1), V^T(t) are in place; K(t), V^T(t-1) are read
    };
    typedef std::true_type Y_;
    typedef std::false_type N_;
    auto run = [&](auto live_tag) __attribute__((always_inline)) {
        if (ntiles == 1) {
            iter(0, live_tag, Y_(), N_(), Y_(), N_());
        } else {
            iter(0, live_tag, Y_(), N_(), N_(), N_());
            for (int t = 1; t + 1 < ntiles; t++) iter(t, live_tag, N_(), N_(), N_(), N_());
            iter(ntiles - 1, live_tag, N_(), N_(), Y_(), N_());
        }
        iter(ntiles, live_tag, N_(), Y_(), N_(), Y_());
    };
    if (wave_live) run(Y_());
    else run(N_());
#undef A4_FETCH_BIAS
#undef A4_STASH_V
#undef A4_STASH_K
#undef A4_FETCH_V
#undef A4_FETCH_K
    if (!wave_live) return;
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
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

// launcher used by ds_attention_fwd (ds_attention.hip)
void at4_launch(const AttnParams &P, int bf16, int has_bias, dim3 grid, hipStream_t st)
{
    if (!bf16) {
        if (has_bias) hipLaunchKernelGGL((k_attention_fwd4<0, 1>), grid, dim3(AT_THREADS), 0, st, P);
        else hipLaunchKernelGGL((k_attention_fwd4<0, 0>), grid, dim3(AT_THREADS), 0, st, P);
    } else {
        if (has_bias) hipLaunchKernelGGL((k_attention_fwd4<1, 1>), grid, dim3(AT_THREADS), 0, st, P);
        else hipLaunchKernelGGL((k_attention_fwd4<1, 0>), grid, dim3(AT_THREADS), 0, st, P);
    }
}
