// ds_linear / ds_conv3x3_nhwc: y = act(x . W^T + b [+ residuals]) as ONE in-tree MFMA kernel with the epilogue fused,
// instead of a library GEMM / convolution followed by element-wise passes.  Two front ends on one K loop:
//   * dense:  the token GEMMs of the ViT encoders.  In the networks it runs `fc1 -> nn.GELU` of every encoder block (timm's
//     Mlp as called from dmidas/backbones/beit.py:93-107; ddepth_anything_v2/depth_anything_v2/dinov2_layers/mlp.py:33-39),
//     the Q/K projection, V^T (ds_linear_vt: written transposed by the epilogue), and the attention / MLP output projections
//     with LayerScale + residual in the epilogue (ds_linear_residual).  Launches whose last round of tiles would be nearly
//     empty hand those tiles to k_linear_ragged (below).
//   * conv:   the 3x3 convolutions of the DPT decoders as an implicit GEMM with bias / ReLU / residual / skip in the epilogue
//     (ResidualConvUnit_custom dmidas/blocks.py:352-377, scratch.layerN_rn :64-80, util/blocks.py:56-85 of Depth-Anything-V2).
//
// Shape of the problem on an MI355X: x is [M, K] (M = batch x padded tokens, 34 816 at the benchmark; pixels for the
// convolution), W is [N, K] (torch Linear layout: both operands K-contiguous, i.e. the "NT" GEMM whose MFMA fragments are
// plain 16-byte reads).
//
// Structure (one workgroup = 256 x 256 tiles of y, 8 waves as 2 (rows) x 4 (columns), wave tile 128 x 64; one PERSISTENT
// workgroup per CU walks a strided list of tiles):
//   * K is walked in tiles of 64.  A K-tile of x and of W is kept in LDS as FOUR half-tiles of 128 rows x 64 k (16 KB):
//     A0/A1 = the first/second 64 rows of every wave-row's 128, B0/B1 = the first/second 32 columns of every
//     wave-column's 64 -- the halves follow the QUADRANTS of the wave tile, so a half-tile is read in exactly one of the
//     four phases of a K-tile and can be re-staged two phases later.  Two K-tiles are resident (128 KB of the 160 KB).
//   * staging is LDS-DMA (`global_load_lds_dwordx4`): no staging registers, no ds_write pass.  The DMA writes
//     lane-linear, so the bank swizzle is applied to the per-lane SOURCE address and undone by the same XOR on the
//     fragment read (16-byte slot ^= (row >> 1) & 7 inside a 128-byte row: every 16-lane group of a ds_read_b128 then
//     covers all 16 slots of the 256-byte bank row; the XOR stays inside one 128-byte line, so global coalescing is intact).
//   * the K loop is 8 phases per two K-tiles.  Each phase = {fragment reads of ONE half-tile, DMA issue of ONE
//     half-tile, counted vmcnt} barrier {8 MFMA 32x32x16 on one quadrant} barrier.  The two wave-rows run staggered
//     by one barrier, so at any time one wave of a SIMD is in its MFMA part while the other one is in its memory part.
//     A half-tile is staged 6 phases before it is read; `s_waitcnt vmcnt(10)` (5 half-tiles stay in flight) never
//     drains the queue.  Hazards (both directions) are argued next to the schedule table below and checked by
//     tools/linear_model.py (tests/test_linear_model.py).
//   * operands go into the MFMA swapped (W fragment as "A", x fragment as "B"), so an accumulator register holds 4
//     consecutive output COLUMNS of one row: the epilogue adds the bias / residuals and applies the activation in fp32 on
//     the accumulator (not on a rounded half); a half-wave exchange (v_permlane32_swap) widens that to 8 columns = one
//     16-byte store per lane.
//
// Workgroup -> tile mapping is XCD-aware: the eight XCDs take contiguous ranges of the tile list, which is ordered so that
// the 32 workgroups resident on one XCD work on 8 row panels x 4 column panels at a time.
#include "ds_common.h"

#include <stdlib.h>
#include <atomic>
#include <type_traits>

typedef _Float16 lf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 lbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 lbf16x4 __attribute__((ext_vector_type(4)));
typedef float lf32x16 __attribute__((ext_vector_type(16)));

#define LN_THREADS 512
#define LN_LDS_BYTES 131072
#define LN_HALF 16384            // one half-tile: 128 rows x 64 k x 2 B
#define LN_B_BASE 65536          // LDS map: A0[2] | A1[2] | B0[2] | B1[2], 16 KB each

template <int BF16> struct ln_traits;
template <> struct ln_traits<0> {
    typedef _Float16 T; typedef lf16x8 V8; typedef lf16x4 V4;
    static __device__ __forceinline__ lf32x16 mfma(V8 a, V8 b, lf32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    // max(v, 0) on the 8 packed halves of a fragment: 4 v_pk_max_f16 (inline asm: the builtin puts a canonicalising v_pk_max in
    // front of every one of them)
    static __device__ __forceinline__ V8 relu(V8 v)
    {
        typedef unsigned lu32x4 __attribute__((ext_vector_type(4)));
        const lu32x4 x = __builtin_bit_cast(lu32x4, v);
        unsigned r0, r1, r2, r3;
        asm("v_pk_max_f16 %0, %1, 0" : "=v"(r0) : "v"(x[0]));
        asm("v_pk_max_f16 %0, %1, 0" : "=v"(r1) : "v"(x[1]));
        asm("v_pk_max_f16 %0, %1, 0" : "=v"(r2) : "v"(x[2]));
        asm("v_pk_max_f16 %0, %1, 0" : "=v"(r3) : "v"(x[3]));
        return __builtin_bit_cast(V8, (lu32x4){r0, r1, r2, r3});
    }
};
template <> struct ln_traits<1> {
    typedef __bf16 T; typedef lbf16x8 V8; typedef lbf16x4 V4;
    static __device__ __forceinline__ lf32x16 mfma(V8 a, V8 b, lf32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    // max(v, 0) on 8 packed bfloat16 (no packed bfloat16 maximum on gfx950): clear every half whose sign bit is set
    static __device__ __forceinline__ V8 relu(V8 v)
    {
        typedef unsigned lu32x4 __attribute__((ext_vector_type(4)));
        lu32x4 x = __builtin_bit_cast(lu32x4, v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned neg = (x[i] >> 15) & 0x00010001u;
            x[i] &= ~(neg * 0xffffu);
        }
        return __builtin_bit_cast(V8, x);
    }
};

struct LinParams {
    const void *x, *w, *bias;
    const void *res1, *res2;    // optional addends of the epilogue, laid out like y
    const void *zeros;          // CONV: >= 128 zero bytes (the padding ring of the image)
    void *y;
    int M, N, K;                // K = reduction length (CONV: 9 * C)
    int nbm, nbn;
    int bnw;                    // columns per tile: 256, or 128 for the NH = 1 instantiations (out_features % 256 == 128)
    int ablate;                 // -DDS_EXPERIMENTS builds only (DS_LIN_ABLATE, results are WRONG): 1 = no epilogue stores, 2 = no K loop
    int stagger;                // -DDS_EXPERIMENTS builds only (DS_LIN_STAGGER_US): every other workgroup of an XCD starts this many 10 ns ticks late
    int H, W, C, cpt, magic;    // CONV: image height / width, input channels, K-tiles per tap (C / 64), 65536 / cpt + 1
    long long ldy;              // row stride of y (and res1 / res2) in elements
    const void *gamma;          // EPI 3: per-column factor applied to (acc + bias) before the addends (LayerScale)
    int early;                  // 1: the next tile's prologue DMAs are issued BEFORE this tile's epilogue (see the tile loop)
    int n_main;                 // the persistent kernel walks positions [0, n_main) of the tile list; k_linear_ragged takes the rest
    int vt_np, vt_c;            // VT: column j of the GEMM is token j % vt_np of batch element j / vt_np, the output is [batch][vt_c rows][vt_np]
    unsigned vt_magic;          // VT: floor(2^32 / vt_np) + 1  (j / vt_np == umulhi(j, vt_magic) for j * vt_np < 2^32)
    // VT 2 (pixel shuffle: ConvTranspose2d with kernel == stride as a GEMM): row m = input pixel (image-major, ps_w pixels per image
    // row), column j = (ky * ps_s + kx) * ps_c + co; the output is NHWC [batch, h * ps_s, ps_w * ps_s, ps_c]
    int ps_w, ps_s, ps_c;
    unsigned ps_w_magic, ps_c_magic;      // floor(2^32 / ps_w) + 1, floor(2^32 / ps_c) + 1
    // VT 3 (read-out): row m = token m % rd_np of image m / rd_np; tokens 1 .. rd_n - 1 go to output row image * (rd_n - 1) + token - 1,
    // the cls token and the pad rows to the DUMMY row rd_dummy behind the output; res1 is one vector per IMAGE ([images, N])
    int rd_np, rd_n, rd_dummy;
    unsigned rd_magic;                   // floor(2^32 / rd_np) + 1
    float *rg_ws;               // ragged round with K split over 2^rg_ksl workgroups per piece: one 32 KB fp32 partial per workgroup ...
    int *rg_cnt;                // ... and one arrival counter per piece (zero between launches)
    int rg_ksl;
    int th_nrb;                 // k_linear_thin: the ragged row panel's new rows as 32-row blocks (the last th_nrb * 32 rows of x)
    int kt_kind;                // host side only: the in-step timer kind of this launch (DS_KT_*)
    // EPI 4 / 5 (LayerNorm folded into the GEMM): x is the UN-normalised residual stream, w holds W . diag(ln_weight); per token
    // {rstd, -mean * rstd} and per output feature colsum = sum_k w[n][k] (of the rounded weights) complete the affine map in the epilogue
    const float2 *ln_stats;     // one pair per token: rows of x (VT: columns of the GEMM)
    const float *ln_colsum;     // one value per output feature: columns of y (VT: rows of the GEMM)
};

// position in the tile list -> origin of the tile.  The list is ordered in groups of 8 row panels, rows fastest inside a group
// (see the kernel); the last row panel is shifted up to end at row M (M >= 256): its first rows are computed twice, identically
__device__ __forceinline__ void ln_tile_origin(const LinParams &P, const int tile, int &bm0, int &bn0)
{
    const int grp8 = tile / (8 * P.nbn), rem8 = tile - grp8 * (8 * P.nbn);
    const int rows8 = min(8, P.nbm - 8 * grp8);
    const int bn = rem8 / rows8, bm = 8 * grp8 + (rem8 - bn * rows8);
    bm0 = min(bm * 256, P.M - 256);
    bn0 = bn * P.bnw;
}

// element offset of output (row, col .. col + 7), col a multiple of 8.  VT: the GEMM computes W_v . h^T with the TOKENS as its
// columns, and the result is stored per batch element as [channels][tokens] -- V transposed, the operand layout of the
// attention kernel's P.V product (a group of 8 columns never straddles two batch elements: vt_np is a multiple of 8)
template <int VT>
__device__ __forceinline__ size_t ln_out_off(const LinParams &P, const int row, const int col)
{
    if (VT == 0) return (size_t)row * P.ldy + col;
    if (VT == 1) {
        const unsigned b = __umulhi((unsigned)col, P.vt_magic);
        const unsigned n = (unsigned)col - b * (unsigned)P.vt_np;
        return ((size_t)b * P.vt_c + row) * (size_t)P.vt_np + n;
    }
    if (VT == 2) {
        // input pixel (q = image * h + y, x) and output tap (ky, kx) of the 8 columns' channel group (ps_c % 8 == 0: a group of 8
        // columns never straddles two taps): output pixel (q * s + ky, x * s + kx) of an image row of ps_w * s pixels
        const unsigned q = __umulhi((unsigned)row, P.ps_w_magic), x = (unsigned)row - q * (unsigned)P.ps_w;
        const unsigned t = __umulhi((unsigned)col, P.ps_c_magic), co = (unsigned)col - t * (unsigned)P.ps_c;
        const unsigned ky = t / (unsigned)P.ps_s, kx = t - ky * (unsigned)P.ps_s;
        const size_t opix = ((size_t)q * P.ps_s + ky) * ((size_t)P.ps_w * P.ps_s) + (size_t)x * P.ps_s + kx;
        return opix * (size_t)P.ps_c + co;
    }
    const unsigned b = __umulhi((unsigned)row, P.rd_magic), t = (unsigned)row - b * (unsigned)P.rd_np;
    const bool live = t >= 1u && t < (unsigned)P.rd_n;
    const size_t orow = live ? (size_t)b * (size_t)(P.rd_n - 1) + (t - 1u) : (size_t)P.rd_dummy;
    return orow * (size_t)P.ldy + col;
}
// element offset of the residual operand's 8 values for output (row, col .. col + 7): laid out like y, except for the read-out
// (VT 3), whose "residual" is one vector per image
template <int VT>
__device__ __forceinline__ size_t ln_res_off(const LinParams &P, const int row, const int col)
{
    if (VT != 3) return ln_out_off<VT>(P, row, col);
    const unsigned b = __umulhi((unsigned)row, P.rd_magic);
    return (size_t)b * (size_t)P.ldy + col;
}

// erf-GELU on the fp32 accumulator.  GELU(v) = v Phi(v) = max(v, 0) - |v| Phi(-|v|), and Phi(-u) = 2^-Q(u) with
// Q(u) = -log2 Phi(-u), a smooth, nearly quadratic function: a degree-7 polynomial on [0, 6] (Chebyshev fit, float64,
// tools/linear_model.py prints the error) gives |GELU - exact| <= 6.6e-7 absolute and <= 5.2e-6 relative for |v| < 5.5;
// beyond u = 6 the argument is clamped (Phi(-6) = 1e-9: |error| < 7e-9).  One transcendental (v_exp_f32) and 10 plain
// VALU operations per element instead of libm erff's ~40 with a divergent branch: the epilogue of a 256 x 256 tile is
// 128 elements per lane and sits on the critical path of a workgroup that owns its CU alone.
__device__ __forceinline__ float ln_gelu(float v)
{
    const float u = fminf(fabsf(v), 6.0f);
    float q = 1.8896206483987044e-06f;
    q = __builtin_fmaf(q, u, -6.268127617659047e-05f);
    q = __builtin_fmaf(q, u, 0.000938866869546473f);
    q = __builtin_fmaf(q, u, -0.008539456874132156f);
    q = __builtin_fmaf(q, u, 0.054020676761865616f);
    q = __builtin_fmaf(q, u, 0.45840978622436523f);
    q = __builtin_fmaf(q, u, 1.1512691974639893f);
    q = __builtin_fmaf(q, u, 0.9999943375587463f);
    const float h = __builtin_amdgcn_exp2f(-q);
    return __builtin_fmaf(-u, h, fmaxf(v, 0.0f));
}
// the same on two values at once: the polynomial and the final multiply-add as packed fp32 operations (v_pk_fma_f32), which
// halves the instruction count of the part of the epilogue that is VALU-bound
typedef float lf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lf32x2 ln_gelu2(lf32x2 v)
{
    const lf32x2 u = {fminf(fabsf(v[0]), 6.0f), fminf(fabsf(v[1]), 6.0f)};
    lf32x2 q = {1.8896206483987044e-06f, 1.8896206483987044e-06f};
#define LN_PK(c) q = __builtin_elementwise_fma(q, u, (lf32x2){c, c})
    LN_PK(-6.268127617659047e-05f); LN_PK(0.000938866869546473f); LN_PK(-0.008539456874132156f); LN_PK(0.054020676761865616f);
    LN_PK(0.45840978622436523f); LN_PK(1.1512691974639893f); LN_PK(0.9999943375587463f);
#undef LN_PK
    const lf32x2 h = {__builtin_amdgcn_exp2f(-q[0]), __builtin_amdgcn_exp2f(-q[1])};
    const lf32x2 r = {fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)};
    return __builtin_elementwise_fma(-u, h, r);
}

// LDS-DMA of 16 bytes per lane (global_load_lds_dwordx4): destination = M0 + 16 * lane.  Written as inline asm so that
// the address is exactly one loop-invariant 32-bit VGPR offset on a per-K-tile SGPR base (hipcc otherwise keeps 64-bit
// per-lane pointers alive across the loop and spills), and so that the compiler's own LDS-DMA bookkeeping (a vmcnt wait in
// front of every later ds_read) stays out of the counted-wait pipeline.
__device__ __forceinline__ void ln_dma_s(const void *base_uniform, unsigned voff, unsigned lds_uniform)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base_uniform), "s"(lds_uniform) : "memory");
}
__device__ __forceinline__ void ln_dma_v(const void *ptr, unsigned lds_uniform)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(ptr), "s"(lds_uniform) : "memory");
}

// Timing experiments that change results (or the start of a workgroup) exist only in -DDS_EXPERIMENTS builds: the shipped
// library has no switch that can produce wrong output.
#ifdef DS_EXPERIMENTS
#define LN_ABLATE(bit) (P.ablate & (bit))
#define LN_STAGGER() (P.stagger)
#else
#define LN_ABLATE(bit) 0
#define LN_STAGGER() 0
#endif
#define LN_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define LN_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define LN_BARRIER()                              \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// EPI: 0 none, 1 erf-GELU, 2 ReLU, 3 per-column scale (LayerScale: y = res1 + gamma * (x.W^T + b)), 4 / 5 = 0 / 1 with the LayerNorm
// in front of the Linear folded in: y = act(rstd[m] * acc + (-mean[m] rstd[m]) * colsum[n] + b[n]) -- LN(x) . W^T = rstd (x . W'^T - mean
// colsum) with W' = W diag(ln_weight), b += W . ln_bias (host), so the GEMM reads the residual stream itself and the LayerNorm pass
// (read x, write h) shrinks to a statistics pass (read x).  RES: number of residual addends (res1, res2).  CONV: 0 = x is a dense [M, K] matrix; 1 = x is an NHWC image [batch, H, W, C] and the
// GEMM is the implicit one of a 3 x 3, stride 1, zero-padded convolution: row m = output pixel, K-tile kt = 64 channels
// kt / 9 of tap kt % 9 (tap-fastest: the pixels of a chunk are fetched once and hit in L2 for the other eight taps), whose
// source is the same 128 bytes of the pixel shifted by (dy, dx) -- or the zero line.
// NH = 1: a tile is 256 rows x 128 columns (the head's 256 -> 128 convolution of the DPT decoders, dmidas/dpt_depth.py:150): the
// wave grid stays 2 x 4, every wave column owns 32 columns = the B0 half alone.  The phase schedule, the LDS map and every
// counted wait stay exactly as they are -- the B1 half-tile is still staged (with B0's source: a dummy that keeps the DMA
// count per phase), only its fragment reads and the MFMAs of the two (.., B1) quadrants are gone, and the epilogue stores 8
// pieces per tile instead of 16 (the early mode's store count follows).  Half the MFMAs for ~2/3 of the phase time.
template <int BF16, int EPI, int CONV, int RES, int VT = 0, int NH = 0>
__global__ __launch_bounds__(LN_THREADS) void k_linear256(LinParams P)
{
    typedef ln_traits<BF16> TR;
    typedef typename TR::T T;
    typedef typename TR::V8 V8;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;

    const int nwg = P.n_main;            // this kernel's share of the tile list (all of it unless a ragged round follows)
    const int K = P.K;
    const int rowbytes = (CONV ? P.C : K) * (int)sizeof(T);                 // bytes of one row of x (one pixel for CONV)

    // ---- staging: this thread's two 16-byte pieces of a half-tile (chunk c = 2*wave + i = LDS rows 8c .. 8c+7) -------
    // LDS row j = 8c + (lane >> 3), LDS slot = lane & 7 holds SOURCE slot (lane & 7) ^ ((j >> 1) & 7).
    unsigned srcA[2], srcB[2];
    int rowA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = 2 * wid + i;
        const int j = 8 * c + (lane >> 3);
        const int slot = (lane & 7) ^ ((j >> 1) & 7);
        rowA[i] = (j >> 6) * 128 + (j & 63);                                // A half h (+ 64 h rows): wave-row j>>6, row j&63 of its 64
        srcA[i] = (unsigned)rowA[i] * (unsigned)rowbytes + slot * 16;
        const int col = NH ? j : (j >> 5) * 64 + (j & 31);                   // B half h adds 32 columns (uniform); NH: B0 = all 128 columns
        srcB[i] = (unsigned)col * (unsigned)K * (unsigned)sizeof(T) + slot * 16;
    }
    const unsigned a_half = 64u * (unsigned)rowbytes;
    const unsigned b_half = NH ? 0u : 32u * (unsigned)K * (unsigned)sizeof(T);      // NH: the B1 slot is staged with B0's rows (unused)
    const unsigned lds_stage = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds + (unsigned)(2 * wid) * 1024u;   // LDS address of chunk 2*wave

    // ---- the workgroup is persistent: it walks tiles orig = blockIdx.x, + gridDim.x, ... of the XCD-aware tile list (XCD x
    // takes tiles [start(x), start(x+1)) of the column-fastest list; gridDim.x is a multiple of 8, so a workgroup stays on
    // its XCD's range).  Per tile: origin, operand bases and (CONV) the border bits of this thread's four pixels.
    int bm0 = 0, bn0 = 0;
    const unsigned char *xb = nullptr, *wb = nullptr;
    unsigned okA = 0;       // CONV: bit 6*(2h+i) + t (t = 0..2): row y-1+t of this piece's pixel is inside the image; + 3 + t: column x-1+t
    auto set_tile = [&](const int orig) {
        const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
        const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
        // the tile list is ordered in groups of 8 row panels, rows fastest inside a group: the 32 tiles an XCD has in flight
        // are then 8 row panels x 4 column panels (12 operand panels per K-slice through that L2 instead of the 18 of a
        // column-fastest list)
        // (measured against chunks of 4 column panels outermost, which would keep W in L2: fc1 316-319 vs 328-331 us)
        ln_tile_origin(P, tile, bm0, bn0);
        xb = (const unsigned char *)P.x + (size_t)bm0 * rowbytes;
        wb = (const unsigned char *)P.w + (size_t)bn0 * K * sizeof(T);
        if (CONV) {
            okA = 0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned pix = (unsigned)(bm0 + rowA[i] + 64 * h) % (unsigned)(P.H * P.W);
                    const int py = (int)(pix / (unsigned)P.W), px = (int)(pix % (unsigned)P.W);
                    const unsigned bits = (py > 0 ? 1u : 0u) | 2u | (py < P.H - 1 ? 4u : 0u) | (px > 0 ? 8u : 0u) | 16u | (px < P.W - 1 ? 32u : 0u);
                    okA |= bits << (6 * (2 * h + i));
                }
        }
    };

    // kind: 0 A0, 1 A1, 2 B0, 3 B1; kt = K-tile; s = LDS buffer
#define LN_STAGE(kind, kt, s)                                                                                            \
    do {                                                                                                                 \
        unsigned koff_ = (unsigned)(kt) * 128u;                                                                          \
        const unsigned dst_ = ((kind) >> 1) * LN_B_BASE + ((kind) & 1) * 2 * LN_HALF + (s) * LN_HALF + lds_stage;  /* LDS address */ \
        int dy_ = 0, dx_ = 0, aoff_ = (int)koff_;                                                                        \
        if (CONV) {              /* K-tile kt = tap kt % 9 of channel chunk kt / 9: the nine taps of one 64-channel chunk */ \
            const int cc_ = ((kt) * 7282) >> 16, tap_ = (kt) - 9 * cc_;      /* follow each other, their pixels stay in L2 */ \
            dy_ = ((tap_ * 11) >> 5) - 1; dx_ = tap_ - 3 * (dy_ + 1) - 1;                                                \
            aoff_ = cc_ * 128 + (dy_ * P.W + dx_) * rowbytes;                                                            \
            koff_ = (unsigned)(tap_ * P.cpt + cc_) * 128u;               /* the weights stay [out][tap][in] */             \
        }                                                                                                                \
        /* uniform 64-bit base of this K-tile (scalar arithmetic) + the thread's loop-invariant 32-bit offset */           \
        const unsigned char *ub_ = ((kind) < 2) ? xb + ((long)aoff_ + (long)(((kind) & 1) * a_half))                     \
                                                : wb + (size_t)(((kind) & 1) * b_half + koff_);                          \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                               \
            if (CONV && (kind) < 2) {                                                                                    \
                const unsigned sh_ = 6 * (2 * ((kind) & 1) + i_);                                                        \
                const unsigned ok_ = (okA >> (sh_ + dy_ + 1)) & (okA >> (sh_ + 4 + dx_)) & 1u;                           \
                const unsigned char *g_ = ok_ ? ub_ + (size_t)srcA[i_] : (const unsigned char *)P.zeros;                 \
                ln_dma_v(g_, dst_ + i_ * 1024);                                                                          \
            } else {                                                                                                     \
                ln_dma_s(ub_, ((kind) < 2) ? srcA[i_] : srcB[i_], dst_ + i_ * 1024);                                     \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)

    // ---- fragment reads: lane reads LDS row (lane & 31) of its 32-row block, slot 2*ks + (lane >> 5), swizzled -----------
    unsigned offA[4], offB[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const unsigned sl = (unsigned)((2 * ks + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4;
        offA[ks] = (unsigned)(wr * 64 + (lane & 31)) * 128u + sl;
        offB[ks] = LN_B_BASE + (unsigned)(wc * 32 + (lane & 31)) * 128u + sl;
    }
    V8 bv[2][2];         // epilogue: bias of this wave's columns, [W half][k]: 8 columns each, kept packed
    // loaded BEFORE the last iteration of a tile (in flight under its MFMAs).  Always exactly 4 loads -- a null bias reads 16
    // valid bytes of W instead and is zeroed afterwards -- because the last iteration's counted waits include them.  (The
    // LayerScale vector of EPI 3 is loaded in the epilogue next to the residuals: that variant waits there anyway.)
#define LN_LOAD_BIAS()                                                                                                   \
    do {                                                                                                                 \
        const T *bias_ = (const T *)P.bias;                                                                              \
        _Pragma("unroll") for (int hb_ = 0; hb_ < 2; ++hb_) _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_) {           \
            const int c_ = bn0 + (NH ? wc * 32 : wc * 64 + hb_ * 32) + 8 * (lane >> 5) + 16 * k_;                        \
            bv[hb_][k_] = *(const V8 *)(bias_ ? bias_ + c_ : (const T *)P.w);                                            \
        }                                                                                                                \
    } while (0)
#define LN_ZERO_NULL_BIAS()                                                                                              \
    do {                                                                                                                 \
        if (!P.bias) {                                                                                                   \
            _Pragma("unroll") for (int hb_ = 0; hb_ < 2; ++hb_) _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_)         \
                _Pragma("unroll") for (int t_ = 0; t_ < 8; ++t_) bv[hb_][k_][t_] = (T)0.f;                               \
        }                                                                                                                \
    } while (0)
    V8 fa[2][2][4];      // [half][row block][k step]   x fragments
    V8 fb[2][4];         // [half][k step]              W fragments
    lf32x16 acc[2][2][2];  // [x half][row block][W half]

#define LN_READ_A(h, s)                                                                                                  \
    _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_) _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)              \
        fa[h][rb_][ks_] = *(const V8 *)(lds + offA[ks_] + ((h) * 2 * LN_HALF + (s) * LN_HALF + rb_ * 4096))
    // CONV 2: ReLU on the x operand (the `conv1(relu(x))` of a residual unit, dmidas/blocks.py:361-363): applied to the fragments
    // in the MEMORY part of the phase that read them -- 32 packed maxima per wave, issued as the reads return, beside the other
    // wave-row's MFMAs -- instead of a pass over the activation in front of the launch
#define LN_RELU_A(h)                                                                                                     \
    do {                                                                                                                 \
        if constexpr (CONV == 2) {                                                                                       \
            _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_) _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)      \
                fa[h][rb_][ks_] = TR::relu(fa[h][rb_][ks_]);                                                             \
        }                                                                                                                \
    } while (0)
#define LN_READ_B(h, s)                                                                                                  \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                                                  \
        if (!(NH && (h) == 1)) fb[h][ks_] = *(const V8 *)(lds + offB[ks_] + ((h) * 2 * LN_HALF + (s) * LN_HALF))
#define LN_MMA_PART(ha, hb, k0, k1)                                                                                      \
    do {                                                                                                                 \
        if (!(NH && (hb) == 1))                                                                                          \
        _Pragma("unroll") for (int ks_ = (k0); ks_ < (k1); ++ks_) _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_)    \
            acc[ha][rb_][hb] = TR::mfma(fb[hb][ks_], fa[ha][rb_][ks_], acc[ha][rb_][hb]);                                \
    } while (0)

    // ---- schedule ------------------------------------------------------------------------------------------------------
    // Iteration i covers K-tiles E = 2i (LDS buffer 0) and O = 2i+1 (buffer 1); phase j = 0..7, quadrant j & 3:
    //   phase  fragment read (this phase's MFMAs need it)    DMA issued                    MFMA quadrant
    //     0    B0(E)                                          A1(O)      -> buf 1           (A0,B0) of E
    //     1    B1(E)                                          A0(E+2)    -> buf 0           (A0,B1)
    //     2    A1(E)                                          B0(E+2)    -> buf 0           (A1,B1)
    //     3    A0(O)   [one phase early: A0 regs are free]    B1(E+2)    -> buf 0           (A1,B0)
    //     4    B0(O)                                          A1(E+2)    -> buf 0           (A0,B0) of O
    //     5    B1(O)                                          A0(O+2)    -> buf 1           (A0,B1)
    //     6    A1(O)                                          B0(O+2)    -> buf 1           (A1,B1)
    //     7    A0(E+2)                                        B1(O+2)    -> buf 1           (A1,B0)
    // Wave-row 1 runs one barrier behind wave-row 0: with barriers b0, b1, ... wave-row 0 has the memory part of
    // phase p in (b[2p-1], b[2p]) and its MFMAs in (b[2p], b[2p+1]); wave-row 1 has them in (b[2p], b[2p+1]) and
    // (b[2p+1], b[2p+2]).
    //   write-after-read: every half-tile is re-staged exactly 2 phases after the phase that read it.  The reads of
    //     phase p are retired (lgkmcnt(0)) by wave-row 0 before b[2p+1] and by wave-row 1 before b[2p+2]; the first DMA
    //     of phase p+2 is issued after b[2p+3].
    //   read-after-write: a half-tile staged in phase s is read in phase s+6.  Every wave ends the memory part of phase
    //     w with vmcnt(10) after having issued phase w's two loads, i.e. its loads of phases <= w-5 have landed; both
    //     wave-rows have done so before b[2w+1], and every read of phase w+1 is issued after b[2w+1].
    const int nt = K >> 6, ni = nt >> 1;

    // the first 7 half-tiles of a tile (everything phases 0..5 read, and A0 of K-tile 0 for the read ahead of phase 0)
#define LN_PROLOGUE()                                                                                                    \
    do {                                                                                                                 \
        LN_STAGE(0, 0, 0); LN_STAGE(2, 0, 0); LN_STAGE(3, 0, 0); LN_STAGE(1, 0, 0);                                      \
        LN_STAGE(0, 1, 1); LN_STAGE(2, 1, 1); LN_STAGE(3, 1, 1);                                                         \
    } while (0)

    // One phase: memory part | barrier | MFMAs | barrier (wave-row 1 one barrier behind).  A variant with ONE barrier per
    // phase (wave-row 0 issues its MFMAs before the barrier, wave-row 1 after it, so the hand-over needs no meeting point)
    // was built and measured: 1.88 us per K-tile against 1.65 us for this one -- the strict alternation wins.
    // Measured and dropped (same box, K = 1024 / 4096 / 8192, us per round of 256 tiles): issuing the last 2 of a phase's 8
    // MFMAs after its closing barrier 43.1 / 122.3 / 238.8 against 39.9 / 113.1 / 216.9 for this form; DMA ahead of the
    // fragment reads and no s_setprio: no difference.
#define LN_PHASE_END(ha, hb)                                                                                             \
    do {                                                                                                                 \
        LN_BARRIER(); LN_WAIT_LGKM0();                                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                                   \
        LN_MMA_PART(ha, hb, 0, 4);                                                                                       \
        __builtin_amdgcn_s_setprio(0);                                                                                   \
        LN_BARRIER();                                                                                                    \
    } while (0)
#define LN_MEM(READ, STAGE)                                                                                              \
    do { READ; STAGE; } while (0)

    // KIND 0: an iteration in the middle of a tile; 1: the last one (nothing left to stage, the queue is drained 10, 8, .. 0);
    // 2: the FIRST iteration of a tile in early mode.  There the queue of this wave holds, oldest first, the 14 prologue DMAs
    // (issued before the previous tile's epilogue), that epilogue's 16 stores, then this iteration's DMAs: "everything but
    // the last 5 half-tiles has landed" is vmcnt(10 + 16) as long as the awaited DMA is OLDER than the stores -- phases 0..4
    // wait for prologue half-tiles -- and the usual vmcnt(10) from phase 5 on (which then also waits for the stores to
    // retire: by that time they have had five phases of MFMA work to drain under).  The counter is in order, so a count
    // that is too high would under-wait: the 16 is exact (one 16-byte store per (ha, rb, hb, k) in every epilogue variant;
    // the residual loads of an epilogue are consumed, i.e. retired, before its stores are issued).
    auto iteration = [&](const int i, auto kind_c) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr bool LAST = KIND == 1;
        const int e2 = 2 * i + 2, o2 = 2 * i + 3;
        // (KIND 1: the 4 bias loads issued just before this iteration are YOUNGER than every DMA its phases
        // 0..4 wait for, so they add to the count; phase 5 waits for everything)
#define LN_WAIT_HEAD()                                                                                                   \
        do {                                                                                                             \
            if constexpr (KIND == 2) { if constexpr (NH) LN_WAIT_VM(18); else LN_WAIT_VM(26); }                          \
            else if constexpr (KIND == 1) LN_WAIT_VM(14);                                                                \
            else LN_WAIT_VM(10);                                                                                         \
        } while (0)
        // phase 0
        LN_MEM(LN_READ_B(0, 0), LN_STAGE(1, 2 * i + 1, 1)); LN_WAIT_HEAD();
        LN_PHASE_END(0, 0);
        // phase 1
        if constexpr (!LAST) { LN_MEM(LN_READ_B(1, 0), LN_STAGE(0, e2, 0)); LN_WAIT_HEAD(); } else { LN_READ_B(1, 0); LN_WAIT_VM(12); }
        LN_PHASE_END(0, 1);
        // phase 2
        if constexpr (!LAST) { LN_MEM(LN_READ_A(1, 0), LN_STAGE(2, e2, 0)); LN_RELU_A(1); LN_WAIT_HEAD(); } else { LN_READ_A(1, 0); LN_RELU_A(1); LN_WAIT_VM(10); }
        LN_PHASE_END(1, 1);
        // phase 3
        if constexpr (!LAST) { LN_MEM(LN_READ_A(0, 1), LN_STAGE(3, e2, 0)); LN_RELU_A(0); LN_WAIT_HEAD(); } else { LN_READ_A(0, 1); LN_RELU_A(0); LN_WAIT_VM(8); }
        LN_PHASE_END(1, 0);
        // phase 4
        if constexpr (!LAST) { LN_MEM(LN_READ_B(0, 1), LN_STAGE(1, e2, 0)); LN_WAIT_HEAD(); } else { LN_READ_B(0, 1); LN_WAIT_VM(6); }
        LN_PHASE_END(0, 0);
#undef LN_WAIT_HEAD
        // phase 5
        if constexpr (!LAST) { LN_MEM(LN_READ_B(1, 1), LN_STAGE(0, o2, 1)); LN_WAIT_VM(10); } else { LN_READ_B(1, 1); LN_WAIT_VM(0); }
        LN_PHASE_END(0, 1);
        // phase 6
        if constexpr (!LAST) { LN_MEM(LN_READ_A(1, 1), LN_STAGE(2, o2, 1)); LN_RELU_A(1); LN_WAIT_VM(10); } else { LN_READ_A(1, 1); LN_RELU_A(1); }
        LN_PHASE_END(1, 1);
        // phase 7
        if constexpr (!LAST) { LN_MEM(LN_READ_A(0, 0), LN_STAGE(3, o2, 1)); LN_RELU_A(0); LN_WAIT_VM(10); }
        LN_PHASE_END(1, 0);
    };
    if (LN_STAGGER() > 0 && ((blockIdx.x >> 3) & 1)) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)LN_STAGGER()) __builtin_amdgcn_s_sleep(32);
    }
    // ---- the tile loop.  Early mode (P.early, the default): the NEXT tile's 14 prologue DMAs are issued right after this
    // tile's last barrier, BEFORE its epilogue -- their latency runs under the epilogue's arithmetic, and the epilogue's 16
    // stores drain under the first phases of the next tile instead of in front of them (iteration KIND 2 counts them).  The
    // bias / LayerScale vectors of the epilogue are loaded before the last iteration and waited for before the DMAs are
    // issued: a compiler-generated wait for a load issued AFTER the DMAs would wait for the DMAs too (the counter is in
    // order and the compiler does not see inline-asm loads).  The residual loads of the RES variants stay where they were:
    // their wait covers the DMAs, which have had the bias arithmetic of the first rows to land under.
    // Late mode (DS_LIN_EARLY=0, round 2's order): prologue after the epilogue's stores.
    const bool early = P.early != 0;
    bool stores_ahead = false;
    set_tile(blockIdx.x);
    LN_PROLOGUE();
    for (int orig = blockIdx.x;;) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;
    // Late mode: in the queue ahead of the prologue's 14 loads is nothing (first tile) or the previous tile's epilogue stores,
    // which retire first (the counter is in order): "all but the last 10" covers A0 and B0 of K-tile 0 either way.
    // Early mode: [14 DMAs, 16 stores]: A0 and B0 of K-tile 0 are the 4 oldest of 30; the first tile of a workgroup has no
    // stores behind its DMAs and simply waits for all of them (iteration KIND 2's counts then never under-wait).
    if (!early) LN_WAIT_VM(10);
    else if (stores_ahead) { if constexpr (NH) LN_WAIT_VM(18); else LN_WAIT_VM(26); }
    else LN_WAIT_VM(0);
    LN_BARRIER();
    LN_READ_A(0, 0);
    LN_WAIT_LGKM0();                      // retired here: A0 of buffer 0 is re-staged in phase 1
    LN_RELU_A(0);
    if (wr == 1) LN_BARRIER();            // the stagger
    if (LN_ABLATE(2)) {
        LN_LOAD_BIAS();
        LN_WAIT_VM(0);
        if (wr == 0) LN_BARRIER();
    } else {
        int i0 = 0;
        if (early && ni >= 2) { iteration(0, std::integral_constant<int, 2>()); i0 = 1; }
        for (int i = i0; i < ni - 1; ++i) iteration(i, std::integral_constant<int, 0>());
        LN_LOAD_BIAS();
        iteration(ni - 1, std::integral_constant<int, 1>());
        if (wr == 0) LN_BARRIER();                         // wave-row 0 arrives at wave-row 1's last barrier
    }
    // every fragment read of this tile was retired before that barrier: LDS is free for the next tile's prologue
    const int cbm0 = bm0, cbn0 = bn0;
    const int next = orig + (int)gridDim.x;
    const int hi8 = 8 * (lane >> 5);
    LN_ZERO_NULL_BIAS();
    T *yb = (T *)P.y;
    const T *r1 = (const T *)P.res1, *r2 = (const T *)P.res2;
    if (early && next < nwg) {
        // the bias vectors have been in flight since before the last iteration: waiting for them HERE (the empty
        // asm statement uses them) keeps every compiler-generated wait for them in front of the DMAs
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                asm volatile("" ::"v"(bv[hb][k]));
            }
        set_tile(next);
        LN_PROLOGUE();
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------------
    // Register r of a 32 x 32 accumulator block = column (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of row lane & 31: a lane holds
    // 4-column groups g = r >> 2, the other half-wave holds the groups in between.  One v_permlane32_swap per register pair
    // (g = 2k, 2k+1) exchanges them so that lanes 0-31 end up with columns 16k .. 16k+7 and lanes 32-63 with 16k+8 .. 16k+15
    // of their row, in fp32: bias, residuals (16-byte loads, all issued before the arithmetic) and the activation are applied
    // on 8 consecutive columns and the result leaves as one 16-byte store per lane (32 bytes of a row per instruction).
    // (Round 2 sent the plain GEMM's results through LDS to store full 128-byte lines: qk 160 -> 152 us.  In early mode that
    // LDS already holds the next tile's operands, and the variant is gone.)
    V8 gv[2][2];                                                // EPI 3: the per-column LayerScale factors, packed like the bias
    if (EPI == 3) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int k = 0; k < 2; ++k) gv[hb][k] = *(const V8 *)((const T *)P.gamma + cbn0 + (NH ? wc * 32 : wc * 64 + hb * 32) + hi8 + 16 * k);
    }
    constexpr int NHB = NH ? 1 : 2;                             // W halves with results (NH: the B0 half alone)
    const int wcol = NH ? wc * 32 : wc * 64;
#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int rl = ha * 64 + rb * 32 + (lane & 31);     // row inside the wave tile
            const size_t o0 = (size_t)(cbm0 + wr * 128 + rl) * P.ldy + cbn0 + wcol + hi8;
            // VT: where the 8 columns of piece (hb, k) go (per batch element [channels][tokens]); else o0 + hb * 32 + 16 * k
            auto out_off = [&](const int hb_, const int k_) -> size_t {
                return VT ? ln_out_off<VT>(P, cbm0 + wr * 128 + rl, cbn0 + wcol + hi8 + hb_ * 32 + 16 * k_) : o0 + hb_ * 32 + 16 * k_;
            };
            V8 ra[2][2], rb2[2][2];                              // residual pieces [W half][k]
            constexpr bool LNF = EPI == 4 || EPI == 5;
            float2 st_row = {1.f, 0.f};                          // LNF, row-major output: {rstd, -mean rstd} of this lane's token
            float cs_row = 0.f;                                  // LNF, VT: colsum of this lane's output channel
            static_assert(!LNF || VT == 0 || VT == 1, "the folded LayerNorm exists for the row-major and the V^T store only");
            if (LNF && VT == 0) st_row = P.ln_stats[cbm0 + wr * 128 + rl];
            if (LNF && VT == 1) cs_row = P.ln_colsum[cbm0 + wr * 128 + rl];
#pragma unroll
            for (int hb = 0; hb < NHB; ++hb)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (RES >= 1) ra[hb][k] = *(const V8 *)(r1 + (VT == 3 ? ln_res_off<VT>(P, cbm0 + wr * 128 + rl, cbn0 + wcol + hi8 + hb * 32 + 16 * k) : o0 + hb * 32 + 16 * k));
                    if (RES >= 2) rb2[hb][k] = *(const V8 *)(r2 + o0 + hb * 32 + 16 * k);
                }
#pragma unroll
            for (int hb = 0; hb < NHB; ++hb)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float v[8];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        // (copies first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this clang)
                        const float fa = acc[ha][rb][hb][8 * k + t], fb = acc[ha][rb][hb][8 * k + 4 + t];
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(fa), __float_as_uint(fb), false, false);
                        v[t] = __uint_as_float(sw[0]);
                        v[4 + t] = __uint_as_float(sw[1]);
                    }
                    V8 o;
                    if (LNF) {                                   // fold the LayerNorm back in (fp32, before bias and activation)
                        const int c0 = cbn0 + wcol + hi8 + hb * 32 + 16 * k;
                        if (!VT) {
                            const float4 s0 = *(const float4 *)(P.ln_colsum + c0), s1 = *(const float4 *)(P.ln_colsum + c0 + 4);
                            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                            for (int t = 0; t < 8; ++t) v[t] = __builtin_fmaf(v[t], st_row.x, st_row.y * sc[t]);
                        } else {                                 // VT: the 8 columns are 8 tokens, the row is one channel
                            const float4 *sp = (const float4 *)(P.ln_stats + c0);
                            const float4 q0 = sp[0], q1 = sp[1], q2 = sp[2], q3 = sp[3];
                            const float rs[8] = {q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, q3.x, q3.z};
                            const float nm[8] = {q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, q3.y, q3.w};
#pragma unroll
                            for (int t = 0; t < 8; ++t) v[t] = __builtin_fmaf(v[t], rs[t], nm[t] * cs_row);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 8; t += 2) {
                        lf32x2 u = {v[t] + (float)bv[hb][k][t], v[t + 1] + (float)bv[hb][k][t + 1]};
                        if (EPI == 3) u *= (lf32x2){(float)gv[hb][k][t], (float)gv[hb][k][t + 1]};
                        if (RES >= 1) u += (lf32x2){(float)ra[hb][k][t], (float)ra[hb][k][t + 1]};
                        if (RES >= 2) u += (lf32x2){(float)rb2[hb][k][t], (float)rb2[hb][k][t + 1]};
                        if (EPI == 1 || EPI == 5) u = ln_gelu2(u);
                        if (EPI == 2) u = (lf32x2){fmaxf(u[0], 0.f), fmaxf(u[1], 0.f)};
                        o[t] = (T)u[0];
                        o[t + 1] = (T)u[1];
                    }
                    if (!LN_ABLATE(1)) *(V8 *)(yb + out_off(hb, k)) = o;
                    else asm volatile("" ::"v"(o));
                }
        }
    if (next >= nwg) break;
    orig = next;
    if (!early) {
        set_tile(orig);
        LN_PROLOGUE();
    }
    stores_ahead = true;
    }
}



// (Round 4 built and measured a second K-loop generation here, k_linear256s: every wave runs ONE software-pipelined stream --
// fragment reads one k-step ahead in two register sets, the last step's MFMAs issued behind the next barrier, the DMA stream
// continuing across output tiles, ONE barrier per K-tile instead of sixteen, 192-245 VGPRs without a spill.  Bit-identical to
// k_linear256 on first run, and not faster: 30.7 vs 30.9 us per round of 256 tiles at K = 1024, 115 vs 106 us at K = 4096
// (profiles/round4_gemm_streamed_experiment.txt).  The ablations taken with it say why neither schedule matters at this
// point: on ZERO operands this kernel runs 1870 TF/s against 1296 on random ones, and on half the chip (128 workgroups) 1958
// TF/s-equivalent per CU -- the full chip on random float16 data is POWER limited (the clock drops), at about 1600 TF/s for the
// MFMA stream alone (no DMA, no fragment reads) and 1300 with its operand traffic.  The kernel is in the history: commit
// "k_linear256s: streamed K loop".)

// ---- the ragged round ----------------------------------------------------------------------------------------------------
// A persistent launch of T tiles on G workgroups costs ceil(T / G) rounds; the token GEMMs of an encoder block at batch 32 are
// 544 tiles (N = 1024: 2.125 -> 3 rounds) and 1088 tiles (N = 2048: 4.25 -> 5).  The last, mostly empty round is replaced: the
// persistent kernel walks the first T - R positions of the tile list (a whole number of rounds), and this kernel renders the R
// left-over tiles as 8 R pieces of 128 x 64 outputs -- one piece per workgroup, two workgroups per CU, so the whole chip works
// on them at once.  A piece is a small LDS-staged GEMM of its own: 8 waves as 4 x 2 blocks of 32 x 32 (one MFMA accumulator
// each), K in tiles of 64 staged by LDS-DMA into a ring of S 24 KB slots (x: 128 rows, W: 64 rows, 128 bytes each, the
// swizzle of k_linear256: applied to the source address, undone on the fragment read), one barrier per K-tile:
//     wait for K-tile t (counted: t + 1 .. t + S - 2 stay in flight) | barrier | stage K-tile t + S - 1 into the slot of t - 1 | fragments | 4 MFMAs
//   read-after-write: every wave has waited for ITS pieces of K-tile t before the barrier, the reads come after it;
//   write-after-read: the slot of K-tile t - 1 is re-staged after the barrier of iteration t, which every wave reaches only
//     after its MFMAs of iteration t - 1 have consumed its fragments of that slot.
// (Round 3's first version split K over the 8 waves and fed every wave from global memory with fragment-shaped loads --
// 32-byte pieces of 32 different lines per instruction: 29 us per piece at K = 1024 and 88 us at K = 4096, slower than the
// round it replaced; profiles/round3_microbench_gemms_v1.txt.)  No inter-workgroup communication, bit-reproducible.
// S = slots of the ring: 3 (72 KB: two workgroups per CU, for launches with more pieces than CUs) or 6 (144 KB, one
// workgroup per CU with five K-tiles in flight: a piece is bound by the latency of its DMA chain, not by its 4 MFMAs per K-tile)
#define RG_SLOT 24576            // one K-tile in LDS: x rows 0..127 (16 KB) | W rows 0..63 (8 KB)
// PIPE = 1 (deep ring only): the fragments of K-tile kt + 1 are requested BEFORE the MFMAs of K-tile kt (two register sets,
// the loop unrolled by two), and the epilogue's operands (bias, LayerScale factor, residual rows) are requested before the
// first DMA -- the chain "barrier -> fragment reads -> dependent MFMAs" of a K-tile loses its middle link.  Bit-identical to the
// plain loop.  Measured: -6 us over a block's five GEMMs in the microbenchmark, nothing in the step (816.5 / 815.5 / 815.0 pairs/s
// with the switch 1 / 0 / 1 on one box): the pieces are bound by their fixed cost (launch, first DMA round trip, epilogue), not by
// the loop, once the long K ranges are split (ln_launch).
template <int BF16, int EPI, int RES, int VT, int S, int PIPE = 0>
__global__ __launch_bounds__(LN_THREADS, S == 3 ? 4 : 2) void k_linear_ragged(LinParams P)
{
    typedef ln_traits<BF16> TR;
    typedef typename TR::T T;
    typedef typename TR::V8 V8;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the 8 pieces of a tile run on ONE XCD (workgroup b lands on XCD b & 7): its L2 serves the shared x rows / W rows
    // (K split, P.rg_ksl > 0: the 2^rg_ksl workgroups of a piece are neighbours in this order, hence on one XCD as well)
    const int q = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    const int ksl = P.rg_ksl;
    const int p = q >> ksl, kpart = q & ((1 << ksl) - 1);
    const int tile = P.n_main + (p >> 3), piece = p & 7;
    int bm0, bn0;
    ln_tile_origin(P, tile, bm0, bn0);
    const int row0 = bm0 + (piece >> 2) * 128, col0 = bn0 + (piece & 3) * 64;
    const int K = P.K, nt = (K >> 6) >> ksl;           // K-tiles of THIS workgroup: [kpart * nt, (kpart + 1) * nt)
    const unsigned rowbytes = (unsigned)K * (unsigned)sizeof(T);
    // staging: a 1 KB chunk = 8 rows x 128 bytes, LDS row j = 8 c + (lane >> 3), LDS slot lane & 7 holds SOURCE slot
    // (lane & 7) ^ ((j >> 1) & 7).  Wave w stages x chunks 2 w, 2 w + 1 (rows 16 w .. 16 w + 15) and W chunk w (rows 8 w .. + 7)
    unsigned srcA[2], srcB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = 8 * (2 * wid + i) + (lane >> 3);
        srcA[i] = (unsigned)j * rowbytes + (unsigned)(((lane & 7) ^ ((j >> 1) & 7)) << 4);
    }
    {
        const int j = 8 * wid + (lane >> 3);
        srcB = (unsigned)j * rowbytes + (unsigned)(((lane & 7) ^ ((j >> 1) & 7)) << 4);
    }
    const unsigned char *xb = (const unsigned char *)P.x + (size_t)row0 * rowbytes + (size_t)(kpart * nt) * 128;
    const unsigned char *wb = (const unsigned char *)P.w + (size_t)col0 * rowbytes + (size_t)(kpart * nt) * 128;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
#define RG_STAGE(kt_)                                                                                                    \
    do {                                                                                                                 \
        const int kc_ = min((kt_), nt - 1);            /* past the end: a harmless re-load into a slot nobody reads */      \
        const unsigned sl_ = lds0 + (unsigned)((kt_) % S) * RG_SLOT;                                                      \
        const unsigned char *xa_ = xb + (size_t)kc_ * 128, *wa_ = wb + (size_t)kc_ * 128;                                 \
        ln_dma_s(xa_, srcA[0], sl_ + (unsigned)(2 * wid) * 1024u);                                                        \
        ln_dma_s(xa_, srcA[1], sl_ + (unsigned)(2 * wid + 1) * 1024u);                                                    \
        ln_dma_s(wa_, srcB, sl_ + 16384u + (unsigned)wid * 1024u);                                                        \
    } while (0)
    // fragment reads: lane reads LDS row l31 of its 32-row block, 16-byte slot 2 ks + hi, swizzled by the row
    const int br = wid >> 1, bc = wid & 1;               // this wave's 32 x 32 block of the piece: rows 32 br, columns 32 bc
    unsigned offA[4], offB[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const unsigned sl = (unsigned)((2 * ks + hi) ^ ((lane >> 1) & 7)) << 4;
        offA[ks] = (unsigned)(br * 32 + l31) * 128u + sl;
        offB[ks] = 16384u + (unsigned)(bc * 32 + l31) * 128u + sl;
    }
    // ONE accumulation chain, the 16-wide slices of K in ascending order: exactly the chain of k_linear256's accumulators (same
    // instruction, same operand order).  Round 6: an output row's value must not depend on whether its tile falls into the main
    // rounds or the ragged one -- the units of a batch are independent images (reference: batch 1, src/core.py:133), and the same
    // image at units 0 and 31 of a batch of 32 has to come out bit-identical (rounds 3-5 ran two chains per K-tile here, even / odd
    // slices summed at the end: another fp32 order, 3.2e-3 of the depth range after 24 blocks).  Price: the four MFMAs of a K-tile
    // are dependent (~64 cycles each instead of two chains of two).
    lf32x16 mine;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[r] = 0.f;

    // epilogue operands of this wave's 32 x 32 block (see the epilogue below): with PIPE they are requested here, ahead of every
    // DMA (vmcnt retires in order: they are the oldest entries, the counted waits of the loop are unaffected)
    const int e_row = row0 + (wid >> 1) * 32 + l31;
    const int e_cb = col0 + (wid & 1) * 32 + 8 * hi;
    V8 e_bv[2], e_gv[2], e_rv[2];
    if constexpr (PIPE) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int col = e_cb + 16 * k;
            if (P.bias) e_bv[k] = *(const V8 *)((const T *)P.bias + col);
            if (EPI == 3) e_gv[k] = *(const V8 *)((const T *)P.gamma + col);
            if (RES >= 1) e_rv[k] = *(const V8 *)((const T *)P.res1 + ln_res_off<VT>(P, e_row, col));
        }
    }
#pragma unroll
    for (int t = 0; t < S - 1; ++t) RG_STAGE(t);
    if constexpr (PIPE) {
        static_assert(!PIPE || S == 6, "the pipelined loop is written for the 6-slot ring");
        V8 fa0[4], fb0[4], fa1[4], fb1[4];
#define RG_READ(fa_, fb_, kt_)                                                                                           \
        do {                                                                                                             \
            const unsigned char *sb_ = lds + ((kt_) % S) * RG_SLOT;                                                      \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                           \
                fa_[ks] = *(const V8 *)(sb_ + offA[ks]);                                                                 \
                fb_[ks] = *(const V8 *)(sb_ + offB[ks]);                                                                 \
            }                                                                                                            \
        } while (0)
        // K-tile kt: [its fragments were requested one step earlier] wait for this wave's pieces of K-tile kt + 1 (the three
        // younger K-tiles stay in flight) | barrier: everybody's have landed, and everybody has consumed K-tile kt - 1 (read
        // in step kt - 2, waited for at the end of it) | stage K-tile kt + 5 into that slot | request the fragments of K-tile
        // kt + 1 (past the end: a slot holding a clamped re-load, never multiplied) | 4 MFMAs on K-tile kt | wait for them
#define RG_STEP(ca_, cb_, na_, nb_, kt_)                                                                                 \
        do {                                                                                                             \
            LN_WAIT_VM(9);                                                                                               \
            LN_BARRIER();                                                                                                \
            RG_STAGE((kt_) + S - 1);                                                                                     \
            RG_READ(na_, nb_, (kt_) + 1);                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            mine = TR::mfma(cb_[0], ca_[0], mine);                                                                       \
            mine = TR::mfma(cb_[1], ca_[1], mine);                                                                       \
            mine = TR::mfma(cb_[2], ca_[2], mine);                                                                       \
            mine = TR::mfma(cb_[3], ca_[3], mine);                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            LN_WAIT_LGKM0();                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
        } while (0)
        LN_WAIT_VM(12);                                  // K-tile 0 has landed (1 .. 4 in flight)
        LN_BARRIER();
        RG_READ(fa0, fb0, 0);
        __builtin_amdgcn_sched_barrier(0);
        LN_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        int kt = 0;
        for (; kt + 1 < nt; kt += 2) {
            RG_STEP(fa0, fb0, fa1, fb1, kt);
            RG_STEP(fa1, fb1, fa0, fb0, kt + 1);
        }
        if (kt < nt) RG_STEP(fa0, fb0, fa1, fb1, kt);
#undef RG_STEP
#undef RG_READ
    } else
    for (int kt = 0; kt < nt; ++kt) {
        // this wave's pieces of K-tile kt have landed (the S - 2 younger K-tiles, 3 DMAs each, stay in flight)
        if constexpr (S == 3) LN_WAIT_VM(3); else LN_WAIT_VM(12);
        LN_BARRIER();
        RG_STAGE(kt + S - 1);                            // into the slot of K-tile kt - 1
        const unsigned char *sb = lds + (kt % S) * RG_SLOT;
        // all 8 fragment reads in flight at once, ONE wait, then the accumulation chain (the first version -- read two, wait,
        // one MFMA, four times -- spent 0.64 us per K-tile)
        V8 fa[4], fb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            fa[ks] = *(const V8 *)(sb + offA[ks]);
            fb[ks] = *(const V8 *)(sb + offB[ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
        LN_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        mine = TR::mfma(fb[0], fa[0], mine);
        mine = TR::mfma(fb[1], fa[1], mine);
        mine = TR::mfma(fb[2], fa[2], mine);
        mine = TR::mfma(fb[3], fa[3], mine);
    }
    LN_WAIT_VM(0);                                       // the trailing re-loads: nothing may still be writing LDS at exit
#undef RG_STAGE

    // ---- K split: every workgroup of a piece leaves its fp32 partial in its own slot of the workspace ([wave][16][64 lanes]),
    // the LAST one to arrive (a counter per piece) adds the slots in the order of the K ranges -- the result does not depend on
    // which workgroup that is: bit-reproducible -- and runs the epilogue.  Nobody waits for anybody; the counter is back at zero
    // when the launch ends.  Visibility: the partials are written and read with agent-scope accesses (the sc1 bit: coherent
    // across the L2s of the XCDs, whichever XCD a workgroup runs on) and are complete (vmcnt(0)) before the arrival is counted.
    // NOT __threadfence(): its agent-scope release / acquire is a write-back + invalidate of the whole L2 (buffer_wbl2 sc1 /
    // buffer_inv sc1) issued by every wave -- measured: +55 us per launch, behind a main kernel that leaves the L2s dirty.
    if (ksl > 0) {
        __shared__ int s_last;
        float *slot = P.rg_ws + ((size_t)q * 8 + wid) * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) __hip_atomic_store(slot + r * 64, mine[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        LN_WAIT_VM(0);
        __syncthreads();
        if (tid == 0) {
            const int arrived = atomicAdd(P.rg_cnt + p, 1);
            const int last = arrived == (1 << ksl) - 1;
            if (last) atomicExch(P.rg_cnt + p, 0);
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[r] = 0.f;
        const float *first = P.rg_ws + ((size_t)(q - kpart) * 8 + wid) * 1024 + lane;
        // four (two) slots in flight per round trip, added in the order of the K ranges
        if (ksl >= 2) {
            for (int kp = 0; kp < (1 << ksl); kp += 4) {
                float part[4][16];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        part[u][r] = __hip_atomic_load(first + (size_t)(kp + u) * 8192 + r * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[r] += part[u][r];
            }
        } else {
            float part[2][16];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    part[u][r] = __hip_atomic_load(first + (size_t)u * 8192 + r * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[r] += part[u][r];
        }
    }

    // ---- epilogue of block wid = (32-row block wid >> 1, 32-column block wid & 1), as in k_linear256: register r = column
    // (r & 3) + 8 (r >> 2) + 4 hi of row l31; one permlane32 swap per register pair gives each lane 8 consecutive columns
    const int row = row0 + (wid >> 1) * 32 + l31;
    const int cb = col0 + (wid & 1) * 32 + 8 * hi;
    T *yb = (T *)P.y;
    const T *bias = (const T *)P.bias;
    const T *r1 = (const T *)P.res1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int col = cb + 16 * k;
        const size_t off = ln_out_off<VT>(P, row, col);
        V8 bv, gv, rv;
        if (bias) bv = PIPE ? e_bv[k] : *(const V8 *)(bias + col);
        else {
#pragma unroll
            for (int t = 0; t < 8; ++t) bv[t] = (T)0.f;
        }
        if (EPI == 3) gv = PIPE ? e_gv[k] : *(const V8 *)((const T *)P.gamma + col);
        if (RES >= 1) rv = PIPE ? e_rv[k] : *(const V8 *)(r1 + ln_res_off<VT>(P, row, col));
        float v[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float fa = mine[8 * k + t], fb = mine[8 * k + 4 + t];
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(fa), __float_as_uint(fb), false, false);
            v[t] = __uint_as_float(sw[0]);
            v[4 + t] = __uint_as_float(sw[1]);
        }
        if (EPI == 4 || EPI == 5) {                          // the folded LayerNorm, as in k_linear256
            if (!VT) {
                const float2 st = P.ln_stats[row];
                const float4 s0 = *(const float4 *)(P.ln_colsum + col), s1 = *(const float4 *)(P.ln_colsum + col + 4);
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = __builtin_fmaf(v[t], st.x, st.y * sc[t]);
            } else {
                const float cs = P.ln_colsum[row];
                const float4 *sp = (const float4 *)(P.ln_stats + col);
                const float4 q0 = sp[0], q1 = sp[1], q2 = sp[2], q3 = sp[3];
                const float rs[8] = {q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, q3.x, q3.z};
                const float nm[8] = {q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, q3.y, q3.w};
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = __builtin_fmaf(v[t], rs[t], nm[t] * cs);
            }
        }
        V8 o;
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            lf32x2 u = {v[t] + (float)bv[t], v[t + 1] + (float)bv[t + 1]};
            if (EPI == 3) u *= (lf32x2){(float)gv[t], (float)gv[t + 1]};
            if (RES >= 1) u += (lf32x2){(float)rv[t], (float)rv[t + 1]};
            if (EPI == 1 || EPI == 5) u = ln_gelu2(u);
            if (EPI == 2) u = (lf32x2){fmaxf(u[0], 0.f), fmaxf(u[1], 0.f)};
            o[t] = (T)u[0];
            o[t + 1] = (T)u[1];
        }
        *(V8 *)(yb + off) = o;
    }
}

// ---- the thin ragged round (round 6) -----------------------------------------------------------------------------------------
// The ragged round of the metric's encoder GEMMs is ONE row panel (32 images x 1032 token rows = 129 x 256: 4 / 8 / 16 tiles), and
// since round 6 it runs one accumulation chain over the whole of K (no K split: a row's value must not depend on the round it falls
// into).  k_linear_ragged stages 128 rows of x and 64 of W per K-tile, five K-tiles in flight: its loop is bound by the round trip
// of its DMA chain, 0.30 us per K-tile -- 19 us for fc2 (K = 4096) behind a main kernel of 240 us, with 32 of 256 CUs at work.
// This kernel renders the same rows as pieces of 32 rows x 64 columns -- four times as many workgroups -- and stages TWO K-tiles
// per step into a slot of the same 24 KB (x: 2 x 32 rows x 128 B, W: 2 x 64 rows x 128 B): twice the K range in flight per round
// trip, half the barriers, and the eight MFMAs of a step form the same chain (same instruction, same operand order, 16-wide slices
// of K ascending) -- bit-identical to k_linear_ragged and to the main rounds.  Waves 0 and 1 own the two 32 x 32 blocks of the
// piece; all eight waves stage (3 DMAs per wave and step, as in k_linear_ragged: its counted waits carry over).  The row blocks
// are the last 32 * th_nrb rows of x: when the panel's new rows are not a multiple of 32 the first block reaches into rows of the
// main rounds and writes them a second time with identical values (as the main kernel's shifted last panel does).  Taken when
// all pieces run at once (one workgroup per CU: 144 KB of LDS).
template <int BF16, int EPI, int RES>
__global__ __launch_bounds__(LN_THREADS) void k_linear_thin(LinParams P)
{
    typedef ln_traits<BF16> TR;
    typedef typename TR::T T;
    typedef typename TR::V8 V8;
    constexpr int S = 6;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncb = P.N >> 6;                            // pieces per row block
    const int rb = (int)blockIdx.x / ncb, cbk = (int)blockIdx.x - rb * ncb;
    const int row0 = P.M - 32 * (P.th_nrb - rb), col0 = cbk * 64;
    const int K = P.K, nst = K >> 7;                     // steps of two K-tiles
    const unsigned rowbytes = (unsigned)K * (unsigned)sizeof(T);
    // staging: a 1 KB chunk = 8 rows x 128 bytes of ONE K-tile, LDS row j = 8 c + (lane >> 3), LDS slot lane & 7 holds SOURCE slot
    // (lane & 7) ^ ((j >> 1) & 7) (the swizzle of k_linear256).  Slot map: x K-tile 0 | x K-tile 1 (4 KB each) | W K-tile 0 | W K-tile 1
    // (8 KB each).  Wave w stages x chunk w (K-tile w >> 2, rows 8 (w & 3) ..) and W chunks 2 w, 2 w + 1 (K-tile w >> 2, rows 16 (w & 3) ..)
    unsigned srcA, srcB[2];
    {
        const int j = 8 * (wid & 3) + (lane >> 3);
        srcA = (unsigned)j * rowbytes + (unsigned)(((lane & 7) ^ ((j >> 1) & 7)) << 4) + (unsigned)(wid >> 2) * 128u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = 8 * ((2 * wid + i) & 7) + (lane >> 3);
        srcB[i] = (unsigned)j * rowbytes + (unsigned)(((lane & 7) ^ ((j >> 1) & 7)) << 4) + (unsigned)(wid >> 2) * 128u;
    }
    const unsigned char *xb = (const unsigned char *)P.x + (size_t)row0 * rowbytes;
    const unsigned char *wb = (const unsigned char *)P.w + (size_t)col0 * rowbytes;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
#define TH_STAGE(st_)                                                                                                    \
    do {                                                                                                                 \
        const int sc_ = min((st_), nst - 1);           /* past the end: a harmless re-load into a slot nobody reads */      \
        const unsigned sl_ = lds0 + (unsigned)((st_) % S) * RG_SLOT;                                                      \
        const unsigned char *xa_ = xb + (size_t)sc_ * 256, *wa_ = wb + (size_t)sc_ * 256;                                 \
        ln_dma_s(xa_, srcA, sl_ + (unsigned)wid * 1024u);                                                                 \
        ln_dma_s(wa_, srcB[0], sl_ + 8192u + (unsigned)(2 * wid) * 1024u);                                                \
        ln_dma_s(wa_, srcB[1], sl_ + 8192u + (unsigned)(2 * wid + 1) * 1024u);                                            \
    } while (0)
    const bool mw = wid < 2;                             // the two waves that multiply: columns 32 wid .. 32 wid + 31 of the piece
    unsigned offA[8], offB[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int j = u >> 2, ks = u & 3;
        const unsigned sl = (unsigned)((2 * ks + hi) ^ ((lane >> 1) & 7)) << 4;
        offA[u] = (unsigned)j * 4096u + (unsigned)l31 * 128u + sl;
        offB[u] = 8192u + (unsigned)j * 8192u + (unsigned)((wid & 1) * 32 + l31) * 128u + sl;
    }
    lf32x16 mine;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[r] = 0.f;
    // epilogue operands first (the oldest vmcnt entries: the counted waits of the loop are unaffected)
    const int e_row = row0 + l31;
    const int e_cb = col0 + (wid & 1) * 32 + 8 * hi;
    V8 e_bv[2], e_gv[2], e_rv[2];
    if (mw) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int col = e_cb + 16 * k;
            if (P.bias) e_bv[k] = *(const V8 *)((const T *)P.bias + col);
            if (EPI == 3) e_gv[k] = *(const V8 *)((const T *)P.gamma + col);
            if (RES >= 1) e_rv[k] = *(const V8 *)((const T *)P.res1 + (size_t)e_row * P.ldy + col);
        }
    }
#pragma unroll
    for (int t = 0; t < S - 1; ++t) TH_STAGE(t);
    V8 fa0[8], fb0[8], fa1[8], fb1[8];
#define TH_READ(fa_, fb_, st_)                                                                                           \
    do {                                                                                                                 \
        const unsigned char *sb_ = lds + ((st_) % S) * RG_SLOT;                                                          \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                  \
            fa_[u] = *(const V8 *)(sb_ + offA[u]);                                                                       \
            fb_[u] = *(const V8 *)(sb_ + offB[u]);                                                                       \
        }                                                                                                                \
    } while (0)
    // step st: wait for this wave's pieces of step st + 1 (three younger steps stay in flight) | barrier | stage step st + 5 into
    // the slot of step st - 1 | request the fragments of step st + 1 | 8 MFMAs on step st | wait for the fragments
#define TH_STEP(ca_, cb_, na_, nb_, st_)                                                                                 \
    do {                                                                                                                 \
        LN_WAIT_VM(9);                                                                                                   \
        LN_BARRIER();                                                                                                    \
        TH_STAGE((st_) + S - 1);                                                                                         \
        if (mw) {                                                                                                        \
            TH_READ(na_, nb_, (st_) + 1);                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) mine = TR::mfma(cb_[u], ca_[u], mine);                         \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            LN_WAIT_LGKM0();                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
        }                                                                                                                \
    } while (0)
    LN_WAIT_VM(12);                                      // step 0 has landed (1 .. 4 in flight)
    LN_BARRIER();
    if (mw) {
        TH_READ(fa0, fb0, 0);
        __builtin_amdgcn_sched_barrier(0);
        LN_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
    }
    int st = 0;
    for (; st + 1 < nst; st += 2) {
        TH_STEP(fa0, fb0, fa1, fb1, st);
        TH_STEP(fa1, fb1, fa0, fb0, st + 1);
    }
    if (st < nst) TH_STEP(fa0, fb0, fa1, fb1, st);
#undef TH_STEP
#undef TH_READ
#undef TH_STAGE
    LN_WAIT_VM(0);                                       // the trailing re-loads: nothing may still be writing LDS at exit
    if (!mw) return;

    // ---- epilogue of this wave's 32 x 32 block, as in k_linear_ragged (VT 0)
    const int row = e_row, cb = e_cb;
    T *yb = (T *)P.y;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int col = cb + 16 * k;
        const size_t off = (size_t)row * P.ldy + col;
        V8 bv;
        if (P.bias) bv = e_bv[k];
        else {
#pragma unroll
            for (int t = 0; t < 8; ++t) bv[t] = (T)0.f;
        }
        float v[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float fa = mine[8 * k + t], fb = mine[8 * k + 4 + t];
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(fa), __float_as_uint(fb), false, false);
            v[t] = __uint_as_float(sw[0]);
            v[4 + t] = __uint_as_float(sw[1]);
        }
        if (EPI == 4 || EPI == 5) {                          // the folded LayerNorm, as in k_linear256
            const float2 stt = P.ln_stats[row];
            const float4 s0 = *(const float4 *)(P.ln_colsum + col), s1 = *(const float4 *)(P.ln_colsum + col + 4);
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = __builtin_fmaf(v[t], stt.x, stt.y * sc[t]);
        }
        V8 o;
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            lf32x2 u = {v[t] + (float)bv[t], v[t + 1] + (float)bv[t + 1]};
            if (EPI == 3) u *= (lf32x2){(float)e_gv[k][t], (float)e_gv[k][t + 1]};
            if (RES >= 1) u += (lf32x2){(float)e_rv[k][t], (float)e_rv[k][t + 1]};
            if (EPI == 1 || EPI == 5) u = ln_gelu2(u);
            if (EPI == 2) u = (lf32x2){fmaxf(u[0], 0.f), fmaxf(u[1], 0.f)};
            o[t] = (T)u[0];
            o[t + 1] = (T)u[1];
        }
        *(V8 *)(yb + off) = o;
    }
}

// ---- C ABI -------------------------------------------------------------------------------------------------------------
// A/B switches of the GEMM path, read from the environment ONCE (hundreds of launches per forward; ds_linear_reload_env re-reads
// them: the tests flip them inside one process).  None of them changes a result, except DS_LIN_RAGGED_KSPLIT > 1.
//   DS_LIN_EARLY    1 (default) the next tile's prologue DMAs are issued before the epilogue, 0 after it
//   DS_LIN_GRID     number of persistent workgroups (default: one per CU)
//   DS_LIN_RAGGED / DS_LIN_RAGGED_DEN / DS_LIN_RAGGED_RING   the ragged round: on / "at most 1/DEN full" / ring depth 3 or 6
//   DS_LIN_RAGGED_KSPLIT   the ragged round: at most this many workgroups share the K range of a piece.  Default 1 = no split since
//                          round 6: a split sums in another fp32 order than the main rounds' single chain, so a row's value would
//                          depend on where in the batch its image sits (8 = rounds 4-5: fc2 at batch 32 -16 us per launch), for
//   DS_LIN_RAGGED_PIPE     the ragged round (deep ring): 1 (default) software-pipelined fragment reads + early epilogue operands
//   DS_LIN_RAGGED_THIN     1 (default) a ragged round that is one row panel with at most 32 new rows runs k_linear_thin (same values)
//   DS_LIN_RAGGED_KSPLIT_MIN / _KEEP   contractions of at least MIN K-tiles (default 32), every workgroup keeping >= KEEP (default 8)
struct LinOptions { int early, grid, ragged, ragged_den, ragged_ring, ragged_ksplit, ragged_ksplit_min, ragged_ksplit_keep, ragged_pipe, ragged_thin; };
static LinOptions g_lin_options;
static std::atomic<int> g_lin_options_state{0};
static void ln_read_options()
{
    auto geti = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    LinOptions o;
    o.early = geti("DS_LIN_EARLY", 1);
    o.grid = geti("DS_LIN_GRID", 0);
    o.ragged = geti("DS_LIN_RAGGED", 1);
    o.ragged_den = geti("DS_LIN_RAGGED_DEN", 4);
    o.ragged_ring = geti("DS_LIN_RAGGED_RING", 0);
    o.ragged_ksplit = geti("DS_LIN_RAGGED_KSPLIT", 1);
    o.ragged_ksplit_min = geti("DS_LIN_RAGGED_KSPLIT_MIN", 32);
    o.ragged_ksplit_keep = geti("DS_LIN_RAGGED_KSPLIT_KEEP", 8);
    if (o.ragged_ksplit_keep < 1) o.ragged_ksplit_keep = 1;
    o.ragged_pipe = geti("DS_LIN_RAGGED_PIPE", 1);
    o.ragged_thin = geti("DS_LIN_RAGGED_THIN", 1);
    g_lin_options = o;
    g_lin_options_state.store(1, std::memory_order_release);
}
static const LinOptions &ln_options()
{
    if (!g_lin_options_state.load(std::memory_order_acquire)) ln_read_options();
    return g_lin_options;
}
DS_API int ds_linear_reload_env(void)
{
    ln_read_options();
    return DS_OK;
}

// Workspace of the ragged round's K split: arrival counters (zeroed once; every launch leaves them at zero) + one 32 KB partial per
// workgroup.  Allocated ONCE per context at its maximum size (LN_RG_MAX_WG workgroups: 32 MB) and never reallocated -- captured
// hipGraphs hold its address (a graph must replay on the context that captured it), and a second allocation could hand the same
// address back with stale counters; a launch with more workgroups than that runs unsplit.  Returns 1 when the block would have to
// be allocated while the stream is being captured into a graph (the caller then launches without the split: the eager warm-up
// calls that precede a capture normally have it in place).
#define LN_RG_COUNTER_BYTES 4096
#define LN_RG_MAX_WG 1024
static int ln_ragged_workspace(ds_ctx *ctx, int grid, hipStream_t stream)
{
    if (grid > LN_RG_MAX_WG) return 1;
    if (ctx->lin_ws && ctx->lin_ws_cleared == ctx->lin_ws) return DS_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return 1; }
    if (!ctx->lin_ws) {
        const int rc = ds_ctx_reserve(ctx, &ctx->lin_ws, &ctx->lin_ws_bytes, LN_RG_COUNTER_BYTES + (size_t)LN_RG_MAX_WG * 32768);
        if (rc != DS_OK) return rc;
    }
    DS_HIP_CHECK(hipMemsetAsync(ctx->lin_ws, 0, LN_RG_COUNTER_BYTES, stream));
    DS_HIP_CHECK(hipStreamSynchronize(stream));
    ctx->lin_ws_cleared = ctx->lin_ws;
    return DS_OK;
}

// The K split's hand-over between workgroups (sc1 stores of the partials, vmcnt(0), a relaxed device-scope counter, sc1 loads by the
// last arriver) is argued on the ISA of gfx942 / gfx950 -- sc1 stores write through to memory and sc1 loads bypass the per-XCD L2 --
// not on the HIP memory model (no agent-scope release / acquire: the cache-wide write-back + invalidate it lowers to costs 55 us per
// launch, see the kernel).  On any other architecture the split stays off.
static int ln_ksplit_arch_ok(ds_ctx *ctx)
{
    static std::atomic<uint64_t> checked{0}, ok{0};
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (!(checked.load(std::memory_order_acquire) & bit)) {
        hipDeviceProp_t prop;
        bool good = false;
        if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess)
            good = strncmp(prop.gcnArchName, "gfx950", 6) == 0 || strncmp(prop.gcnArchName, "gfx942", 6) == 0;
        else (void)hipGetLastError();
        if (good) ok.fetch_or(bit, std::memory_order_relaxed);
        checked.fetch_or(bit, std::memory_order_release);
    }
    return (ok.load(std::memory_order_relaxed) & bit) != 0;
}

template <int BF16, int EPI, int CONV, int RES, int VT = 0, int NH = 0>
static int ln_launch(ds_ctx *ctx, const LinParams &P0, hipStream_t stream)
{
    // per DEVICE, not per process: the dynamic-LDS attribute belongs to the function on one device, and the grid is that
    // device's CU count (a process may drive several GPUs through several contexts).  Setting the attribute twice is harmless,
    // so the bit mask needs no lock.
    static std::atomic<uint64_t> attr_done{0};
    auto fn = k_linear256<BF16, EPI, CONV, RES, VT, NH>;
    DS_HIP_CHECK(hipSetDevice(ctx->device));
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (!(attr_done.load(std::memory_order_relaxed) & bit)) {
        DS_HIP_CHECK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LN_LDS_BYTES));
        if constexpr (CONV == 0) {
            DS_HIP_CHECK(hipFuncSetAttribute((const void *)k_linear_ragged<BF16, EPI, RES, VT, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * RG_SLOT));
            DS_HIP_CHECK(hipFuncSetAttribute((const void *)k_linear_ragged<BF16, EPI, RES, VT, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * RG_SLOT));
            DS_HIP_CHECK(hipFuncSetAttribute((const void *)k_linear_ragged<BF16, EPI, RES, VT, 6, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * RG_SLOT));
            if constexpr (VT == 0 && RES <= 1)
                DS_HIP_CHECK(hipFuncSetAttribute((const void *)k_linear_thin<BF16, EPI, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * RG_SLOT));
        }
        attr_done.fetch_or(bit, std::memory_order_relaxed);
    }
    if (!ctx->ncu) {
        int ncu = 0;
        DS_HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
        ctx->ncu = ncu >= 8 ? ncu / 8 * 8 : 8;            // one workgroup per CU (128 KB of LDS each), a multiple of the 8 XCDs
    }
    const LinOptions &O = ln_options();
    int grid = ctx->ncu;
    if (O.grid >= 8) grid = O.grid / 8 * 8;             // tests: a small grid makes every workgroup walk many tiles
    LinParams P = P0;
    P.bnw = NH ? 128 : 256;
    P.early = O.early;                                   // A/B switch (both orders give the same values)
#ifdef DS_EXPERIMENTS
    if (P.ablate) P.early = 0;               // the no-store ablation changes the store count the early mode's waits rely on
#endif
    const int ntiles = P.nbm * P.nbn;
    // the ragged round (k_linear_ragged): when the last round of the persistent walk would be at most a quarter full, its
    // tiles are rendered as 128 x 64 pieces by the whole chip instead.  DS_LIN_RAGGED=0 switches it off (A/B runs).
    int ragged = 0;
    if (CONV == 0 && ntiles > grid) {
        const int r = ntiles % grid;
        if (O.ragged && r > 0 && O.ragged_den > 0 && O.ragged_den * r <= grid) ragged = r;
    }
    P.n_main = ntiles - ragged;
    // the ragged round's K split: a launch of few pieces (4 tiles = 32 pieces on 256 CUs) is bound by the latency of one
    // piece's K loop (0.35 us per K-tile), so up to 8 workgroups share a long one, as long as all of them run at once (one per
    // CU: the deep ring) and each keeps at least 8 K-tiles.  Measured at batch 32 (profiles/round4_microbench_gemms_ksplit.txt):
    // fc2 (K = 4096, 4 tiles) 270.1 -> 253.9 us; at K = 1024 the loop is 6 of the launch's 12 us and the split buys nothing
    // (fc1 274.4 / 275.0, qk 148.2 / 148.1, proj 85.9 / 87.6 with 2 K-tiles each), hence the lower bound of 32 K-tiles.
    const int deep = O.ragged_ring ? O.ragged_ring == 6 : 8 * ragged <= grid;
    // the thin ragged round (k_linear_thin): the left-over tiles are exactly the last row panel (a group of its own in the tile
    // list: (nbm - 1) % 8 == 0), and its 32 x 64 pieces all run at once (one workgroup per CU)
    bool thin = false;
    int th_nrb = 0;
    if constexpr (CONV == 0 && VT == 0 && RES <= 1) {
        th_nrb = (P.M - (P.nbm - 1) * 256 + 31) / 32;
        thin = O.ragged_thin && ragged > 0 && ragged == P.nbn && P.nbm >= 2 && (P.nbm - 1) % 8 == 0 && P.K % 128 == 0 && P.N % 64 == 0 &&
               O.ragged_ksplit <= 1 && th_nrb * (P.N / 64) <= ctx->ncu;
    }
    P.th_nrb = th_nrb;
    int ksl = 0;
    if (CONV == 0 && ragged && !thin && deep && 8 * ragged * (int)sizeof(int) <= LN_RG_COUNTER_BYTES && ln_ksplit_arch_ok(ctx)) {
        const int nt = P.K / 64;
        while (nt >= O.ragged_ksplit_min && (2 << ksl) <= O.ragged_ksplit && 8 * ragged * (2 << ksl) <= grid && nt % (2 << ksl) == 0 &&
               nt / (2 << ksl) >= O.ragged_ksplit_keep) ++ksl;
    }
    if (ksl > 0) {
        const int rc = ln_ragged_workspace(ctx, grid, stream);
        if (rc == DS_OK) { P.rg_cnt = (int *)ctx->lin_ws; P.rg_ws = (float *)((char *)ctx->lin_ws + LN_RG_COUNTER_BYTES); }
        else if (rc == 1) ksl = 0;                       // not available inside a stream capture before its first eager use
        else return rc;
    }
    P.rg_ksl = ksl;
    const int kt0 = ds_kt_begin(ctx, P.kt_kind, stream);
    hipLaunchKernelGGL(fn, dim3(P.n_main < grid ? P.n_main : grid), dim3(LN_THREADS), LN_LDS_BYTES, stream, P);
    ds_kt_end(ctx, P.kt_kind, kt0, stream);
    if constexpr (CONV == 0) {
        const int kt1 = ragged ? ds_kt_begin(ctx, P.kt_kind + DS_KT_RAGGED, stream) : -1;
        if (thin) {
            if constexpr (VT == 0 && RES <= 1) hipLaunchKernelGGL((k_linear_thin<BF16, EPI, RES>), dim3(th_nrb * (P.N / 64)), dim3(LN_THREADS), 6 * RG_SLOT, stream, P);
        } else if (ragged && deep && O.ragged_pipe) hipLaunchKernelGGL((k_linear_ragged<BF16, EPI, RES, VT, 6, 1>), dim3((8 * ragged) << ksl), dim3(LN_THREADS), 6 * RG_SLOT, stream, P);
        else if (ragged && deep) hipLaunchKernelGGL((k_linear_ragged<BF16, EPI, RES, VT, 6>), dim3((8 * ragged) << ksl), dim3(LN_THREADS), 6 * RG_SLOT, stream, P);
        else if (ragged) hipLaunchKernelGGL((k_linear_ragged<BF16, EPI, RES, VT, 3>), dim3(8 * ragged), dim3(LN_THREADS), 3 * RG_SLOT, stream, P);
        ds_kt_end(ctx, P.kt_kind + DS_KT_RAGGED, kt1, stream);
    }
    DS_HIP_CHECK(hipGetLastError());
    return DS_OK;
}

template <int BF16>
static int ln_dispatch_dense(ds_ctx *ctx, const LinParams &P, int act, hipStream_t st)
{
    if (act == 1) return ln_launch<BF16, 1, 0, 0>(ctx, P, st);
    if (act == 2) return ln_launch<BF16, 2, 0, 0>(ctx, P, st);
    return ln_launch<BF16, 0, 0, 0>(ctx, P, st);
}

template <int BF16>
static int ln_dispatch_conv(ds_ctx *ctx, const LinParams &P, int act, hipStream_t st)
{
    const int res = P.res1 ? (P.res2 ? 2 : 1) : 0;
    if (act & 4) {                           // ReLU on x (CONV 2): the first convolution of a residual unit -- ReLU out, no addends
        if (P.N % 256 == 0 && (act & 3) == 2 && res == 0) return ln_launch<BF16, 2, 2, 0>(ctx, P, st);
        ds_set_error("ds_conv3x3_nhwc: act 4 (ReLU on x) is built for act 2 | 4 without residual operands and out_channels %% 256 == 0");
        return DS_EUNSUPPORTED;
    }
    if (P.N % 256 != 0) {                    // 128-column tiles (NH = 1): the head convolution, no residual operands
        if (act == 2) return ln_launch<BF16, 2, 1, 0, 0, 1>(ctx, P, st);
        return ln_launch<BF16, 0, 1, 0, 0, 1>(ctx, P, st);
    }
    if (act == 2) {
        if (res == 2) return ln_launch<BF16, 2, 1, 2>(ctx, P, st);
        if (res == 1) return ln_launch<BF16, 2, 1, 1>(ctx, P, st);
        return ln_launch<BF16, 2, 1, 0>(ctx, P, st);
    }
    if (res == 2) return ln_launch<BF16, 0, 1, 2>(ctx, P, st);
    if (res == 1) return ln_launch<BF16, 0, 1, 1>(ctx, P, st);
    return ln_launch<BF16, 0, 1, 0>(ctx, P, st);
}

DS_API int ds_linear(ds_ctx *ctx, const void *x, const void *w, const void *bias, void *y, int64_t rows, int64_t out_features,
                     int64_t in_features, int64_t ldy, int act, int dtype, void *stream)
{
    DS_REQUIRE(ctx && x && w && y, DS_EINVAL, "ds_linear: null argument");
    DS_REQUIRE(rows >= 256 && rows < (1ll << 31) - 256, DS_EINVAL, "ds_linear: rows must be >= 256 (one tile)");
    DS_REQUIRE(out_features > 0 && out_features % 256 == 0, DS_EINVAL, "ds_linear: out_features must be a multiple of 256");
    DS_REQUIRE(in_features >= 128 && in_features % 128 == 0 && in_features <= 16384, DS_EINVAL,
               "ds_linear: in_features must be a multiple of 128 (<= 16384)");
    DS_REQUIRE(ldy >= out_features && ldy % 8 == 0, DS_EINVAL, "ds_linear: ldy must be >= out_features and a multiple of 8");
    DS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0), DS_EINVAL,
               "ds_linear: x, w (the LDS-DMA sources), y and bias must be 16-byte aligned");
    DS_REQUIRE(act >= 0 && act <= 2, DS_EINVAL, "ds_linear: act must be 0 (none), 1 (erf-GELU) or 2 (ReLU)");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_linear: dtype must be f16 or bf16");
    LinParams P;
    memset(&P, 0, sizeof(P));
    P.x = x; P.w = w; P.bias = bias; P.y = y;
    P.M = (int)rows; P.N = (int)out_features; P.K = (int)in_features;
    P.nbm = (int)((rows + 255) / 256); P.nbn = (int)(out_features / 256);
    P.ldy = ldy;
    P.kt_kind = act == 1 ? DS_KT_LINEAR_GELU : DS_KT_LINEAR;
#ifdef DS_EXPERIMENTS
    P.ablate = getenv("DS_LIN_ABLATE") ? atoi(getenv("DS_LIN_ABLATE")) : 0;
    P.stagger = getenv("DS_LIN_STAGGER_US") ? (int)(atof(getenv("DS_LIN_STAGGER_US")) * 100.0) : 0;
#endif
    return dtype == DS_DTYPE_F16 ? ln_dispatch_dense<0>(ctx, P, act, (hipStream_t)stream) : ln_dispatch_dense<1>(ctx, P, act, (hipStream_t)stream);
}

// y = res + [gamma *] (x . W^T + b): the output projection of an encoder block with its LayerScale and residual add in the
// epilogue (prepared at the end of round 2, opt-in through DS_LINEAR=proj; not yet measured on hardware).
DS_API int ds_linear_residual(ds_ctx *ctx, const void *x, const void *w, const void *bias, const void *gamma, const void *res, void *y,
                              int64_t rows, int64_t out_features, int64_t in_features, int dtype, void *stream)
{
    DS_REQUIRE(ctx && x && w && y && res, DS_EINVAL, "ds_linear_residual: null argument");
    DS_REQUIRE(rows >= 256 && rows < (1ll << 31) - 256, DS_EINVAL, "ds_linear_residual: rows must be >= 256 (one tile)");
    DS_REQUIRE(out_features > 0 && out_features % 256 == 0, DS_EINVAL, "ds_linear_residual: out_features must be a multiple of 256");
    DS_REQUIRE(in_features >= 128 && in_features % 128 == 0 && in_features <= 16384, DS_EINVAL,
               "ds_linear_residual: in_features must be a multiple of 128 (<= 16384)");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_linear_residual: dtype must be f16 or bf16");
    DS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)res & 15) == 0 &&
               ((uintptr_t)bias & 15) == 0 && ((uintptr_t)gamma & 15) == 0 && y != res,
               DS_EINVAL, "ds_linear_residual: x, w, y, res, bias and gamma must be 16-byte aligned, and y must not alias res");
    LinParams P;
    memset(&P, 0, sizeof(P));
    P.x = x; P.w = w; P.bias = bias; P.gamma = gamma; P.res1 = res; P.y = y;
    P.M = (int)rows; P.N = (int)out_features; P.K = (int)in_features;
    P.nbm = (int)((rows + 255) / 256); P.nbn = (int)(out_features / 256);
    P.ldy = out_features;
    P.kt_kind = DS_KT_LINEAR_RESIDUAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DS_DTYPE_F16) return gamma ? ln_launch<0, 3, 0, 1>(ctx, P, st) : ln_launch<0, 0, 0, 1>(ctx, P, st);
    return gamma ? ln_launch<1, 3, 0, 1>(ctx, P, st) : ln_launch<1, 0, 0, 1>(ctx, P, st);
}

DS_API int ds_conv3x3_nhwc(ds_ctx *ctx, const void *x, const void *w, const void *bias, const void *res1, const void *res2, void *y,
                           int batch, int height, int width, int in_channels, int out_channels, int act, int dtype, void *stream)
{
    DS_REQUIRE(ctx && x && w && y, DS_EINVAL, "ds_conv3x3_nhwc: null argument");
    DS_REQUIRE(batch > 0 && height > 0 && width > 0 && height < 32768 && width < 32768, DS_EINVAL, "ds_conv3x3_nhwc: bad image shape");
    DS_REQUIRE((int64_t)batch * height * width < (1ll << 31) - 256 && (int64_t)batch * height * width >= 256, DS_EINVAL,
               "ds_conv3x3_nhwc: batch * height * width must be in [256, 2^31)");
    DS_REQUIRE(in_channels >= 64 && in_channels % 64 == 0 && (9 * in_channels / 64) % 2 == 0 && in_channels <= 4096, DS_EINVAL,
               "ds_conv3x3_nhwc: in_channels must be a multiple of 128 (<= 4096)");
    DS_REQUIRE(out_channels > 0 && out_channels % 128 == 0, DS_EINVAL, "ds_conv3x3_nhwc: out_channels must be a multiple of 128");
    DS_REQUIRE(out_channels % 256 == 0 || (!res1 && !res2), DS_EUNSUPPORTED,
               "ds_conv3x3_nhwc: residual operands need out_channels to be a multiple of 256 (the 128-column tiles have no residual epilogue)");
    DS_REQUIRE(act == 0 || act == 2 || act == 6, DS_EINVAL, "ds_conv3x3_nhwc: act must be 0 (none), 2 (ReLU) or 6 (ReLU, and ReLU on x)");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_conv3x3_nhwc: dtype must be f16 or bf16");
    DS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) &&
               ((uintptr_t)res1 & 15) == 0 && ((uintptr_t)res2 & 15) == 0,
               DS_EINVAL, "ds_conv3x3_nhwc: x, w, y, bias, res1 and res2 must be 16-byte aligned");
    int rc = ds_ctx_reserve(ctx, &ctx->zero_line, &ctx->zero_line_bytes, 256);
    if (rc != DS_OK) return rc;
    if (!ctx->zero_line_cleared) {          // once per context; waited for, so that a second stream of the same context never
        DS_HIP_CHECK(hipMemsetAsync(ctx->zero_line, 0, 256, (hipStream_t)stream));          // reads the line before it is zero
        DS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        ctx->zero_line_cleared = 1;
    }
    LinParams P;
    memset(&P, 0, sizeof(P));
    if (!res1 && res2) { res1 = res2; res2 = nullptr; }
    P.x = x; P.w = w; P.bias = bias; P.res1 = res1; P.res2 = res2; P.y = y; P.zeros = ctx->zero_line;
    P.M = batch * height * width; P.N = out_channels; P.K = 9 * in_channels;
    P.nbm = (P.M + 255) / 256; P.nbn = out_channels % 256 == 0 ? out_channels / 256 : out_channels / 128;
    P.H = height; P.W = width; P.C = in_channels; P.cpt = in_channels / 64; P.magic = 65536 / P.cpt + 1;
    for (int kt = 0; kt < P.K / 64; ++kt)
        DS_REQUIRE(((kt * 7282) >> 16) == kt / 9, DS_EUNSUPPORTED, "ds_conv3x3_nhwc: K-tile arithmetic does not cover %d channels", in_channels);
    P.ldy = out_channels;
    P.kt_kind = DS_KT_CONV3X3;
#ifdef DS_EXPERIMENTS
    P.ablate = getenv("DS_LIN_ABLATE") ? atoi(getenv("DS_LIN_ABLATE")) : 0;
    P.stagger = getenv("DS_LIN_STAGGER_US") ? (int)(atof(getenv("DS_LIN_STAGGER_US")) * 100.0) : 0;
#endif
    return dtype == DS_DTYPE_F16 ? ln_dispatch_conv<0>(ctx, P, act, (hipStream_t)stream) : ln_dispatch_conv<1>(ctx, P, act, (hipStream_t)stream);
}

// V^T of an encoder block straight out of the GEMM: vt[b][c][n] = sum_k w_v[c][k] h[b][n][k] (the reference computes
// qkv = Linear(h) and permutes, dmidas/backbones/beit.py:71-74, dinov2_layers/attention.py:52-55; the attention kernel wants V
// with the key index contiguous).  The GEMM runs with W_v as the row operand and ALL tokens of the batch as columns; the
// epilogue scatters every group of 8 columns to its batch element (ln_out_off<1>).  The V bias is not added here: it commutes
// with the attention and is folded into the projection bias on the host.
DS_API int ds_linear_vt(ds_ctx *ctx, const void *w_v, const void *h, void *vt, int64_t channels, int64_t batch, int64_t tokens,
                        int64_t in_features, int dtype, void *stream)
{
    DS_REQUIRE(ctx && w_v && h && vt, DS_EINVAL, "ds_linear_vt: null argument");
    DS_REQUIRE(channels >= 256, DS_EINVAL, "ds_linear_vt: channels must be >= 256 (one tile)");
    DS_REQUIRE(batch > 0 && tokens > 0 && tokens % 8 == 0 && (batch * tokens) % 256 == 0 && batch * tokens < (1ll << 31) - 256, DS_EINVAL,
               "ds_linear_vt: tokens must be a multiple of 8 and batch * tokens a multiple of 256");
    DS_REQUIRE(batch * tokens * tokens < (1ll << 32), DS_EUNSUPPORTED, "ds_linear_vt: batch * tokens^2 must stay below 2^32");
    DS_REQUIRE(in_features >= 128 && in_features % 128 == 0 && in_features <= 16384, DS_EINVAL,
               "ds_linear_vt: in_features must be a multiple of 128 (<= 16384)");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_linear_vt: dtype must be f16 or bf16");
    DS_REQUIRE(((uintptr_t)w_v & 15) == 0 && ((uintptr_t)h & 15) == 0 && ((uintptr_t)vt & 15) == 0, DS_EINVAL,
               "ds_linear_vt: w_v, h and vt must be 16-byte aligned");
    LinParams P;
    memset(&P, 0, sizeof(P));
    P.x = w_v; P.w = h; P.y = vt;
    P.M = (int)channels; P.N = (int)(batch * tokens); P.K = (int)in_features;
    P.nbm = (int)((channels + 255) / 256); P.nbn = (int)(batch * tokens / 256);
    P.ldy = batch * tokens;
    P.vt_np = (int)tokens; P.vt_c = (int)channels; P.vt_magic = (unsigned)((1ull << 32) / (unsigned long long)tokens + 1ull);
    P.kt_kind = DS_KT_LINEAR_VT;
    hipStream_t st = (hipStream_t)stream;
    return dtype == DS_DTYPE_F16 ? ln_launch<0, 0, 0, 0, 1>(ctx, P, st) : ln_launch<1, 0, 0, 0, 1>(ctx, P, st);
}

// ConvTranspose2d with kernel_size == stride (no overlap between the taps of neighbouring pixels) is a plain GEMM whose output
// columns are (ky, kx, co): y[b, y*s + ky, x*s + kx, co] = bias[co] + sum_ci x[b, y, x, ci] * w[ci, co, ky, kx] -- the 4x4-s4 and
// 2x2-s2 transposed convolutions of the reassemble stage (dmidas/backbones/utils.py:196-205,215-224; ddepth_anything_v2/
// depth_anything_v2/dpt.py:57-71).  The epilogue stores every 8-channel group straight to its output pixel (pixel shuffle in the
// store address): no [pixels, s*s*C] intermediate, no shuffle pass.
DS_API int ds_linear_shuffle(ds_ctx *ctx, const void *x, const void *w, const void *bias, void *y, int64_t pixels, int64_t in_features,
                             int width, int stride, int out_channels, int dtype, void *stream)
{
    DS_REQUIRE(ctx && x && w && y, DS_EINVAL, "ds_linear_shuffle: null argument");
    DS_REQUIRE(stride >= 1 && stride <= 8 && width >= 1 && out_channels >= 8 && out_channels % 8 == 0, DS_EINVAL,
               "ds_linear_shuffle: stride must be 1..8, out_channels a multiple of 8");
    const int64_t n = (int64_t)stride * stride * out_channels;
    DS_REQUIRE(pixels >= 256 && pixels % width == 0 && pixels < (1ll << 31) - 256, DS_EINVAL,
               "ds_linear_shuffle: pixels must be >= 256 (one tile) and a whole number of image rows");
    DS_REQUIRE(n % 256 == 0, DS_EUNSUPPORTED, "ds_linear_shuffle: stride^2 * out_channels must be a multiple of 256 (got %lld)", (long long)n);
    DS_REQUIRE(in_features >= 128 && in_features % 128 == 0 && in_features <= 16384, DS_EINVAL,
               "ds_linear_shuffle: in_features must be a multiple of 128 (<= 16384)");
    DS_REQUIRE(pixels * width < (1ll << 32) && n * out_channels < (1ll << 32) && pixels * n < (1ll << 40), DS_EUNSUPPORTED,
               "ds_linear_shuffle: shape too large for the index arithmetic");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_linear_shuffle: dtype must be f16 or bf16");
    DS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0), DS_EINVAL,
               "ds_linear_shuffle: x, w, y and bias must be 16-byte aligned");
    LinParams P;
    memset(&P, 0, sizeof(P));
    P.x = x; P.w = w; P.bias = bias; P.y = y;
    P.M = (int)pixels; P.N = (int)n; P.K = (int)in_features;
    P.nbm = (int)((pixels + 255) / 256); P.nbn = (int)(n / 256);
    P.ldy = n;
    P.ps_w = width; P.ps_s = stride; P.ps_c = out_channels;
    P.ps_w_magic = (unsigned)((1ull << 32) / (unsigned long long)width + 1ull);
    P.ps_c_magic = (unsigned)((1ull << 32) / (unsigned long long)out_channels + 1ull);
    P.kt_kind = DS_KT_LINEAR_SHUFFLE;
    hipStream_t st = (hipStream_t)stream;
    return dtype == DS_DTYPE_F16 ? ln_launch<0, 0, 0, 0, 2>(ctx, P, st) : ln_launch<1, 0, 0, 0, 2>(ctx, P, st);
}

// The read-out of the reassemble stage (ProjectReadout, dmidas/backbones/utils.py:28-39: cat(token, cls) -> Linear(2C -> C) -> GELU,
// then the Transpose / Unflatten of :165-169) as ONE GEMM on the padded token sequence the encoder leaves behind:
//     y[b, t - 1, :] = GELU( x[b, t, :] . w_tok^T + cls_vec[b, :] )      for the tokens t = 1 .. tokens - 1 of image b
// with cls_vec[b] = w_cls . x[b, 0] + bias (one small GEMM on the host side of the C ABI).  The cls row and the pad rows of every
// image are computed like any other row and stored to the dummy row behind the output, so the store count of a tile is constant
// (the early mode's counted waits rely on it): y must hold images * (tokens - 1) + 1 rows.
DS_API int ds_linear_readout(ds_ctx *ctx, const void *x, const void *w_tok, const void *cls_vec, void *y, int64_t images,
                             int64_t tokens_padded, int64_t tokens, int64_t out_features, int64_t in_features, int dtype, void *stream)
{
    DS_REQUIRE(ctx && x && w_tok && cls_vec && y, DS_EINVAL, "ds_linear_readout: null argument");
    DS_REQUIRE(images > 0 && tokens >= 2 && tokens <= tokens_padded, DS_EINVAL, "ds_linear_readout: need 2 <= tokens <= tokens_padded");
    const int64_t rows = images * tokens_padded;
    DS_REQUIRE(rows >= 256 && rows < (1ll << 31) - 256 && rows * tokens_padded < (1ll << 32), DS_EINVAL,
               "ds_linear_readout: images * tokens_padded must be in [256, 2^31) and images * tokens_padded^2 below 2^32");
    DS_REQUIRE(out_features > 0 && out_features % 256 == 0, DS_EINVAL, "ds_linear_readout: out_features must be a multiple of 256");
    DS_REQUIRE(in_features >= 128 && in_features % 128 == 0 && in_features <= 16384, DS_EINVAL,
               "ds_linear_readout: in_features must be a multiple of 128 (<= 16384)");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_linear_readout: dtype must be f16 or bf16");
    DS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_tok & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)cls_vec & 15) == 0, DS_EINVAL,
               "ds_linear_readout: x, w_tok, cls_vec and y must be 16-byte aligned");
    LinParams P;
    memset(&P, 0, sizeof(P));
    P.x = x; P.w = w_tok; P.res1 = cls_vec; P.y = y;
    P.M = (int)rows; P.N = (int)out_features; P.K = (int)in_features;
    P.nbm = (int)((rows + 255) / 256); P.nbn = (int)(out_features / 256);
    P.ldy = out_features;
    P.rd_np = (int)tokens_padded; P.rd_n = (int)tokens; P.rd_dummy = (int)(images * (tokens - 1));
    P.rd_magic = (unsigned)((1ull << 32) / (unsigned long long)tokens_padded + 1ull);
    P.kt_kind = DS_KT_LINEAR_READOUT;
    hipStream_t st = (hipStream_t)stream;
    return dtype == DS_DTYPE_F16 ? ln_launch<0, 1, 0, 1, 3>(ctx, P, st) : ln_launch<1, 1, 0, 1, 3>(ctx, P, st);
}

// LayerNorm folded into the Linear that follows it (the norm1 -> qkv and norm2 -> fc1 pairs of every encoder block: timm's Block as
// run by dmidas/backbones/beit.py:94-107, ddepth_anything_v2/depth_anything_v2/dinov2_layers/block.py:82-107): with
// W' = W diag(ln_weight), colsum[n] = sum_k W'[n][k] and b' = b + W . ln_bias (all three prepared once per module by the host),
//     LN(x) . W^T + b = rstd[m] (x . W'^T)[m][n] - mean[m] rstd[m] colsum[n] + b'[n],
// so the GEMM reads the residual stream x itself and the LayerNorm pass (read x, write the normalised copy) is replaced by
// ds_row_stats (read x, write 8 bytes per token).  act: 0 none, 1 erf-GELU.
DS_API int ds_linear_ln(ds_ctx *ctx, const void *x, const void *w_scaled, const float *colsum, const void *bias, const void *stats, void *y,
                        int64_t rows, int64_t out_features, int64_t in_features, int64_t ldy, int act, int dtype, void *stream)
{
    DS_REQUIRE(ctx && x && w_scaled && colsum && stats && y, DS_EINVAL, "ds_linear_ln: null argument");
    DS_REQUIRE(rows >= 256 && rows < (1ll << 31) - 256, DS_EINVAL, "ds_linear_ln: rows must be >= 256 (one tile)");
    DS_REQUIRE(out_features > 0 && out_features % 256 == 0, DS_EINVAL, "ds_linear_ln: out_features must be a multiple of 256");
    DS_REQUIRE(in_features >= 128 && in_features % 128 == 0 && in_features <= 16384, DS_EINVAL, "ds_linear_ln: in_features must be a multiple of 128 (<= 16384)");
    DS_REQUIRE(ldy >= out_features && ldy % 8 == 0, DS_EINVAL, "ds_linear_ln: ldy must be >= out_features and a multiple of 8");
    DS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_scaled & 15) == 0 && ((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) &&
               ((uintptr_t)colsum & 15) == 0 && ((uintptr_t)stats & 15) == 0, DS_EINVAL, "ds_linear_ln: operands must be 16-byte aligned");
    DS_REQUIRE(act == 0 || act == 1, DS_EINVAL, "ds_linear_ln: act must be 0 (none) or 1 (erf-GELU)");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_linear_ln: dtype must be f16 or bf16");
    LinParams P;
    memset(&P, 0, sizeof(P));
    P.x = x; P.w = w_scaled; P.bias = bias; P.y = y; P.ln_stats = (const float2 *)stats; P.ln_colsum = colsum;
    P.M = (int)rows; P.N = (int)out_features; P.K = (int)in_features;
    P.nbm = (int)((rows + 255) / 256); P.nbn = (int)(out_features / 256);
    P.ldy = ldy;
    P.kt_kind = act == 1 ? DS_KT_LINEAR_GELU : DS_KT_LINEAR;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DS_DTYPE_F16) return act == 1 ? ln_launch<0, 5, 0, 0>(ctx, P, st) : ln_launch<0, 4, 0, 0>(ctx, P, st);
    return act == 1 ? ln_launch<1, 5, 0, 0>(ctx, P, st) : ln_launch<1, 4, 0, 0>(ctx, P, st);
}

// ds_linear_vt with the LayerNorm folded in: vt[b][c][n] = rstd[b, n] (w_v' . x[b, n]) - mean[b, n] rstd[b, n] colsum[c]; the constant
// W_v . ln_bias (like the V bias) commutes with the attention and is folded into the output projection's bias by the host.
DS_API int ds_linear_vt_ln(ds_ctx *ctx, const void *w_v_scaled, const float *colsum, const void *x, const void *stats, void *vt,
                           int64_t channels, int64_t batch, int64_t tokens, int64_t in_features, int dtype, void *stream)
{
    DS_REQUIRE(ctx && w_v_scaled && colsum && x && stats && vt, DS_EINVAL, "ds_linear_vt_ln: null argument");
    DS_REQUIRE(channels >= 256, DS_EINVAL, "ds_linear_vt_ln: channels must be >= 256 (one tile)");
    DS_REQUIRE(batch > 0 && tokens > 0 && tokens % 8 == 0 && (batch * tokens) % 256 == 0 && batch * tokens < (1ll << 31) - 256, DS_EINVAL,
               "ds_linear_vt_ln: tokens must be a multiple of 8 and batch * tokens a multiple of 256");
    DS_REQUIRE(batch * tokens * tokens < (1ll << 32), DS_EUNSUPPORTED, "ds_linear_vt_ln: batch * tokens^2 must stay below 2^32");
    DS_REQUIRE(in_features >= 128 && in_features % 128 == 0 && in_features <= 16384, DS_EINVAL, "ds_linear_vt_ln: in_features must be a multiple of 128 (<= 16384)");
    DS_REQUIRE(dtype == DS_DTYPE_F16 || dtype == DS_DTYPE_BF16, DS_EINVAL, "ds_linear_vt_ln: dtype must be f16 or bf16");
    DS_REQUIRE(((uintptr_t)w_v_scaled & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)vt & 15) == 0 && ((uintptr_t)colsum & 15) == 0 &&
               ((uintptr_t)stats & 15) == 0, DS_EINVAL, "ds_linear_vt_ln: operands must be 16-byte aligned");
    LinParams P;
    memset(&P, 0, sizeof(P));
    P.x = w_v_scaled; P.w = x; P.y = vt; P.ln_stats = (const float2 *)stats; P.ln_colsum = colsum;
    P.M = (int)channels; P.N = (int)(batch * tokens); P.K = (int)in_features;
    P.nbm = (int)((channels + 255) / 256); P.nbn = (int)(batch * tokens / 256);
    P.ldy = batch * tokens;
    P.vt_np = (int)tokens; P.vt_c = (int)channels; P.vt_magic = (unsigned)((1ull << 32) / (unsigned long long)tokens + 1ull);
    P.kt_kind = DS_KT_LINEAR_VT;
    hipStream_t st = (hipStream_t)stream;
    return dtype == DS_DTYPE_F16 ? ln_launch<0, 4, 0, 0, 1>(ctx, P, st) : ln_launch<1, 4, 0, 0, 1>(ctx, P, st);
}
