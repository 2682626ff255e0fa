// apply_stereo_divergence_polylines (reference: src/stereoimage_generation.py:162-283) on gfx950.
//
// The reference sweeps every image row sequentially: it morphs the row into a polyline of points
// (x = col + 0.5 + d + sep [+-0.45], closeness = |d|, colour index = col), sorts the points, and for
// every output pixel walks the sub-intervals between consecutive sorted points, keeping an "active
// set" of segments and taking the colour of the closest one.
//
// Here every OUTPUT PIXEL is evaluated independently (one lane per pixel).  Points stay in ORIGINAL
// order in one LDS array pt[q]; segment q runs from pt[q-1] to pt[q].
//
//   * SIMPLE pixels.  Crossings of a vertical line x = t by the polyline (a continuous path from the
//     -w sentinel to the 2w sentinel) alternate forward, backward, forward, ...  If no backward (or
//     zero-length) segment overlaps the closed strip [col, col+1] the path is monotone inside it: the
//     segments overlapping the strip are a contiguous run of forward segments, the sub-intervals of
//     the pixel are exactly their pieces [max(x0,col), min(x1,col+1)] in that order, and each
//     sub-interval has exactly one active segment, which the reference takes without looking at its
//     closeness (csg_end == 1, :259).  The run starts at the one forward segment that ENTERS the pixel
//     (floor(x0) < col <= floor(x1)); entering segments write their index into a per-pixel table while
//     the points are computed (one LDS max per pixel entered), backward segments write a BAD mark into
//     the same table, so "simple" costs the pixel one LDS read: no sort, no search, no binning pass.
//   * GENERAL pixels (BAD: folds at occlusion boundaries, noisy depth; or more than 6 pieces) collect
//     the forward segments that overlap the strip and the points inside it from a conservative source
//     window (the wave tests 64 candidates at a time and hands the pixel's lane two bit masks), visit the
//     points in sorted order, and apply the reference's rule per sub-interval: active <=> x0 < c and
//     not (x1 < c), which is the reference's active set as long as the centres c are non-decreasing
//     along the row, i.e. every sub-interval has positive length.  The winner rule (strict '>' on
//     interpolated closeness, 0 < ip < 1) is order independent unless two valid candidates tie exactly
//     or none is valid.
// A pixel that hits a history-dependent situation (non-positive significance, empty active set,
// exact tie, no valid candidate among >= 2, list overflow, NaN) raises a flag for its ROW, and flagged
// rows are re-rendered by k_polylines_exact: a statement-by-statement sequential transliteration of
// the reference (one lane per row).  The result is therefore bit-identical to the reference for every
// input; the fallback only costs time.  All arithmetic is IEEE binary64, compiled with
// -ffp-contract=off, in the reference's operation order.
//
// Mapping to the machine.  A 256-thread workgroup is persistent and walks (image, row, supertile) work
// items; a supertile is S output pixels (S = the whole row when it fits) rendered for BOTH eyes from one
// staging of what the eyes share:
//   P01  each thread takes 4 consecutive source columns of the window: one 8-byte depth load + one
//        12-byte pixel load, float64 normalisation (one division per column), both eyes' points written
//        to LDS as aligned 16-byte pairs, entry table updated.                          -- barrier --
//   P2   one lane per output pixel and eye: entry table -> 7 point coordinates + 4 packed RGBX words
//        from LDS -> up to 6 predicated pieces -> bytes into an LDS output row.  General pixels are APPENDED TO A
//        QUEUE IN HBM (one private segment per persistent workgroup: no global atomics) and rendered afterwards by
//        k_polylines_general, densely packed 8 per wave.  Round 1 rendered them inside this kernel (phase P3, one
//        wave at work, three waiting at an extra barrier): with a network's noisy prediction almost every work item
//        holds a few such pixels, and that serial phase cost 1 ms of the 2.4 ms launch.
//   S0   (start of the next work item) the finished output rows leave LDS as 16-byte stores.
// HBM sees each input byte once per supertile (+ the window halo) and each output byte once.
#include <stdlib.h>
#include <atomic>

#include "ds_common.h"

#define PL_THREADS 256
#define PL_EPS 1e-7
#define PL_BAD 0x40000000u
#define PL_KMAX 4            // general pixels: candidate segments are examined 64 at a time, up to PL_KMAX words
#define PL_PT_PAD 10         // pt[] entries past the last real point (tail sentinel + fast-path over-read)

// timing ablations that produce WRONG pixels exist only in -DDS_EXPERIMENTS builds
#ifdef DS_EXPERIMENTS
#define PL_DBG() (P.dbg)
#else
#define PL_DBG() 0
#endif

struct PolyParams {
    const uint8_t *img;
    const void *depth;
    const double *minmax;      // n * {min,max}
    const double *lut;         // optional n*65536 table of norm**exponent (uint16 depth only)
    int depth_dtype;
    int n, h, w;
    int n_eyes;
    int S;                     // supertile: output pixels per work item (multiple of 64)
    int nwmax;                 // capacity of the source window in columns (multiple of 4)
    int al4;                   // rows start 4-element aligned: vector loads allowed
    int nseg;                  // candidate segments per general pixel (max over the eyes)
    int K;                     // 64-segment words covering the per-pixel candidate window (> PL_KMAX: general pixels go to the exact sweep)
    double div_px[2], sep_px[2];
    uint8_t *out[2];
    int64_t ors[2], ois[2];
    int offL[2], offU[2];      // per-pixel source-column window [col+offL, col+offU]
    int *row_flags;            // one int per (image, eye, row)
    int *row_list;             // flagged rows, compacted
    int *counters;             // [0] = number of flagged rows, [1] = number of general pixels (statistics)
    unsigned long long *gq;    // queue of general pixels: one segment of gq_cap entries per workgroup of k_polylines
    int *gq_count;             // entries in each segment
    int gq_cap, gq_segments;   // entry = (row id << 32) | column, row id = (image * n_eyes + eye) * h + row
    int dbg;                   // -DDS_EXPERIMENTS builds only: DS_PL_DEBUG ablation knob (0 = off); results are WRONG when set
    unsigned long long *prof;  // optional (DS_PL_PROF=1): 8 cycle accumulators, wave 0 of every workgroup
};

// coord_d of stereoimage_generation.py:182 for one depth element (used by the exact kernel)
template <int DT>
__device__ __forceinline__ double pl_coord_d(const PolyParams &P, int img, const void *depth_row, int col, double mn, double mx, double div_px)
{
    typedef typename ds_depth_traits<DT>::T T;
    const T v = ((const T *)depth_row)[col];
    double nd;
    if (DT == DS_DEPTH_U16 && P.lut != nullptr) nd = P.lut[(size_t)img * 65536 + (unsigned)v];   // norm ** exponent
    else nd = ds_depth_traits<DT>::norm(v, mn, mx);                                            // exponent == 1.0: pow(x, 1.0) == x
    return nd * div_px;
}

__device__ __forceinline__ void pl_flag_row(const PolyParams &P, int img, int eye, int row)
{
    const int rowid = (img * P.n_eyes + eye) * P.h + row;
    if (atomicExch(&P.row_flags[rowid], 1) == 0) {
        const int slot = atomicAdd(&P.counters[0], 1);
        P.row_list[slot] = rowid;
    }
}

// float64 -> uint8 for a colour accumulator that is known to be finite and small
__device__ __forceinline__ uint32_t pl_u8(double v) { return (uint32_t)(int)v & 0xffu; }

// ---- P01 helpers ------------------------------------------------------------------------------------
// Normalised depth (:79-81) of 4 consecutive columns j..j+3 (the first nv are real) and of column j-1.
//
// uint16: numerator a = v - min and denominator b = max - min are integers <= 65535.  With y = 1/b (one correctly
// rounded division per thread and work item), q0 = a*y, r = fma(-b, q0, a), q = fma(r, y, q0) IS the correctly
// rounded a/b for every such pair -- checked exhaustively (all 2^32 pairs) by the test suite (check_u16_division.c) -- so the
// per-column float64 division costs 3 instructions instead of the ~11 of the generic expansion.
__device__ __forceinline__ void pl_load_nd(const PolyParams &P, const void *depth_row, int j, int nv, bool hasprev,
                                           double mn, double mx, const double *lut, double *nd, double &ndp)
{
    if (P.depth_dtype == DS_DEPTH_U16) {
        const uint16_t *d = (const uint16_t *)depth_row;
        uint32_t v[4] = { 0, 0, 0, 0 };
        if (nv == 4 && P.al4) {
            const uint2 raw = *reinterpret_cast<const uint2 *>(d + j);
            v[0] = raw.x & 0xffffu; v[1] = raw.x >> 16; v[2] = raw.y & 0xffffu; v[3] = raw.y >> 16;
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) if (m < nv) v[m] = d[j + m];
        }
        const uint32_t vp = hasprev ? d[j - 1] : 0u;
        if (lut != nullptr) {                                                       // norm ** exponent, host-built
#pragma unroll
            for (int m = 0; m < 4; m++) nd[m] = lut[v[m]];
            ndp = lut[vp];
        } else {
            const uint32_t mn16 = (uint32_t)mn & 0xffffu;
            const double b = (double)(((uint32_t)mx - mn16) & 0xffffu);
            const double y = 1.0 / b;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const double a = (double)((v[m] - mn16) & 0xffffu);                 // uint16 subtraction (:81)
                const double q0 = a * y;
                nd[m] = fma(fma(-b, q0, a), y, q0);
            }
            const double a = (double)((vp - mn16) & 0xffffu);
            const double q0 = a * y;
            ndp = fma(fma(-b, q0, a), y, q0);
        }
    } else if (P.depth_dtype == DS_DEPTH_F32) {
        const float *d = (const float *)depth_row;
        float v[4] = { 0.f, 0.f, 0.f, 0.f };
        if (nv == 4 && P.al4) {
            const float4 raw = *reinterpret_cast<const float4 *>(d + j);
            v[0] = raw.x; v[1] = raw.y; v[2] = raw.z; v[3] = raw.w;
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) if (m < nv) v[m] = d[j + m];
        }
        const float vp = hasprev ? d[j - 1] : 0.f;
        const float mnf = (float)mn, den = (float)mx - (float)mn;
#pragma unroll
        for (int m = 0; m < 4; m++) nd[m] = (double)((v[m] - mnf) / den);
        ndp = (double)((vp - mnf) / den);
    } else {
        const double *d = (const double *)depth_row;
        double v[4] = { 0., 0., 0., 0. };
        if (nv == 4 && P.al4) {
            const double2 a = *reinterpret_cast<const double2 *>(d + j), b = *reinterpret_cast<const double2 *>(d + j + 2);
            v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) if (m < nv) v[m] = d[j + m];
        }
        const double vp = hasprev ? d[j - 1] : 0.;
        const double den = mx - mn;
#pragma unroll
        for (int m = 0; m < 4; m++) nd[m] = (v[m] - mn) / den;
        ndp = (vp - mn) / den;
    }
}

// Pixel bytes of 4 consecutive columns -> one RGBX word per column (channel k in byte k).
template <int C>
__device__ __forceinline__ void pl_load_rgbx(const uint8_t *src, int nv, bool al4, uint32_t *rgbx)
{
    uint32_t raw[C];
    if (nv == 4 && al4) {
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
#pragma unroll
        for (int k = 0; k < C; k++) raw[k] = s32[k];
    } else {
#pragma unroll
        for (int k = 0; k < C; k++) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) if ((k * 4 + b) < nv * C) v |= (uint32_t)src[k * 4 + b] << (8 * b);
            raw[k] = v;
        }
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < C; k++) {
            const int b = m * C + k;
            v |= ((raw[b >> 2] >> (8 * (b & 3))) & 0xffu) << (8 * k);
        }
        rgbx[m] = v;
    }
}

// Entry table.  A forward segment q = (x0 -> x1) ENTERS pixels floor(x0)+1 .. floor(x1); a backward or zero-length
// one marks every pixel it touches as BAD.  g0 = table of the current eye, indexed by pixel - c0; f0/f1 = floors.
// Common case first (a forward segment that enters exactly one pixel, or none); everything else is the slow path.
__device__ __forceinline__ void pl_scatter_slow(uint32_t *g0, double x0, double x1, int f0, int f1, uint32_t q, int c0, int tn)
{
    if (x0 < x1) {
        const int lo = max(f0 + 1, c0) - c0;
        const int hi = min(f1, c0 + tn - 1) - c0;
        for (int p = lo; p <= hi; p++) atomicMax(&g0[p], q);
    } else {
        const int lo = max(min(f0, f1), c0) - c0;
        const int hi = min(max(f0, f1), c0 + tn - 1) - c0;
        for (int p = lo; p <= hi; p++) atomicMax(&g0[p], PL_BAD);
    }
}
// returns true when the segment needs the slow path (the caller collects those and runs them in one shared loop)
__device__ __forceinline__ bool pl_scatter(uint32_t *g0, double x0, double x1, int f0, int f1, uint32_t q, int c0, int tn)
{
    const bool fwd = x0 < x1;
    const int d = f1 - f0;
    if (fwd && d == 1) {
        const unsigned p = (unsigned)(f1 - c0);
        if (p < (unsigned)tn) atomicMax(&g0[p], q);
    }
    return !(fwd && (unsigned)d <= 1u);
}
__device__ __forceinline__ int pl_floor_i(double x)
{
    // v_floor_f64 + v_cvt_i32_f64 (saturating).  Rows with |x| >= 4e9 or NaN are re-rendered by the exact sweep, and
    // every loop over pixels is clipped to the tile, so an out-of-range value cannot do harm here.
    return (int)floor(x);
}

// ---- P2/P3 helpers -----------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ void pl_unpack(uint32_t v, double *px)
{
#pragma unroll
    for (int k = 0; k < C; k++) px[k] = (double)((v >> (8 * k)) & 0xffu);
}

template <int C>
__device__ __forceinline__ void pl_add_flat(const double *px, double significance, double *color)
{
#pragma unroll
    for (int k = 0; k < C; k++) color[k] += px[k] * significance;                    // :273
}

template <int C>
__device__ __forceinline__ void pl_add_lerp(const double *pl, const double *pr, double x0, double x1, double coord_center,
                                            double significance, double *color)
{
    const double ip_k = (coord_center - x0) / (x1 - x0);                             // :276
    const double om = 1.0 - ip_k;
#pragma unroll
    for (int k = 0; k < C; k++) {
        const double u = pl[k] * om;
        const double v = pr[k] * ip_k;
        color[k] += (u + v) * significance;                                          // :277-279
    }
}

// The reference's sub-interval geometry (:235-239).  0.5*significance is exact, so the fused form of
// coord_from + 0.5*significance rounds exactly like the reference's two operations.
struct PlSub { double significance, coord_center; };
__device__ __forceinline__ PlSub pl_sub(double a, double b)
{
    const double coord_from = a + PL_EPS;                                            // :235
    const double coord_to = b - PL_EPS;                                              // :236
    PlSub s;
    s.significance = coord_to - coord_from;                                          // :237
    s.coord_center = fma(0.5, s.significance, coord_from);                           // :239
    return s;
}

// what P2/P3 need to know about the staged window
struct PlWin {
    const double *pt;          // this eye's points, pt[0] .. pt[NP*ncols + 1]
    const double *nd;          // normalised depth per window column
    const uint32_t *rgbx;      // rgbx[i] = window column i, rgbx[-1] exists
    int ncols;
    bool head, tail;
    double div_px;
};

// colour index (window column) of point q: the sentinels carry the edge columns (:179,:191)
template <int NP>
__device__ __forceinline__ int pl_col_of(const PlWin &V, int q)
{
    if (q == 0) return V.head ? 0 : -1;
    if (q == NP * V.ncols + 1) return V.ncols - 1;
    return (q - 1) >> (NP - 1);
}
// closeness |coord_d| of point q (:184,:188-189); 0 for the sentinels
template <int NP>
__device__ __forceinline__ double pl_dd_of(const PlWin &V, int q)
{
    if ((q == 0 && V.head) || q == NP * V.ncols + 1) return 0.0;
    return fabs(V.nd[q == 0 ? -1 : ((q - 1) >> (NP - 1))] * V.div_px);
}

// colour of segment q over one sub-interval (:270-279)
template <int C, int NP>
__device__ __forceinline__ void pl_add_segment(const PlWin &V, int q, double x0, double x1, const PlSub &s, double *color)
{
    const int cl = pl_col_of<NP>(V, q - 1), cr = pl_col_of<NP>(V, q);               // :270-271
    double pl[C], pr[C];
    pl_unpack<C>(V.rgbx[cr], pr);
    if (cl == cr) pl_add_flat<C>(pr, s.significance, color);                         // :272
    else { pl_unpack<C>(V.rgbx[cl], pl); pl_add_lerp<C>(pl, pr, x0, x1, s.coord_center, s.significance, color); }
}

// One piece of segment q inside pixel [fq, fq1], given that it is the only active segment
template <int C, int NP>
__device__ __forceinline__ void pl_piece(const PlWin &V, int q, double x0, double x1, double fq, double fq1, double *color, int &flag)
{
    const PlSub s = pl_sub(fmax(x0, fq), fmin(x1, fq1));                             // max(col, pt[pt_i][0]), min(col+1, pt[pt_i+1][0])
    if (!(s.significance > 0.0)) flag = 1;
    pl_add_segment<C, NP>(V, q, x0, x1, s, color);
}

// GENERAL pixel.  fm/bm: bit t of word k <=> segment / point q = qlo + 64k + t is a forward segment overlapping the
// closed strip [fq, fq1] / a point inside [fq, fq1).  Sub-intervals are visited in (x, original index) order of the
// points -- the order of the reference's stable insertion sort (:214-219).
template <int C, int NP, int KW = PL_KMAX>
__device__ __forceinline__ void pl_general_pixel_lds(const PlWin &V, int qlo, int K, const unsigned long long *fm, unsigned long long *bm,
                                                  double fq, double fq1, double *color, int &flag)
{
    double a = fq;
    for (;;) {
        // next point: smallest (x, q) among the remaining ones
        double b = fq1;
        int bk = -1, bt = 0;
#pragma unroll
        for (int k = 0; k < KW; k++) {
            if (k < K) {
                unsigned long long m = bm[k];
                while (m != 0ull) {
                    const int t = __ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    const double x = V.pt[qlo + 64 * k + t];
                    if (bk < 0 || x < b) { b = x; bk = k; bt = t; }
                }
            }
        }
        const bool more = bk >= 0;
        if (more) {
#pragma unroll
            for (int k = 0; k < KW; k++) if (k == bk) bm[k] &= ~(1ull << bt);
        } else {
            b = fq1;
        }
        const PlSub s = pl_sub(a, b);
        a = b;
        if (!(s.significance > 0.0)) flag = 1;     // centres may stop being monotone: history dependent
        // active set = { q : x0 < c and not (x1 < c) }  (:242-253); winner by interpolated closeness (:259-268)
        int count = 0, first = -1, win = -1;
        double best = -PL_EPS;                                                       // :261
        bool have = false;
#pragma unroll
        for (int k = 0; k < KW; k++) {
            if (k < K) {
                unsigned long long m = fm[k];
                while (m != 0ull) {
                    const int t = __ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    const int q = qlo + 64 * k + t;
                    const double x0 = V.pt[q - 1], x1 = V.pt[q];
                    if (x0 < s.coord_center && !(x1 < s.coord_center)) {
                        if (count == 0) first = q;
                        count++;
                        const double d0 = pl_dd_of<NP>(V, q - 1), d1 = pl_dd_of<NP>(V, q);
                        const double ip_k = (s.coord_center - x0) / (x1 - x0);       // :263
                        const double closeness = (1.0 - ip_k) * d0 + ip_k * d1;      // :265
                        const bool valid = 0.0 < ip_k && ip_k < 1.0;
                        if (valid && have && closeness == best) flag = 1;            // exact tie: csg order decides
                        if (best < closeness && valid) { best = closeness; win = q; have = true; }   // :266
                    }
                }
            }
        }
        if (count == 1) win = first;                                                 // :259
        if (count == 0 || (count > 1 && !have)) flag = 1;   // reference reads a stale / falls back to csg[0]
        else pl_add_segment<C, NP>(V, win, V.pt[win - 1], V.pt[win], s, color);
        if (!more) break;
    }
}


// SIMPLE sharp pixel, straight line: the run starts at segment g0 and is assumed to have at most 6 pieces inside the
// pairs { gap, body } of window columns ib, ib+1 (straight line) and ib+2 (wave-uniform branch).  Every slot is computed and selected
// (no branches: the divisions of the two gaps and of the other eye's pixel overlap).  Returns false when the pixel needs
// the walker instead (a fifth piece, or a sentinel segment in the run); color is then meaningless.
template <int C>
__device__ __forceinline__ bool pl_simple4(const double *pt, const uint32_t *rgbx, uint32_t g0, bool ok, bool head, int qT,
                                           double fq, double fq1, double *color, int &flag)
{
    const int ib = ok ? (int)((g0 - 1) >> 1) : 0;
    const double *pp = pt + 2 * ib;
    const double2 p01 = *reinterpret_cast<const double2 *>(pp);
    const double2 p23 = *reinterpret_cast<const double2 *>(pp + 2);
    const double x4 = pp[4];
    double px0[C], px1[C], px2[C];
    pl_unpack<C>(rgbx[ib - 1], px0);
    pl_unpack<C>(rgbx[ib], px1);
    pl_unpack<C>(rgbx[ib + 1], px2);
    const double x0 = p01.x, x1 = p01.y, x2 = p23.x, x3 = p23.y;
    const bool a0 = ok && (g0 & 1u) != 0;                  // gap of column ib
    const bool a1 = ok && (!(g0 & 1u) || x1 < fq1);        // body of column ib
    const bool a2 = a1 && x2 < fq1;                        // gap of column ib+1
    const bool a3 = a2 && x3 < fq1;                        // body of column ib+1
    const bool a4 = a3 && x4 < fq1;                        // gap of column ib+2 (rare: compressed stretches)
    const int q0 = 2 * ib + 1;
    bool sentinel = (a0 && ((q0 == 1 && head) || q0 == qT)) || (a2 && q0 + 2 == qT) || (a4 && q0 + 4 == qT);
    // geometry of the four pieces (:235-239)
    const PlSub s0 = pl_sub(fmax(x0, fq), fmin(x1, fq1));
    const PlSub s2 = pl_sub(fmax(x2, fq), fmin(x3, fq1));
    const double sg1 = (fmin(x2, fq1) - PL_EPS) - (fmax(x1, fq) + PL_EPS);
    const double sg3 = (fmin(x4, fq1) - PL_EPS) - (fmax(x3, fq) + PL_EPS);
    const double ip0 = (s0.coord_center - x0) / (x1 - x0), om0 = 1.0 - ip0;          // :276
    const double ip2 = (s2.coord_center - x2) / (x3 - x2), om2 = 1.0 - ip2;
    if ((a0 && !(s0.significance > 0.0)) || (a1 && !(sg1 > 0.0)) || (a2 && !(s2.significance > 0.0)) || (a3 && !(sg3 > 0.0))) flag = 1;
#pragma unroll
    for (int k = 0; k < C; k++) {
        const double t0 = (px0[k] * om0 + px1[k] * ip0) * s0.significance;           // :277-279
        const double t1 = px1[k] * sg1;                                              // :273
        const double t2 = (px1[k] * om2 + px2[k] * ip2) * s2.significance;
        const double t3 = px2[k] * sg3;
        double c = color[k];
        c = a0 ? c + t0 : c;
        c = a1 ? c + t1 : c;
        c = a2 ? c + t2 : c;
        c = a3 ? c + t3 : c;
        color[k] = c;
    }
    bool more = false;
    if (__any(a4)) {                                       // wave-uniform: the pair of column ib+2
        const double x5 = pp[5], x6 = pp[6];
        double px3[C];
        pl_unpack<C>(rgbx[ib + 2], px3);
        const bool a5 = a4 && x5 < fq1;
        more = a5 && x6 < fq1;
        const PlSub s4 = pl_sub(fmax(x4, fq), fmin(x5, fq1));
        const double sg5 = (fmin(x6, fq1) - PL_EPS) - (fmax(x5, fq) + PL_EPS);
        const double ip4 = (s4.coord_center - x4) / (x5 - x4), om4 = 1.0 - ip4;
        if ((a4 && !(s4.significance > 0.0)) || (a5 && !(sg5 > 0.0))) flag = 1;
#pragma unroll
        for (int k = 0; k < C; k++) {
            const double t4 = (px2[k] * om4 + px3[k] * ip4) * s4.significance;
            const double t5 = px3[k] * sg5;
            double c = color[k];
            c = a4 ? c + t4 : c;
            c = a5 ? c + t5 : c;
            color[k] = c;
        }
    }
    return !(more || sentinel);
}

// SIMPLE pixel, any number of pieces: walk the run from the entering segment g0 (:234-280 with a single active segment)
template <int C, int NP>
__device__ __forceinline__ void pl_walk(const PlWin &V, uint32_t g0, double fq, double fq1, double *color, int &flag)
{
    int q = (int)g0;
    const int qT = NP * V.ncols + 1;
    double x0 = V.pt[q - 1];
#pragma unroll
    for (int k = 0; k < C; k++) color[k] = 0.5;                                      // :229
    for (;;) {
        const double x1 = V.pt[q];
        pl_piece<C, NP>(V, q, x0, x1, fq, fq1, color, flag);
        if (!(x1 < fq1)) break;
        if (q >= qT) { flag = 1; break; }
        x0 = x1; q++;
    }
}

// LDS carve-up, shared by the kernel and the host-side size computation
struct PlLds {
    int npt, off_nd, off_rgbx, off_g0, off_out, outstride, off_queue, off_misc, total;
};
__host__ __device__ inline PlLds pl_lds_layout(int S, int nwmax, int c, int np)
{
    PlLds L;
    L.npt = np * nwmax + PL_PT_PAD;                       // doubles per eye (even)
    L.off_nd = 2 * L.npt * 8;                              // nd[-4 .. nwmax): 4 doubles of lead-in (nd[-1] is used)
    L.off_rgbx = L.off_nd + (nwmax + 4) * 8;               // rgbx[-4 .. nwmax + 4)
    L.off_g0 = L.off_rgbx + (nwmax + 8) * 4;
    L.off_out = L.off_g0 + 2 * S * 4;
    L.outstride = (S * c + 15) & ~15;
    L.off_queue = L.off_out + 2 * L.outstride;
    L.off_misc = L.off_queue;                              // (the round-1 in-LDS queue of general pixels is gone)
    L.total = L.off_misc + 8 * 4;
    return L;
}

template <int C, int SHARP, int NE>
__global__ __launch_bounds__(PL_THREADS, 5) void k_polylines(PolyParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NP = SHARP ? 2 : 1;
    const int tid = threadIdx.x;
    const int w = P.w, S = P.S;
    const PlLds L = pl_lds_layout(S, P.nwmax, C, NP);
    double *s_pt = reinterpret_cast<double *>(smem);                         // [eye][npt]
    double *s_nd = reinterpret_cast<double *>(smem + L.off_nd) + 4;
    uint32_t *s_rgbx = reinterpret_cast<uint32_t *>(smem + L.off_rgbx) + 4;
    uint32_t *s_g0 = reinterpret_cast<uint32_t *>(smem + L.off_g0);          // [eye][S]
    uint8_t *s_out = smem + L.off_out;                                       // [eye][outstride]
    int *s_misc = reinterpret_cast<int *>(smem + L.off_misc);                // [0]: entries in this workgroup's queue segment, [2],[3]: row flag per eye

    // per-eye constants pinned in scalar registers: without this the compiler re-reads them from the kernel-argument
    // segment inside the per-column code (s_load + s_waitcnt lgkmcnt(0) in the hot path)
    double k_div[NE], k_sep[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) {
        int dl = __double2loint(P.div_px[e]), dh = __double2hiint(P.div_px[e]);
        int sl = __double2loint(P.sep_px[e]), sh = __double2hiint(P.sep_px[e]);
        asm volatile("" : "+s"(dl), "+s"(dh), "+s"(sl), "+s"(sh));
        k_div[e] = __hiloint2double(dh, dl);
        k_sep[e] = __hiloint2double(sh, sl);
    }
    const int tiles = (w + S - 1) / S;
    const int nwork = P.n * P.h * tiles;                    // < 2^31, checked by the host
    int uL = P.offL[0], uU = P.offU[0];                    // union of the two eyes' window offsets
    if (NE > 1) { uL = min(uL, P.offL[1]); uU = max(uU, P.offU[1]); }

    for (int i = tid; i < 2 * S; i += PL_THREADS) s_g0[i] = 0;
    if (tid < 8) s_misc[tid] = 0;
    __syncthreads();

    int prev = -1, prev_img = 0, prev_row = 0, prev_c0 = 0;
#ifdef DS_PL_PROFILE        // build with -DDS_PL_PROFILE to get per-phase cycle counts (DS_PL_PROF=1 at run time)
    unsigned long long tacc[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, tl = P.prof ? __builtin_readcyclecounter() : 0ull;
#define PL_TICK(i) do { if (P.prof) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tl; tl = t_; } } while (0)
#define PL_TICKW(w, i) do { if (P.prof) { asm volatile(w ::: "memory"); PL_TICK(i); } } while (0)
#else
#define PL_TICK(i) do { } while (0)
#define PL_TICKW(w, i) do { } while (0)
#endif
    for (int work = blockIdx.x;; work += gridDim.x) {
        // ---- S0: the previous work item's rows leave LDS ------------------------------------------------
        if (prev >= 0) {
            const int row = prev_row, img = prev_img;
            const int c0 = prev_c0, tn = min(S, w - c0);
            const int nbytes = tn * C;
#pragma unroll
            for (int e = 0; e < NE; e++) {
                if (PL_DBG() == 3) break;
                uint8_t *dst = P.out[e] + (int64_t)img * P.ois[e] + (int64_t)row * P.ors[e] + (size_t)c0 * C;
                const uint8_t *src = s_out + e * L.outstride;
                if ((((uintptr_t)dst) & 15) == 0) {
                    const int n16 = nbytes >> 4;
                    for (int i = tid; i < n16; i += PL_THREADS)
                        reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
                    for (int i = (n16 << 4) + tid; i < nbytes; i += PL_THREADS) dst[i] = src[i];
                } else if ((((uintptr_t)dst) & 3) == 0) {
                    const int n4 = nbytes >> 2;
                    for (int i = tid; i < n4; i += PL_THREADS)
                        reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
                    for (int i = (n4 << 2) + tid; i < nbytes; i += PL_THREADS) dst[i] = src[i];
                } else {
                    for (int i = tid; i < nbytes; i += PL_THREADS) dst[i] = src[i];
                }
            }
            if (tid == 0)
                for (int e = 0; e < NE; e++)
                    if (s_misc[2 + e]) { pl_flag_row(P, img, e, row); s_misc[2 + e] = 0; }
            prev = -1;
        }
        PL_TICK(0);
        if (work >= nwork) break;

        const unsigned uw = (unsigned)work, rr = uw / (unsigned)tiles;
        const int ts = (int)(uw - rr * (unsigned)tiles);
        const int img = (int)(rr / (unsigned)P.h), row = (int)(rr - (unsigned)img * (unsigned)P.h);
        const int c0 = ts * S, c1 = min(c0 + S, w), tn = c1 - c0;
        const double mn = P.minmax[img * 2], mx = P.minmax[img * 2 + 1];
        const uint8_t *src_row = P.img + ((size_t)img * P.h + row) * (size_t)w * C;
        const size_t esz = P.depth_dtype == DS_DEPTH_U16 ? 2 : (P.depth_dtype == DS_DEPTH_F32 ? 4 : 8);
        const void *depth_row = (const char *)P.depth + ((size_t)img * P.h + row) * (size_t)w * esz;

        // 0/0: constant depth gives NaN for every point (stereoimage_generation.py:81); the sweep then
        // only ever sees the segment from the -w sentinel, whose colour index is 0 on both ends.
        if (!(mx > mn)) {
            for (int p = tid; p < tn; p += PL_THREADS) {
                const int col = c0 + p;
                const double coord_from = (double)col + PL_EPS;
                const double coord_to = (double)(col + 1) - PL_EPS;
                const double significance = coord_to - coord_from;
                for (int e = 0; e < NE; e++) {
                    uint8_t *out_row = P.out[e] + (int64_t)img * P.ois[e] + (int64_t)row * P.ors[e];
                    for (int k = 0; k < C; k++) {
                        double color = 0.5;
                        color += (double)src_row[k] * significance;
                        out_row[(size_t)col * C + k] = ds_f64_to_u8(color);
                    }
                }
            }
            continue;
        }

        // ---- P01: stage the source window, both eyes' points, entry tables --------------------------------
        const int j0 = max(0, c0 + uL - 1) & ~3;
        const int j1 = max(j0, min(w - 1, c1 - 1 + uU));
        const int ncols = j1 - j0 + 1;
        const bool head = j0 == 0, tail = j1 == w - 1;
        const double *lut = (P.lut != nullptr && P.depth_dtype == DS_DEPTH_U16) ? P.lut + (size_t)img * 65536 : nullptr;
        int bad = 0;                                        // non-finite / absurd coordinates seen by this thread
        for (int i4 = tid * 4; i4 < ncols; i4 += 4 * PL_THREADS) {
            const int j = j0 + i4;
            const int nv = min(4, ncols - i4);
            double nd[4], ndp;
            PL_TICK(7);
            pl_load_nd(P, depth_row, j, nv, j > 0, mn, mx, lut, nd, ndp);
            uint32_t rgbx[4];
            pl_load_rgbx<C>(src_row + (size_t)j * C, nv, P.al4 != 0, rgbx);
            PL_TICKW("s_waitcnt vmcnt(0)", 8);
            if (nv == 4) {
                *reinterpret_cast<double2 *>(&s_nd[i4]) = make_double2(nd[0], nd[1]);
                *reinterpret_cast<double2 *>(&s_nd[i4 + 2]) = make_double2(nd[2], nd[3]);
                *reinterpret_cast<uint4 *>(&s_rgbx[i4]) = make_uint4(rgbx[0], rgbx[1], rgbx[2], rgbx[3]);
            } else {
                for (int m = 0; m < nv; m++) { s_nd[i4 + m] = nd[m]; s_rgbx[i4 + m] = rgbx[m]; }
            }
            if (i4 == 0) {                                  // window column -1 (never selected when the halo is right)
                s_nd[-1] = j > 0 ? ndp : 0.0;
                uint32_t v = 0;
                if (j > 0) for (int k = 0; k < C; k++) v |= (uint32_t)src_row[(size_t)(j - 1) * C + k] << (8 * k);
                s_rgbx[-1] = v;
            }
            PL_TICKW("s_waitcnt lgkmcnt(0)", 9);
            double colx[4];                                 // (col + 0.5), shared by the eyes (:183)
#pragma unroll
            for (int m = 0; m < 4; m++) colx[m] = (double)(j + m) + 0.5;
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const double div_px = k_div[e], sep_px = k_sep[e];
                double *pt = s_pt + e * L.npt;
                uint32_t *g0 = s_g0 + e * S;
                uint32_t slow = 0;                          // bit t: segment NP*i4 + 1 + t of this thread needs the slow path
                double xprev;
                if (j == 0) xprev = -1.0 * (double)w;                                           // :179
                else {
                    const double coord_d = ndp * div_px;
                    const double coord_x = (double)(j - 1) + 0.5 + coord_d + sep_px;
                    xprev = SHARP ? coord_x + 0.45 : coord_x;
                }
                int fprev = pl_floor_i(xprev);
                if (!SHARP) pt[i4] = xprev;              // also stored (same bits) by the previous thread: see the slow loop
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    if (m < nv) {
                        const double coord_d = nd[m] * div_px;                                  // :182
                        const double coord_x = colx[m] + coord_d + sep_px;                      // :183
                        if (!(fabs(coord_x) < 4.0e9)) bad = 1;
                        if (SHARP) {
                            const double xl = coord_x - 0.45, xr = coord_x + 0.45;              // :188-189
                            const int fl = pl_floor_i(xl), fr = pl_floor_i(xr);
                            const int q = 2 * (i4 + m) + 1;
                            *reinterpret_cast<double2 *>(&pt[q - 1]) = make_double2(xprev, xl);
                            if (pl_scatter(g0, xprev, xl, fprev, fl, (uint32_t)q, c0, tn)) slow |= 1u << (2 * m);
                            if (pl_scatter(g0, xl, xr, fl, fr, (uint32_t)(q + 1), c0, tn)) slow |= 2u << (2 * m);
                            xprev = xr; fprev = fr;
                        } else {
                            const int fx = pl_floor_i(coord_x);
                            const int q = i4 + m + 1;
                            pt[q] = coord_x;                                                    // :185
                            if (pl_scatter(g0, xprev, coord_x, fprev, fx, (uint32_t)q, c0, tn)) slow |= 1u << m;
                            xprev = coord_x; fprev = fx;
                        }
                    }
                }
                if (SHARP) pt[2 * (i4 + nv)] = xprev;       // also stored (same bits) by the next thread: see the slow loop
                if (i4 + nv == ncols) {                     // owner of the last column: tail sentinel, padding
                    const int qT = NP * ncols + 1;
                    for (int k = 0; k < PL_PT_PAD - 1; k++) pt[qT + k] = 2.0 * (double)w;        // :191 (or harmless padding)
                    if (tail) slow |= 1u << (NP * nv);      // the tail sentinel segment qT follows this thread's last point
                }
                PL_TICKW("s_waitcnt lgkmcnt(0)", 10);
                // long gaps, backward segments, the tail sentinel: rare, one shared loop (the points are re-read from LDS;
                // a thread reads back only what it stored itself)
                while (__any(slow != 0u)) {
                    if (slow != 0u) {
                        const int t = __ffs((int)slow) - 1;
                        slow &= slow - 1u;
                        const int q = NP * i4 + 1 + t;
                        const double x0 = pt[q - 1], x1 = pt[q];
                        pl_scatter_slow(g0, x0, x1, (int)floor(x0), (int)floor(x1), (uint32_t)q, c0, tn);
                    }
                }
            }
        }
        PL_TICK(1);
        __syncthreads();                                                                        // ---- A ----
        PL_TICK(2);
        if (bad) { atomicOr(&s_misc[2], 1); atomicOr(&s_misc[3], 1); }

        // ---- P2: one lane per output pixel, both eyes in one straight line -----------------------------------
        if (PL_DBG() != 1)
        for (int p0 = 0; p0 < tn; p0 += PL_THREADS) {
            const int p = p0 + tid;
            const bool inb = p < tn;
            const double fq = (double)(c0 + p), fq1 = (double)(c0 + p + 1);
            const int qT = NP * ncols + 1;
            double color[NE][C];
            uint32_t g0[NE];
            int flag[NE];
            bool walk[NE], queued[NE];
#pragma unroll
            for (int e = 0; e < NE; e++) {
                g0[e] = 0;
                if (inb) { g0[e] = s_g0[e * S + p]; s_g0[e * S + p] = 0; }
                flag[e] = (inb && g0[e] == 0) ? 1 : 0;     // nothing enters this pixel: only possible with non-finite coordinates
                queued[e] = inb && g0[e] >= PL_BAD;
                const bool ok = inb && g0[e] != 0 && g0[e] < PL_BAD;
#pragma unroll
                for (int k = 0; k < C; k++) color[e][k] = 0.5;                                   // :229
                if (SHARP) walk[e] = ok && !pl_simple4<C>(s_pt + e * L.npt, s_rgbx, g0[e], ok, head, qT, fq, fq1, color[e], flag[e]);
                else walk[e] = ok;
            }
            // runs with a fifth piece or a sentinel segment (image borders), and every polylines_soft pixel: the walker
#pragma unroll
            for (int e = 0; e < NE; e++) {
                if (__any(walk[e])) {
                    if (walk[e]) {
                        PlWin V;
                        V.pt = s_pt + e * L.npt; V.nd = s_nd; V.rgbx = s_rgbx; V.ncols = ncols; V.head = head; V.tail = tail; V.div_px = k_div[e];
                        flag[e] = 0;
                        pl_walk<C, NP>(V, g0[e], fq, fq1, color[e], flag[e]);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < NE; e++) {
                if (inb) {
                    uint8_t *o = s_out + e * L.outstride + p * C;
#pragma unroll
                    for (int k = 0; k < C; k++) o[k] = (uint8_t)pl_u8(color[e][k]);              // :281
                }
                if (flag[e]) atomicOr(&s_misc[2 + e], 1);
                const unsigned long long qmask = __ballot(queued[e]);
                if (qmask != 0ull) {
                    if (P.K > PL_KMAX) {                     // candidate window too wide for the bit masks: exact sweep
                        if (queued[e]) atomicOr(&s_misc[2 + e], 1);
                    } else {
                        const int lane = tid & 63;
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&s_misc[0], __popcll(qmask));
                        base = __shfl(base, 0, 64);
                        if (queued[e]) {
                            const int slot = base + __popcll(qmask & ((1ull << lane) - 1ull));
                            if (slot < P.gq_cap) {
                                const unsigned rowid = (unsigned)((img * P.n_eyes + e) * P.h + row);
                                P.gq[(size_t)blockIdx.x * P.gq_cap + slot] = ((unsigned long long)rowid << 32) | (unsigned)(c0 + p);
                            } else {
                                atomicOr(&s_misc[2 + e], 1);     // segment full: the row goes to the exact sweep instead
                            }
                        }
                    }
                }
            }
        }
        PL_TICK(3);
        __syncthreads();                                                                        // ---- B ----
        PL_TICK(4);
        prev = work; prev_img = img; prev_row = row; prev_c0 = c0;
    }
    if (tid == 0) {                                        // s_misc[0] was last written before the final barrier B
        const int nq = s_misc[0];
        P.gq_count[blockIdx.x] = min(nq, P.gq_cap);
        if (nq > 0) atomicAdd(&P.counters[1], nq);
    }
#ifdef DS_PL_PROFILE
    if (P.prof && tid == 0) {
        for (int i = 0; i < 7; i++) atomicAdd(&P.prof[i], tacc[i]);
        atomicAdd(&P.prof[7], 1ull);
        for (int i = 7; i < 12; i++) atomicAdd(&P.prof[i + 1], tacc[i]);
    }
#endif
#undef PL_TICK
#undef PL_TICKW
}

// ------------------------------------------------------------------------------------------------
// GENERAL pixels, second pass.  k_polylines queued them (row id, column) in HBM; here they are rendered densely packed,
// EIGHT LANES PER PIXEL, 8 pixels per wave, one wave per workgroup (no workgroup barrier is ever shared with idle waves).
// A group first rebuilds the pixel's own source window in LDS (since round 3: ONE window per wave when its 8 pixels sit in one
// row, see the kernel) -- the <= offU - offL + 2 columns that can reach it: normalised
// depth, points, packed colours, with the arithmetic of phase P01, so every value is bit-identical to what the main kernel
// staged -- then the 8 lanes test the candidate segments together (8 per step) and share two bit masks (forward segments
// overlapping the strip, points inside the pixel); lane u takes the u-th point, the group ranks its points (the reference's
// stable sort, :214-219), lane t evaluates sub-interval t, and the terms are added in order (float64 addition is not
// associative).  More than 8 sub-intervals: one lane walks them all (pl_general_pixel_lds).  History-dependent corners flag
// the row for the exact sweep, exactly as in the main kernel.
struct PlGenLds { int ptn, ndn, stride; };
__host__ __device__ inline PlGenLds pl_gen_layout(int ncmax, int np)
{
    PlGenLds G;
    G.ptn = (np * ncmax + PL_PT_PAD + 2 + 1) & ~1;         // doubles
    G.ndn = (ncmax + 2 + 1) & ~1;                          // doubles (nd[-1] exists)
    G.stride = (G.ptn + G.ndn) * 8 + ((ncmax + 2 + 3) & ~3) * 4;   // + rgbx words; multiple of 16 bytes
    return G;
}

// normalised depth of ONE column, same expressions as pl_load_nd; y = 1 / (max - min) of the uint16 path is computed once
// per pixel by the caller
__device__ __forceinline__ double pl_nd_one(const PolyParams &P, const void *depth_row, int j, double mn, double mx, const double *lut,
                                            double b, double y)
{
    if (P.depth_dtype == DS_DEPTH_U16) {
        const uint32_t v = ((const uint16_t *)depth_row)[j];
        if (lut != nullptr) return lut[v];
        const uint32_t mn16 = (uint32_t)mn & 0xffffu;
        const double a = (double)((v - mn16) & 0xffffu);
        const double q0 = a * y;
        return fma(fma(-b, q0, a), y, q0);
    }
    if (P.depth_dtype == DS_DEPTH_F32) {
        const float v = ((const float *)depth_row)[j];
        const float mnf = (float)mn, den = (float)mx - (float)mn;
        return (double)((v - mnf) / den);
    }
    const double v = ((const double *)depth_row)[j];
    return (v - mn) / (mx - mn);
}

// (launch bounds: 4 waves per SIMD = 128 VGPRs, no spill -- the pass is a chain of dependent LDS reads, ballots and shuffles, i.e.
// latency bound, and the unconstrained allocation of 170 registers left it two waves per SIMD to hide that latency with)
// WPE = waves per SIMD the register allocation is bounded for: 4 is the default; 6 (80 VGPRs, a few spills) exists for A/B runs
// (DS_PL_GEN_WPE=6)
// KW = 64-segment words of the candidate masks the kernel is compiled for: 1 when P.K == 1 (windows of up to 64 segments: every
// default-parameter launch at 1024 columns), 2 when P.K == 2 (1080p) -- popping the next candidate out of a four-word mask cost ~40 vector instructions per candidate,
// and the pass is bound by vector-instruction issue (profiles/round3_pmc_c5_polylines.json) -- PL_KMAX otherwise.
template <int C, int SHARP, int WPE = 4, int KW = PL_KMAX>
__global__ __launch_bounds__(64, WPE) void k_polylines_general(PolyParams P, int ncmax, int per_seg, int one_window)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NP = SHARP ? 2 : 1;
    const int lane = threadIdx.x, grp = lane >> 3, u = lane & 7;
    const PlGenLds G = pl_gen_layout(ncmax, NP);
    unsigned char *gbase = smem + grp * G.stride;
    double *g_pt = reinterpret_cast<double *>(gbase);
    double *g_nd = reinterpret_cast<double *>(gbase) + G.ptn + 1;            // g_nd[-1] exists
    uint32_t *g_rgbx = reinterpret_cast<uint32_t *>(gbase + (G.ptn + G.ndn) * 8) + 1;   // g_rgbx[-1] exists
    const int w = P.w;
    // ONE window for the whole wave when its pixels sit in one row (they almost always do: the main kernel queues a row's
    // general pixels in column order, folds come in clusters, and a wave takes 8 consecutive entries): the 8 private windows
    // overlap almost completely, and building them costs every lane ~4 columns of dependent global loads.  The shared window
    // (columns cmin + offL - 1 .. cmax + offU, the same LDS bytes, laid out as ONE window of up to `cap` columns) costs a lane
    // one column; a group then works on its own slice of it through the same PlWin view (V.pt = s_pt + NP * (j0 - js0), ...):
    // every value and every index a group reads is what its private window would have held.
    const int cap = (8 * G.stride - 176) / (8 * NP + 12);
    const PlGenLds S = pl_gen_layout(cap, NP);
    double *s_pt = reinterpret_cast<double *>(smem);
    double *s_nd = reinterpret_cast<double *>(smem) + S.ptn + 1;
    uint32_t *s_rgbx = reinterpret_cast<uint32_t *>(smem + (S.ptn + S.ndn) * 8) + 1;
    // workgroup (seg, part): entries part*8, part*8 + 8*per_seg, ... of queue segment seg
    const int seg = blockIdx.x / per_seg, part = blockIdx.x - seg * per_seg;
    const int count = P.gq_count[seg];
    const unsigned long long *q = P.gq + (size_t)seg * P.gq_cap;
    // the queue entry of the NEXT group of 8 is requested while this one is rendered: one global round trip less on the
    // dependent chain entry -> depth / pixel loads -> window of every iteration
    unsigned long long ent_next = (part * 8 + grp < count) ? q[part * 8 + grp] : 0ull;
    for (int base = part * 8; base < count; base += 8 * per_seg) {
        const int idx = base + grp;
        const bool have = idx < count;
        const unsigned long long ent = ent_next;
        {
            const int nidx = idx + 8 * per_seg;
            ent_next = nidx < count ? q[nidx] : 0ull;
        }
        const int rowid = (int)(ent >> 32), col = (int)(ent & 0xffffffffull);
        const int row = rowid % P.h, ie = rowid / P.h;
        const int e = ie % P.n_eyes, img = ie / P.n_eyes;
        const double div_px = P.div_px[e], sep_px = P.sep_px[e];
        const double mn = P.minmax[img * 2], mx = P.minmax[img * 2 + 1];
        const uint8_t *src_row = P.img + ((size_t)img * P.h + row) * (size_t)w * C;
        const size_t esz = P.depth_dtype == DS_DEPTH_U16 ? 2 : (P.depth_dtype == DS_DEPTH_F32 ? 4 : 8);
        const void *depth_row = (const char *)P.depth + ((size_t)img * P.h + row) * (size_t)w * esz;
        const double *lut = (P.lut != nullptr && P.depth_dtype == DS_DEPTH_U16) ? P.lut + (size_t)img * 65536 : nullptr;
        // the pixel's source window: columns j0 .. j1 (what phase P3 of round 1 selected out of the supertile's staging)
        const int j0 = max(0, col + P.offL[e] - 1);
        const int j1 = max(j0, min(w - 1, col + P.offU[e]));
        const int ncols = have ? j1 - j0 + 1 : 1;
        const bool head = j0 == 0, tail = j1 == w - 1;
        // all pixels of this wave in one row?  (group 0 always has an entry here; its row id is the reference)
        const int rowid0 = __builtin_amdgcn_readfirstlane(rowid);
        int cmin = have ? col : 0x7fffffff, cmax = have ? col : -1;
#pragma unroll
        for (int sft = 8; sft < 64; sft <<= 1) { cmin = min(cmin, __shfl_xor(cmin, sft, 64)); cmax = max(cmax, __shfl_xor(cmax, sft, 64)); }
        const int e0 = (rowid0 / P.h) % P.n_eyes;
        const int js0 = max(0, cmin + P.offL[e0] - 1);
        const int js1 = max(js0, min(w - 1, cmax + P.offU[e0]));
        const int ncs = js1 - js0 + 1;
        const bool shared = one_window != 0 && ncs <= cap && __all(!have || rowid == rowid0);
        __syncthreads();                                    // the previous pixel's window is no longer read
        const double b16 = (double)(((uint32_t)mx - ((uint32_t)mn & 0xffffu)) & 0xffffu), y16 = 1.0 / b16;   // uint16 path only
        // column i of a window that starts at column jw (i = -1: the column before it: point 0, colour -1)
#define PL_GEN_COLUMN(PT, ND, RGBX, jw, i, hd, srow, drow, mn_, mx_, lut_, b_, y_, div_, sep_) do {                      \
            const int j = (jw) + (i);                                                                                     \
            double nd = 0.0;                                                                                              \
            uint32_t px = 0;                                                                                              \
            if (j >= 0) {                                                                                                 \
                nd = pl_nd_one(P, drow, j, mn_, mx_, lut_, b_, y_);                                                       \
                _Pragma("unroll") for (int k = 0; k < C; k++) px |= (uint32_t)(srow)[(size_t)j * C + k] << (8 * k);       \
            }                                                                                                             \
            (ND)[i] = nd;                                                                                                 \
            (RGBX)[i] = px;                                                                                               \
            const double coord_d = nd * (div_);                                                 /* :182 */                \
            const double coord_x = ((double)j + 0.5) + coord_d + (sep_);                        /* :183 */                \
            if ((i) < 0) {                                                                                                \
                (PT)[0] = (hd) ? -1.0 * (double)w : (SHARP ? coord_x + 0.45 : coord_x);       /* :179 / the previous point */ \
            } else if (SHARP) {                                                                                           \
                (PT)[2 * (i) + 1] = coord_x - 0.45;                                             /* :188 */                \
                (PT)[2 * (i) + 2] = coord_x + 0.45;                                             /* :189 */                \
            } else {                                                                                                      \
                (PT)[(i) + 1] = coord_x;                                                        /* :185 */                \
            }                                                                                                             \
        } while (0)
        if (shared) {
            // the row's constants from group 0's entry (every group that has an entry agrees; the others must not be used)
            const int row0 = rowid0 % P.h, ie0 = rowid0 / P.h, img0 = ie0 / P.n_eyes;
            const double div0 = P.div_px[e0], sep0 = P.sep_px[e0];
            const double mn0 = P.minmax[img0 * 2], mx0 = P.minmax[img0 * 2 + 1];
            const uint8_t *srow0 = P.img + ((size_t)img0 * P.h + row0) * (size_t)w * C;
            const void *drow0 = (const char *)P.depth + ((size_t)img0 * P.h + row0) * (size_t)w * esz;
            const double *lut0 = (P.lut != nullptr && P.depth_dtype == DS_DEPTH_U16) ? P.lut + (size_t)img0 * 65536 : nullptr;
            const double b0 = (double)(((uint32_t)mx0 - ((uint32_t)mn0 & 0xffffu)) & 0xffffu), y0 = 1.0 / b0;
            const bool heads = js0 == 0;
            for (int i = lane - 1; i < ncs; i += 64)
                PL_GEN_COLUMN(s_pt, s_nd, s_rgbx, js0, i, heads, srow0, drow0, mn0, mx0, lut0, b0, y0, div0, sep0);
            if (lane == 0) {
                const int qT = NP * ncs + 1;
                for (int k = 0; k < PL_PT_PAD - 1; k++) s_pt[qT + k] = 2.0 * (double)w;         // :191 (or harmless padding)
            }
        } else if (have) {
            for (int i = u - 1; i < ncols; i += 8)
                PL_GEN_COLUMN(g_pt, g_nd, g_rgbx, j0, i, head, src_row, depth_row, mn, mx, lut, b16, y16, div_px, sep_px);
            if (u == 0) {
                const int qT = NP * ncols + 1;
                for (int k = 0; k < PL_PT_PAD - 1; k++) g_pt[qT + k] = 2.0 * (double)w;         // :191 (or harmless padding)
            }
        }
#undef PL_GEN_COLUMN
        __syncthreads();
        PlWin V;
        const int woff = have ? j0 - js0 : 0;              // this group's slice of the shared window (groups without an entry read nothing)
        V.pt = shared ? s_pt + NP * woff : g_pt; V.nd = shared ? s_nd + woff : g_nd; V.rgbx = shared ? s_rgbx + woff : g_rgbx; V.ncols = ncols; V.head = head; V.tail = tail; V.div_px = div_px;
        const int qlo = 1;
        const int qhi = !have ? 0 : (tail ? NP * ncols + 1 : NP * ncols);
        const double fq = (double)col, fq1 = (double)(col + 1);
        unsigned long long fm[KW], bm[KW];
#pragma unroll
        for (int k = 0; k < KW; k++) { fm[k] = 0ull; bm[k] = 0ull; }
#pragma unroll
        for (int k = 0; k < KW; k++) {
            if (k < P.K) {
                for (int st = 0; st < 8 && 64 * k + 8 * st < P.nseg; st++) {
                    const int qq = qlo + 64 * k + 8 * st + u;
                    const bool valid = qq <= qhi;
                    const double x0 = V.pt[valid ? qq - 1 : 0], x1 = V.pt[valid ? qq : 0];
                    const unsigned long long mf = __ballot(valid && x0 < x1 && x0 < fq1 && !(x1 < fq));
                    const unsigned long long mb = __ballot(valid && !(x1 < fq) && x1 < fq1);
                    fm[k] |= ((mf >> (8 * grp)) & 0xffull) << (8 * st);
                    bm[k] |= ((mb >> (8 * grp)) & 0xffull) << (8 * st);
                }
            }
        }
        int nb = 0;
#pragma unroll
        for (int k = 0; k < KW; k++) nb += __popcll(bm[k]);
        const bool ovf = nb > 7;                             // more than 8 sub-intervals: one lane walks them all
        const bool par = have && !ovf;
        // lane u's point, its rank among the group's points, the sorted sequence
        const double BIG = 1.0e300;
        double bx = BIG;
        {
            int n = u, qb = -1;
#pragma unroll
            for (int k = 0; k < KW; k++) {
                const int c = __popcll(bm[k]);
                if (qb < 0 && n < c) {
                    unsigned long long mm = bm[k];
#pragma unroll
                    for (int i = 0; i < 7; i++) if (i < n) mm &= mm - 1ull;
                    qb = 64 * k + __ffsll((long long)mm) - 1;
                }
                n -= c;
            }
            if (par && qb >= 0) bx = V.pt[qlo + qb];
        }
        int rank = 0;
#pragma unroll
        for (int i = 1; i < 8; i++) {
            const int ou = (u + i) & 7;
            const double ox = __shfl(bx, (lane & ~7) | ou, 64);
            rank += (ox < bx || (ox == bx && ou < u)) ? 1 : 0;
        }
        double sx;
        {
            const int tgt = ((lane & ~7) | rank) << 2;
            const int lo = __builtin_amdgcn_ds_permute(tgt, __double2loint(bx));
            const int hi = __builtin_amdgcn_ds_permute(tgt, __double2hiint(bx));
            sx = __hiloint2double(hi, lo);
        }
        const double sprev = __shfl(sx, (lane & ~7) | ((u + 7) & 7), 64);
        const double a = u == 0 ? fq : sprev;
        const double b = u < nb ? sx : fq1;
        double term[C];
#pragma unroll
        for (int k = 0; k < C; k++) term[k] = 0.0;
        int flag = 0;
        if (par && u <= nb) {
            // sub-interval u of the pixel: active set, winner, colour term (:235-279)
            const PlSub sb = pl_sub(a, b);
            if (!(sb.significance > 0.0)) flag = 1;          // centres may stop being monotone: history dependent
            int count_a = 0, first = -1, win = -1;
            double best = -PL_EPS;                                                       // :261
            bool hv = false;
#pragma nounroll
            for (;;) {
                int qq = -1;                                 // next forward candidate: lowest set bit of the multi-word mask
#pragma unroll
                for (int k = 0; k < KW; k++) {
                    if (qq < 0 && fm[k] != 0ull) {
                        qq = qlo + 64 * k + __ffsll((long long)fm[k]) - 1;
                        fm[k] &= fm[k] - 1ull;
                    }
                }
                if (qq < 0) break;
                const double x0 = V.pt[qq - 1], x1 = V.pt[qq];
                if (x0 < sb.coord_center && !(x1 < sb.coord_center)) {                  // :242-253
                    if (count_a == 0) first = qq;
                    count_a++;
                    const double d0 = pl_dd_of<NP>(V, qq - 1), d1 = pl_dd_of<NP>(V, qq);
                    const double ip_k = (sb.coord_center - x0) / (x1 - x0);             // :263
                    const double closeness = (1.0 - ip_k) * d0 + ip_k * d1;             // :265
                    const bool valid = 0.0 < ip_k && ip_k < 1.0;
                    if (valid && hv && closeness == best) flag = 1;                     // exact tie: csg order decides
                    if (best < closeness && valid) { best = closeness; win = qq; hv = true; }   // :266
                }
            }
            if (count_a == 1) win = first;                                               // :259
            if (count_a == 0 || (count_a > 1 && !hv)) flag = 1;  // reference reads a stale / falls back to csg[0]
            else pl_add_segment<C, NP>(V, win, V.pt[win - 1], V.pt[win], sb, term);
        }
        double color[C];
#pragma unroll
        for (int k = 0; k < C; k++) color[k] = 0.5;                                      // :229
        for (int t = 0; __any(par && t <= nb); t++) {
#pragma unroll
            for (int k = 0; k < C; k++) color[k] += __shfl(term[k], (lane & ~7) | (t & 7), 64);   // lanes past nb hold 0.0
        }
        if (have && ovf && u == 0) pl_general_pixel_lds<C, NP, KW>(V, qlo, P.K, fm, bm, fq, fq1, color, flag);
        if (have && u == 0) {
            uint8_t *o = P.out[e] + (int64_t)img * P.ois[e] + (int64_t)row * P.ors[e] + (size_t)col * C;
#pragma unroll
            for (int k = 0; k < C; k++) o[k] = ds_f64_to_u8(color[k]);
        }
        if (have && flag) pl_flag_row(P, img, e, row);
    }
}

// ------------------------------------------------------------------------------------------------
// Exact fallback: the reference's row sweep, statement by statement, one lane per flagged row.
// Scratch (per worker, interleaved across workers so lock-step lanes coalesce):
//   OX[np] OD[np]  points in original order        (x, |d|)
//   SX[np] SK[np]  points in sorted order          (x, original index); segment k travels with point k
//   CSG[np]        active set as original segment indices
struct ExactScratch { double *ox, *od, *sx; int *sk, *csg; int nworkers; int np_max; int c; };

template <int DT, int SHARP>
__global__ __launch_bounds__(64) void k_polylines_exact(PolyParams P, ExactScratch S)
{
    const int worker = blockIdx.x * 64 + threadIdx.x;
    const int count = P.counters[0];
    const int w = P.w, c = S.c;
    const int NW_ = S.nworkers;
#define A_(arr, i) arr[(size_t)(i) * NW_ + worker]
    for (int it = worker; it < count; it += NW_) {
        const int rowid = P.row_list[it];
        const int row = rowid % P.h;
        const int ie = rowid / P.h;
        const int eye = ie % P.n_eyes, img = ie / P.n_eyes;
        const double div_px = P.div_px[eye], sep_px = P.sep_px[eye];
        const double mn = P.minmax[img * 2], mx = P.minmax[img * 2 + 1];
        const uint8_t *src = P.img + ((size_t)img * P.h + row) * (size_t)w * c;
        typedef typename ds_depth_traits<DT>::T DTy;
        const DTy *depth_row = (const DTy *)P.depth + ((size_t)img * P.h + row) * (size_t)w;
        uint8_t *dst = P.out[eye] + (int64_t)img * P.ois[eye] + (int64_t)row * P.ors[eye];

        int pt_end = 0;
        A_(S.ox, 0) = -1.0 * (double)w; A_(S.od, 0) = 0.0; pt_end = 1;                       // :179
        for (int col = 0; col < w; col++) {                                                 // :181
            const double coord_d = pl_coord_d<DT>(P, img, depth_row, col, mn, mx, div_px);
            const double coord_x = (double)col + 0.5 + coord_d + sep_px;
            if (SHARP) {
                A_(S.ox, pt_end) = coord_x - 0.45; A_(S.od, pt_end) = fabs(coord_d);
                A_(S.ox, pt_end + 1) = coord_x + 0.45; A_(S.od, pt_end + 1) = fabs(coord_d);
                pt_end += 2;
            } else {
                A_(S.ox, pt_end) = coord_x; A_(S.od, pt_end) = fabs(coord_d);
                pt_end += 1;
            }
        }
        A_(S.ox, pt_end) = 2.0 * (double)w; A_(S.od, pt_end) = 0.0; pt_end++;               // :191
        const int sg_end = pt_end - 1;                                                      // :196
        for (int i = 0; i < pt_end; i++) { A_(S.sx, i) = A_(S.ox, i); A_(S.sk, i) = i; }
        for (int i = 1; i < sg_end; i++) {                                                  // :214
            int u = i - 1;
            while (u >= 0 && A_(S.sx, u) > A_(S.sx, u + 1)) {
                const double tx = A_(S.sx, u); A_(S.sx, u) = A_(S.sx, u + 1); A_(S.sx, u + 1) = tx;
                const int tk = A_(S.sk, u); A_(S.sk, u) = A_(S.sk, u + 1); A_(S.sk, u + 1) = tk;
                u--;
            }
        }
        // colour index of original point p
#define CI_(p) ((p) == 0 ? 0 : ((p) == pt_end - 1 ? (w - 1) : (SHARP ? ((p) - 1) >> 1 : (p) - 1)))
        int csg_end = 0, sg_pointer = 0, pt_i = 0;
        bool slot0_written = false;   // csg[0] is a row of np.zeros until something is stored there
        for (int col = 0; col < w; col++) {                                                 // :228
            double color[4] = { 0.5, 0.5, 0.5, 0.5 };
            while (A_(S.sx, pt_i) < (double)col) pt_i++;                                    // :230
            pt_i--;
            while (A_(S.sx, pt_i) < (double)(col + 1)) {                                    // :234
                const double pa = A_(S.sx, pt_i), pb = A_(S.sx, pt_i + 1);
                const double coord_from = (pa > (double)col ? pa : (double)col) + PL_EPS;
                const double coord_to = (pb < (double)(col + 1) ? pb : (double)(col + 1)) - PL_EPS;
                const double significance = coord_to - coord_from;
                const double coord_center = coord_from + 0.5 * significance;
                while (sg_pointer < sg_end && A_(S.sx, sg_pointer) < coord_center) {        // :242
                    A_(S.csg, csg_end) = A_(S.sk, sg_pointer);
                    if (csg_end == 0) slot0_written = true;
                    sg_pointer++; csg_end++;
                }
                int csg_i = 0;                                                              // :247
                while (csg_i < csg_end) {
                    const int k = A_(S.csg, csg_i);
                    if (A_(S.ox, k + 1) < coord_center) { A_(S.csg, csg_i) = A_(S.csg, csg_end - 1); csg_end--; }
                    else csg_i++;
                }
                int best = 0;
                if (csg_end != 1) {                                                         // :259
                    double best_closeness = -PL_EPS;
                    for (csg_i = 0; csg_i < csg_end; csg_i++) {
                        const int k = A_(S.csg, csg_i);
                        const double x0 = A_(S.ox, k), x1 = A_(S.ox, k + 1);
                        const double ip_k = (coord_center - x0) / (x1 - x0);
                        const double closeness = (1.0 - ip_k) * A_(S.od, k) + ip_k * A_(S.od, k + 1);
                        if (best_closeness < closeness && 0.0 < ip_k && ip_k < 1.0) { best_closeness = closeness; best = csg_i; }
                    }
                }
                // csg[best]; with an empty set best == 0 and the reference reads whatever row 0 still holds
                // (the removal loop never clears a slot, and neither do we)
                const int k = (csg_end > 0 || slot0_written) ? A_(S.csg, best) : -1;
                if (k >= 0) {
                    const int col_l = CI_(k), col_r = CI_(k + 1);                           // :270-271
                    if (col_l == col_r) {
                        for (int q = 0; q < 4; q++) if (q < c) color[q] += (double)src[(size_t)col_l * c + q] * significance;
                    } else {
                        const double x0 = A_(S.ox, k), x1 = A_(S.ox, k + 1);
                        const double ip_k = (coord_center - x0) / (x1 - x0);
                        for (int q = 0; q < 4; q++) if (q < c) {
                            const double u = (double)src[(size_t)col_l * c + q] * (1.0 - ip_k);
                            const double v = (double)src[(size_t)col_r * c + q] * ip_k;
                            color[q] += (u + v) * significance;
                        }
                    }
                } else {
                    // zero row: col_l == col_r == 0
                    for (int q = 0; q < 4; q++) if (q < c) color[q] += (double)src[q] * significance;
                }
                pt_i++;                                                                     // :280
            }
            for (int q = 0; q < 4; q++) if (q < c) dst[(size_t)col * c + q] = ds_f64_to_u8(color[q]);
        }
#undef CI_
    }
#undef A_
}



// ------------------------------------------------------------------------------------------------
// Exact fallback, LDS resident: ONE WORKGROUP PER FLAGGED ROW (used whenever the row's arrays fit the CU's LDS; the kernel
// above is what remains for wider rows).  A network's prediction flags one or two rows of a 1080p batch now and then, and
// the lock-step kernel above -- one LANE per row, every array in global scratch -- took 14-17 ms for such a launch: a
// single lane walking 1920 columns with a dependent global round trip behind every access.  Here the row's points are built
// by all 64 lanes, sorted in parallel, and only the sweep itself -- the history-dependent part -- runs on one lane, against
// LDS.  The sort: the reference's insertion sort (:214-219, strict '>': stable) puts point i at position
// #{j : x_j < x_i} + #{j < i : x_j == x_i}; a point can only be overtaken by points of columns within |divergence_px| + 2 of
// its own (x = col + 0.5 + d + sep +- 0.45 with d between 0 and divergence_px), so each lane counts inside that window.
// NaN coordinates (a constant depth map: 0 / 0) compare false like in the reference's loop: such points stay where they are.
// Round 5: the LDS image of a row is COMPACT -- the points in original order are not stored at all: x and |d| of point p are
// recomputed from the column's coord_d (one double per column; the same three additions in the reference's order, so the same bits),
// and the active set gets a bounded array instead of one slot per point (a segment in the set covers the current centre, and a
// segment is at most |divergence_px| + 0.1 long: at most NP (3 |divergence_px| + 8) segments can be in the set, transient adds
// included) -- 36 bytes per column instead of 68: a 3840-column row (Boost on a 4K image, BASELINE config 4) fits the CU's 160 KB,
// where it used to fall back to the one-lane-per-row kernel above at 143 ms per launch (19 % of config 4's step).
//
// COOP = 1 (round 5, the default): the sweep is run by the WHOLE wave instead of one lane.  Control flow and every double stay
// uniform (all lanes compute the same values; lane 0 stores); the three scans over the active set -- the parts that made a row with a
// wide divergence cost 12 us per column -- are spread over the lanes with the reference's sequential semantics restated exactly:
//   * entering segments (:242-245): the sorted start points are appended in order -- a ballot finds how many consecutive ones lie left
//     of the centre;
//   * the swap-remove scan (:247-255: `csg[i] = csg[end - 1]; end -= 1` while scanning upwards) leaves, for K kept entries, the kept
//     entries of positions < K in place and fills the removed positions < K (ascending) with the kept entries of positions >= K taken
//     from the END downwards -- a permutation that only depends on the keep flags (checked against the sequential loop on 3 * 10^5
//     random cases in tools, and by the byte-identity tests); if nothing is kept, slot 0 is left holding what the sequential loop
//     leaves there (the old entry 1: the reference later reads that stale slot when the set is empty);
//   * the winner (:259-267, strict `<` in index order = the FIRST index that attains the maximal closeness among valid candidates):
//     per lane the first maximum of its strided candidates, across lanes the maximum with ties going to the lower index.
#define PLX_SCR 512             // COOP: capacity of the hole / donor lists of one removal (more than that: the sequential loop, by lane 0)
template <int DT, int SHARP, int COOP>
__global__ __launch_bounds__(64) void k_polylines_exact_lds(PolyParams P, int c, int win_pts, int csg_cap, int coop_min)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NP = SHARP ? 2 : 1;
    const int lane = threadIdx.x;
    const int w = P.w;
    const int pt_end = NP * w + 2, sg_end = pt_end - 1;
    double *cd = reinterpret_cast<double *>(smem);           // coord_d per column
    double *sx = cd + w;                                     // sorted x
    int *sk = reinterpret_cast<int *>(sx + pt_end);          // original index of the sorted point
    uint32_t *rgbx = reinterpret_cast<uint32_t *>(sk + pt_end);
    int *csg = reinterpret_cast<int *>(rgbx + w);            // the active set, csg_cap entries
    int *holes = csg + csg_cap;                              // COOP scratch: PLX_SCR + PLX_SCR positions, then the keep masks of the chunks
    int *donors = holes + PLX_SCR;
    unsigned long long *kmask = reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(donors + PLX_SCR) + 7) & ~(uintptr_t)7);   // (odd widths leave the lists 4-byte aligned)
    const int count = P.counters[0];
    for (int it = blockIdx.x; it < count; it += gridDim.x) {
        const int rowid = P.row_list[it];
        const int row = rowid % P.h;
        const int ie = rowid / P.h;
        const int eye = ie % P.n_eyes, img = ie / P.n_eyes;
        const double div_px = P.div_px[eye], sep_px = P.sep_px[eye];
        const double mn = P.minmax[img * 2], mx = P.minmax[img * 2 + 1];
        const uint8_t *src = P.img + ((size_t)img * P.h + row) * (size_t)w * c;
        typedef typename ds_depth_traits<DT>::T DTy;
        const DTy *depth_row = (const DTy *)P.depth + ((size_t)img * P.h + row) * (size_t)w;
        uint8_t *dst = P.out[eye] + (int64_t)img * P.ois[eye] + (int64_t)row * P.ors[eye];
        // x and |d| of ORIGINAL point p (:179-191): point 0 and the last one are the sentinels, the others belong to column
        // (p - 1) / NP; coord_x = col + 0.5 + coord_d + sep evaluated left to right like the reference's expression
        auto OX = [&](const int p) -> double {
            if (p == 0) return -1.0 * (double)w;
            if (p == pt_end - 1) return 2.0 * (double)w;
            const int col = SHARP ? (p - 1) >> 1 : p - 1;
            const double coord_x = (double)col + 0.5 + cd[col] + sep_px;
            if (!SHARP) return coord_x;
            return ((p - 1) & 1) ? coord_x + 0.45 : coord_x - 0.45;
        };
        auto OD = [&](const int p) -> double {
            if (p == 0 || p == pt_end - 1) return 0.0;
            return fabs(cd[SHARP ? (p - 1) >> 1 : p - 1]);
        };
        __syncthreads();                                     // the previous row's arrays are no longer read
        // ---- coord_d and the colour of every column, all lanes ----
        for (int col = lane; col < w; col += 64) {
            cd[col] = pl_coord_d<DT>(P, img, depth_row, col, mn, mx, div_px);
            uint32_t px = 0;
            for (int q = 0; q < c; q++) px |= (uint32_t)src[(size_t)col * c + q] << (8 * q);
            rgbx[col] = px;
        }
        __syncthreads();
        // ---- stable sort of points 0 .. sg_end - 1 by x; the last point keeps its place (:214 sorts range(1, sg_end)) ----
        for (int i = lane; i < pt_end; i += 64) {
            const double xi = OX(i);
            int pos = i;
            if (i < sg_end && xi == xi) {
                const int lo = max(0, i - win_pts), hi = min(sg_end - 1, i + win_pts);
                int less = lo;                               // everything left of the window is smaller (or NaN: see below)
                for (int j = lo; j <= hi; j++) {
                    const double xj = OX(j);
                    less += (xj < xi || (xj == xi && j < i)) ? 1 : 0;
                }
                pos = less;
            }
            sx[pos] = xi;
            sk[pos] = i;
        }
        __syncthreads();
#define CI_(p) ((p) == 0 ? 0 : ((p) == pt_end - 1 ? (w - 1) : (SHARP ? ((p) - 1) >> 1 : (p) - 1)))
        if (COOP) {
            // ---- the sweep (:228-282) by the whole wave: uniform control flow, the active-set scans across the lanes ----
            int csg_end = 0, sg_pointer = 0, pt_i = 0;
            bool slot0_written = false;
            for (int col = 0; col < w; col++) {
                double color[4] = { 0.5, 0.5, 0.5, 0.5 };
                while (sx[pt_i] < (double)col) pt_i++;
                pt_i--;
                while (sx[pt_i] < (double)(col + 1)) {
                    const double pa = sx[pt_i], pb = sx[pt_i + 1];
                    const double coord_from = (pa > (double)col ? pa : (double)col) + PL_EPS;
                    const double coord_to = (pb < (double)(col + 1) ? pb : (double)(col + 1)) - PL_EPS;
                    const double significance = coord_to - coord_from;
                    const double coord_center = coord_from + 0.5 * significance;
                    // (a) entering segments, in sorted order
                    for (;;) {
                        const int idx = sg_pointer + lane;
                        const bool ok = idx < sg_end && sx[idx < sg_end ? idx : sg_end - 1] < coord_center;
                        const unsigned long long m = __ballot(ok);
                        const int nadd = m == ~0ull ? 64 : __builtin_ctzll(~m);
                        if (lane < nadd && csg_end + lane < csg_cap) csg[csg_end + lane] = sk[idx];
                        if (nadd > 0 && csg_end == 0) slot0_written = true;
                        sg_pointer += nadd; csg_end += nadd;
                        if (nadd < 64) break;
                    }
                    if (csg_end > csg_cap) csg_end = csg_cap;
                    __syncthreads();
                    int best = 0;
                    if (csg_end < coop_min) {
                        // a small active set: the two scans as the reference writes them, executed by every lane alike (the same
                        // values into the same LDS words) -- ballots, list building and barriers cost more than they save here
                        int csg_i = 0;
                        while (csg_i < csg_end) {
                            const int k = csg[csg_i];
                            if (OX(k + 1) < coord_center) { csg[csg_i] = csg[csg_end - 1]; csg_end--; }
                            else csg_i++;
                        }
                        if (csg_end != 1) {
                            double best_closeness = -PL_EPS;
                            for (csg_i = 0; csg_i < csg_end; csg_i++) {
                                const int k = csg[csg_i];
                                const double x0 = OX(k), x1 = OX(k + 1);
                                const double ip_k = (coord_center - x0) / (x1 - x0);
                                const double closeness = (1.0 - ip_k) * OD(k) + ip_k * OD(k + 1);
                                if (best_closeness < closeness && 0.0 < ip_k && ip_k < 1.0) { best_closeness = closeness; best = csg_i; }
                            }
                        }
                    } else {
                    // (b) leaving segments: the swap-remove scan as a permutation of the keep flags
                    if (csg_end > 0) {
                        const int n = csg_end, nch = (n + 63) >> 6;
                        int K = 0;
                        for (int ch = 0; ch < nch; ++ch) {
                            const int p = 64 * ch + lane;
                            bool keep = false;
                            if (p < n) keep = !(OX(csg[p] + 1) < coord_center);
                            const unsigned long long m = __ballot(keep);
                            if (lane == 0) kmask[ch] = m;
                            K += __builtin_popcountll(m);
                        }
                        __syncthreads();
                        if (K == 0) {
                            if (n >= 2 && lane == 0) csg[0] = csg[1];                // what the sequential scan leaves in the stale slot
                        } else if (K < n) {
                            const int nholes = K - [&] { int kept = 0; for (int ch = 0; ch <= (K - 1) >> 6; ++ch) { unsigned long long m = kmask[ch]; const int top = K - 64 * ch; if (top < 64) m &= (1ull << top) - 1ull; kept += __builtin_popcountll(m); } return kept; }();
                            if (nholes > PLX_SCR) {
                                if (lane == 0) {                                     // a removal larger than the scratch lists: the scan itself
                                    int e = n, i = 0;
                                    while (i < e) { const int k = csg[i]; if (OX(k + 1) < coord_center) { csg[i] = csg[e - 1]; e--; } else i++; }
                                }
                            } else if (nholes > 0) {
                                int base = 0;
                                for (int ch = 0; ch <= (K - 1) >> 6; ++ch) {         // removed positions below K, ascending
                                    const int p = 64 * ch + lane;
                                    unsigned long long hm = ~kmask[ch];
                                    const int top = K - 64 * ch;
                                    if (top < 64) hm &= (1ull << top) - 1ull;
                                    if ((hm >> lane) & 1ull) holes[base + __builtin_popcountll(hm & ((1ull << lane) - 1ull))] = p;
                                    base += __builtin_popcountll(hm);
                                }
                                base = 0;
                                for (int ch = nch - 1; ch >= K >> 6; --ch) {         // kept positions from K up, descending
                                    const int p = 64 * ch + lane;
                                    unsigned long long dm = kmask[ch];
                                    const int lo = K - 64 * ch;
                                    if (lo > 0) dm &= ~((1ull << lo) - 1ull);
                                    if ((dm >> lane) & 1ull) donors[base + __builtin_popcountll(lane == 63 ? 0ull : dm >> (lane + 1))] = p;
                                    base += __builtin_popcountll(dm);
                                }
                                __syncthreads();
                                for (int j = lane; j < nholes; j += 64) csg[holes[j]] = csg[donors[j]];      // disjoint: holes < K <= donors
                            }
                        }
                        csg_end = K;
                        __syncthreads();
                    }
                    // (c) the winner: the first index that attains the maximal closeness among the valid candidates
                    if (csg_end != 1) {
                        double bc = -PL_EPS;
                        int bi = 0x7fffffff;
                        for (int i = lane; i < csg_end; i += 64) {
                            const int k = csg[i];
                            const double x0 = OX(k), x1 = OX(k + 1);
                            const double ip_k = (coord_center - x0) / (x1 - x0);
                            const double closeness = (1.0 - ip_k) * OD(k) + ip_k * OD(k + 1);
                            if (bc < closeness && 0.0 < ip_k && ip_k < 1.0) { bc = closeness; bi = i; }
                        }
#pragma unroll
                        for (int sft = 32; sft > 0; sft >>= 1) {
                            const double oc = __shfl_xor(bc, sft, 64);
                            const int oi = __shfl_xor(bi, sft, 64);
                            if (oi != 0x7fffffff && (bi == 0x7fffffff || oc > bc || (oc == bc && oi < bi))) { bc = oc; bi = oi; }
                        }
                        if (bi != 0x7fffffff) best = bi;
                    }
                    }
                    const int k = (csg_end > 0 || slot0_written) ? csg[best] : -1;
                    if (k >= 0) {
                        const int col_l = CI_(k), col_r = CI_(k + 1);
                        const uint32_t pl = rgbx[col_l];
                        if (col_l == col_r) {
                            for (int q = 0; q < 4; q++) if (q < c) color[q] += (double)((pl >> (8 * q)) & 0xffu) * significance;
                        } else {
                            const uint32_t pr = rgbx[col_r];
                            const double x0 = OX(k), x1 = OX(k + 1);
                            const double ip_k = (coord_center - x0) / (x1 - x0);
                            for (int q = 0; q < 4; q++) if (q < c) {
                                const double u = (double)((pl >> (8 * q)) & 0xffu) * (1.0 - ip_k);
                                const double v = (double)((pr >> (8 * q)) & 0xffu) * ip_k;
                                color[q] += (u + v) * significance;
                            }
                        }
                    } else {
                        const uint32_t p0 = rgbx[0];
                        for (int q = 0; q < 4; q++) if (q < c) color[q] += (double)((p0 >> (8 * q)) & 0xffu) * significance;
                    }
                    pt_i++;
                }
                if (lane == 0) for (int q = 0; q < 4; q++) if (q < c) dst[(size_t)col * c + q] = ds_f64_to_u8(color[q]);
            }
        } else if (lane == 0) {
            // ---- the sweep (:228-282), statement by statement as in k_polylines_exact, by ONE lane (rounds 3-4; DS_PL_EXACT_COOP=0) ----
            int csg_end = 0, sg_pointer = 0, pt_i = 0;
            bool slot0_written = false;
            for (int col = 0; col < w; col++) {
                double color[4] = { 0.5, 0.5, 0.5, 0.5 };
                while (sx[pt_i] < (double)col) pt_i++;
                pt_i--;
                while (sx[pt_i] < (double)(col + 1)) {
                    const double pa = sx[pt_i], pb = sx[pt_i + 1];
                    const double coord_from = (pa > (double)col ? pa : (double)col) + PL_EPS;
                    const double coord_to = (pb < (double)(col + 1) ? pb : (double)(col + 1)) - PL_EPS;
                    const double significance = coord_to - coord_from;
                    const double coord_center = coord_from + 0.5 * significance;
                    while (sg_pointer < sg_end && sx[sg_pointer] < coord_center) {
                        if (csg_end < csg_cap) csg[csg_end] = sk[sg_pointer];      // (the host sized csg_cap above the bound: never clamps)
                        if (csg_end == 0) slot0_written = true;
                        sg_pointer++; csg_end++;
                    }
                    if (csg_end > csg_cap) csg_end = csg_cap;
                    int csg_i = 0;
                    while (csg_i < csg_end) {
                        const int k = csg[csg_i];
                        if (OX(k + 1) < coord_center) { csg[csg_i] = csg[csg_end - 1]; csg_end--; }
                        else csg_i++;
                    }
                    int best = 0;
                    if (csg_end != 1) {
                        double best_closeness = -PL_EPS;
                        for (csg_i = 0; csg_i < csg_end; csg_i++) {
                            const int k = csg[csg_i];
                            const double x0 = OX(k), x1 = OX(k + 1);
                            const double ip_k = (coord_center - x0) / (x1 - x0);
                            const double closeness = (1.0 - ip_k) * OD(k) + ip_k * OD(k + 1);
                            if (best_closeness < closeness && 0.0 < ip_k && ip_k < 1.0) { best_closeness = closeness; best = csg_i; }
                        }
                    }
                    const int k = (csg_end > 0 || slot0_written) ? csg[best] : -1;
                    if (k >= 0) {
                        const int col_l = CI_(k), col_r = CI_(k + 1);
                        const uint32_t pl = rgbx[col_l];
                        if (col_l == col_r) {
                            for (int q = 0; q < 4; q++) if (q < c) color[q] += (double)((pl >> (8 * q)) & 0xffu) * significance;
                        } else {
                            const uint32_t pr = rgbx[col_r];
                            const double x0 = OX(k), x1 = OX(k + 1);
                            const double ip_k = (coord_center - x0) / (x1 - x0);
                            for (int q = 0; q < 4; q++) if (q < c) {
                                const double u = (double)((pl >> (8 * q)) & 0xffu) * (1.0 - ip_k);
                                const double v = (double)((pr >> (8 * q)) & 0xffu) * ip_k;
                                color[q] += (u + v) * significance;
                            }
                        }
                    } else {
                        const uint32_t p0 = rgbx[0];
                        for (int q = 0; q < 4; q++) if (q < c) color[q] += (double)((p0 >> (8 * q)) & 0xffu) * significance;
                    }
                    pt_i++;
                }
                for (int q = 0; q < 4; q++) if (q < c) dst[(size_t)col * c + q] = ds_f64_to_u8(color[q]);
            }
#undef CI_
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Exact fallback, LDS resident, SPECULATIVE CHUNKS (round 6).  The sweep of a flagged row is a chain of dependent LDS round
// trips -- ~5 us per column on one wave: 9.5 ms for one 1920-column row of a 1080p batch, with one or two workgroups at work and the
// stream waiting.  What is history in the reference's sweep (:228-281) is the ORDER of the active set (the swap-remove scan) and the
// two pointers; WHICH segments are in the set at a centre is a local property (start < centre <= end), and where the set holds exactly
// ONE segment its order is trivial.  So NW waves of the workgroup sweep NW chunks of the row side by side: wave k > 0 looks for a
// column near its nominal start whose first sub-interval has, computed locally, exactly one covering segment, ASSUMES the state
// {pt_i, sg_pointer, csg = [that segment]} there and sweeps from it; every wave, at the end of its chunk, runs the first
// sub-interval of the next chunk's start column up to the removal step ("peek": adding and removing are idempotent at one centre) and
// compares its TRUE state with what the successor assumed.  A chunk is valid when its predecessor is valid and the hand-over matches
// -- from identical state the same deterministic code produces the same bytes.  After a barrier wave 0 walks the chain and re-sweeps,
// from the true state, every chunk whose assumption did not hold (anomalies of the history -- non-monotone centres, a set emptied
// and refilled -- can reach at most one segment length ahead, so this is rare; correctness never depends on the speculation).
// Rows with NaN coordinates (a constant depth map) and rows whose LDS image leaves no room for NW active sets take one chunk.
// The sweep itself is k_polylines_exact_lds's cooperative one, statement for statement, with its workgroup barriers replaced by
// wavefront fences (one wave's LDS operations execute in order).
struct PlxState { int pt_i, sg_pointer, csg_end, slot0; };
#define PLX_MAXW 16
#define PLX_WSYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
template <int DT, int SHARP>
__global__ __launch_bounds__(512) void k_polylines_exact_chunked(PolyParams P, int c, int win_pts, int csg_cap, int coop_min, int scan_pts, int break_spec, int scr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NP = SHARP ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
    const int w = P.w;
    const int pt_end = NP * w + 2, sg_end = pt_end - 1;
    double *cd = reinterpret_cast<double *>(smem);           // coord_d per column
    double *sx = cd + w;                                     // sorted x
    int *sk = reinterpret_cast<int *>(sx + pt_end);          // original index of the sorted point
    uint32_t *rgbx = reinterpret_cast<uint32_t *>(sk + pt_end);
    int *shared_i = reinterpret_cast<int *>(rgbx + w);       // per-row control block
    int *start_col = shared_i;                               // [PLX_MAXW + 1]: start column of every wave's chunk (-1: none), then w
    int *a_pt = start_col + PLX_MAXW + 1, *a_sgp = a_pt + PLX_MAXW, *a_seg = a_sgp + PLX_MAXW;      // assumed state at the chunk's start
    int *match = a_seg + PLX_MAXW;                           // [k]: the predecessor's true state equals chunk k's assumption
    int *fin = match + PLX_MAXW;                             // [4 k ..]: pt_i, sg_pointer, csg_end, slot0 of wave k after its peek
    int *s_nan = fin + 4 * PLX_MAXW;
    // per-wave arrays: active set, hole / donor lists, keep masks
    const int per_wave_ints = csg_cap + 2 * scr + 2 * ((csg_cap + 63) / 64) + 2;   // scr: capacity of the hole / donor lists (PLX_SCR, or less on wide rows)
    int *wave_base = reinterpret_cast<int *>((reinterpret_cast<uintptr_t>(s_nan + 2) + 7) & ~(uintptr_t)7);
    auto CSG = [&](int k) -> int * { return wave_base + (size_t)k * per_wave_ints; };
    const int count = P.counters[0];
    for (int it = blockIdx.x; it < count; it += gridDim.x) {
        const int rowid = P.row_list[it];
        const int row = rowid % P.h;
        const int ie = rowid / P.h;
        const int eye = ie % P.n_eyes, img = ie / P.n_eyes;
        const double div_px = P.div_px[eye], sep_px = P.sep_px[eye];
        const double mn = P.minmax[img * 2], mx = P.minmax[img * 2 + 1];
        const uint8_t *src = P.img + ((size_t)img * P.h + row) * (size_t)w * c;
        typedef typename ds_depth_traits<DT>::T DTy;
        const DTy *depth_row = (const DTy *)P.depth + ((size_t)img * P.h + row) * (size_t)w;
        uint8_t *dst = P.out[eye] + (int64_t)img * P.ois[eye] + (int64_t)row * P.ors[eye];
        auto OX = [&](const int p) -> double {
            if (p == 0) return -1.0 * (double)w;
            if (p == pt_end - 1) return 2.0 * (double)w;
            const int col = SHARP ? (p - 1) >> 1 : p - 1;
            const double coord_x = (double)col + 0.5 + cd[col] + sep_px;
            if (!SHARP) return coord_x;
            return ((p - 1) & 1) ? coord_x + 0.45 : coord_x - 0.45;
        };
        auto OD = [&](const int p) -> double {
            if (p == 0 || p == pt_end - 1) return 0.0;
            return fabs(cd[SHARP ? (p - 1) >> 1 : p - 1]);
        };
        __syncthreads();                                     // the previous row's arrays are no longer read
        if (tid == 0) s_nan[0] = 0;
        __syncthreads();
        // ---- coord_d and the colour of every column, all threads ----
        bool nan_here = false;
        for (int col = tid; col < w; col += nthr) {
            const double v = pl_coord_d<DT>(P, img, depth_row, col, mn, mx, div_px);
            cd[col] = v;
            nan_here |= !(v == v);
            uint32_t px = 0;
            for (int q = 0; q < c; q++) px |= (uint32_t)src[(size_t)col * c + q] << (8 * q);
            rgbx[col] = px;
        }
        if (nan_here) s_nan[0] = 1;
        __syncthreads();
        // ---- stable sort of points 0 .. sg_end - 1 by x (as in k_polylines_exact_lds) ----
        for (int i = tid; i < pt_end; i += nthr) {
            const double xi = OX(i);
            int pos = i;
            if (i < sg_end && xi == xi) {
                const int lo = max(0, i - win_pts), hi = min(sg_end - 1, i + win_pts);
                int less = lo;
                for (int j = lo; j <= hi; j++) {
                    const double xj = OX(j);
                    less += (xj < xi || (xj == xi && j < i)) ? 1 : 0;
                }
                pos = less;
            }
            sx[pos] = xi;
            sk[pos] = i;
        }
        __syncthreads();
        const bool one_chunk = s_nan[0] != 0 || nw == 1;
#define CI_(p) ((p) == 0 ? 0 : ((p) == pt_end - 1 ? (w - 1) : (SHARP ? ((p) - 1) >> 1 : (p) - 1)))
        // first index in [lo, hi) whose sorted x is not below v (the sorted array has no NaN here)
        auto lower_bound = [&](int lo, int hi, const double v) -> int {
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (sx[mid] < v) lo = mid + 1; else hi = mid; }
            return lo;
        };
        // ---- the sweep (:228-282) of columns [cb, ce) by ONE wave on the active set `csg`; peek: only the first sub-interval of column
        // cb up to the removal step (state after it in st, nothing written) ----
        auto sweep = [&](int *csg, const int cb, const int ce, PlxState &st, const bool peek) {
            int *holes = csg + csg_cap, *donors = holes + scr;
            unsigned long long *kmask = reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(donors + scr) + 7) & ~(uintptr_t)7);
            int csg_end = st.csg_end, sg_pointer = st.sg_pointer, pt_i = st.pt_i;
            bool slot0_written = st.slot0 != 0;
            for (int col = cb; col < (peek ? cb + 1 : ce); col++) {
                double color[4] = { 0.5, 0.5, 0.5, 0.5 };
                while (sx[pt_i] < (double)col) pt_i++;
                pt_i--;
                while (sx[pt_i] < (double)(col + 1)) {
                    const double pa = sx[pt_i], pb = sx[pt_i + 1];
                    const double coord_from = (pa > (double)col ? pa : (double)col) + PL_EPS;
                    const double coord_to = (pb < (double)(col + 1) ? pb : (double)(col + 1)) - PL_EPS;
                    const double significance = coord_to - coord_from;
                    const double coord_center = coord_from + 0.5 * significance;
                    // (a) entering segments, in sorted order
                    for (;;) {
                        const int idx = sg_pointer + lane;
                        const bool ok = idx < sg_end && sx[idx < sg_end ? idx : sg_end - 1] < coord_center;
                        const unsigned long long m = __ballot(ok);
                        const int nadd = m == ~0ull ? 64 : __builtin_ctzll(~m);
                        if (lane < nadd && csg_end + lane < csg_cap) csg[csg_end + lane] = sk[idx];
                        if (nadd > 0 && csg_end == 0) slot0_written = true;
                        sg_pointer += nadd; csg_end += nadd;
                        if (nadd < 64) break;
                    }
                    if (csg_end > csg_cap) csg_end = csg_cap;
                    PLX_WSYNC();
                    int best = 0;
                    if (csg_end < coop_min) {
                        int csg_i = 0;
                        while (csg_i < csg_end) {
                            const int k = csg[csg_i];
                            if (OX(k + 1) < coord_center) { csg[csg_i] = csg[csg_end - 1]; csg_end--; }
                            else csg_i++;
                        }
                        if (peek) break;
                        if (csg_end != 1) {
                            double best_closeness = -PL_EPS;
                            for (csg_i = 0; csg_i < csg_end; csg_i++) {
                                const int k = csg[csg_i];
                                const double x0 = OX(k), x1 = OX(k + 1);
                                const double ip_k = (coord_center - x0) / (x1 - x0);
                                const double closeness = (1.0 - ip_k) * OD(k) + ip_k * OD(k + 1);
                                if (best_closeness < closeness && 0.0 < ip_k && ip_k < 1.0) { best_closeness = closeness; best = csg_i; }
                            }
                        }
                    } else {
                    // (b) leaving segments: the swap-remove scan as a permutation of the keep flags
                    if (csg_end > 0) {
                        const int n = csg_end, nch = (n + 63) >> 6;
                        int K = 0;
                        for (int ch = 0; ch < nch; ++ch) {
                            const int p = 64 * ch + lane;
                            bool keep = false;
                            if (p < n) keep = !(OX(csg[p] + 1) < coord_center);
                            const unsigned long long m = __ballot(keep);
                            if (lane == 0) kmask[ch] = m;
                            K += __builtin_popcountll(m);
                        }
                        PLX_WSYNC();
                        if (K == 0) {
                            if (n >= 2 && lane == 0) csg[0] = csg[1];                // what the sequential scan leaves in the stale slot
                        } else if (K < n) {
                            const int nholes = K - [&] { int kept = 0; for (int ch = 0; ch <= (K - 1) >> 6; ++ch) { unsigned long long m = kmask[ch]; const int top = K - 64 * ch; if (top < 64) m &= (1ull << top) - 1ull; kept += __builtin_popcountll(m); } return kept; }();
                            if (nholes > scr) {
                                if (lane == 0) {
                                    int e = n, i = 0;
                                    while (i < e) { const int k = csg[i]; if (OX(k + 1) < coord_center) { csg[i] = csg[e - 1]; e--; } else i++; }
                                }
                            } else if (nholes > 0) {
                                int base = 0;
                                for (int ch = 0; ch <= (K - 1) >> 6; ++ch) {         // removed positions below K, ascending
                                    const int p = 64 * ch + lane;
                                    unsigned long long hm = ~kmask[ch];
                                    const int top = K - 64 * ch;
                                    if (top < 64) hm &= (1ull << top) - 1ull;
                                    if ((hm >> lane) & 1ull) holes[base + __builtin_popcountll(hm & ((1ull << lane) - 1ull))] = p;
                                    base += __builtin_popcountll(hm);
                                }
                                base = 0;
                                for (int ch = nch - 1; ch >= K >> 6; --ch) {         // kept positions from K up, descending
                                    const int p = 64 * ch + lane;
                                    unsigned long long dm = kmask[ch];
                                    const int lo = K - 64 * ch;
                                    if (lo > 0) dm &= ~((1ull << lo) - 1ull);
                                    if ((dm >> lane) & 1ull) donors[base + __builtin_popcountll(lane == 63 ? 0ull : dm >> (lane + 1))] = p;
                                    base += __builtin_popcountll(dm);
                                }
                                PLX_WSYNC();
                                for (int j = lane; j < nholes; j += 64) csg[holes[j]] = csg[donors[j]];      // disjoint: holes < K <= donors
                            }
                        }
                        csg_end = K;
                        PLX_WSYNC();
                    }
                    if (peek) break;
                    // (c) the winner: the first index that attains the maximal closeness among the valid candidates
                    if (csg_end != 1) {
                        double bc = -PL_EPS;
                        int bi = 0x7fffffff;
                        for (int i = lane; i < csg_end; i += 64) {
                            const int k = csg[i];
                            const double x0 = OX(k), x1 = OX(k + 1);
                            const double ip_k = (coord_center - x0) / (x1 - x0);
                            const double closeness = (1.0 - ip_k) * OD(k) + ip_k * OD(k + 1);
                            if (bc < closeness && 0.0 < ip_k && ip_k < 1.0) { bc = closeness; bi = i; }
                        }
#pragma unroll
                        for (int sft = 32; sft > 0; sft >>= 1) {
                            const double oc = __shfl_xor(bc, sft, 64);
                            const int oi = __shfl_xor(bi, sft, 64);
                            if (oi != 0x7fffffff && (bi == 0x7fffffff || oc > bc || (oc == bc && oi < bi))) { bc = oc; bi = oi; }
                        }
                        if (bi != 0x7fffffff) best = bi;
                    }
                    }
                    const int k = (csg_end > 0 || slot0_written) ? csg[best] : -1;
                    if (k >= 0) {
                        const int col_l = CI_(k), col_r = CI_(k + 1);
                        const uint32_t pl = rgbx[col_l];
                        if (col_l == col_r) {
                            for (int q = 0; q < 4; q++) if (q < c) color[q] += (double)((pl >> (8 * q)) & 0xffu) * significance;
                        } else {
                            const uint32_t pr = rgbx[col_r];
                            const double x0 = OX(k), x1 = OX(k + 1);
                            const double ip_k = (coord_center - x0) / (x1 - x0);
                            for (int q = 0; q < 4; q++) if (q < c) {
                                const double u = (double)((pl >> (8 * q)) & 0xffu) * (1.0 - ip_k);
                                const double v = (double)((pr >> (8 * q)) & 0xffu) * ip_k;
                                color[q] += (u + v) * significance;
                            }
                        }
                    } else {
                        const uint32_t p0 = rgbx[0];
                        for (int q = 0; q < 4; q++) if (q < c) color[q] += (double)((p0 >> (8 * q)) & 0xffu) * significance;
                    }
                    pt_i++;
                }
                if (!peek && lane == 0) for (int q = 0; q < 4; q++) if (q < c) dst[(size_t)col * c + q] = ds_f64_to_u8(color[q]);
            }
            st.csg_end = csg_end; st.sg_pointer = sg_pointer; st.pt_i = pt_i; st.slot0 = slot0_written ? 1 : 0;
        };
        if (one_chunk) {
            if (wv == 0) { PlxState st = { 0, 0, 0, 0 }; sweep(CSG(0), 0, w, st, false); }
            continue;
        }
        // ---- where every wave starts: wave 0 at column 0 with the empty state, wave k at the first column at or behind k w / nw whose
        // first sub-interval is covered by exactly one segment (set membership computed locally) ----
        if (lane == 0) { start_col[wv] = wv == 0 ? 0 : -1; match[wv] = 0; if (wv == 0) start_col[nw] = w; }
        if (wv > 0) {
            const int nominal = (int)(((long long)wv * w) / nw), limit = min((int)(((long long)(wv + 1) * w) / nw), nominal + 96);
            for (int col = nominal; col < limit; col++) {
                const int lb = lower_bound(0, pt_end, (double)col);
                if (lb < 1 || lb >= pt_end) continue;
                const int pt_i = lb - 1;
                const double pa = sx[pt_i], pb = sx[pt_i + 1];
                const double coord_from = (pa > (double)col ? pa : (double)col) + PL_EPS;
                const double coord_to = (pb < (double)(col + 1) ? pb : (double)(col + 1)) - PL_EPS;
                const double significance = coord_to - coord_from;
                if (!(significance > 1.0e-3)) continue;      // a sliver: its neighbourhood is where centres may stop being monotone
                const double coord_center = coord_from + 0.5 * significance;
                const int sgp0 = lower_bound(0, sg_end, coord_center);
                int covering = 0, seg = -1;
                for (int base = max(0, sgp0 - scan_pts); base < sgp0; base += 64) {
                    const int idx = base + lane;
                    bool in = false;
                    int k = -1;
                    if (idx < sgp0) { k = sk[idx]; in = !(OX(k + 1) < coord_center); }
                    const unsigned long long m = __ballot(in);
                    covering += __builtin_popcountll(m);
                    if (m != 0ull) seg = __shfl(k, __builtin_ctzll(m), 64);
                }
                if (covering == 1) {
                    if (lane == 0) { start_col[wv] = col; a_pt[wv] = pt_i; a_sgp[wv] = sgp0 + (break_spec ? -1 : 0); a_seg[wv] = seg; }
                    break;
                }
            }
        }
        __syncthreads();
        // ---- every wave with a start sweeps its chunk, peeks into the next one and compares ----
        int my_start = start_col[wv], nxt = -1, my_end = w;
        for (int k = wv + 1; k < nw; k++) if (start_col[k] >= 0) { nxt = k; my_end = start_col[k]; break; }
        if (my_start >= 0) {
            int *csg = CSG(wv);
            PlxState st = { 0, 0, 0, 0 };
            if (wv > 0) {
                st.pt_i = a_pt[wv]; st.sg_pointer = a_sgp[wv]; st.csg_end = 1; st.slot0 = 1;
                if (lane == 0) csg[0] = a_seg[wv];
                PLX_WSYNC();
            }
            sweep(csg, my_start, my_end, st, false);
            if (nxt >= 0) {
                sweep(csg, my_end, my_end + 1, st, true);
                PLX_WSYNC();
                const bool ok = st.pt_i == a_pt[nxt] && st.sg_pointer == a_sgp[nxt] && st.csg_end == 1 && csg[0] == a_seg[nxt];
                if (lane == 0) match[nxt] = ok ? 1 : 0;
            }
            if (lane == 0) { fin[4 * wv] = st.pt_i; fin[4 * wv + 1] = st.sg_pointer; fin[4 * wv + 2] = st.csg_end; fin[4 * wv + 3] = st.slot0; }
        }
        __syncthreads();
        // ---- wave 0 walks the chain: a chunk whose assumption did not hold is swept again from the true state, which lives in the
        // active-set array (and the saved scalars) of the wave that swept up to it ----
        if (wv == 0) {
            int holder = 0;                                  // its array + fin hold the TRUE state at the start of the next chunk
            for (int k = 1; k < nw; k++) {
                if (start_col[k] < 0) continue;
                if (match[k]) { holder = k; continue; }
                int ke = w, kn = -1;
                for (int j = k + 1; j < nw; j++) if (start_col[j] >= 0) { kn = j; ke = start_col[j]; break; }
                int *csg = CSG(holder);
                PlxState st = { fin[4 * holder], fin[4 * holder + 1], fin[4 * holder + 2], fin[4 * holder + 3] };
                sweep(csg, start_col[k], ke, st, false);
                if (kn >= 0) {
                    sweep(csg, ke, ke + 1, st, true);
                    PLX_WSYNC();
                    const bool ok = st.pt_i == a_pt[kn] && st.sg_pointer == a_sgp[kn] && st.csg_end == 1 && csg[0] == a_seg[kn];
                    if (lane == 0) match[kn] = ok ? 1 : 0;
                    PLX_WSYNC();
                }
                if (lane == 0) { fin[4 * holder] = st.pt_i; fin[4 * holder + 1] = st.sg_pointer; fin[4 * holder + 2] = st.csg_end; fin[4 * holder + 3] = st.slot0; }
                PLX_WSYNC();
            }
        }
#undef CI_
    }
}

template <int C, int SHARP, int NE>
static int pl_main_blocks(int ncu, long long nwork, size_t lds, long long *nblocks_out)
{
    auto kfn = k_polylines<C, SHARP, NE>;
    DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    DS_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, PL_THREADS, lds));
    DS_REQUIRE(per_cu > 0, DS_EUNSUPPORTED, "ds_stereo_warp: polylines kernel does not fit a compute unit (%zu bytes of LDS)", lds);
    long long nblocks = (long long)per_cu * ncu;
    { const char *e = getenv("DS_PL_BLOCKS"); if (e && atoi(e) > 0) nblocks = atoi(e); }
    if (nblocks > nwork) nblocks = nwork;
    *nblocks_out = nblocks;
    return DS_OK;
}

template <int C, int SHARP, int NE>
static int pl_launch_main(const PolyParams &P, long long nblocks, size_t lds, hipStream_t st)
{
    if (getenv("DS_PL_VERBOSE")) fprintf(stderr, "k_polylines<%d,%d>: S=%d nwmax=%d K=%d lds=%zu grid=%lld queue segment %d entries\n", C, SHARP, P.S, P.nwmax, P.K, lds, nblocks, P.gq_cap);
    hipLaunchKernelGGL((k_polylines<C, SHARP, NE>), dim3((unsigned)nblocks), dim3(PL_THREADS), lds, st, P);
    return DS_OK;
}

// dispatch on (channels, eyes): op 0 sizes the grid of the main kernel, 1 launches it, 2 launches the general-pixel pass
template <int SHARP>
static int pl_dispatch(int op, const PolyParams &P, int c, int ncu, long long nwork, size_t lds, long long *nblocks, int ncmax,
                       int per_seg, size_t glds, hipStream_t st)
{
#define PL_CASE(C_, NE_)                                                                                                \
    do {                                                                                                                \
        if (op == 0) return pl_main_blocks<C_, SHARP, NE_>(ncu, nwork, lds, nblocks);                                   \
        if (op == 1) return pl_launch_main<C_, SHARP, NE_>(P, *nblocks, lds, st);                                       \
        const int one_window = !(getenv("DS_PL_GEN_SHARED") && atoi(getenv("DS_PL_GEN_SHARED")) == 0);   /* A/B switch, read per call */ \
        if (getenv("DS_PL_GEN_WPE") && atoi(getenv("DS_PL_GEN_WPE")) == 6) {                                            \
            DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_polylines_general<C_, SHARP, 6>),         \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));                   \
            hipLaunchKernelGGL((k_polylines_general<C_, SHARP, 6>), dim3((unsigned)(P.gq_segments * per_seg)), dim3(64), glds, st, P, ncmax, per_seg, one_window); \
            return DS_OK;                                                                                               \
        }                                                                                                               \
        const bool kw_auto = !(getenv("DS_PL_GEN_KW") && atoi(getenv("DS_PL_GEN_KW")) == 4);   /* DS_PL_GEN_KW=4: always the four-word kernel */ \
        if (P.K == 1 && kw_auto) {                                                                                      \
            DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_polylines_general<C_, SHARP, 4, 1>),      \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));                   \
            hipLaunchKernelGGL((k_polylines_general<C_, SHARP, 4, 1>), dim3((unsigned)(P.gq_segments * per_seg)), dim3(64), glds, st, P, ncmax, per_seg, one_window); \
            return DS_OK;                                                                                               \
        }                                                                                                               \
        if (P.K == 2 && kw_auto) {                                                                                      \
            DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_polylines_general<C_, SHARP, 4, 2>),      \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));                   \
            hipLaunchKernelGGL((k_polylines_general<C_, SHARP, 4, 2>), dim3((unsigned)(P.gq_segments * per_seg)), dim3(64), glds, st, P, ncmax, per_seg, one_window); \
            return DS_OK;                                                                                               \
        }                                                                                                               \
        DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_polylines_general<C_, SHARP>),                \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));                       \
        hipLaunchKernelGGL((k_polylines_general<C_, SHARP>), dim3((unsigned)(P.gq_segments * per_seg)), dim3(64), glds, st, P, ncmax, per_seg, one_window); \
        return DS_OK;                                                                                                   \
    } while (0)
    if (P.n_eyes == 2) {
        switch (c) {
        case 1: PL_CASE(1, 2);
        case 2: PL_CASE(2, 2);
        case 3: PL_CASE(3, 2);
        default: PL_CASE(4, 2);
        }
    }
    switch (c) {
    case 1: PL_CASE(1, 1);
    case 2: PL_CASE(2, 1);
    case 3: PL_CASE(3, 1);
    default: PL_CASE(4, 1);
    }
#undef PL_CASE
}

// LDS of k_polylines_exact_lds for a row of w columns: coord_d (8 B) and a packed colour (4 B) per column, sorted x (8 B) and
// original index (4 B) per point, + the bounded active set
static size_t pl_exact_lds_fixed(int w, int sharp) { return (size_t)w * 12 + (size_t)((sharp ? 2 : 1) * w + 2) * 12 + 16; }
// + the cooperative sweep's scratch behind the active set: two position lists of PLX_SCR entries and one keep mask per 64 entries
static size_t pl_exact_lds_scratch(long long csg_cap) { return (size_t)2 * PLX_SCR * 4 + (size_t)((csg_cap + 63) / 64) * 8 + 16; }

template <int DT>
static int pl_launch_exact(const PolyParams &P, int sharp, const ExactScratch &S, int exact_blocks, int c, double max_div_px, int device, hipStream_t st)
{
    const size_t fixed = pl_exact_lds_fixed(P.w, sharp);
    const int force_global = getenv("DS_PL_EXACT_GLOBAL") ? atoi(getenv("DS_PL_EXACT_GLOBAL")) : 0;      // A/B switch (tests)
    const int coop = !(getenv("DS_PL_EXACT_COOP") && atoi(getenv("DS_PL_EXACT_COOP")) == 0);            // A/B switch: 0 = the one-lane sweep
    // active-set size from which the cooperative sweep spreads its scans over the lanes (below it: the sequential scans, by every lane).
    // Measured on 3840-column rows (tools/exact_sweep_probe.py, 80 flagged rows): always parallel 27.5 ms, parallel from 24 entries 70.0 ms,
    // one lane 72.1 ms -- the parallel scans win on small sets too (one round trip instead of one per entry); sets of 0 / 1 entries have
    // nothing to scan.  On config 5 (1080p, 0-2 flagged rows per launch) the choice is inside the run-to-run noise.
    const int coop_min = getenv("DS_PL_EXACT_COOP_MIN") ? atoi(getenv("DS_PL_EXACT_COOP_MIN")) : 2;
    // the active set holds at most NP (3 |divergence_px| + 8) segments (see the kernel); whatever the CU's LDS leaves after the
    // row's arrays (and the cooperative sweep's scratch), up to 8192 entries, is its capacity
    const long long csg_need = (long long)(sharp ? 2 : 1) * (3 * (long long)ceil(fabs(max_div_px)) + 8);
    long long csg_cap = 0;
    const size_t room = 160 * 1024;
    if (fixed + pl_exact_lds_scratch(8192) + 8192 * 4 <= room) csg_cap = 8192;
    else if (fixed + pl_exact_lds_scratch(64) + 64 * 4 <= room) {
        csg_cap = (long long)((room - fixed - 2 * PLX_SCR * 4 - 32) / 4);
        csg_cap = csg_cap * 64 / 66 / 64 * 64;               // leave 8 bytes of mask per 64 entries, whole chunks
        if (csg_cap > 8192) csg_cap = 8192;
    }
    // round 6: speculative chunks (k_polylines_exact_chunked) -- NW waves sweep NW chunks of the row side by side, each with an active set
    // of its own: as many as the CU's LDS leaves room for behind the row's arrays (8 at 1920 columns, one or two at 3840: then the
    // single-chunk kernel below).  DS_PL_EXACT_CHUNKS=<n> caps the count (1 = off), DS_PL_EXACT_BREAK=1 makes every assumption wrong
    // (tests: every chunk is then swept again from the true state and the bytes must not change).
    {
        const int want = getenv("DS_PL_EXACT_CHUNKS") ? atoi(getenv("DS_PL_EXACT_CHUNKS")) : 8;
        const long long cap_w = (csg_need + 64 + 63) / 64 * 64;
        const size_t ctrl = (size_t)(PLX_MAXW + 1 + 4 * PLX_MAXW + 4 * PLX_MAXW + 2) * 4 + 16;
        // the hole / donor lists of the cooperative removal: PLX_SCR entries each where the LDS allows eight chunks with them, 128 on
        // wide rows (3840 columns: six chunks instead of three; a removal of more holes than that takes the sequential scan)
        int scr = PLX_SCR;
        size_t per_wave = (size_t)(cap_w + 2 * scr + 2 * ((cap_w + 63) / 64) + 2) * 4;
        if (fixed + ctrl + 8 * per_wave > room) { scr = 128; per_wave = (size_t)(cap_w + 2 * scr + 2 * ((cap_w + 63) / 64) + 2) * 4; }
        int nw = 0;
        if (fixed + ctrl + 2 * per_wave <= room) nw = (int)((room - fixed - ctrl) / per_wave);
        if (nw > 8) nw = 8;
        if (nw > want) nw = want;
        if (coop && !force_global && nw >= 2 && cap_w <= 8192 && P.w >= 64 * nw) {
            static std::atomic<uint64_t> attr2_done{0};
            const uint64_t bit2 = 1ull << (device & 63);
            if (!(attr2_done.load(std::memory_order_relaxed) & bit2)) {
                DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_polylines_exact_chunked<DT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_polylines_exact_chunked<DT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr2_done.fetch_or(bit2, std::memory_order_relaxed);
            }
            const int win_pts2 = (sharp ? 2 : 1) * ((int)ceil(fabs(max_div_px)) + 3);
            const int scan_pts = (sharp ? 2 : 1) * (2 * (int)ceil(fabs(max_div_px)) + 6);
            const int break_spec = getenv("DS_PL_EXACT_BREAK") ? atoi(getenv("DS_PL_EXACT_BREAK")) : 0;
            const size_t lds2 = fixed + ctrl + (size_t)nw * per_wave;
            if (sharp) hipLaunchKernelGGL((k_polylines_exact_chunked<DT, 1>), dim3(1024), dim3(64 * nw), lds2, st, P, c, win_pts2, (int)cap_w, coop_min, scan_pts, break_spec, scr);
            else hipLaunchKernelGGL((k_polylines_exact_chunked<DT, 0>), dim3(1024), dim3(64 * nw), lds2, st, P, c, win_pts2, (int)cap_w, coop_min, scan_pts, break_spec, scr);
            return DS_OK;
        }
    }
    if (csg_cap >= csg_need && csg_cap >= 64 && !force_global && fixed + (size_t)csg_cap * 4 + pl_exact_lds_scratch(csg_cap) <= room) {
        const size_t lds = fixed + (size_t)csg_cap * 4 + pl_exact_lds_scratch(csg_cap);
        // one workgroup per flagged row, rows taken round-robin by a fixed grid (the count lives on the device)
        const int win_pts = (sharp ? 2 : 1) * ((int)ceil(fabs(max_div_px)) + 3);
        static std::atomic<uint64_t> attr_done{0};           // per device (bit) -- the limit is raised to the CU's whole LDS once
        const uint64_t bit = 1ull << (device & 63);
        if (!(attr_done.load(std::memory_order_relaxed) & bit)) {
            DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_polylines_exact_lds<DT, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_polylines_exact_lds<DT, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_polylines_exact_lds<DT, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            DS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_polylines_exact_lds<DT, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_done.fetch_or(bit, std::memory_order_relaxed);
        }
        if (sharp && coop) hipLaunchKernelGGL((k_polylines_exact_lds<DT, 1, 1>), dim3(1024), dim3(64), lds, st, P, c, win_pts, (int)csg_cap, coop_min);
        else if (sharp) hipLaunchKernelGGL((k_polylines_exact_lds<DT, 1, 0>), dim3(1024), dim3(64), lds, st, P, c, win_pts, (int)csg_cap, coop_min);
        else if (coop) hipLaunchKernelGGL((k_polylines_exact_lds<DT, 0, 1>), dim3(1024), dim3(64), lds, st, P, c, win_pts, (int)csg_cap, coop_min);
        else hipLaunchKernelGGL((k_polylines_exact_lds<DT, 0, 0>), dim3(1024), dim3(64), lds, st, P, c, win_pts, (int)csg_cap, coop_min);
        return DS_OK;
    }
    if (sharp) hipLaunchKernelGGL((k_polylines_exact<DT, 1>), dim3(exact_blocks), dim3(64), 0, st, P, S);
    else hipLaunchKernelGGL((k_polylines_exact<DT, 0>), dim3(exact_blocks), dim3(64), 0, st, P, S);
    return DS_OK;
}

// called from ds_stereo_warp (ds_stereo.hip)
int ds_polylines_launch(ds_ctx *ctx, const uint8_t *image, const void *depth, int depth_dtype, const double *minmax,
                        const double *lut, int n, int h, int w, int c, int sharp, const ds_eye *eyes, int n_eyes,
                        hipStream_t st)
{
    PolyParams P;
    memset(&P, 0, sizeof(P));
    P.img = image; P.depth = depth; P.depth_dtype = depth_dtype; P.minmax = minmax; P.lut = lut;
    P.n = n; P.h = h; P.w = w; P.n_eyes = n_eyes;
#ifdef DS_EXPERIMENTS
    { const char *e = getenv("DS_PL_DEBUG"); P.dbg = e ? atoi(e) : 0; }
#endif
    DS_REQUIRE(depth_dtype == DS_DEPTH_U16 || depth_dtype == DS_DEPTH_F32 || depth_dtype == DS_DEPTH_F64, DS_EINVAL,
               "unknown depth dtype %d", depth_dtype);
    for (int e = 0; e < n_eyes; e++) {
        const double dv = eyes[e].divergence_px, sp = eyes[e].separation_px;
        DS_REQUIRE(dv == dv && sp == sp && fabs(dv) < 1e6 && fabs(sp) < 1e6, DS_EINVAL, "ds_stereo_warp: divergence/separation not finite");
        // the reference's insertion sort never moves the 2w sentinel and its sweep assumes the -w one stays first
        DS_REQUIRE(fabs(dv) + fabs(sp) + 2.0 < (double)w, DS_EUNSUPPORTED,
                   "ds_stereo_warp: |divergence_px| + |separation_px| must stay below the image width for polylines");
        const double dmin = dv < 0 ? dv : 0.0, dmax = dv > 0 ? dv : 0.0;
        P.div_px[e] = dv; P.sep_px[e] = sp;
        P.out[e] = eyes[e].out; P.ors[e] = eyes[e].out_row_stride; P.ois[e] = eyes[e].out_img_stride;
        // source columns that can reach output pixel col: [col + offL, col + offU] (one column of slack each side)
        P.offL[e] = (int)floor(-1.95 - sp - dmax) - 1;
        P.offU[e] = (int)ceil(0.95 - sp - dmin) + 1;
    }
    int uL = P.offL[0], uU = P.offU[0];
    if (n_eyes > 1) { uL = P.offL[1] < uL ? P.offL[1] : uL; uU = P.offU[1] > uU ? P.offU[1] : uU; }
    P.K = 1; P.nseg = 1;
    for (int e = 0; e < n_eyes; e++) {
        const int nseg = (sharp ? 2 : 1) * (P.offU[e] - P.offL[e] + 3) + 2;
        if (nseg > P.nseg) P.nseg = nseg;
        if ((nseg + 63) / 64 > P.K) P.K = (nseg + 63) / 64;
    }
    P.al4 = ((w & 3) == 0 && (((uintptr_t)depth) & 15) == 0 && (((uintptr_t)image) & 3) == 0) ? 1 : 0;

    // supertile: the largest S (multiple of 64, <= 32704 so a pixel fits the 15-bit queue entry) whose staging fits the LDS budget
    const int np = sharp ? 2 : 1;
    const int wr = (w + 63) / 64 * 64;
    int S = 512;
    { const char *e = getenv("DS_PL_S"); if (e && atoi(e) >= 64) S = atoi(e) / 64 * 64; }
    if (S > wr) S = wr;
    size_t lds_budget = 64 * 1024;
    { const char *e = getenv("DS_PL_LDS"); if (e && atoi(e) >= 8192) lds_budget = (size_t)atoi(e); }
    PlLds L;
    for (;;) {
        long long nw = (long long)S + (uU - uL + 1) + 8;
        if (nw > (long long)w + 4) nw = (long long)w + 4;
        nw = (nw + 3) / 4 * 4;
        DS_REQUIRE(nw < (1 << 20), DS_EUNSUPPORTED, "ds_stereo_warp: source window too large");
        P.S = S; P.nwmax = (int)nw;
        L = pl_lds_layout(S, P.nwmax, c, np);
        if ((size_t)L.total <= lds_budget || S == 64) break;
        S = (S / 2 + 63) / 64 * 64;
    }
    DS_REQUIRE((size_t)L.total <= 160 * 1024 && np * (long long)P.nwmax + PL_PT_PAD < 65536, DS_EUNSUPPORTED,
               "ds_stereo_warp: divergence/separation window of %d columns does not fit the LDS", uU - uL + 1);

    const int64_t nrows = (int64_t)n * n_eyes * h;
    DS_REQUIRE(nrows < (1ll << 30), DS_EUNSUPPORTED, "ds_stereo_warp: too many rows in one call");
    int rc = ds_ctx_reserve(ctx, &ctx->row_flags, &ctx->row_flags_bytes, (size_t)(nrows + 64) * sizeof(int));
    if (rc) return rc;
    rc = ds_ctx_reserve(ctx, &ctx->row_list, &ctx->row_list_bytes, (size_t)nrows * sizeof(int));
    if (rc) return rc;
    // exact-sweep scratch: fixed worker pool
    const int np_max = 2 * w + 5;
    int nworkers = 4096;
    if (nworkers > nrows) nworkers = (int)((nrows + 63) / 64 * 64);
    const size_t per_worker = (size_t)np_max * (3 * sizeof(double) + 2 * sizeof(int));
    rc = ds_ctx_reserve(ctx, &ctx->exact_ws, &ctx->exact_ws_bytes, per_worker * nworkers);
    if (rc) return rc;
    ExactScratch X;
    X.nworkers = nworkers; X.np_max = np_max; X.c = c;
    X.ox = (double *)ctx->exact_ws;
    X.od = X.ox + (size_t)np_max * nworkers;
    X.sx = X.od + (size_t)np_max * nworkers;
    X.sk = (int *)(X.sx + (size_t)np_max * nworkers);
    X.csg = X.sk + (size_t)np_max * nworkers;

    // row_flags[0..nrows) flags, then 16 ints of counters
    P.row_flags = (int *)ctx->row_flags;
    P.counters = P.row_flags + nrows;
    P.row_list = (int *)ctx->row_list;
    static const bool s_prof = getenv("DS_PL_PROF") != nullptr;
    P.prof = s_prof ? reinterpret_cast<unsigned long long *>(P.row_flags + ((nrows + 17) & ~(int64_t)1)) : nullptr;
    DS_HIP_CHECK(hipMemsetAsync(ctx->row_flags, 0, (size_t)(nrows + 64) * sizeof(int), st));

    static int s_ncu[64];
    DS_REQUIRE(ctx->device >= 0 && ctx->device < 64, DS_EUNSUPPORTED, "device ordinal %d", ctx->device);
    if (s_ncu[ctx->device] == 0) {
        int v = 0;
        DS_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device));
        s_ncu[ctx->device] = v > 0 ? v : 256;
    }
    const int tiles = (w + S - 1) / S;
    const long long nwork = (long long)n * h * tiles;
    DS_REQUIRE(nwork < (1ll << 31) - (1 << 20), DS_EUNSUPPORTED, "ds_stereo_warp: too many row tiles in one call");

    // grid of the main kernel first: the queue of general pixels has one segment per workgroup
    long long nblocks = 0;
    rc = sharp ? pl_dispatch<1>(0, P, c, s_ncu[ctx->device], nwork, (size_t)L.total, &nblocks, 0, 0, 0, st)
               : pl_dispatch<0>(0, P, c, s_ncu[ctx->device], nwork, (size_t)L.total, &nblocks, 0, 0, 0, st);
    if (rc) return rc;
    // capacity: half of the pixel-eyes of the call, at least 64 K and at most 32 M entries in total (a full segment sends
    // the row to the exact sweep: only speed is at stake)
    {
        long long total = (long long)n * n_eyes * h * (long long)w / 2;
        if (total < 65536) total = 65536;
        if (total > (32ll << 20)) total = 32ll << 20;
        { const char *e = getenv("DS_PL_QUEUE"); if (e && atoll(e) > 0) total = atoll(e); }
        long long per = (total + nblocks - 1) / nblocks;
        per = (per + 7) & ~7ll;
        P.gq_cap = (int)per; P.gq_segments = (int)nblocks;
        rc = ds_ctx_reserve(ctx, &ctx->tmp_b, &ctx->tmp_b_bytes, (size_t)nblocks * (size_t)per * 8 + (size_t)nblocks * sizeof(int) + 64);
        if (rc) return rc;
        P.gq = (unsigned long long *)ctx->tmp_b;
        P.gq_count = (int *)(P.gq + (size_t)nblocks * (size_t)per);
    }
    // the per-pixel source window of the second pass
    const int ncmax = (uU - uL) + 4;
    const PlGenLds G = pl_gen_layout(ncmax, np);
    const size_t glds = (size_t)G.stride * 8;
    if (glds > 160 * 1024) P.K = PL_KMAX + 1;             // window too large for the second pass too: rows go to the exact sweep
    const bool wide = P.K > PL_KMAX;                       // the main kernel then queues nothing
    int per_seg = 8;                                        // measured: 1.97 / 1.53 / 1.49 / 1.46 ms per launch at 1 / 2 / 4 / 8
    { const char *e = getenv("DS_PL_PER_SEG"); if (e && atoi(e) > 0) per_seg = atoi(e); }

    if (ctx->profile) (void)hipEventRecord(ctx->ev[0], st);
    rc = sharp ? pl_dispatch<1>(1, P, c, 0, 0, (size_t)L.total, &nblocks, 0, 0, 0, st)
               : pl_dispatch<0>(1, P, c, 0, 0, (size_t)L.total, &nblocks, 0, 0, 0, st);
    if (rc) return rc;
    if (!wide) {
        rc = sharp ? pl_dispatch<1>(2, P, c, 0, 0, 0, &nblocks, ncmax, per_seg, glds, st)
                   : pl_dispatch<0>(2, P, c, 0, 0, 0, &nblocks, ncmax, per_seg, glds, st);
        if (rc) return rc;
    }
    if (ctx->profile) { (void)hipEventRecord(ctx->ev[1], st); (void)hipEventRecord(ctx->ev[2], st); }
    {
        double max_div = 0.0;
        for (int e = 0; e < n_eyes; e++) max_div = fmax(max_div, fabs(eyes[e].divergence_px));
        switch (depth_dtype) {
        case DS_DEPTH_U16: rc = pl_launch_exact<DS_DEPTH_U16>(P, sharp, X, nworkers / 64, c, max_div, ctx->device, st); break;
        case DS_DEPTH_F32: rc = pl_launch_exact<DS_DEPTH_F32>(P, sharp, X, nworkers / 64, c, max_div, ctx->device, st); break;
        default: rc = pl_launch_exact<DS_DEPTH_F64>(P, sharp, X, nworkers / 64, c, max_div, ctx->device, st); break;
        }
        if (rc) return rc;
    }
    if (ctx->profile) { (void)hipEventRecord(ctx->ev[3], st); ctx->ev_recorded = 1; }
    DS_HIP_CHECK(hipGetLastError());
    if (s_prof) {
        unsigned long long hp[16];
        DS_HIP_CHECK(hipMemcpyAsync(hp, P.prof, sizeof(hp), hipMemcpyDeviceToHost, st));
        DS_HIP_CHECK(hipStreamSynchronize(st));
        const double nb = hp[7] ? (double)hp[7] : 1.0;
        fprintf(stderr, "pl prof (cycles per workgroup, wave 0): flush %.0f  P01 %.0f  barA %.0f  P2 %.0f  barB %.0f  P3 %.0f  barC %.0f  (%llu workgroups)\n",
                hp[0] / nb, hp[1] / nb, hp[2] / nb, hp[3] / nb, hp[4] / nb, hp[5] / nb, hp[6] / nb, hp[7]);
        fprintf(stderr, "pl prof P01 split: setup %.0f  loads %.0f  normalise+store %.0f  points+scatter(2 eyes, +slow loop of eye 0) %.0f  rest %.0f\n", hp[8] / nb, hp[9] / nb, hp[10] / nb, hp[11] / nb, hp[1] / nb);
    }
    ctx->last_exact_rows_valid = nrows;
    return DS_OK;
}
