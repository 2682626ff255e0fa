// apply_stereo_divergence_polylines (reference: src/stereoimage_generation.py:162-283) on gfx950.
//
// The reference sweeps every image row sequentially: it morphs the row into a polyline of points
// (x = col + 0.5 + d + sep [+-0.45], closeness = |d|, colour index = col), sorts the points, and for
// every output pixel walks the sub-intervals between consecutive sorted points, keeping an "active
// set" of segments and taking the colour of the closest one.
//
// Here every OUTPUT PIXEL is evaluated independently (one lane per pixel).  Points are kept in
// ORIGINAL order in one array pt[k]; segment g runs from pt[g-1] to pt[g].
//
//   * SIMPLE pixels.  Crossings of a vertical line x = c by the polyline (a continuous path from the
//     -w sentinel to the 2w sentinel) alternate forward, backward, forward, ...  If no backward (or
//     zero-length) segment overlaps the strip [col, col+1] the path is monotone inside the strip:
//     the segments overlapping it are a contiguous run g0..g1 of forward segments, the sub-intervals
//     of the pixel are exactly their pieces [max(x0,col), min(x1,col+1)] in that order, and each
//     sub-interval has exactly one active segment, which the reference takes without looking at its
//     closeness (csg_end == 1, :259).  The lane just walks g0..g1: no sort, no search.
//     Which pixels are simple is established by binning (LDS atomics: min g, max g, forward/backward
//     counts per pixel) and is self-checking: simple <=> no backward overlap and #forward == g1-g0+1.
//   * GENERAL pixels (folds: occlusion boundaries, noisy depth) enumerate their breakpoints
//     (points with floor(x) == col among pt[g0-1..g1]) and test every forward segment in g0..g1
//     with the reference's rule: active <=> x0 < c and not (x1 < c), which is the reference's active
//     set as long as the centres c are non-decreasing along the row, i.e. every sub-interval has
//     positive length.  The winner rule (strict '>' on interpolated closeness, 0 < ip < 1) is order
//     independent unless two valid candidates tie exactly or none is valid.
// A pixel that hits a history-dependent situation (non-positive significance, empty active set,
// exact tie, no valid candidate among >= 2) raises a flag for its ROW, and flagged rows are
// re-rendered by k_polylines_exact: a statement-by-statement sequential transliteration of the
// reference (one lane per row).  The result is therefore bit-identical to the reference for every
// input; the fallback only costs time.  All arithmetic is IEEE binary64, compiled with
// -ffp-contract=off, in the reference's operation order.
//
// Mapping to the machine: one WAVE (64-thread workgroup) renders 64 consecutive pixels of one row for
// BOTH eyes; LDS scratch is private to the wave and there is no cross-wave barrier.  Workgroups are
// persistent and walk (image, row, tile) work items with a grid stride.  Per work item the wave stages
// what the eyes share (normalised depth: one float64 division per source column; the pixel bytes) with
// coalesced loads -- |shift| <= |divergence_px| bounds the source window to tile + ~|divergence_px| + 7
// columns -- then per eye computes the point coordinates, bins, renders out of LDS and writes the
// tile's bytes with dword stores.
#include <stdlib.h>

#include "ds_common.h"

#define PL_TILE 64
#define PL_EPS 1e-7
#define PL_QCHUNK 128      // queue chunk: entry 0 is the header {count}, entries 1..count are pixels

struct PolyParams {
    const uint8_t *img;
    const void *depth;
    const double *minmax;      // n * {min,max}
    const double *lut;         // optional n*65536 table of norm**exponent (uint16 depth only)
    int n, h, w, c;
    int n_eyes;
    double div_px[2], sep_px[2];
    uint8_t *out[2];
    int64_t ors[2], ois[2];
    int offL[2], offU[2];      // per-pixel source-column window [col+offL, col+offU]
    int *row_flags;            // one int per (image, eye, row)
    int *row_list;             // flagged rows, compacted
    int *counters;             // [0] = number of flagged rows, [1] = number of queued (general) pixels
    int4 *queue;               // general pixels in chunks of PL_QCHUNK: {rowid, col, g0, g1}, g absolute (col*NP + side)
    int queue_chunks;          // capacity in chunks
    int dbg;                   // DS_PL_DEBUG ablation knob (0 = off); results are WRONG when set
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

// ---- wave-private LDS view ------------------------------------------------------------------------
// Window columns are indexed i = j - ju0.  Points: k = i*NP + side (NP = 2 sharp: xl, xr; 1 soft), plus the
// sentinels k = -1 (x = -w) when the window starts at column 0 and k = kend (x = 2w) when it ends at w-1.
template <int SHARP> struct PlView {
    static constexpr int NP = SHARP ? 2 : 1;
    const double *ptp;         // ptp[k], k >= -1
    const double *ad;          // |coord_d| per window column
    const uint8_t *src;        // c bytes per window column
    int ju0, c, ncol;          // ncol = number of window columns staged (union window of the two eyes)
    int kfirst, klast;         // real points of this eye: kfirst..klast
    int ghead, gtail;          // sentinel segment ids (or -1000000 when absent)

    __device__ __forceinline__ double pt(int k) const { return ptp[k]; }
    __device__ __forceinline__ int col_of(int k) const { const int kk = min(max(k, kfirst), klast); return SHARP ? (kk >> 1) : kk; }
    __device__ __forceinline__ double dd(int k) const { return (k < kfirst || k > klast) ? 0.0 : ad[SHARP ? (k >> 1) : k]; }
    __device__ __forceinline__ double pix(int colw, int ch) const { return (double)src[(size_t)colw * c + ch]; }
};

// The same interface over GLOBAL memory for one (image, eye, row): point k is an absolute index
// (k = col*NP + side, -1 and NP*w are the sentinels); coordinates are recomputed on the fly.
template <int DT, int SHARP> struct PlRowView {
    static constexpr int NP = SHARP ? 2 : 1;
    const typename ds_depth_traits<DT>::T *depth_row;
    const uint8_t *src_row;
    const double *lut;         // already offset to the image, or null
    double mn, mx, div_px, sep_px;
    int w, c;
    int kfirst, klast;         // 0 .. NP*w-1
    int ghead, gtail;          // 0 and NP*w

    __device__ __forceinline__ double coord_d(int col) const {
        const typename ds_depth_traits<DT>::T v = depth_row[col];
        double nd;
        if (DT == DS_DEPTH_U16 && lut != nullptr) nd = lut[(unsigned)v];
        else nd = ds_depth_traits<DT>::norm(v, mn, mx);
        return nd * div_px;                                                          // :182
    }
    __device__ __forceinline__ double pt(int k) const {
        if (k < 0) return -1.0 * (double)w;                                          // :179
        if (k > klast) return 2.0 * (double)w;                                       // :191
        const int col = SHARP ? (k >> 1) : k;
        const double coord_x = (double)col + 0.5 + coord_d(col) + sep_px;            // :183
        if (!SHARP) return coord_x;
        return (k & 1) ? coord_x + 0.45 : coord_x - 0.45;                            // :188-189
    }
    __device__ __forceinline__ int col_of(int k) const { const int kk = min(max(k, kfirst), klast); return SHARP ? (kk >> 1) : kk; }
    __device__ __forceinline__ double dd(int k) const { return (k < kfirst || k > klast) ? 0.0 : fabs(coord_d(SHARP ? (k >> 1) : k)); }
    __device__ __forceinline__ double pix(int col, int ch) const { return (double)src_row[(size_t)col * c + ch]; }
};

// colour of one sub-interval once its segment is known (stereoimage_generation.py:270-279)
template <class View>
__device__ __forceinline__ void pl_add_flat(const View &V, int colw, double significance, double *color)
{
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (k < V.c) color[k] += V.pix(colw, k) * significance;                      // :273
}

template <class View>
__device__ __forceinline__ void pl_add_lerp(const View &V, int cl, int cr, double x0, double x1, double coord_center,
                                            double significance, double *color)
{
    const double ip_k = (coord_center - x0) / (x1 - x0);                             // :276
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (k < V.c) {
            const double u = V.pix(cl, k) * (1.0 - ip_k);
            const double v = V.pix(cr, k) * ip_k;
            color[k] += (u + v) * significance;                                      // :277-279
        }
}

// SIMPLE pixel: one piece of segment g inside pixel [fq, fq1]
template <bool FLAT, class View>
__device__ __forceinline__ void pl_piece(const View &V, int g, double fq, double fq1, double *color, int &flag)
{
    const double x0 = V.pt(g - 1), x1 = V.pt(g);
    const double a = x0 > fq ? x0 : fq;                                              // max(col, pt[pt_i][0])      :235
    const double b = x1 < fq1 ? x1 : fq1;                                            // min(col + 1, pt[pt_i+1][0]) :236
    const double coord_from = a + PL_EPS;
    const double coord_to = b - PL_EPS;
    const double significance = coord_to - coord_from;                               // :237
    const double coord_center = coord_from + 0.5 * significance;                     // :239
    if (!(significance > 0.0)) flag = 1;
    if (FLAT) pl_add_flat(V, V.col_of(g), significance, color);
    else pl_add_lerp(V, V.col_of(g - 1), V.col_of(g), x0, x1, coord_center, significance, color);
}

// GENERAL pixel: one sub-interval [a, b]; candidates are the forward segments among g0..g1 (:235-279)
template <class View>
__device__ __forceinline__ void pl_subinterval(const View &V, int g0, int g1, double a, double b, double *color, int &flag)
{
    const double coord_from = a + PL_EPS;                                            // :235
    const double coord_to = b - PL_EPS;                                              // :236
    const double significance = coord_to - coord_from;                               // :237
    const double coord_center = coord_from + 0.5 * significance;                     // :239
    if (!(significance > 0.0)) flag = 1;     // centres may stop being monotone: history dependent

    // active set = { g : x0 < c and not (x1 < c) }  (:242-253)
    int count = 0, win = -1;
    for (int g = g0; g <= g1; g++) {
        const double x0 = V.pt(g - 1), x1 = V.pt(g);
        if (x0 < coord_center && !(x1 < coord_center)) { if (count == 0) win = g; count++; }
    }
    if (count == 0) { flag = 1; return; }    // reference reads a stale csg[0]
    if (count != 1) {                                                                // :259
        double best = -PL_EPS;                                                       // :261
        bool have = false;
        win = -1;
        for (int g = g0; g <= g1; g++) {
            const double x0 = V.pt(g - 1), x1 = V.pt(g);
            if (x0 < coord_center && !(x1 < coord_center)) {
                const double ip_k = (coord_center - x0) / (x1 - x0);                 // :263
                const double closeness = (1.0 - ip_k) * V.dd(g - 1) + ip_k * V.dd(g);   // :265
                const bool valid = 0.0 < ip_k && ip_k < 1.0;
                if (valid && have && closeness == best) flag = 1;                    // exact tie: csg order decides
                if (best < closeness && valid) { best = closeness; win = g; have = true; }   // :266
            }
        }
        if (!have) { flag = 1; return; }     // reference falls back to csg[0]
    }
    const int cl = V.col_of(win - 1), cr = V.col_of(win);
    if (cl == cr) pl_add_flat(V, cl, significance, color);                           // :272
    else pl_add_lerp(V, cl, cr, V.pt(win - 1), V.pt(win), coord_center, significance, color);
}

template <class View>
__device__ __forceinline__ void pl_render_general(const View &V, int g0, int g1, double fq, double fq1, double *color, int &flag)
{
    // breakpoints: points k in g0-1..g1 with floor(x) == col, visited in (x, k) order
    double a = fq, last_x = 0.0;
    int last_k = -1000000;
    bool first = true;
    for (;;) {
        double b = fq1;
        int bk = -1000000;
        for (int k = max(g0 - 1, V.kfirst); k <= min(g1, V.klast); k++) {
            const double x = V.pt(k);
            if (!(x < fq) && x < fq1) {
                const bool after = first || x > last_x || (x == last_x && k > last_k);
                if (after && (bk == -1000000 || x < b)) { b = x; bk = k; }
            }
        }
        const bool more = bk != -1000000;
        if (!more) b = fq1;
        pl_subinterval(V, g0, g1, a, b, color, flag);
        if (!more) break;
        a = b; last_x = b; last_k = bk; first = false;
    }
}

// SIMPLE pixel, SHARP, at most 3 column pairs (6 segments): everything the lane needs is loaded up front
// (7 point coordinates, 4 columns of pixel bytes) so the LDS latency is paid once, then the pieces are evaluated
// with predication instead of loops.  Slot layout: pair j (j = 0,1,2) = { even g = ge0 + 2j : incoming (lerp),
// odd g + 1 : body (flat) }.
struct PlPx { double v[4]; };

template <int SHARP>
__device__ __forceinline__ void pl_render_simple_sharp(const PlView<SHARP> &V, int g0, int g1, double fq, double fq1, bool third,
                                                       double *color, int &flag)
{
    const int ge0 = g0 & ~1;
    const int ib = ge0 >> 1;                         // window column of pair 0
    const int c = V.c;
    const int ncm1 = V.ncol - 1;
    // points ge0-1 .. ge0+5, indices clamped into the staged range (clamped values are never used)
    const int klo = V.ghead == 0 ? -1 : V.kfirst, khi = V.gtail >= 0 ? V.gtail : V.klast;
    double p[7];
#pragma unroll
    for (int t = 0; t < 7; t++) {
        if (t < 5 || third) p[t] = V.ptp[min(max(ge0 - 1 + t, klo), khi)];
        else p[t] = 0.0;
    }
    // pixel bytes of columns ib-1 .. ib+2
    PlPx px[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (j < 3 || third) {
            const uint8_t *q = V.src + (size_t)min(max(ib - 1 + j, 0), ncm1) * c;
#pragma unroll
            for (int k = 0; k < 4; k++) px[j].v[k] = (k < c) ? (double)q[k] : 0.0;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) px[j].v[k] = 0.0;
        }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (j == 2 && !third) break;
        const int ge = ge0 + 2 * j;
        // ---- even slot: incoming segment (pt[ge-1] -> pt[ge]), colours of columns ib+j-1 and ib+j
        if (ge >= g0 && ge <= g1) {
            const double x0 = p[2 * j], x1 = p[2 * j + 1];
            const double a = x0 > fq ? x0 : fq;                                              // :235
            const double b = x1 < fq1 ? x1 : fq1;                                            // :236
            const double coord_from = a + PL_EPS;
            const double coord_to = b - PL_EPS;
            const double significance = coord_to - coord_from;                               // :237
            const double coord_center = coord_from + 0.5 * significance;                     // :239
            if (!(significance > 0.0)) flag = 1;
            if (ge == V.ghead || ge == V.gtail) {
                // sentinel segment: both colour indices are the edge column (:179,:191) -> flat
                const int cc = V.col_of(ge);
                const uint8_t *q = V.src + (size_t)cc * c;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (k < c) color[k] += (double)q[k] * significance;                      // :273
            } else {
                const double ip_k = (coord_center - x0) / (x1 - x0);                         // :276
                const double om = 1.0 - ip_k;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (k < c) {
                        const double u = px[j].v[k] * om;
                        const double v = px[j + 1].v[k] * ip_k;
                        color[k] += (u + v) * significance;                                  // :277-279
                    }
            }
        }
        // ---- odd slot: body segment (pt[ge] -> pt[ge+1]), colour of column ib+j
        const int go = ge + 1;
        if (go >= g0 && go <= g1) {
            const double x0 = p[2 * j + 1], x1 = p[2 * j + 2];
            const double a = x0 > fq ? x0 : fq;
            const double b = x1 < fq1 ? x1 : fq1;
            const double coord_from = a + PL_EPS;
            const double coord_to = b - PL_EPS;
            const double significance = coord_to - coord_from;
            if (!(significance > 0.0)) flag = 1;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k < c) color[k] += px[j + 1].v[k] * significance;                        // :273
        }
    }
}

__device__ __forceinline__ void pl_store_tile(const uint8_t *s_out, uint8_t *dst, int nbytes, int tid)
{
    if ((((uintptr_t)dst) & 3) == 0) {
        const int nw4 = nbytes >> 2;
        const uint32_t *s4 = reinterpret_cast<const uint32_t *>(s_out);
        uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
        for (int i = tid; i < nw4; i += PL_TILE) d4[i] = s4[i];
        for (int i = (nw4 << 2) + tid; i < nbytes; i += PL_TILE) dst[i] = s_out[i];
    } else {
        for (int i = tid; i < nbytes; i += PL_TILE) dst[i] = s_out[i];
    }
}

__device__ __forceinline__ void pl_flag_row(const PolyParams &P, int img, int eye, int row)
{
    const int rowid = (img * P.n_eyes + eye) * P.h + row;
    if (atomicExch(&P.row_flags[rowid], 1) == 0) {
        const int slot = atomicAdd(&P.counters[0], 1);
        P.row_list[slot] = rowid;
    }
}

static size_t pl_lds_bytes(int ncu_max, int c)
{
    // nd, ad (ncu_max doubles each), pt (2*ncu_max+2 doubles), gmin/gmax/cnt (PL_TILE ints each), out (PL_TILE*4), src
    return (size_t)(4 * ncu_max + 2) * sizeof(double) + 3 * PL_TILE * sizeof(int) + PL_TILE * 4 + (size_t)ncu_max * c + 16;
}

template <int DT, int SHARP>
__global__ __launch_bounds__(PL_TILE) void k_polylines(PolyParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NP = SHARP ? 2 : 1;
    const int tid = threadIdx.x;
    const int w = P.w, c = P.c;
    const int tiles_x = (w + PL_TILE - 1) / PL_TILE;
    const long long nwork = (long long)P.n * P.h * tiles_x;

    int uL = P.offL[0], uU = P.offU[0];                    // union of the two eyes' window offsets
    if (P.n_eyes > 1) { uL = min(uL, P.offL[1]); uU = max(uU, P.offU[1]); }
    const int ncu_max = PL_TILE + (uU - uL + 1) + 2;
    double *s_nd = reinterpret_cast<double *>(smem);
    double *s_ad = s_nd + ncu_max;
    double *s_ptb = s_ad + ncu_max;                         // pt[k] = s_ptb[k + 1]
    int *s_gmin = reinterpret_cast<int *>(s_ptb + 2 * ncu_max + 2);
    int *s_gmax = s_gmin + PL_TILE;
    int *s_cnt = s_gmax + PL_TILE;                          // low 16 bits: forward segments, high 16: backward/zero-length
    uint8_t *s_out = reinterpret_cast<uint8_t *>(s_cnt + PL_TILE);
    uint8_t *s_src = s_out + PL_TILE * 4;
    double *s_pt = s_ptb + 1;

    typedef typename ds_depth_traits<DT>::T DTy;
    // this wave's open queue chunk (one global atomic per PL_QCHUNK-1 queued pixels instead of one per pixel)
    int chunk = -1, chunk_used = 0;
    for (long long work = blockIdx.x; work < nwork; work += gridDim.x) {
        const int tx = (int)(work % tiles_x);
        const long long rr = work / tiles_x;
        const int row = (int)(rr % P.h), img = (int)(rr / P.h);
        const int c0 = tx * PL_TILE;
        const int c1 = min(c0 + PL_TILE, w);               // exclusive
        const int tn = c1 - c0;
        const double mn = P.minmax[img * 2], mx = P.minmax[img * 2 + 1];
        const uint8_t *src_row = P.img + ((size_t)img * P.h + row) * (size_t)w * c;
        const DTy *depth_row = (const DTy *)P.depth + ((size_t)img * P.h + row) * (size_t)w;

        // 0/0: constant depth gives NaN for every point (stereoimage_generation.py:81); the sweep then
        // only ever sees the segment from the -w sentinel, whose colour index is 0 on both ends.
        if (!(mx > mn)) {
            const int col = c0 + tid;
            if (col < w) {
                const double coord_from = (double)col + PL_EPS;
                const double coord_to = (double)(col + 1) - PL_EPS;
                const double significance = coord_to - coord_from;
                for (int eye = 0; eye < P.n_eyes; eye++) {
                    uint8_t *out_row = P.out[eye] + (int64_t)img * P.ois[eye] + (int64_t)row * P.ors[eye];
                    for (int k = 0; k < c; k++) {
                        double color = 0.5;
                        color += (double)src_row[k] * significance;
                        out_row[(size_t)col * c + k] = ds_f64_to_u8(color);
                    }
                }
            }
            continue;
        }

        // ---- stage what both eyes share --------------------------------------------------------------
        const int ju0 = max(0, min(c0 + uL - 1, w - 2));
        const int ju1 = max(ju0, max(0, min(c1 - 1 + uU, w - 1)));
        const int ncu = ju1 - ju0 + 1;
        __syncthreads();                                    // previous work item is done with the LDS
        for (int i = tid; i < ncu; i += PL_TILE) {
            const DTy v = depth_row[ju0 + i];
            double nd;
            if (DT == DS_DEPTH_U16 && P.lut != nullptr) nd = P.lut[(size_t)img * 65536 + (unsigned)v];   // norm ** exponent
            else nd = ds_depth_traits<DT>::norm(v, mn, mx);                                            // pow(x, 1.0) == x
            s_nd[i] = nd;
        }
        for (int i = tid; i < ncu * c; i += PL_TILE) s_src[i] = src_row[(size_t)ju0 * c + i];

        for (int eye = 0; eye < P.n_eyes; eye++) {
            const double div_px = P.div_px[eye], sep_px = P.sep_px[eye];
            const int offL = P.offL[eye], offU = P.offU[eye];
            // this eye's source window (inclusive), always at least one real column
            const int jt0 = max(0, min(c0 + offL - 1, w - 2));
            const int jt1 = max(jt0, max(0, min(c1 - 1 + offU, w - 1)));
            const int i0 = jt0 - ju0, i1 = jt1 - ju0;
            const bool head = jt0 == 0, tail = jt1 == w - 1;

            __syncthreads();                                // nd ready / previous eye done with pt, bins, out
            s_gmin[tid] = 0x7fffffff; s_gmax[tid] = -0x7fffffff; s_cnt[tid] = 0;
            for (int i = i0 + tid; i <= i1; i += PL_TILE) {
                const double coord_d = s_nd[i] * div_px;                                       // :182
                const double coord_x = (double)(ju0 + i) + 0.5 + coord_d + sep_px;             // :183
                if (SHARP) { s_pt[2 * i] = coord_x - 0.45; s_pt[2 * i + 1] = coord_x + 0.45; }   // :188-189
                else s_pt[i] = coord_x;                                                        // :185
                s_ad[i] = fabs(coord_d);
            }
            if (tid == 0) {
                if (head) s_pt[-1] = -1.0 * (double)w;                                         // :179
                if (tail) s_pt[NP * (i1 + 1)] = 2.0 * (double)w;                               // :191
            }
            __syncthreads();
            if (P.dbg == 1) continue;

            PlView<SHARP> V;
            V.ptp = s_pt; V.ad = s_ad; V.src = s_src; V.ju0 = ju0; V.c = c; V.ncol = ncu;
            V.kfirst = NP * i0; V.klast = NP * i1 + NP - 1;
            V.ghead = head ? 0 : -1000000; V.gtail = tail ? NP * (i1 + 1) : -1000000;

            // ---- bin segments by the output pixels they overlap ----------------------------------------
            // segment g = (pt[g-1], pt[g]); real segments g = kfirst+1 .. klast, plus the sentinel segments
            const int gfirst = head ? 0 : V.kfirst + 1, glast = tail ? V.klast + 1 : V.klast;
            for (int g = gfirst + tid; g <= glast; g += PL_TILE) {
                const double x0 = s_pt[g - 1], x1 = s_pt[g];
                const bool fwd = x0 < x1;
                const double lo = fwd ? x0 : x1, hi = fwd ? x1 : x0;
                const double qa = fmax(floor(lo), (double)c0), qb = fmin(floor(hi), (double)(c1 - 1));
                const int code = fwd ? 1 : (1 << 16);
                for (double q = qa; q <= qb; q += 1.0) {
                    const int p = (int)q - c0;
                    atomicMin(&s_gmin[p], g);
                    atomicMax(&s_gmax[p], g);
                    atomicAdd(&s_cnt[p], code);
                }
            }
            __syncthreads();
            if (P.dbg == 2) continue;

            // ---- render ------------------------------------------------------------------------------
            // SIMPLE pixels are rendered here; pixels under a fold are queued for k_polylines_general, which
            // writes their bytes itself (one lane per queued pixel, so folds do not stall this wave).
            const int col = c0 + tid;
            int flag = 0;
            bool queued = false;
            int4 qe = make_int4(0, 0, 0, 0);
            if (col < w) {
                const double fq = (double)col, fq1 = (double)(col + 1);
                double color[4] = { 0.5, 0.5, 0.5, 0.5 };                                       // :229
                const int g0 = s_gmin[tid], g1 = s_gmax[tid], cnt = s_cnt[tid];
                const bool simple = (cnt >> 16) == 0 && (cnt & 0xffff) == g1 - g0 + 1 && g1 >= g0;
                const bool fast = SHARP && simple && g1 - (g0 & ~1) < 6;
                const bool third = __any(fast && g1 - (g0 & ~1) >= 4) != 0;   // wave-uniform: does any lane need pair 2
                if (P.dbg == 3) { color[0] += s_pt[NP * i0 + (tid & 7)]; }
                else if (fast) {
                    pl_render_simple_sharp<SHARP>(V, g0, g1, fq, fq1, third, color, flag);
                } else if (simple && !SHARP) {
                    for (int g = g0; g <= g1; g++) {
                        if (g == V.ghead || g == V.gtail) pl_piece<true>(V, g, fq, fq1, color, flag);
                        else pl_piece<false>(V, g, fq, fq1, color, flag);
                    }
                } else if (g1 >= g0) {
                    queued = true;
                    qe = make_int4((img * P.n_eyes + eye) * P.h + row, col, NP * ju0 + g0, NP * ju0 + g1);
                } else {
                    flag = 1;        // nothing overlaps this pixel: the reference reads a stale csg[0]
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (k < c) s_out[tid * c + k] = ds_f64_to_u8(color[k]);                      // :281
            }
            {   // append this tile's general pixels to the wave's queue chunk
                const unsigned long long qmask = __ballot(queued);
                if (qmask != 0ull) {
                    const int nq = __popcll(qmask);
                    if (chunk < 0 || chunk_used + nq > PL_QCHUNK - 1) {
                        int fresh = 0;
                        if (tid == 0) {
                            if (chunk >= 0) P.queue[(size_t)chunk * PL_QCHUNK] = make_int4(chunk_used, 0, 0, 0);
                            fresh = atomicAdd(&P.counters[1], 1);
                        }
                        chunk = __shfl(fresh, 0, 64);
                        chunk_used = 0;
                    }
                    if (queued) {
                        const int rank = __popcll(qmask & ((1ull << tid) - 1ull));
                        P.queue[(size_t)chunk * PL_QCHUNK + 1 + chunk_used + rank] = qe;
                    }
                    chunk_used += nq;
                }
            }
            const int any = __syncthreads_or(flag);
            uint8_t *out_row = P.out[eye] + (int64_t)img * P.ois[eye] + (int64_t)row * P.ors[eye];
            // queued pixels get placeholder bytes here; k_polylines_general (next on the stream) overwrites them
            pl_store_tile(s_out, out_row + (size_t)c0 * c, tn * c, tid);
            if (any && tid == 0) pl_flag_row(P, img, eye, row);
        }
    }
    if (chunk >= 0 && tid == 0) P.queue[(size_t)chunk * PL_QCHUNK] = make_int4(chunk_used, 0, 0, 0);
}

// ---- general pixels: one LANE per queued pixel, coordinates recomputed on the fly from global memory -----
template <int DT, int SHARP>
__global__ __launch_bounds__(64) void k_polylines_general(PolyParams P)
{
    constexpr int NP = SHARP ? 2 : 1;
    const int nchunks = P.counters[1];
    const int w = P.w, c = P.c;
    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x)
    for (int it = 1 + threadIdx.x, cnt = P.queue[(size_t)ch * PL_QCHUNK].x; it <= cnt; it += 64) {
        const int4 e = P.queue[(size_t)ch * PL_QCHUNK + it];
        const int rowid = e.x, col = e.y, g0 = e.z, g1 = e.w;
        const int row = rowid % P.h;
        const int ie = rowid / P.h;
        const int eye = ie % P.n_eyes, img = ie / P.n_eyes;
        PlRowView<DT, SHARP> V;
        V.depth_row = (const typename ds_depth_traits<DT>::T *)P.depth + ((size_t)img * P.h + row) * (size_t)w;
        V.src_row = P.img + ((size_t)img * P.h + row) * (size_t)w * c;
        V.lut = P.lut ? P.lut + (size_t)img * 65536 : nullptr;
        V.mn = P.minmax[img * 2]; V.mx = P.minmax[img * 2 + 1];
        V.div_px = P.div_px[eye]; V.sep_px = P.sep_px[eye];
        V.w = w; V.c = c; V.kfirst = 0; V.klast = NP * w - 1; V.ghead = 0; V.gtail = NP * w;
        double color[4] = { 0.5, 0.5, 0.5, 0.5 };
        int flag = 0;
        pl_render_general(V, g0, g1, (double)col, (double)(col + 1), color, flag);
        uint8_t *dst = P.out[eye] + (int64_t)img * P.ois[eye] + (int64_t)row * P.ors[eye] + (size_t)col * c;
        for (int k = 0; k < 4; k++)
            if (k < c) dst[k] = ds_f64_to_u8(color[k]);
        if (flag) pl_flag_row(P, img, eye, row);
    }
}


// ------------------------------------------------------------------------------------------------
// Exact fallback: the reference's row sweep, statement by statement, one lane per flagged row.
// Scratch (per worker, interleaved across workers so lock-step lanes coalesce):
//   OX[np] OD[np]  points in original order        (x, |d|)
//   SX[np] SK[np]  points in sorted order          (x, original index); segment k travels with point k
//   CSG[np]        active set as original segment indices
struct ExactScratch { double *ox, *od, *sx; int *sk, *csg; int nworkers; int np_max; };

template <int DT, int SHARP>
__global__ __launch_bounds__(64) void k_polylines_exact(PolyParams P, ExactScratch S)
{
    const int worker = blockIdx.x * 64 + threadIdx.x;
    const int count = P.counters[0];
    const int w = P.w, c = P.c;
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


template <int DT>
static int launch_dt(ds_ctx *ctx, const PolyParams &P, int sharp, dim3 grid, size_t lds, const ExactScratch &S, int exact_blocks, hipStream_t st)
{
    if (ctx->profile) (void)hipEventRecord(ctx->ev[0], st);
    if (sharp) hipLaunchKernelGGL((k_polylines<DT, 1>), grid, dim3(PL_TILE), lds, st, P);
    else hipLaunchKernelGGL((k_polylines<DT, 0>), grid, dim3(PL_TILE), lds, st, P);
    if (ctx->profile) { (void)hipEventRecord(ctx->ev[1], st); (void)hipEventRecord(ctx->ev[2], st); }
    if (sharp) hipLaunchKernelGGL((k_polylines_general<DT, 1>), dim3(256 * 64), dim3(64), 0, st, P);
    else hipLaunchKernelGGL((k_polylines_general<DT, 0>), dim3(256 * 64), dim3(64), 0, st, P);
    if (sharp) hipLaunchKernelGGL((k_polylines_exact<DT, 1>), dim3(exact_blocks), dim3(64), 0, st, P, S);
    else hipLaunchKernelGGL((k_polylines_exact<DT, 0>), dim3(exact_blocks), dim3(64), 0, st, P, S);
    if (ctx->profile) { (void)hipEventRecord(ctx->ev[3], st); ctx->ev_recorded = 1; }
    return DS_OK;
}

// called from ds_stereo_warp (ds_stereo.hip)
int ds_polylines_launch(ds_ctx *ctx, const uint8_t *image, const void *depth, int depth_dtype, const double *minmax,
                        const double *lut, int n, int h, int w, int c, int sharp, const ds_eye *eyes, int n_eyes,
                        hipStream_t st)
{
    PolyParams P;
    memset(&P, 0, sizeof(P));
    P.img = image; P.depth = depth; P.minmax = minmax; P.lut = lut;
    P.n = n; P.h = h; P.w = w; P.c = c; P.n_eyes = n_eyes;
    { const char *e = getenv("DS_PL_DEBUG"); P.dbg = e ? atoi(e) : 0; }
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
    const int ncu_max = PL_TILE + (uU - uL + 1) + 2;
    const size_t lds = pl_lds_bytes(ncu_max, c);
    DS_REQUIRE(lds <= 64 * 1024, DS_EUNSUPPORTED,
               "ds_stereo_warp: divergence/separation window of %d columns does not fit the LDS budget", uU - uL + 1);

    const int64_t nrows = (int64_t)n * n_eyes * h;
    DS_REQUIRE(nrows < (1ll << 30), DS_EUNSUPPORTED, "ds_stereo_warp: too many rows in one call");
    int rc = ds_ctx_reserve(ctx, &ctx->row_flags, &ctx->row_flags_bytes, (size_t)(nrows + 16) * sizeof(int));
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
    ExactScratch S;
    S.nworkers = nworkers; S.np_max = np_max;
    S.ox = (double *)ctx->exact_ws;
    S.od = S.ox + (size_t)np_max * nworkers;
    S.sx = S.od + (size_t)np_max * nworkers;
    S.sk = (int *)(S.sx + (size_t)np_max * nworkers);
    S.csg = S.sk + (size_t)np_max * nworkers;

    // row_flags[0..nrows) flags, then 16 ints of counters
    P.row_flags = (int *)ctx->row_flags;
    P.counters = P.row_flags + nrows;
    P.row_list = (int *)ctx->row_list;
    DS_HIP_CHECK(hipMemsetAsync(ctx->row_flags, 0, (size_t)(nrows + 16) * sizeof(int), st));

    // queue of general pixels: worst case every pixel of every eye
    // persistent grid of single-wave workgroups
    const int tiles_x = (w + PL_TILE - 1) / PL_TILE;
    const long long nwork = (long long)n * h * tiles_x;
    long long nblocks = 256 * 128;
    { const char *e = getenv("DS_PL_BLOCKS"); if (e && atoi(e) > 0) nblocks = atoi(e); }
    if (nblocks > nwork) nblocks = nwork;
    // queue of general pixels: worst case every pixel of every eye, plus one partly filled chunk per wave and the
    // slack of closing a chunk early (at most one tile's worth per chunk)
    const long long qchunks = ((long long)nrows * w) / (PL_QCHUNK - 1 - PL_TILE) + nblocks + 2;
    rc = ds_ctx_reserve(ctx, &ctx->tmp_b, &ctx->tmp_b_bytes, (size_t)qchunks * PL_QCHUNK * sizeof(int4));
    if (rc) return rc;
    P.queue = (int4 *)ctx->tmp_b;
    P.queue_chunks = (int)qchunks;
    dim3 grid((unsigned)nblocks, 1, 1);

    switch (depth_dtype) {
    case DS_DEPTH_U16: rc = launch_dt<DS_DEPTH_U16>(ctx, P, sharp, grid, lds, S, nworkers / 64, st); break;
    case DS_DEPTH_F32: rc = launch_dt<DS_DEPTH_F32>(ctx, P, sharp, grid, lds, S, nworkers / 64, st); break;
    case DS_DEPTH_F64: rc = launch_dt<DS_DEPTH_F64>(ctx, P, sharp, grid, lds, S, nworkers / 64, st); break;
    default: ds_set_error("unknown depth dtype %d", depth_dtype); return DS_EINVAL;
    }
    DS_HIP_CHECK(hipGetLastError());
    ctx->last_exact_rows_valid = nrows;
    return DS_OK;
}
