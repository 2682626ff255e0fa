// apply_stereo_divergence_polylines (reference: src/stereoimage_generation.py:162-283) on gfx950.
//
// The reference sweeps every image row sequentially: it morphs the row into a polyline of points
// (x = col + 0.5 + d + sep [+-0.45], closeness = |d|, colour index = col), sorts the points, and for
// every output pixel walks the sub-intervals between consecutive sorted points, keeping an "active
// set" of segments and taking the colour of the closest one.
//
// Here every OUTPUT PIXEL is evaluated independently (one lane per pixel):
//   * its breakpoints are the points with col <= x < col+1, found by scanning the bounded window of
//     source columns that can reach this pixel (|shift| <= |divergence_px|, so the window is
//     ~|divergence_px|+7 columns) -- no row sort is needed;
//   * the active set of the reference at a sub-interval centre c is exactly
//         { segment k : x0_k < c and not (x1_k < c) }
//     as long as the centres are non-decreasing along the row, which holds whenever every
//     sub-interval has positive length (significance > 0);
//   * the winner rule (strict '>' on the interpolated closeness, 0 < ip < 1) is order independent
//     unless two valid candidates tie exactly or no candidate is valid.
// A pixel that hits one of those history-dependent situations (non-positive significance, empty
// active set, exact tie, no valid candidate among >= 2) raises a flag for its ROW, and flagged rows
// are re-rendered by k_polylines_exact: a statement-by-statement sequential transliteration of the
// reference (one lane per row).  The result is therefore bit-identical to the reference for every
// input; the fallback only costs time.  All arithmetic is IEEE binary64, compiled with
// -ffp-contract=off, in the reference's operation order.
//
// Data movement: a 256-thread workgroup renders 256 consecutive pixels of one row of one eye.  It
// stages the source window of the row (coord_x and |d| as float64, the RGB bytes) in LDS with
// coalesced global loads; all per-pixel scanning then runs out of LDS.
#include "ds_common.h"

#define PL_TILE 256
#define PL_EPS 1e-7

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
    int *counters;             // [0] = number of flagged rows
};

// coord_d of stereoimage_generation.py:182 for one depth element
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

// Segment / point geometry of the morphed polyline, addressed by source column.
//   SHARP: column j owns points  left = cx-0.45, right = cx+0.45  and segments
//          "incoming" (right edge of j-1, or the -w sentinel, -> left edge of j) and "body" (left -> right)
//   SOFT : column j owns one point cx and the incoming segment (point j-1 or sentinel -> point j)
//   the tail segment runs from the last point of column w-1 to the 2w sentinel.
struct PlSeg { double x0, x1, d0, d1; int cl, cr; };

template <int SHARP>
__device__ __forceinline__ PlSeg pl_segment(const double *s_cx, const double *s_ad, int jt0, int w, int j, int body)
{
    PlSeg s;
    if (body) {            // SHARP only
        const double cx = s_cx[j - jt0];
        s.x0 = cx - 0.45; s.x1 = cx + 0.45; s.d0 = s.d1 = s_ad[j - jt0]; s.cl = s.cr = j;
        return s;
    }
    const double cx = s_cx[j - jt0];
    s.x1 = SHARP ? cx - 0.45 : cx; s.d1 = s_ad[j - jt0]; s.cr = j;
    if (j == 0) { s.x0 = -1.0 * (double)w; s.d0 = 0.0; s.cl = 0; }
    else {
        const double px = s_cx[j - 1 - jt0];
        s.x0 = SHARP ? px + 0.45 : px; s.d0 = s_ad[j - 1 - jt0]; s.cl = j - 1;
    }
    return s;
}

template <int SHARP>
__device__ __forceinline__ PlSeg pl_tail(const double *s_cx, const double *s_ad, int jt0, int w)
{
    PlSeg s;
    const double cx = s_cx[w - 1 - jt0];
    s.x0 = SHARP ? cx + 0.45 : cx; s.d0 = s_ad[w - 1 - jt0]; s.cl = w - 1;
    s.x1 = 2.0 * (double)w; s.d1 = 0.0; s.cr = w - 1;
    return s;
}

template <int DT, int SHARP, int NW>
__global__ __launch_bounds__(PL_TILE) void k_polylines(PolyParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NPTS = SHARP ? 2 : 1;
    const int tid = threadIdx.x;
    const int eye = blockIdx.z % P.n_eyes, img = blockIdx.z / P.n_eyes, row = blockIdx.y;
    const int w = P.w, c = P.c;
    const int c0 = blockIdx.x * PL_TILE;
    const int c1 = min(c0 + PL_TILE, w);               // exclusive
    const double div_px = P.div_px[eye], sep_px = P.sep_px[eye];
    const int offL = P.offL[eye], offU = P.offU[eye];

    // source window of this tile (inclusive), always at least one real column
    const int jt0 = max(0, min(c0 + offL - 1, w - 2));
    const int jt1 = max(jt0, max(0, min(c1 - 1 + offU, w - 1)));
    const int ncw = jt1 - jt0 + 1;
    double *s_cx = reinterpret_cast<double *>(smem);
    double *s_ad = s_cx + ncw;
    uint8_t *s_src = reinterpret_cast<uint8_t *>(s_ad + ncw);

    const double mn = P.minmax[img * 2], mx = P.minmax[img * 2 + 1];
    const uint8_t *src_row = P.img + ((size_t)img * P.h + row) * (size_t)w * c;
    typedef typename ds_depth_traits<DT>::T DTy;
    const DTy *depth_row = (const DTy *)P.depth + ((size_t)img * P.h + row) * (size_t)w;
    uint8_t *out_row = P.out[eye] + (int64_t)img * P.ois[eye] + (int64_t)row * P.ors[eye];

    // 0/0: constant depth gives NaN for every point (stereoimage_generation.py:81); the sweep then
    // only ever sees the segment from the -w sentinel, whose colour index is 0 on both ends.
    const bool nan_image = !(mx > mn);
    if (nan_image) {
        const int col = c0 + tid;
        if (col < w) {
            const double coord_from = (double)col + PL_EPS;
            const double coord_to = (double)(col + 1) - PL_EPS;
            const double significance = coord_to - coord_from;
            for (int k = 0; k < c; k++) {
                double color = 0.5;
                color += (double)src_row[k] * significance;
                out_row[(size_t)col * c + k] = ds_f64_to_u8(color);
            }
        }
        return;
    }

    for (int i = tid; i < ncw; i += PL_TILE) {
        const int j = jt0 + i;
        const double coord_d = pl_coord_d<DT>(P, img, depth_row, j, mn, mx, div_px);      // :182
        s_cx[i] = (double)j + 0.5 + coord_d + sep_px;                                      // :183
        s_ad[i] = fabs(coord_d);
    }
    for (int i = tid; i < ncw * c; i += PL_TILE) s_src[i] = src_row[(size_t)jt0 * c + i];
    __syncthreads();

    const int col = c0 + tid;
    int flag = 0;
    if (col < w) {
        const int L = max(0, min(col + offL, w - 1));
        const int U = max(L, max(0, min(col + offU, w - 1)));
        const double fcol = (double)col, fcol1 = (double)(col + 1);
        unsigned long long bp[NW], cd[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) { bp[k] = 0ull; cd[k] = 0ull; }
        double prev;
        if (L == 0) prev = -1.0 * (double)w;
        else prev = SHARP ? s_cx[L - 1 - jt0] + 0.45 : s_cx[L - 1 - jt0];

        // ---- window scan: breakpoints of this pixel and the segments that can touch it ----
#pragma unroll
        for (int wi = 0; wi < NW; wi++) {
            unsigned long long bpw = 0ull, cdw = 0ull;
            const int jj0 = wi * (64 / NPTS);
            const int jjn = min(64 / NPTS, U - L + 1 - jj0);
            for (int q = 0; q < jjn; q++) {
                const double cx = s_cx[L + jj0 + q - jt0];
                if (SHARP) {
                    const double xl = cx - 0.45, xr = cx + 0.45;
                    const unsigned long long b0 = 1ull << (2 * q), b1 = 2ull << (2 * q);
                    if (!(xl < fcol) && xl < fcol1) bpw |= b0;
                    if (!(xr < fcol) && xr < fcol1) bpw |= b1;
                    if (prev < fcol1 && !(xl < fcol)) cdw |= b0;      // incoming segment
                    if (xl < fcol1 && !(xr < fcol)) cdw |= b1;        // body segment
                    prev = xr;
                } else {
                    const unsigned long long b0 = 1ull << q;
                    if (!(cx < fcol) && cx < fcol1) bpw |= b0;
                    if (prev < fcol1 && !(cx < fcol)) cdw |= b0;
                    prev = cx;
                }
            }
            bp[wi] = bpw; cd[wi] = cdw;
        }
        const bool tail = (U == w - 1) && (prev < fcol1);

        double color[4] = { 0.5, 0.5, 0.5, 0.5 };                                           // :229
        double a = fcol;           // max(col, pt[pt_i].x) of the current sub-interval
        bool more = true;
        while (more) {
            // ---- next breakpoint: smallest x among the remaining ones (stable on ties) ----
            double b = fcol1;
            int bw = -1, bb = 0;
#pragma unroll
            for (int wi = 0; wi < NW; wi++) {
                unsigned long long m = bp[wi];
                while (m) {
                    const int t = __builtin_ctzll(m);
                    m &= m - 1;
                    const int jj = wi * (64 / NPTS) + (SHARP ? (t >> 1) : t);
                    const double cx = s_cx[L + jj - jt0];
                    const double x = SHARP ? ((t & 1) ? cx + 0.45 : cx - 0.45) : cx;
                    if (bw < 0 ? true : (x < b)) { b = x; bw = wi; bb = t; }
                }
            }
            if (bw < 0) { b = fcol1; more = false; }
            else {
#pragma unroll
                for (int wi = 0; wi < NW; wi++) if (wi == bw) bp[wi] &= ~(1ull << bb);
            }
            const double coord_from = a + PL_EPS;                                            // :235
            const double coord_to = b - PL_EPS;                                              // :236
            const double significance = coord_to - coord_from;                               // :237
            const double coord_center = coord_from + 0.5 * significance;                     // :239
            if (!(significance > 0.0)) flag = 1;     // centres may stop being monotone: history dependent

            // ---- active set = { k : x0 < c and not (x1 < c) } ----
            int count = 0, first_id = -1;
#pragma unroll
            for (int wi = 0; wi < NW; wi++) {
                unsigned long long m = cd[wi];
                while (m) {
                    const int t = __builtin_ctzll(m);
                    m &= m - 1;
                    const int jj = wi * (64 / NPTS) + (SHARP ? (t >> 1) : t);
                    const PlSeg s = pl_segment<SHARP>(s_cx, s_ad, jt0, w, L + jj, SHARP ? (t & 1) : 0);
                    if (s.x0 < coord_center && !(s.x1 < coord_center)) {
                        if (count == 0) first_id = wi * 64 + t;
                        count++;
                    }
                }
            }
            if (tail) {
                const PlSeg s = pl_tail<SHARP>(s_cx, s_ad, jt0, w);
                if (s.x0 < coord_center && !(s.x1 < coord_center)) {
                    if (count == 0) first_id = 64 * NW;
                    count++;
                }
            }

            int win_id = first_id;
            if (count == 0) flag = 1;                 // reference reads a stale csg[0]
            else if (count != 1) {                                                           // :259
                double best = -PL_EPS;                                                       // :261
                bool have = false;
                win_id = -1;
#pragma unroll
                for (int wi = 0; wi < NW; wi++) {
                    unsigned long long m = cd[wi];
                    while (m) {
                        const int t = __builtin_ctzll(m);
                        m &= m - 1;
                        const int jj = wi * (64 / NPTS) + (SHARP ? (t >> 1) : t);
                        const PlSeg s = pl_segment<SHARP>(s_cx, s_ad, jt0, w, L + jj, SHARP ? (t & 1) : 0);
                        if (s.x0 < coord_center && !(s.x1 < coord_center)) {
                            const double ip_k = (coord_center - s.x0) / (s.x1 - s.x0);       // :263
                            const double closeness = (1.0 - ip_k) * s.d0 + ip_k * s.d1;      // :265
                            const bool valid = 0.0 < ip_k && ip_k < 1.0;
                            if (valid && have && closeness == best) flag = 1;                // exact tie: csg order decides
                            if (best < closeness && valid) { best = closeness; win_id = wi * 64 + t; have = true; }   // :266
                        }
                    }
                }
                if (tail) {
                    const PlSeg s = pl_tail<SHARP>(s_cx, s_ad, jt0, w);
                    if (s.x0 < coord_center && !(s.x1 < coord_center)) {
                        const double ip_k = (coord_center - s.x0) / (s.x1 - s.x0);
                        const double closeness = (1.0 - ip_k) * s.d0 + ip_k * s.d1;
                        const bool valid = 0.0 < ip_k && ip_k < 1.0;
                        if (valid && have && closeness == best) flag = 1;
                        if (best < closeness && valid) { best = closeness; win_id = 64 * NW; have = true; }
                    }
                }
                if (!have) flag = 1;                  // reference falls back to csg[0]
            }

            if (win_id >= 0) {
                PlSeg s;
                if (win_id == 64 * NW) s = pl_tail<SHARP>(s_cx, s_ad, jt0, w);
                else {
                    const int wi = win_id >> 6, t = win_id & 63;
                    const int jj = wi * (64 / NPTS) + (SHARP ? (t >> 1) : t);
                    s = pl_segment<SHARP>(s_cx, s_ad, jt0, w, L + jj, SHARP ? (t & 1) : 0);
                }
                if (s.cl == s.cr) {                                                          // :272
                    const uint8_t *px = s_src + (size_t)(s.cl - jt0) * c;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (k < c) color[k] += (double)px[k] * significance;                 // :273
                } else {
                    const double ip_k = (coord_center - s.x0) / (s.x1 - s.x0);               // :276
                    const uint8_t *pl = s_src + (size_t)(s.cl - jt0) * c;
                    const uint8_t *pr = s_src + (size_t)(s.cr - jt0) * c;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (k < c) {
                            const double u = (double)pl[k] * (1.0 - ip_k);
                            const double v = (double)pr[k] * ip_k;
                            color[k] += (u + v) * significance;                              // :277-279
                        }
                }
            }
            a = b;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k < c) out_row[(size_t)col * c + k] = ds_f64_to_u8(color[k]);                // :281
    }

    const int any = __syncthreads_or(flag);
    if (any && tid == 0) {
        const int rowid = (img * P.n_eyes + eye) * P.h + row;
        if (atomicExch(&P.row_flags[rowid], 1) == 0) {
            const int slot = atomicAdd(&P.counters[0], 1);
            P.row_list[slot] = rowid;
        }
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

// ------------------------------------------------------------------------------------------------
template <int DT, int SHARP>
static int launch_fast(const PolyParams &P, int nw, dim3 grid, size_t lds, hipStream_t st)
{
    switch (nw) {
    case 1: hipLaunchKernelGGL((k_polylines<DT, SHARP, 1>), grid, dim3(PL_TILE), lds, st, P); break;
    case 2: hipLaunchKernelGGL((k_polylines<DT, SHARP, 2>), grid, dim3(PL_TILE), lds, st, P); break;
    case 4: hipLaunchKernelGGL((k_polylines<DT, SHARP, 4>), grid, dim3(PL_TILE), lds, st, P); break;
    default: return DS_EUNSUPPORTED;
    }
    return DS_OK;
}

template <int DT>
static int launch_dt(ds_ctx *ctx, const PolyParams &P, int sharp, int nw, dim3 grid, size_t lds, const ExactScratch &S, int exact_blocks, hipStream_t st)
{
    if (ctx->profile) (void)hipEventRecord(ctx->ev[0], st);
    int rc = sharp ? launch_fast<DT, 1>(P, nw, grid, lds, st) : launch_fast<DT, 0>(P, nw, grid, lds, st);
    if (rc) return rc;
    if (ctx->profile) { (void)hipEventRecord(ctx->ev[1], st); (void)hipEventRecord(ctx->ev[2], st); }
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
    int max_cols = 0;
    for (int e = 0; e < n_eyes; e++) {
        const double dv = eyes[e].divergence_px, sp = eyes[e].separation_px;
        DS_REQUIRE(dv == dv && sp == sp && fabs(dv) < 1e6 && fabs(sp) < 1e6, DS_EINVAL, "ds_stereo_warp: divergence/separation not finite");
        const double dmin = dv < 0 ? dv : 0.0, dmax = dv > 0 ? dv : 0.0;
        P.div_px[e] = dv; P.sep_px[e] = sp;
        P.out[e] = eyes[e].out; P.ors[e] = eyes[e].out_row_stride; P.ois[e] = eyes[e].out_img_stride;
        // source columns that can reach output pixel col: [col + offL, col + offU] (one column of slack each side)
        P.offL[e] = (int)floor(-1.95 - sp - dmax) - 1;
        P.offU[e] = (int)ceil(0.95 - sp - dmin) + 1;
        const int ncols = P.offU[e] - P.offL[e] + 1;
        if (ncols > max_cols) max_cols = ncols;
    }
    const int npts = sharp ? 2 : 1;
    int nw = 1;
    while (nw <= 4 && max_cols * npts > 64 * nw) nw *= 2;
    DS_REQUIRE(nw <= 4, DS_EUNSUPPORTED,
               "ds_stereo_warp: |divergence_px| too large for the polylines kernel (window of %d columns > %d)", max_cols, 256 / npts);

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

    DS_REQUIRE(h <= 65535 && (int64_t)n * n_eyes <= 65535, DS_EUNSUPPORTED, "ds_stereo_warp: h and n*n_eyes must be <= 65535");
    dim3 grid((w + PL_TILE - 1) / PL_TILE, h, n * n_eyes);
    const int ncw_max = PL_TILE + max_cols + 2;
    const size_t lds = (size_t)ncw_max * (2 * sizeof(double) + c);
    DS_REQUIRE(lds <= 160 * 1024, DS_EUNSUPPORTED, "ds_stereo_warp: LDS window too large");

    switch (depth_dtype) {
    case DS_DEPTH_U16: rc = launch_dt<DS_DEPTH_U16>(ctx, P, sharp, nw, grid, lds, S, nworkers / 64, st); break;
    case DS_DEPTH_F32: rc = launch_dt<DS_DEPTH_F32>(ctx, P, sharp, nw, grid, lds, S, nworkers / 64, st); break;
    case DS_DEPTH_F64: rc = launch_dt<DS_DEPTH_F64>(ctx, P, sharp, nw, grid, lds, S, nworkers / 64, st); break;
    default: ds_set_error("unknown depth dtype %d", depth_dtype); return DS_EINVAL;
    }
    if (rc) { ds_set_error("ds_stereo_warp: no polylines kernel for this window size"); return rc; }
    DS_HIP_CHECK(hipGetLastError());
    ctx->last_exact_rows_valid = nrows;
    return DS_OK;
}
