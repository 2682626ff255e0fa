"""Pure-Python CPU restatement of the polylines stereo path -- TEST / BASELINE INFRASTRUCTURE ONLY.

What the reference runs when numba is missing is its own kernel interpreted by CPython
(src/stereoimage_generation.py:1-8 fallback decorators, :162-283 the kernel; README: "much slower").  /root/reference
does not exist on the GPU box, so bench.py's ``cpu_baseline.python_fallback`` leg times THIS restatement there: plain
Python floats and lists, one core, same algorithm in the same operation order, bit-identical to the C port
(oracle/depthstereo_oracle.c) and to the reference-made goldens (tests/test_oracle_golden.py).  The reference's own
fallback -- numpy row objects instead of Python floats -- is timed in the build container by
tools/time_reference_fallback.py (profiles/round2_reference_fallback.json); it is several times slower still.

Layout (same as the exact HIP fallback k_polylines_exact): points in original order (ox, od), the sorted view (sx) with
the original index travelling along (sk); segment k = (point k, point k+1); the active set holds segment indices.
"""
import numpy as np

EPS = 1e-7


def _row_points(nd_row, div_px, sep_px, half):
    """:175-192 -> (ox, od) lists: x coordinate and closeness |coord_d| of every point, sentinels included."""
    w = len(nd_row)
    ox, od = [-1.0 * w], [0.0]
    for col in range(w):
        coord_d = nd_row[col] * div_px                       # exponent folded into nd_row by the caller (:182)
        coord_x = col + 0.5 + coord_d + sep_px               # :183
        if half < EPS:
            ox.append(coord_x); od.append(abs(coord_d))
        else:
            ox.append(coord_x - half); od.append(abs(coord_d))
            ox.append(coord_x + half); od.append(abs(coord_d))
    ox.append(2.0 * w); od.append(0.0)
    return ox, od


def polylines_row(px_row, nd_row, div_px, sep_px, sharp):
    """One image row (:173-282).  px_row: list of per-pixel channel tuples (ints); nd_row: list of floats."""
    w = len(px_row)
    c = len(px_row[0])
    half = 0.45 if sharp else 0.0
    ox, od = _row_points(nd_row, div_px, sep_px, half)
    n_pt = len(ox)
    n_sg = n_pt - 1                                           # :196
    last = n_pt - 1

    def colour_index(p):                                      # third component of a point (:179,:185,:188-191)
        if p == 0:
            return 0
        if p == last:
            return w - 1
        return (p - 1) >> 1 if sharp else p - 1

    sx, sk = list(ox), list(range(n_pt))
    for i in range(1, n_sg):                                  # :214-219 insertion sort, stable
        u = i - 1
        while u >= 0 and sx[u] > sx[u + 1]:
            sx[u], sx[u + 1] = sx[u + 1], sx[u]
            sk[u], sk[u + 1] = sk[u + 1], sk[u]
            u -= 1
    csg = [None] * (5 * int(abs(div_px)) + 25)                # :223; None = a row of zeros nobody wrote yet
    n_act = 0
    nxt = 0                                                   # sg_pointer
    pt_i = 0
    out = []
    for col in range(w):                                      # :228
        colour = [0.5] * c
        while sx[pt_i] < col:
            pt_i += 1
        pt_i -= 1
        while sx[pt_i] < col + 1:                             # :234
            a, b = sx[pt_i], sx[pt_i + 1]
            lo = (a if a > col else float(col)) + EPS         # max(col, .) with numpy's operand order
            hi = (b if b < col + 1 else float(col + 1)) - EPS
            weight = hi - lo                                  # significance
            centre = lo + 0.5 * weight
            while nxt < n_sg and sx[nxt] < centre:            # :242
                csg[n_act] = sk[nxt]
                nxt += 1
                n_act += 1
            i = 0
            while i < n_act:                                  # :247
                if ox[csg[i] + 1] < centre:
                    csg[i] = csg[n_act - 1]
                    n_act -= 1
                else:
                    i += 1
            best = 0
            if n_act != 1:                                    # :259
                best_close = -EPS
                for i in range(n_act):
                    k = csg[i]
                    t = (centre - ox[k]) / (ox[k + 1] - ox[k])
                    close = (1.0 - t) * od[k] + t * od[k + 1]
                    if best_close < close and 0.0 < t < 1.0:
                        best_close, best = close, i
            k = csg[best]
            if k is None:                                     # the all-zero row: both colour indices 0, flat colour
                cl = cr = 0
            else:
                cl, cr = colour_index(k), colour_index(k + 1)
            if cl == cr:                                      # :272
                p = px_row[cl]
                for q in range(c):
                    colour[q] += p[q] * weight
            else:
                t = (centre - ox[k]) / (ox[k + 1] - ox[k])
                pl, pr = px_row[cl], px_row[cr]
                for q in range(c):
                    colour[q] += (pl[q] * (1.0 - t) + pr[q] * t) * weight
            pt_i += 1
        out.append([int(v) & 0xff for v in colour])           # :281 float64 -> uint8
    return out


def apply_stereo_divergence(original_image, depth, divergence, separation, stereo_offset_exponent, fill_technique):
    """:77-92 for the two polylines fills."""
    assert fill_technique in ('polylines_sharp', 'polylines_soft'), "oracle_py restates the polylines kernels only"
    img = np.asarray(original_image)
    depth = np.asarray(depth)
    h, w, c = img.shape
    with np.errstate(all='ignore'):
        dmin, dmax = depth.min(), depth.max()
        nd = (depth - dmin) / (dmax - dmin)                   # :79-81 in the depth's own dtype arithmetic
    div_px = (divergence / 100.0) * w
    sep_px = (separation / 100.0) * w
    out = np.zeros_like(img)
    rows_px = img.tolist()
    for row in range(h):
        nd_row = [float(v) ** stereo_offset_exponent for v in nd[row]] if stereo_offset_exponent != 1.0 else [float(v) for v in nd[row]]
        out[row] = polylines_row([tuple(p) for p in rows_px[row]], nd_row, div_px, sep_px, fill_technique == 'polylines_sharp')
    return out


def create_stereoimages_arrays(original_image, depthmap, divergence, separation=0.0, modes=None, stereo_balance=0.0,
                               stereo_offset_exponent=1.0, fill_technique='polylines_sharp'):
    """:13-74, 'left-right' only (the metric's output)."""
    modes = ['left-right'] if modes is None else modes
    assert modes == ['left-right']
    img = np.asarray(original_image)
    balance = (stereo_balance + 1) / 2
    left = img if balance < 0.001 else apply_stereo_divergence(
        img, depthmap, +1 * divergence * balance, -1 * separation, stereo_offset_exponent, fill_technique)
    right = img if balance > 0.999 else apply_stereo_divergence(
        img, depthmap, -1 * divergence * (1 - balance), separation, stereo_offset_exponent, fill_technique)
    return [np.hstack([left, right])]
