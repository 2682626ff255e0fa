"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Python face of ``oracle/depthstereo_oracle.c`` (ctypes) plus numpy restatements of the pieces whose
arithmetic lives in OpenCV in the reference.  Function names and signatures mirror the reference
(``/root/reference`` = thygate/stable-diffusion-webui-depthmap-script v0.4.8):

* ``create_stereoimages`` / ``apply_stereo_divergence`` / ``overlap_red_cyan``
  -- src/stereoimage_generation.py:13-74, :77-92, :286-307
* ``create_normalmap``   -- src/normalmap_generation.py:5-56
* ``convert_to_i16``     -- src/core.py:44-50
* ``depth_normalize01``  -- src/core.py:189-206 (no-clip branch)

Pinned: the stereo functions against the reference's own Python code executed in the build
container (``tests/golden/*.npz`` made by ``tests/golden/make_golden.py``) and SURVEY.md Appendix A.
The normal map's default path (Sobel 3x3, no blur) and every integer-kernel Sobel are exactly
representable in float64, so the restatement is authoritative; the Gaussian blur paths depend on
OpenCV's summation order and are **parity unpinned** (cv2 is not installed here).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so with gcc (see oracle/Makefile)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "depthstereo_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        u8p, u16p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint16)
        f64p, f32p = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float)
        ci, cd, i64 = ctypes.c_int, ctypes.c_double, ctypes.c_int64
        L.orc_normalize_depth_u16.argtypes = [u16p, i64, f64p]
        L.orc_normalize_depth_f64.argtypes = [f64p, i64, f64p]
        L.orc_normalize_depth_f32.argtypes = [f32p, i64, f64p]
        L.orc_stereo_naive.argtypes = [u8p, ci, ci, ci, f64p, cd, cd, cd, ci, u8p]
        L.orc_stereo_naive.restype = ci
        L.orc_stereo_polylines.argtypes = [u8p, ci, ci, ci, f64p, cd, cd, cd, ci, u8p]
        L.orc_stereo_polylines.restype = ci
        L.orc_overlap_red_cyan.argtypes = [u8p, u8p, ci, ci, ci, u8p]
        L.orc_normalmap_sobel3.argtypes = [u16p, ci, ci, ci, u8p]
        L.orc_normalmap_gradient.argtypes = [u16p, ci, ci, ci, u8p]
        L.orc_convert_to_i16_f64.argtypes = [f64p, i64, u16p]
        L.orc_convert_to_i16_f32.argtypes = [f32p, i64, u16p]
        L.orc_depth_normalize01_f32.argtypes = [f32p, i64, ci, f32p]
        L.orc_num_threads.restype = ci
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


FILLS_NAIVE = {"none": 0, "naive": 1, "naive_interpolating": 2}
FILLS_POLY = {"polylines_soft": 0, "polylines_sharp": 1}


def normalize_depth(depth):
    """stereoimage_generation.py:79-81 -> float64 HxW."""
    depth = np.asarray(depth)
    out = np.empty(depth.shape, np.float64)
    L = lib()
    if depth.dtype == np.uint16:
        d = np.ascontiguousarray(depth)
        L.orc_normalize_depth_u16(_p(d, ctypes.c_uint16), d.size, _p(out, ctypes.c_double))
    elif depth.dtype == np.float32:
        d = np.ascontiguousarray(depth)
        L.orc_normalize_depth_f32(_p(d, ctypes.c_float), d.size, _p(out, ctypes.c_double))
    elif depth.dtype == np.float64:
        d = np.ascontiguousarray(depth)
        L.orc_normalize_depth_f64(_p(d, ctypes.c_double), d.size, _p(out, ctypes.c_double))
    else:  # other integer dtypes: numpy semantics directly
        with np.errstate(all="ignore"):
            out = ((depth - depth.min()) / (depth.max() - depth.min())).astype(np.float64)
    return out


def apply_stereo_divergence(original_image, depth, divergence, separation, stereo_offset_exponent, fill_technique):
    """stereoimage_generation.py:77-92."""
    original_image = np.ascontiguousarray(original_image)
    assert original_image.shape[:2] == depth.shape, 'Depthmap and the image must have the same size'
    h, w, c = original_image.shape
    norm = normalize_depth(depth)
    divergence_px = (divergence / 100.0) * w
    separation_px = (separation / 100.0) * w
    out = np.empty_like(original_image)
    L = lib()
    if fill_technique in FILLS_NAIVE:
        rc = L.orc_stereo_naive(_p(original_image, ctypes.c_uint8), h, w, c, _p(norm, ctypes.c_double),
                                divergence_px, separation_px, float(stereo_offset_exponent),
                                FILLS_NAIVE[fill_technique], _p(out, ctypes.c_uint8))
        assert rc == 0
        return out
    if fill_technique in FILLS_POLY:
        rc = L.orc_stereo_polylines(_p(original_image, ctypes.c_uint8), h, w, c, _p(norm, ctypes.c_double),
                                    divergence_px, separation_px, float(stereo_offset_exponent),
                                    FILLS_POLY[fill_technique], _p(out, ctypes.c_uint8))
        assert rc == 0
        return out
    return None  # reference: unknown fill silently yields None (:85-92)


def overlap_red_cyan(im1, im2):
    im1, im2 = np.ascontiguousarray(im1), np.ascontiguousarray(im2)
    h, w, c = im2.shape
    out = np.zeros((h, w, 3), np.uint8)
    lib().orc_overlap_red_cyan(_p(im1, ctypes.c_uint8), _p(im2, ctypes.c_uint8), h, w, c, _p(out, ctypes.c_uint8))
    return out


def create_stereoimages_arrays(original_image, depthmap, divergence, separation=0.0, modes=None,
                               stereo_balance=0.0, stereo_offset_exponent=1.0, fill_technique='polylines_sharp'):
    """stereoimage_generation.py:13-74, returning ndarrays instead of PIL images."""
    if modes is None:
        modes = ['left-right']
    if not isinstance(modes, list):
        modes = [modes]
    if len(modes) == 0:
        return []
    original_image = np.asarray(original_image)
    balance = (stereo_balance + 1) / 2
    left_eye = original_image if balance < 0.001 else \
        apply_stereo_divergence(original_image, depthmap, +1 * divergence * balance, -1 * separation,
                                stereo_offset_exponent, fill_technique)
    right_eye = original_image if balance > 0.999 else \
        apply_stereo_divergence(original_image, depthmap, -1 * divergence * (1 - balance), separation,
                                stereo_offset_exponent, fill_technique)
    results = []
    for mode in modes:
        if mode == 'left-right':
            results.append(np.hstack([left_eye, right_eye]))
        elif mode == 'right-left':
            results.append(np.hstack([right_eye, left_eye]))
        elif mode == 'top-bottom':
            results.append(np.vstack([left_eye, right_eye]))
        elif mode == 'bottom-top':
            results.append(np.vstack([right_eye, left_eye]))
        elif mode == 'red-cyan-anaglyph':
            results.append(overlap_red_cyan(left_eye, right_eye))
        elif mode == 'left-only':
            results.append(left_eye)
        elif mode == 'only-right':
            results.append(right_eye)
        elif mode == 'cyan-red-reverseanaglyph':
            results.append(overlap_red_cyan(right_eye, left_eye))
        else:
            raise Exception('Unknown mode')
    return results


def create_stereoimages(*args, **kwargs):
    from PIL import Image
    return [Image.fromarray(r) for r in create_stereoimages_arrays(*args, **kwargs)]


# ------------------------------------------------------------------------------------------------
# normal map
def _reflect101_idx(n, r):
    idx = np.arange(-r, n + r)
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * (n - 1)
    idx = np.abs(idx) % period
    return np.where(idx >= n, period - idx, idx)


def sobel_kernels(ksize, order):
    """OpenCV getDerivKernels/getSobelKernels for one axis (order 0 = smoothing, 1 = derivative)."""
    if ksize == 1:
        return np.array([[0., 1., 0.], [-1., 0., 1.]][order]) if order else np.array([1.0])
    if ksize == 3:
        return np.array([[1., 2., 1.], [-1., 0., 1.]][order])
    ker = np.zeros(ksize + 1)
    ker[0] = 1
    for _ in range(ksize - order - 1):
        old = ker[0]
        for j in range(1, ksize + 1):
            new = ker[j] + ker[j - 1]
            ker[j - 1] = old
            old = new
    for _ in range(order):
        old = -ker[0]
        for j in range(1, ksize + 1):
            new = ker[j - 1] - ker[j]
            ker[j - 1] = old
            old = new
    return ker[:ksize].copy()


def _sep_filter(a, kx, ky):
    """Separable correlation with BORDER_REFLECT_101: rows (kx) first, then columns (ky); taps summed in order."""
    h, w = a.shape[:2]
    rx, ry = len(kx) // 2, len(ky) // 2
    ix, iy = _reflect101_idx(w, rx), _reflect101_idx(h, ry)
    t = np.zeros_like(a)
    for k, cf in enumerate(kx):
        t = t + cf * a[:, ix[k:k + w]]
    o = np.zeros_like(a)
    for k, cf in enumerate(ky):
        o = o + cf * t[iy[k:k + h]]
    return o


def gaussian_kernel(ksize, sigma):
    """cv2.getGaussianKernel(ksize, sigma>0, CV_64F)."""
    import math
    scale2x = -0.5 / (sigma * sigma)
    cf, total = [], 0.0
    for i in range(ksize):                       # OpenCV sums the taps sequentially in double
        x = i - (ksize - 1) * 0.5
        t = math.exp(scale2x * x * x)
        cf.append(t)
        total += t
    inv = 1.0 / total
    return np.array([t * inv for t in cf], dtype=np.float64)


def gaussian_kernel_f32(ksize, sigma):
    """cv2.getGaussianKernel(ksize, sigma>0, CV_32F): exp() rounded to float, the floats summed in double, scaled by the double
    reciprocal and rounded to float again."""
    import math
    scale2x = -0.5 / (sigma * sigma)
    cf, total = [], 0.0
    for i in range(ksize):
        x = i - (ksize - 1) * 0.5
        t = float(np.float32(math.exp(scale2x * x * x)))
        cf.append(t)
        total += t
    inv = 1.0 / total
    return np.array([t * inv for t in cf], dtype=np.float32)


def create_normalmap_array(depthmap, pre_blur=None, sobel_gradient=3, post_blur=None, invert=False):
    depthmap = np.asarray(depthmap)
    if (pre_blur is None or pre_blur <= 0) and (post_blur is None or post_blur <= 0) and depthmap.dtype == np.uint16:
        h, w = depthmap.shape
        d = np.ascontiguousarray(depthmap)
        out = np.empty((h, w, 3), np.uint8)
        if sobel_gradient == 3:
            lib().orc_normalmap_sobel3(_p(d, ctypes.c_uint16), h, w, int(bool(invert)), _p(out, ctypes.c_uint8))
            return out
        if sobel_gradient is None or sobel_gradient <= 0:
            lib().orc_normalmap_gradient(_p(d, ctypes.c_uint16), h, w, int(bool(invert)), _p(out, ctypes.c_uint8))
            return out
    normalmap = depthmap if invert else depthmap * (-1.0)
    normalmap = normalmap / 256.0
    gradient = sobel_gradient is None or sobel_gradient <= 0
    # float32 data on the np.gradient path stays float32 through cv2.GaussianBlur (CV_32F coefficients and accumulators); every other
    # combination reaches the blurs as float64 (or is promoted for cv2.Sobel right behind the first one: restated in float64)
    f32 = gradient and normalmap.dtype == np.float32
    if pre_blur is not None and pre_blur > 0:
        g = gaussian_kernel_f32(pre_blur, float(pre_blur)) if f32 else gaussian_kernel(pre_blur, float(pre_blur))
        normalmap = _sep_filter(normalmap if f32 else np.float64(normalmap), g, g)
    if not gradient:
        kd, ks = sobel_kernels(sobel_gradient, 1), sobel_kernels(sobel_gradient, 0)
        zx = _sep_filter(np.float64(normalmap), kd, ks)
        zy = _sep_filter(np.float64(normalmap), ks, kd)
    else:
        zy, zx = np.gradient(normalmap)
    normal = np.dstack((zx, -zy, np.ones_like(normalmap)))

    def _norm(v):
        # np.linalg.norm(v, axis=2) (:34, :45): sqrt(add.reduce(v * v)) in the array's own dtype -- float16 reduces with a float32
        # accumulator and one rounding, which the ufunc does by itself
        return np.sqrt(np.add.reduce(v * v, axis=2))
    n = _norm(normal)
    normal[:, :, 0] /= n
    normal[:, :, 1] /= n
    normal[:, :, 2] /= n
    if post_blur is not None and post_blur > 0:
        g = gaussian_kernel_f32(post_blur, float(post_blur)) if f32 else gaussian_kernel(post_blur, float(post_blur))
        normal = np.dstack([_sep_filter(np.ascontiguousarray(normal[:, :, k]), g, g) for k in range(3)])
        n = _norm(normal)
        normal[:, :, 0] /= n
        normal[:, :, 1] /= n
        normal[:, :, 2] /= n
    normal += 1
    normal /= 2
    normal = np.clip(normal * 256, 0, 256 - 0.1)
    return normal.astype(np.uint8)


def create_normalmap(depthmap, pre_blur=None, sobel_gradient=3, post_blur=None, invert=False):
    from PIL import Image
    return Image.fromarray(create_normalmap_array(depthmap, pre_blur, sobel_gradient, post_blur, invert))


# ------------------------------------------------------------------------------------------------
def convert_to_i16(arr):
    """core.py:44-50."""
    arr = np.asarray(arr)
    out = np.empty(arr.shape, np.uint16)
    if arr.dtype == np.float32:
        a = np.ascontiguousarray(arr)
        lib().orc_convert_to_i16_f32(_p(a, ctypes.c_float), a.size, _p(out, ctypes.c_uint16))
    else:
        a = np.ascontiguousarray(arr, dtype=np.float64)
        lib().orc_convert_to_i16_f64(_p(a, ctypes.c_double), a.size, _p(out, ctypes.c_uint16))
    return out


def depth_normalize01(raw_prediction, invert=False):
    """core.py:189-206 without clipping, float32 predictions."""
    a = np.ascontiguousarray(raw_prediction, dtype=np.float32)
    out = np.empty(a.shape, np.float32)
    lib().orc_depth_normalize01_f32(_p(a, ctypes.c_float), a.size, int(bool(invert)), _p(out, ctypes.c_float))
    return out


def depth_postprocess(raw_prediction, invert=False, clipdepth=False, clipdepth_mode='Range', far=0.0, near=1.0):
    """core.py:189-206 with the clipping branches (:196-201), plain numpy (NumPy >= 2 promotion rules: the float64
    percentile bounds of 'Outliers' promote the float32 prediction to float64; the python-float bounds of 'Range' do not)."""
    raw = np.asarray(raw_prediction)
    if not abs(raw.max() - raw.min()) > np.finfo("float").eps:
        return np.zeros(raw.shape)
    d = -raw if invert else raw.copy()
    if clipdepth and clipdepth_mode == 'Range':
        lo, hi = d.min(), d.max()
        d = np.clip((d - lo) / (hi - lo), far, near)
    elif clipdepth and clipdepth_mode == 'Outliers':
        bounds = np.percentile(d, [far * 100.0, near * 100.0])
        d = np.clip(d, bounds[0], bounds[1])
    lo, hi = d.min(), d.max()
    return (d - lo) / (hi - lo)


def depth_postprocess_f16_numpy1(raw_prediction_f16, invert=False, clipdepth=False, far=0.0, near=1.0):
    """core.py:189-211 for a FLOAT16 prediction -- what estimatemidas hands over on the reference's default GPU path
    (src/depthmap_generation.py:484-497: `.half()` network, the bicubic upsample in half, `.cpu().numpy()`) -- under the
    promotion rules of NumPy 1.x, the NumPy the reference was written for, spelled out with explicit casts so that the result
    does not depend on the NumPy that runs this file:
      * `out - out.min()`, `out.max() - out.min()` and their quotient are float16 operations (array and numpy-scalar operands
        of one dtype; each correctly rounded -- NumPy computes them in float32 and rounds, which is the same thing for + - /
        at 24 >= 2 * 11 + 2 bits);  np.clip(float16, python floats in [0, 1]) ('Range') stays float16;
      * `arr * 65536 + 0.0001` (convert_to_i16, :44-50): value-based casting makes the python int 65536 a uint32, and
        promote_types(float16, uint32) = float64, so the product, the sum and the clip are float64, truncated to uint16.
    I.e. the depth map is quantised to float16's ~2 k levels per octave and then scaled exactly.  Under NumPy >= 2 (NEP 50) the
    same reference code multiplies in float16 by float16(65536) = inf and stores zeros (and NaNs for a zero) -- see
    tests/test_host_logic.py::test_float16_prediction_corner.  Returns the uint16 depth map."""
    raw = np.asarray(raw_prediction_f16)
    assert raw.dtype == np.float16
    with np.errstate(over='ignore', invalid='ignore'):
        if not abs(raw.max() - raw.min()) > np.finfo("float").eps:
            return np.zeros(raw.shape, np.uint16)
        d = (-raw if invert else raw.copy())
        if clipdepth:                                                     # 'Range' (:197-198)
            lo, hi = d.min(), d.max()
            d = ((d - lo).astype(np.float16) / np.float16(hi - lo)).astype(np.float16)
            d = np.clip(d, np.float16(far), np.float16(near)).astype(np.float16)
        lo, hi = d.min(), d.max()
        norm = ((d - lo).astype(np.float16) / np.float16(hi - lo)).astype(np.float16)
        out = np.clip(norm.astype(np.float64) * 65536.0 + 0.0001, 0, 65536 - 0.1)
    return out.astype(np.uint16)


def colorize_u16(depth, lut_rgba, lo=2.0, hi=85.0):
    """dzoedepth/utils/misc.py:97-150 (colorize with default arguments, as src/core.py:271-274 calls it) on a uint16
    depth; lut_rgba = the colormap's bytes=True table [N, 4].  Percentile normalisation (:121-127) in float64, then
    matplotlib's Colormap.__call__ on floats: index = trunc(value * N), negative -> first entry, >= N -> last."""
    d = np.asarray(depth).squeeze()
    n = lut_rgba.shape[0]
    vmin, vmax = np.percentile(d, lo), np.percentile(d, hi)
    v = (d - vmin) / (vmax - vmin) if vmin != vmax else d * 0.0
    x = v * n
    idx = np.where(x < 0, 0, np.where(x >= n, n - 1, np.minimum(x, n - 1).astype(np.int64)))
    return lut_rgba[idx]


def num_threads():
    return int(lib().orc_num_threads())


# ---- Boost patch blend (src/depthmap_generation.py:915-937) --------------------------------------------------------------------
def _cv_cubic_resize(src, out_hw):
    """cv2.resize(src, (w, h), interpolation=cv2.INTER_CUBIC) restated from OpenCV's documentation: half-pixel centres,
    4 taps with a = -0.75, replicated border, no antialiasing.  float64.  PARITY UNPINNED (cv2 is not installable here)."""
    src = np.asarray(src, dtype=np.float64)
    sh, sw = src.shape
    oh, ow = out_hw

    def taps(n_out, n_in):
        f = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        i = np.floor(f).astype(np.int64)
        t = f - i
        A = -0.75
        w = np.empty((n_out, 4))
        w[:, 0] = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A
        w[:, 1] = ((A + 2) * t - (A + 3)) * t * t + 1
        w[:, 2] = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1
        w[:, 3] = 1.0 - w[:, 0] - w[:, 1] - w[:, 2]
        idx = np.clip(i[:, None] - 1 + np.arange(4)[None, :], 0, n_in - 1)
        return idx, w

    iy, wy = taps(oh, sh)
    ix, wx = taps(ow, sw)
    rows = np.zeros((sh, ow))
    for k in range(4):
        rows += src[:, ix[:, k]] * wx[:, k][None, :]
    out = np.zeros((oh, ow))
    for k in range(4):
        out += rows[iy[:, k], :] * wy[:, k][:, None]
    return out


def _cv_linear_resize(src, out_hw):
    """cv2.resize INTER_LINEAR (half-pixel centres, edge clamp), float64 accumulation, float32 result.  UNPINNED."""
    src = np.asarray(src, dtype=np.float64)
    sh, sw = src.shape
    oh, ow = out_hw

    def taps(n_out, n_in):
        f = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        i = np.floor(f).astype(np.int64)
        t = f - i
        t = np.where(i < 0, 0.0, t)
        i = np.where(i < 0, 0, i)
        t = np.where(i >= n_in - 1, 1.0, t)
        i = np.where(i >= n_in - 1, n_in - 2, i)
        return i, t

    iy, ty = taps(oh, sh)
    ix, tx = taps(ow, sw)
    top = (1 - tx)[None, :] * src[iy][:, ix] + tx[None, :] * src[iy][:, ix + 1]
    bot = (1 - tx)[None, :] * src[iy + 1][:, ix] + tx[None, :] * src[iy + 1][:, ix + 1]
    return ((1 - ty)[:, None] * top + ty[:, None] * bot).astype(np.float32)


def boost_blend(dst, rects, coefs, preds, mask_template):
    """The per-patch loop of estimateboost (:915-937): polyval, cubic resize to the rectangle, bilinear resize of the mask
    template, `dst[rect] = dst[rect] * (1 - mask) + merged * mask` with numpy's promotion (float32 * float32, float64 *
    float32, float64 sum stored into the float32 image).  Returns a new float32 array."""
    out = np.array(dst, dtype=np.float32, copy=True)
    for (x0, y0, w, h), (p0, p1), pred in zip(rects, coefs, preds):
        merged = _cv_cubic_resize(pred, (h, w)) * p0 + p1      # affine map and cubic resize commute (taps sum to one)
        mask = _cv_linear_resize(mask_template, (h, w))
        reg = out[y0:y0 + h, x0:x0 + w]
        out[y0:y0 + h, x0:x0 + w] = np.multiply(reg, np.float32(1) - mask) + np.multiply(merged, mask)
    return out


def process_predicitons(predictions, smoothening='none'):
    """numpy restatement of video mode's normalisation, src/video_mode.py:103-128 (the spelling is the reference's).
    Pinned: tests/golden/video_cases.npz holds the outputs of the reference's own function
    (tests/golden/make_golden_video.py) and tests/test_oracle_golden.py reproduces them bit for bit.
      'none' (:115-116)          every frame scaled with the clip's global min / max
      'experimental' (:117-127)  the global 0.5 / 99.5 percentiles of the temporally smoothed clip (5 taps 0.1 0.2 0.4 0.2
                                 0.1, frame index clamped at both ends) replace min / max; the raw frames are scaled"""
    frames = [np.asarray(p) for p in predictions]
    if smoothening == 'none':                                                   # :105-111 with a = b = None
        lo = min(f.min() for f in frames)
        hi = max(f.max() for f in frames)
    elif smoothening == 'experimental':
        n = len(frames)
        smoothed = []
        for i in range(n):                                                      # :120-124
            acc = np.zeros_like(frames[i])
            for tap, weight in zip(range(-2, 3), (0.10, 0.20, 0.40, 0.20, 0.10)):
                acc += weight * frames[min(max(0, i + tap), n - 1)]
            smoothed.append(acc)
        lo, hi = np.percentile(np.stack(smoothed), [0.5, 99.5])                 # :126
    else:
        return predictions                                                      # :128
    return [(f - lo) / (hi - lo) for f in frames]
