#!/usr/bin/env python3
"""Golden normal maps from the REFERENCE's own ``create_normalmap`` (src/normalmap_generation.py:5-56).

Build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_normalmap.py

The reference module is imported unmodified.  Its only absent import is ``cv2`` (opencv-python is not installed and not
even pinned by the reference: SURVEY.md section 8c), which it uses for exactly two calls:

* ``cv2.Sobel(np.float64(img), cv2.CV_64F, dx, dy, ksize=k)`` (:28-29).  The stub below implements OpenCV's documented
  definition -- the separable kernels of ``getDerivKernels`` for k = 1, 3, 5, 7, written out as literal tables, applied as a
  2-D correlation with the default ``BORDER_REFLECT_101`` -- as a DIRECT 2-D sum over an ``np.pad(mode='reflect')`` image
  (not the separable two-pass order the oracle uses).  The tables are small integers, and every Sobel case below feeds
  values that are integer multiples of 2**-8 with magnitude < 2**8 (uint16 / 256, or floats built that way): every product
  and every partial sum is then exactly representable in float64 (< 2**8 * 2**12 with 8 fractional bits), so ANY summation
  order -- OpenCV's, this stub's, the oracle's, the HIP kernel's -- gives the same bits.  These cases are therefore
  reference-executed goldens, not stand-ins.
* ``np.gradient`` path (``sobel_gradient=None``, :31): no cv2 call at all; arbitrary float64 input is pinned bit for bit.
* ``cv2.GaussianBlur(img, (k, k), k)`` (:24, :43).  The stub follows ``getGaussianKernel``'s documented formula for
  sigma > 0 (``exp(-(i-(k-1)/2)**2 / (2 sigma**2))`` normalised to sum 1) and REFLECT_101; OpenCV's own summation order and
  its fixed small-kernel tables are NOT reproducible here, so the blur cases are stored with ``standin=1`` and the tests
  compare them with a one-LSB tolerance: they pin the structure (which blur, where, re-normalisation), not the rounding.

Output: normalmap_cases.npz.
"""
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'

# getDerivKernels (Sobel), written out: (smoothing, derivative) per aperture size
SOBEL = {
    1: ([1.0], [-1.0, 0.0, 1.0]),
    3: ([1.0, 2.0, 1.0], [-1.0, 0.0, 1.0]),
    5: ([1.0, 4.0, 6.0, 4.0, 1.0], [-1.0, -2.0, 0.0, 2.0, 1.0]),
    7: ([1.0, 6.0, 15.0, 20.0, 15.0, 6.0, 1.0], [-1.0, -4.0, -5.0, 0.0, 5.0, 4.0, 1.0]),
}


def _correlate2d_reflect101(img, k2d):
    kh, kw = k2d.shape
    ph, pw = kh // 2, kw // 2
    p = np.pad(img, ((ph, ph), (pw, pw)), mode='reflect')       # numpy 'reflect' == BORDER_REFLECT_101
    h, w = img.shape
    out = np.zeros((h, w), np.float64)
    for i in range(kh):
        for j in range(kw):
            if k2d[i, j] != 0.0:
                out += k2d[i, j] * p[i:i + h, j:j + w]
    return out


def make_stub_cv2():
    cv2 = types.ModuleType('cv2')
    cv2.CV_64F = 6

    def Sobel(src, ddepth, dx, dy, ksize=3):
        assert ddepth == cv2.CV_64F and src.dtype == np.float64 and (dx, dy) in ((1, 0), (0, 1))
        smooth, deriv = SOBEL[ksize]
        kx = np.array(deriv if dx else smooth)        # along x (columns)
        ky = np.array(deriv if dy else smooth)        # along y (rows)
        return _correlate2d_reflect101(src, np.outer(ky, kx))

    def GaussianBlur(src, ksize, sigmaX):
        k = ksize[0]
        assert ksize[0] == ksize[1] and k % 2 == 1 and sigmaX > 0
        x = np.arange(k, dtype=np.float64) - (k - 1) / 2.0
        g = np.exp(-(x * x) / (2.0 * float(sigmaX) ** 2))
        g /= g.sum()
        k2 = np.outer(g, g)
        src = np.asarray(src)
        if src.ndim == 2:
            return _correlate2d_reflect101(np.float64(src), k2).astype(src.dtype if src.dtype.kind == 'f' else np.float64)
        return np.dstack([_correlate2d_reflect101(np.float64(src[:, :, c]), k2) for c in range(src.shape[2])])

    cv2.Sobel = Sobel
    cv2.GaussianBlur = GaussianBlur
    return cv2


def depth_cases():
    rng = np.random.default_rng(11)
    H, W = 48, 64
    yy, xx = np.mgrid[0:H, 0:W].astype(np.int64)
    d = (xx * 30000) // (W - 1) + ((xx // 8 + yy // 8) % 2) * 8000
    d[H // 4: H // 2, W // 3: 2 * W // 3] = 60000
    d[(3 * H) // 4:, : W // 5] = 1000
    out = {
        'survey48x64': d.astype(np.uint16),                                           # SURVEY.md Appendix A pattern
        'noise37x53': rng.integers(0, 65536, (37, 53), dtype=np.uint16),
        'extremes9x11': rng.choice(np.array([0, 1, 255, 256, 32767, 32768, 65534, 65535], np.uint16), (9, 11)),
        'smooth64x80': (32768 + 30000 * np.sin(np.mgrid[0:64, 0:80][1] / 9.0) * np.cos(np.mgrid[0:64, 0:80][0] / 7.0)).astype(np.uint16),
        'tiny3x3': rng.integers(0, 65536, (3, 3), dtype=np.uint16),
    }
    return out


def main():
    sys.modules['cv2'] = make_stub_cv2()
    sys.path.insert(0, REF)
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_normalmap_generation', os.path.join(REF, 'src', 'normalmap_generation.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    out, index = {}, []

    def add(name, depth, pre, sob, post, inv, standin=0):
        r = np.asarray(ref.create_normalmap(depth.copy(), pre, sob, post, inv))
        key = f'{name}__p{pre}_s{sob}_q{post}_i{int(inv)}'
        out[key + '__out'] = r
        index.append({'key': key, 'depth': name, 'pre_blur': pre, 'sobel': sob, 'post_blur': post, 'invert': bool(inv), 'standin': standin})

    deps = depth_cases()
    for name, d in deps.items():
        out[name + '__depth'] = d
        for inv in (False, True):
            for sob in (3, None, 0, 1, 5, 7):
                if d.shape[0] < 7 and sob in (5, 7):
                    continue                                   # REFLECT_101 needs size > radius
                add(name, d, None, sob, None, inv)
    # other dtypes the reference accepts (:20-21 promote): exactly representable values for Sobel, anything for np.gradient
    rng = np.random.default_rng(12)
    f64 = rng.integers(-(1 << 15), 1 << 15, (21, 34)).astype(np.float64) / 4.0        # multiples of 1/4, |v| < 2**13
    out['f64quarter21x34__depth'] = f64
    for sob in (3, 5, 7, 1):
        add('f64quarter21x34', f64, None, sob, None, False)
    f64r = rng.normal(0.0, 3000.0, (29, 31))
    out['f64normal29x31__depth'] = f64r
    add('f64normal29x31', f64r, None, None, None, False)
    add('f64normal29x31', f64r, None, None, None, True)
    f32 = (rng.integers(0, 1 << 16, (19, 23)).astype(np.float32))                      # float32 depth, Sobel path (np.float64(...) at :28)
    out['f32int19x23__depth'] = f32
    add('f32int19x23', f32, None, 3, None, False)
    i32 = rng.integers(0, 1 << 16, (17, 20)).astype(np.int32)
    out['i32_17x20__depth'] = i32
    add('i32_17x20', i32, None, 3, None, True)
    add('i32_17x20', i32, None, None, None, False)
    u8 = rng.integers(0, 256, (16, 18), dtype=np.uint8)
    out['u8_16x18__depth'] = u8
    add('u8_16x18', u8, None, 3, None, False)
    # float32 depth with np.gradient: the one combination the reference evaluates in FLOAT32 end to end (:20-21 keep float32,
    # :31 np.gradient, :34-39 np.linalg.norm and the divisions, :51-54 the quantisation) -- arbitrary values, no cv2 call
    f32r = rng.normal(0.0, 3000.0, (27, 33)).astype(np.float32)
    out['f32normal27x33__depth'] = f32r
    f32m = (rng.random((22, 45)) * 37.0 + 5.0).astype(np.float32)                       # a MiDaS-like raw prediction: small gradients
    f32m[5:11, 10:30] += 11.5
    out['f32midas22x45__depth'] = f32m
    f32s = np.zeros((8, 9), np.float32)                                                # flat areas (n == 1 exactly) and huge steps
    f32s[2:5, 3:7] = 6.0e4
    f32s[6, :] = -3.0e38 / 1.0e33
    out['f32steps8x9__depth'] = f32s
    for nm_, arr in (('f32normal27x33', f32r), ('f32midas22x45', f32m), ('f32steps8x9', f32s)):
        for inv in (False, True):
            for sob in (None, 0):
                add(nm_, arr, None, sob, None, inv)
    # float16 depth with np.gradient (round 6): numpy stays in float16 from end to end (each ufunc loop computes in float32 and rounds
    # to half; np.linalg.norm's add.reduce keeps a float32 accumulator) -- arbitrary values, no cv2 call: reference-exact
    f16r = rng.normal(0.0, 3000.0, (31, 37)).astype(np.float16)
    f16r[0:3, 0:5] = 60000.0
    f16r[10, 10] = -60000.0
    f16r[10, 11] = 60000.0                                                              # a step whose square overflows half: n = inf
    out['f16normal31x37__depth'] = f16r
    f16m = (rng.random((26, 41)) * 37.0 + 5.0).astype(np.float16)
    f16m[5:11, 10:30] += np.float16(11.5)
    out['f16midas26x41__depth'] = f16m
    f16t = (rng.random((12, 14)) * 0.01).astype(np.float16)                             # quotients by 256 land in the subnormals
    f16t[3:6, 4:9] = 0.0
    out['f16tiny12x14__depth'] = f16t
    with np.errstate(all='ignore'):
        for nm_, arr in (('f16normal31x37', f16r), ('f16midas26x41', f16m), ('f16tiny12x14', f16t)):
            for inv in (False, True):
                for sob in (None, 0):
                    add(nm_, arr, None, sob, None, inv)
    f16i = rng.integers(0, 2048, (15, 22)).astype(np.float16)                           # integers: exact through / 256 and the Sobel stub
    out['f16int15x22__depth'] = f16i
    add('f16int15x22', f16i, None, 3, None, False)
    add('f16int15x22', f16i, None, 5, None, True)
    # float32 depth, np.gradient AND Gaussian blurs: cv2.GaussianBlur on CV_32F data (stand-in: float64 sums cast back to float32)
    for args in ((3, None, None), (None, None, 3), (5, 0, 3)):
        add('f32midas22x45', f32m, args[0], args[1], args[2], False, standin=1)
        add('f32normal27x33', f32r, args[0], args[1], args[2], True, standin=1)
    # blur structure (stand-in arithmetic, see the module docstring)
    for args in ((3, 3, None), (None, 3, 3), (5, None, 3), (3, 5, 5)):
        add('survey48x64', deps['survey48x64'], args[0], args[1], args[2], False, standin=1)
        add('noise37x53', deps['noise37x53'], args[0], args[1], args[2], True, standin=1)

    out['__index__'] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'normalmap_cases.npz'), **out)
    print('wrote', len(index), 'cases,', sum(1 for c in index if not c['standin']), 'reference-exact')


if __name__ == '__main__':
    main()
