"""CPU: the C oracle reproduces every reference-generated golden vector bit for bit.

The fixtures under tests/golden/ were produced by running the reference's own Python code
(tests/golden/make_golden.py); SURVEY.md Appendix A hashes are asserted when they are generated and again here.
"""
import hashlib
import warnings

import numpy as np
import pytest

import util

h16 = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def test_appendix_a_input_hashes():
    img, dep = util.survey_inputs(48, 64, 1)
    assert h16(img[0]) == 'c2f6d254fbd9afe6' and h16(dep[0]) == 'dcbaca95416947e6'
    img, dep = util.survey_inputs(96, 128, 2)
    assert h16(img[0]) == '962470ffe997cea7' and h16(dep[0]) == 'f7f1494b03dcb398'


def test_oracle_matches_reference_golden(oracle):
    z, index = util.load_stereo_golden()
    assert len(index) >= 80
    for case in index:
        img, dep = util.golden_inputs(case)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            res = oracle.create_stereoimages_arrays(img, dep, case['div'], case['sep'], case['modes'], case['bal'],
                                                    case['exp'], case['fill'])
        key = case['name'].replace('/', '__')
        for m, o in zip(case['modes'], res):
            g = z[f'{key}__{m}']
            assert g.shape == o.shape, (case['name'], m)
            assert np.array_equal(g, o), (case['name'], m, int((g != o).sum()))


def test_oracle_appendix_a_hashes(oracle):
    import make_golden as mg
    for (prefix, fill), (sbs, ana) in mg.SURVEY_A.items():
        case = next(c for c in mg.stereo_case_list() if c['name'] == f'{prefix}/{fill}')
        img, dep = mg.gen_inputs(case)
        r = oracle.create_stereoimages_arrays(img, dep, case['div'], case['sep'], case['modes'], case['bal'], case['exp'], fill)
        assert (h16(r[0]), h16(r[1])) == (sbs, ana), (prefix, fill)


def test_oracle_normalmap_known_answers(oracle):
    # SURVEY.md Appendix A (restatement-derived): default path on the Appendix A depth arrays
    _, dep = util.survey_inputs(48, 64, 1)
    nm = oracle.create_normalmap_array(dep[0])
    assert h16(nm) == 'bc1264e3fb8eb3d2' and int(nm.sum()) == 1027584 and tuple(nm[0, 0]) == (128, 128, 255)
    _, dep = util.survey_inputs(96, 128, 2)
    nm = oracle.create_normalmap_array(dep[0])
    assert h16(nm) == 'c3a9ce0f64a2f5e3' and int(nm.sum()) == 4218191


def test_oracle_normalmap_matches_reference_golden(oracle):
    """oracle.create_normalmap_array against outputs of the REFERENCE's own create_normalmap (src/normalmap_generation.py
    :5-56, imported unmodified by tests/golden/make_golden_normalmap.py with an exact-integer stub for cv2.Sobel): every
    Sobel aperture (1/3/5/7), np.gradient, both inversions, uint16 / uint8 / int32 / float32 / float64 depth -- bit for bit.
    The Gaussian-blur cases were made with a documented-formula stand-in for cv2.GaussianBlur (standin=1): one LSB."""
    z, index = util.load_normalmap_golden()
    exact = 0
    for c in index:
        d = z[c['depth'] + '__depth']
        got = oracle.create_normalmap_array(d, c['pre_blur'], c['sobel'], c['post_blur'], c['invert'])
        want = z[c['key'] + '__out']
        assert got.shape == want.shape and got.dtype == np.uint8, c['key']
        if c['standin']:
            assert np.abs(got.astype(np.int16) - want.astype(np.int16)).max() <= 1, c['key']
        else:
            assert np.array_equal(got, want), (c['key'], int((got != want).sum()))
            exact += 1
    assert exact >= 60


def test_oracle_normalmap_c_equals_numpy(oracle):
    # the fused C paths (Sobel 3, np.gradient) against the generic numpy restatement
    rng = np.random.default_rng(5)
    dep = rng.integers(0, 65536, (37, 53), dtype=np.uint16)
    for inv in (False, True):
        a = oracle.create_normalmap_array(dep, None, 3, None, inv)
        zx = oracle._sep_filter(np.float64(dep if inv else dep * -1.0) / 256.0, oracle.sobel_kernels(3, 1), oracle.sobel_kernels(3, 0))
        assert a.shape == (37, 53, 3) and zx.shape == (37, 53)
        # numpy path via a fake float depth to force the generic branch
        normalmap = (dep if inv else dep * (-1.0)) / 256.0
        zy_, zx_ = np.gradient(normalmap)
        g = oracle.create_normalmap_array(dep, None, None, None, inv)
        n = np.sqrt(zx_ ** 2 + zy_ ** 2 + 1.0)
        ref = np.clip(((np.dstack((zx_ / n, -zy_ / n, 1.0 / n)) + 1) / 2) * 256, 0, 255.9).astype(np.uint8)
        assert np.array_equal(g, ref)


def test_oracle_convert_to_i16(oracle):
    rng = np.random.default_rng(3)
    a = np.concatenate([rng.random(1000), [0.0, 1.0, 0.999999, 1e-9, -0.5, 1.5]])
    ref = np.clip(a * 65536 + 0.0001, 0, 65536 - 0.1).astype("uint16")
    assert np.array_equal(oracle.convert_to_i16(a), ref)
    a32 = a.astype(np.float32)
    ref32 = np.clip(a32 * 65536 + 0.0001, 0, 65536 - 0.1).astype("uint16")
    assert np.array_equal(oracle.convert_to_i16(a32), ref32)


def test_oracle_depth_normalize(oracle):
    p = util.smooth_depth(40, 60, 1)
    for inv in (False, True):
        out = np.copy(p)
        if inv:
            out *= -1
        ref = (out - out.min()) / (out.max() - out.min())
        assert np.array_equal(oracle.depth_normalize01(p, inv), ref)
    assert not oracle.depth_normalize01(np.full((4, 4), 3.0, np.float32)).any()


def test_heatmap_oracle_matches_reference_colorize():
    """oracle.colorize_u16 against the reference's own colorize outputs (tests/golden/make_golden_heatmap.py)."""
    import os
    from oracle import oracle as orc
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'heatmap_cases.npz'))
    lut = z['inferno_lut']
    names = sorted(k[:-7] for k in z.files if k.endswith('__depth'))
    assert len(names) == 7
    for name in names:
        got = orc.colorize_u16(z[f'{name}__depth'], lut)
        assert got.dtype == np.uint8 and np.array_equal(got, z[f'{name}__rgba']), name


def test_python_port_reproduces_reference_goldens(oracle):
    """oracle/oracle_py.py (the pure-Python port bench.py times as cpu_baseline.python_fallback) against every small
    reference-made polylines golden with a left-right output, and against the C port on a seeded case."""
    from oracle import oracle_py
    z, index = util.load_stereo_golden()
    n = 0
    for case in index:
        if not case['fill'].startswith('polylines') or 'left-right' not in case['modes']:
            continue
        img, dep = util.golden_inputs(case)
        if img.shape[0] * img.shape[1] > 8000:
            continue
        got = oracle_py.create_stereoimages_arrays(img, dep, case['div'], case['sep'], ['left-right'], case['bal'], case['exp'], case['fill'])[0]
        assert np.array_equal(z[case['name'].replace('/', '__') + '__left-right'], got), case['name']
        n += 1
    assert n >= 10
    img, dep = util.survey_inputs(40, 96, 3)
    for fill in ('polylines_sharp', 'polylines_soft'):
        a = oracle_py.create_stereoimages_arrays(img[0], dep[0], 4.0, 0.5, ['left-right'], 0.2, 1.0, fill)[0]
        b = oracle.create_stereoimages_arrays(img[0], dep[0], 4.0, 0.5, ['left-right'], 0.2, 1.0, fill)[0]
        assert np.array_equal(a, b), fill


def _f16_kernel_model(depth, invert):
    """k_nm_gradient_f16 (csrc/ds_normalmap.hip) written out in numpy float32 with an explicit rounding to half behind every
    operation -- the arithmetic the device kernel performs, operation by operation."""
    f, hf = np.float32, np.float16

    def r(x):
        return x.astype(hf).astype(f)
    with np.errstate(all='ignore'):
        p = r(r(depth.astype(f) * f(1.0 if invert else -1.0)) / f(256))
        gx, gy = np.empty_like(p), np.empty_like(p)
        gx[:, 1:-1] = r(r(p[:, 2:] - p[:, :-2]) / f(2)); gx[:, 0] = r(p[:, 1] - p[:, 0]); gx[:, -1] = r(p[:, -1] - p[:, -2])
        gy[1:-1] = r(r(p[2:] - p[:-2]) / f(2)); gy[0] = r(p[1] - p[0]); gy[-1] = r(p[-1] - p[-2])
        a, b, c = gx, -gy, np.ones_like(p)
        n = r(np.sqrt(r((r(a * a) + r(b * b)) + r(c * c))))
        out = []
        for v in (a, b, c):
            t = r(r(r(r(v / n) + f(1)) / f(2)) * f(256))
            t = np.where(t < 0, f(0), t)
            t = np.where(t > f(hf(256 - 0.1)), f(hf(256 - 0.1)), t)
            out.append(np.where(np.isnan(t), f(0), t).astype(np.uint8))
    return np.dstack(out)


def test_float16_normalmap_kernel_arithmetic_matches_reference_golden():
    """Round 6: the float16 + np.gradient kernel's arithmetic (float32 operation, then one rounding to half, per numpy ufunc) restated
    in numpy and held to the reference-made float16 goldens bit for bit -- the GPU test then only has to show that the device
    executes this arithmetic."""
    z, index = util.load_normalmap_golden()
    seen = 0
    for c in index:
        d = z[c['depth'] + '__depth']
        if d.dtype != np.float16 or (c['sobel'] is not None and c['sobel'] > 0):
            continue
        got = _f16_kernel_model(d, c['invert'])
        want = z[c['key'] + '__out']
        assert np.array_equal(got, want), (c['key'], int((got != want).sum()))
        seen += 1
    assert seen == 12


def test_numpy_half_add_reduce_is_a_float32_chain_rounded_once():
    """np.linalg.norm(float16, axis=2) = sqrt(add.reduce(x * x)): the reduce adds the three squares left to right in a float32
    accumulator and rounds to half once.  Operand triples on which (s0 + s1) + s2, s0 + (s1 + s2) and the half-rounded chain all
    differ (found by search; s2 = 1 as in the normal's third component) pin which one numpy computes."""
    f, hf = np.float32, np.float16
    rng = np.random.default_rng(1)
    n = 4_000_000
    s0 = rng.integers(0x0400, 0x7bff, n, dtype=np.uint16).view(hf)
    s1 = rng.integers(0x0001, 0x7bff, n, dtype=np.uint16).view(hf)
    with np.errstate(all='ignore'):
        left = ((s0.astype(f) + s1.astype(f)) + f(1)).astype(hf)
        right = (s0.astype(f) + (s1.astype(f) + f(1))).astype(hf)
        chain = (s0 + s1) + hf(1)
        idx = np.nonzero((left != right) | (left != chain))[0]
        assert len(idx) > 100
        x = np.ones((3, len(idx), 3), hf)
        x[:, :, 0] = s0[idx]
        x[:, :, 1] = s1[idx]
        got = np.add.reduce(x, axis=2)
    assert np.array_equal(got, np.broadcast_to(left[idx], got.shape))
    assert (left[idx] != chain[idx]).sum() > 50
