#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE's own code.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is executed is the reference unmodified:
  * ``src/stereoimage_generation.py`` in its documented numba-less fallback (:1-8) with the two
    outside shims of SURVEY.md Appendix A (``np.float_`` alias for NumPy 2; ``sum`` accumulating in
    int64 like numba does),
  * ``src/core.py::core_generation_funnel`` (custom-depth branch) with the absent third-party modules
    (cv2, skimage, torchvision, timm, diffusers, ...) pre-seeded as MagicMock stubs.

Outputs: ``stereo_cases.npz`` (inputs + reference outputs for every case), ``funnel_cases.npz``.
Nothing is written under /root/reference.
"""
import hashlib
import json
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def load_reference_stereo():
    sys.path.insert(0, REF)
    np.float_ = np.float64                                   # NumPy-2 shim (reference uses np.float_)
    import src.stereoimage_generation as sg                  # prints the "Numba failed to import" warning
    sg.sum = lambda a: int(np.sum(a, dtype=np.int64))        # numba accumulates uint8 sums in int64
    return sg


def make_survey(H, W, seed):
    """SURVEY.md Appendix A synthetic input."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.int64)
    d = (xx * 30000) // (W - 1) + ((xx // 8 + yy // 8) % 2) * 8000
    d[H // 4: H // 2, W // 3: 2 * W // 3] = 60000
    d[(3 * H) // 4:, : W // 5] = 1000
    return img, d.astype(np.uint16)


def make_noise(H, W, seed, c=3, levels=None):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (H, W, c), dtype=np.uint8)
    if levels is None:
        dep = rng.integers(0, 65536, (H, W), dtype=np.uint16)
    else:
        dep = (rng.integers(0, levels, (H, W)) * (65535 // (levels - 1))).astype(np.uint16)
    return img, dep


def make_smooth(H, W, seed, c=3):
    """Smooth blobs + a few hard occluders: closest to a real depth map."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (H, W, c), dtype=np.uint8)
    img[:, ::7] = 0            # black columns exercise naive_interpolating's "black == gap" rule
    yy, xx = np.mgrid[0:H, 0:W]
    d = 20000 + 15000 * np.sin(xx / 9.0) * np.cos(yy / 5.0) + 40 * xx
    d[H // 3: 2 * H // 3, W // 4: W // 2] = 61000
    d[: H // 4, 3 * W // 4:] = 300
    return img, np.clip(d, 0, 65535).astype(np.uint16)


FILLS = ['none', 'naive', 'naive_interpolating', 'polylines_soft', 'polylines_sharp']
ALL_MODES = ['left-right', 'right-left', 'top-bottom', 'bottom-top', 'red-cyan-anaglyph',
             'left-only', 'only-right', 'cyan-red-reverseanaglyph']


def stereo_case_list():
    cases = []

    def add(name, gen, gargs, div, sep, bal, exp, fills, modes=('left-right', 'red-cyan-anaglyph')):
        for f in fills:
            cases.append(dict(name=f'{name}/{f}', gen=gen, gargs=list(gargs), div=div, sep=sep, bal=bal, exp=exp,
                              fill=f, modes=list(modes)))

    # SURVEY.md Appendix A rows (known-answer hashes are re-checked below)
    add('surveyA_48x64_s1_d2.5', 'survey', (48, 64, 1), 2.5, 0.0, 0.0, 1.0, FILLS)
    add('surveyA_48x64_s1_d10', 'survey', (48, 64, 1), 10.0, 1.5, 0.3, 2.0, FILLS)
    add('surveyA_96x128_s2_d5', 'survey', (96, 128, 2), 5.0, -1.0, -0.5, 1.0, FILLS)
    # white-noise depth: maximal folding, many candidates per sub-interval
    add('noise_24x80_s3', 'noise', (24, 80, 3), 9.0, 0.0, 0.0, 1.0, FILLS)
    add('noise_24x80_s4_neg', 'noise', (24, 80, 4), -14.0, 2.0, 0.2, 1.0, FILLS)
    add('noise_rgba_20x53_s5', 'noise', (20, 53, 5, 4), 7.0, -0.7, 0.0, 1.0,
        ['none', 'naive', 'polylines_soft', 'polylines_sharp'])
    # 4-level depth with power-of-two divergence_px: coincident breakpoints, exact closeness ties
    add('quant4_16x64_s6', 'noise', (16, 64, 6, 3, 4), 50.0, 0.0, 0.0, 1.0, FILLS)
    add('quant3_16x64_s7', 'noise', (16, 64, 7, 3, 3), 25.0, 0.0, -1.0, 1.0, FILLS)
    add('quant5_12x96_s8', 'noise', (12, 96, 8, 3, 5), 33.333, 3.0, 1.0, 1.0, ['polylines_soft', 'polylines_sharp'])
    # smooth depth with occluders and black columns, odd sizes, all output modes
    add('smooth_37x53_s9', 'smooth', (37, 53, 9), 6.0, 0.5, 0.0, 1.0, FILLS, ALL_MODES)
    add('smooth_30x130_s10', 'smooth', (30, 130, 10), 3.0, 0.0, 0.5, 1.5, FILLS)
    add('smooth_16x200_s11_big', 'smooth', (16, 200, 11), 30.0, 0.0, 0.0, 1.0, FILLS)
    # balance extremes: one eye is the untouched original
    add('smooth_20x40_s12_balL', 'smooth', (20, 40, 12), 5.0, 1.0, -1.0, 1.0, ['naive', 'polylines_sharp'])
    add('smooth_20x40_s12_balR', 'smooth', (20, 40, 12), 5.0, 1.0, 1.0, 1.0, ['naive', 'polylines_sharp'])
    # large separation: part of the picture leaves the frame (kept below the reference's csg capacity, :223); tiny divergence; zero divergence
    add('smooth_16x48_s13_sep', 'smooth', (16, 48, 13), 4.0, 20.0, 0.0, 1.0, FILLS)
    add('smooth_16x48_s14_tiny', 'smooth', (16, 48, 14), 0.05, 0.0, 0.0, 1.0, FILLS)
    add('smooth_16x48_s15_zero', 'smooth', (16, 48, 15), 0.0, 0.0, 0.0, 1.0, FILLS)
    # constant depth (0/0 -> NaN): only the polylines path is defined in the fallback
    add('const_8x32_s16', 'const', (8, 32, 16), 5.0, 0.0, 0.0, 1.0, ['polylines_soft', 'polylines_sharp'])
    # float64 depth input
    add('f64depth_12x40_s17', 'f64', (12, 40, 17), 8.0, 0.0, 0.0, 1.0, FILLS)
    return cases


def gen_inputs(case):
    g, a = case['gen'], case['gargs']
    if g == 'survey':
        return make_survey(*a)
    if g == 'noise':
        H, W, seed = a[:3]
        c = a[3] if len(a) > 3 else 3
        levels = a[4] if len(a) > 4 else None
        return make_noise(H, W, seed, c, levels)
    if g == 'smooth':
        return make_smooth(*a)
    if g == 'const':
        img, _ = make_noise(*a)
        return img, np.full(img.shape[:2], 1234, np.uint16)
    if g == 'f64':
        img, dep = make_smooth(*a)
        rng = np.random.default_rng(a[2])
        return img, dep.astype(np.float64) / 65535.0 + rng.random(dep.shape) * 1e-3
    raise ValueError(g)


SURVEY_A = {  # (name prefix, fill) -> (sbs sha256[:16], anaglyph sha256[:16])  -- SURVEY.md Appendix A
    ('surveyA_48x64_s1_d2.5', 'none'): ('0f03e0b21d7e4a87', 'c2f6d254fbd9afe6'),
    ('surveyA_48x64_s1_d2.5', 'naive'): ('0f03e0b21d7e4a87', 'c2f6d254fbd9afe6'),
    ('surveyA_48x64_s1_d2.5', 'naive_interpolating'): ('0f03e0b21d7e4a87', 'c2f6d254fbd9afe6'),
    ('surveyA_48x64_s1_d2.5', 'polylines_soft'): ('781553291ed1fb53', '80d9c7eb03ddcc3b'),
    ('surveyA_48x64_s1_d2.5', 'polylines_sharp'): ('226c30db9b5c0e30', 'ae3e7a631f216e97'),
    ('surveyA_48x64_s1_d10', 'none'): ('346ea2fe09fcc368', 'ab532e8e073a8f26'),
    ('surveyA_48x64_s1_d10', 'naive'): ('a3750570d7691d12', '9c46e4adf6202e8e'),
    ('surveyA_48x64_s1_d10', 'naive_interpolating'): ('d94d51a986e6fe9b', '3d2c82d0cb510cd6'),
    ('surveyA_48x64_s1_d10', 'polylines_soft'): ('f4098dfd934311bf', '78ffbec62f2e9371'),
    ('surveyA_48x64_s1_d10', 'polylines_sharp'): ('3e54843b212173b3', '383d2af0e38a8849'),
    ('surveyA_96x128_s2_d5', 'none'): ('ec80e2af00586db7', '6cbafeaec3fc8eae'),
    ('surveyA_96x128_s2_d5', 'naive'): ('c87e551cf1cac811', '46f895c6acf45322'),
    ('surveyA_96x128_s2_d5', 'naive_interpolating'): ('b0dbdd5185d2cf7f', 'fe9f3d15011cb1d6'),
    ('surveyA_96x128_s2_d5', 'polylines_soft'): ('1b51cbd6eebe1d44', '10d86e53b7d8e780'),
    ('surveyA_96x128_s2_d5', 'polylines_sharp'): ('12724e38fa30af2d', '952faa0fe82077da'),
}
h16 = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def run_stereo(sg):
    store = {}
    index = []
    for case in stereo_case_list():
        img, dep = gen_inputs(case)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            res = sg.create_stereoimages(img, dep, case['div'], case['sep'], case['modes'], case['bal'], case['exp'],
                                         case['fill'])
        outs = [np.asarray(r) for r in res]
        key = case['name'].replace('/', '__')
        for m, o in zip(case['modes'], outs):
            store[f'{key}__{m}'] = o
        prefix, fill = case['name'].split('/')
        if (prefix, fill) in SURVEY_A:
            exp = SURVEY_A[(prefix, fill)]
            got = (h16(outs[0]), h16(outs[1]))
            assert got == exp, (case['name'], got, exp)
        index.append(case)
        print('stereo', case['name'], [o.shape for o in outs], flush=True)
    store['__index__'] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'stereo_cases.npz'), **store)


def load_reference_core():
    """Import the reference's src/core.py unchanged; absent third-party modules become MagicMock stubs."""
    import importlib.machinery
    from unittest.mock import MagicMock

    def stub(name):
        m = MagicMock()
        m.__path__ = []                     # lets "import a.b" treat the stub as a package
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = m

    stub('transformers')                    # installed here, but probes torchvision on import
    for _ in range(64):
        try:
            import src.core as core
            return core
        except ModuleNotFoundError as e:
            missing = e.name
            if missing is None or missing.startswith('src'):
                raise
            parts = missing.split('.')
            for i in range(1, len(parts) + 1):      # stub the package and every parent
                name = '.'.join(parts[:i])
                if not isinstance(sys.modules.get(name), MagicMock):
                    stub(name)
            for k in [k for k in sys.modules if k == 'src.core' or k.startswith('src.depthmap')]:
                del sys.modules[k]
    raise RuntimeError('could not import the reference core')


def run_funnel(core):
    from PIL import Image
    store, index = {}, []
    img, dep = make_smooth(24, 56, 21)
    pil = Image.fromarray(img)
    dfloat = dep.astype(np.float64) / 65536.0 * 0.9
    cases = [
        ('ndarray_depth', dfloat, {'gen_stereo': True, 'stereo_fill_algo': 'polylines_sharp', 'compute_device': 'CPU'}),
        ('pil16_depth', Image.fromarray((dfloat * 65536).astype(np.uint16)),
         {'gen_stereo': True, 'stereo_fill_algo': 'naive', 'stereo_modes': ['top-bottom'], 'stereo_divergence': 4.0,
          'compute_device': 'CPU', 'unknown_key_is_ignored': 1}),
        ('pil8_depth', Image.fromarray((dfloat * 256).astype(np.uint8)),
         {'gen_stereo': True, 'stereo_fill_algo': 'polylines_soft', 'stereo_modes': ['left-right'],
          'compute_device': 'CPU', 'do_output_depth': False}),
    ]
    for name, d, opts in cases:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            got = list(core.core_generation_funnel(None, [pil.copy()], [d], None, opts))
        kinds = []
        for i, (idx, kind, res) in enumerate(got):
            kinds.append([int(idx), kind])
            store[f'{name}__{i}__{kind}'] = np.asarray(res)
        index.append(dict(name=name, opts=opts, kinds=kinds))
        if isinstance(d, np.ndarray):
            store[f'{name}__depth_in'] = d
        else:
            store[f'{name}__depth_in'] = np.asarray(d)
        print('funnel', name, kinds, flush=True)
    store['image'] = img
    store['__index__'] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'funnel_cases.npz'), **store)


if __name__ == '__main__':
    assert os.path.isdir(REF), 'the reference tree is only available in the build container'
    sg = load_reference_stereo()
    run_stereo(sg)
    core = load_reference_core()
    run_funnel(core)
    print('done')
