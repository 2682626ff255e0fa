"""GPU parity: the HIP path (through the C ABI / the drop-in Python API) against
  * the golden vectors produced by the reference's own code (tests/golden/), and
  * the CPU oracle on seeded sweeps,
bit-exact (uint8/uint16 outputs).  Nothing here reads /root/reference.
"""
import warnings

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

FILLS = ['none', 'naive', 'naive_interpolating', 'polylines_soft', 'polylines_sharp']


@pytest.fixture(scope="module")
def sg(gpu):
    import src.stereoimage_generation as sg
    return sg


@pytest.fixture(scope="module")
def native(gpu):
    import src._native as nat
    nat.lib()
    return nat


def test_library_loaded_is_in_tree(native):
    import os
    assert os.path.exists(native.LIB_PATH) and native.lib().ds_version() == 100


def test_golden_reference_vectors(sg):
    """Every case the reference itself produced (82 cases: all fills, all modes, RGBA, odd sizes, quantised depth with
    coincident breakpoints, constant depth, float64 depth, balance extremes, off-frame separation)."""
    z, index = util.load_stereo_golden()
    for case in index:
        img, dep = util.golden_inputs(case)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            res = sg.create_stereoimages(img, dep, case['div'], case['sep'], case['modes'], case['bal'], case['exp'], case['fill'])
        key = case['name'].replace('/', '__')
        for m, r in zip(case['modes'], res):
            g = z[f'{key}__{m}']
            o = np.asarray(r)
            assert g.shape == o.shape, (case['name'], m)
            assert np.array_equal(g, o), (case['name'], m, int((g != o).sum()))


def _rand_case(rng, H, W, c=3, kind='noise'):
    img = rng.integers(0, 256, (H, W, c), dtype=np.uint8)
    if kind == 'noise':
        dep = rng.integers(0, 65536, (H, W), dtype=np.uint16)
    elif kind == 'smooth':
        dep = (util.smooth_depth(H, W, int(rng.integers(1 << 30))) * 20000 + 20000).clip(0, 65535).astype(np.uint16)
    elif kind == 'quant':
        dep = (rng.integers(0, 5, (H, W)) * 16383).astype(np.uint16)
    else:
        raise ValueError(kind)
    if rng.random() < 0.5:
        img[:, rng.integers(0, W, 3)] = 0
    return img, dep


@pytest.mark.parametrize("fill", FILLS)
def test_sweep_vs_oracle(sg, oracle, fill):
    rng = np.random.default_rng(1234)
    shapes = [(5, 17), (9, 64), (7, 65), (12, 129), (6, 255), (4, 300), (3, 513), (2, 1024), (16, 96)]
    for i, (H, W) in enumerate(shapes):
        for kind in ('noise', 'smooth', 'quant'):
            c = 4 if (i % 4 == 3 and fill != 'naive_interpolating') else 3
            img, dep = _rand_case(rng, H, W, c, kind)
            div = float(rng.choice([0.3, 2.5, 5.0, -4.0, 9.0]))
            sep = float(rng.choice([0.0, 0.0, 1.0, -2.5]))
            bal = float(rng.choice([0.0, 0.0, 0.4, -0.7]))
            ex = float(rng.choice([1.0, 1.0, 1.0, 2.0, 0.5]))
            modes = ['left-right', 'red-cyan-anaglyph'] if c == 3 else ['left-right']
            want = oracle.create_stereoimages_arrays(img, dep, div, sep, modes, bal, ex, fill)
            got = sg.create_stereoimages(img, dep, div, sep, modes, bal, ex, fill)
            for m, w_, g_ in zip(modes, want, got):
                g_ = np.asarray(g_)
                assert np.array_equal(w_, g_), (fill, H, W, kind, div, sep, bal, ex, m, int((w_ != g_).sum()))


def test_large_divergence_windows(sg, oracle):
    """Window sizes that need the 2- and 4-word masks of the polylines kernel (|divergence_px| up to ~100)."""
    rng = np.random.default_rng(77)
    for W, div in ((640, 12.0), (900, 20.0), (1200, 16.0)):
        img, dep = _rand_case(rng, 6, W, 3, 'smooth')
        for fill in ('polylines_sharp', 'polylines_soft', 'naive'):
            want = oracle.create_stereoimages_arrays(img, dep, div, 0.0, ['left-right'], 0.0, 1.0, fill)[0]
            got = np.asarray(sg.create_stereoimages(img, dep, div, 0.0, ['left-right'], 0.0, 1.0, fill)[0])
            assert np.array_equal(want, got), (W, div, fill, int((want != got).sum()))


def test_exact_fallback_is_exercised(sg, native, oracle, gpu):
    """Quantised depth with power-of-two divergence_px makes coincident breakpoints and exact closeness ties:
    rows must go through the sequential fallback (and still match); smooth depth should not need it."""
    torch = gpu
    rng = np.random.default_rng(5)
    H, W = 32, 256
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    dep_q = (rng.integers(0, 4, (H, W)) * 21845).astype(np.uint16)
    it = torch.from_numpy(img).cuda().unsqueeze(0)
    res = sg.create_stereoimages_batch(it, torch.from_numpy(dep_q).cuda().unsqueeze(0), 12.5, 0.0, ['left-right'], 0.0, 1.0,
                                       'polylines_soft')
    rows_q = native.last_exact_rows(it)
    want = oracle.create_stereoimages_arrays(img, dep_q, 12.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_soft')[0]
    assert np.array_equal(want, res[0][0].cpu().numpy())
    assert rows_q > 0
    dep_s = (util.smooth_depth(H, W, 3) * 20000 + 20000).clip(0, 65535).astype(np.uint16)
    res = sg.create_stereoimages_batch(it, torch.from_numpy(dep_s).cuda().unsqueeze(0), 5.0, 0.0, ['left-right'], 0.0, 1.0,
                                       'polylines_sharp')
    rows_s = native.last_exact_rows(it)
    want = oracle.create_stereoimages_arrays(img, dep_s, 5.0, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')[0]
    assert np.array_equal(want, res[0][0].cpu().numpy())
    assert rows_s <= 2 * H // 8, rows_s


def test_float_depth_inputs(sg, oracle):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (10, 90, 3), dtype=np.uint8)
    d32 = util.smooth_depth(10, 90, 4)
    d64 = d32.astype(np.float64) + rng.random((10, 90)) * 1e-6
    for dep in (d32, d64, (d64 * 200).astype(np.uint8), (d64 * 1000).astype(np.int32)):
        for fill in ('naive', 'polylines_sharp'):
            want = oracle.create_stereoimages_arrays(img, dep, 6.0, 0.0, ['left-right'], 0.0, 1.0, fill)[0]
            got = np.asarray(sg.create_stereoimages(img, dep, 6.0, 0.0, ['left-right'], 0.0, 1.0, fill)[0])
            assert np.array_equal(want, got), (dep.dtype, fill, int((want != got).sum()))


def test_api_edge_behaviour(sg):
    img = np.zeros((4, 8, 3), np.uint8)
    dep = np.arange(32, dtype=np.uint16).reshape(4, 8)
    assert sg.create_stereoimages(img, dep, 2.5, modes=[]) == []
    with pytest.raises(Exception, match='Unknown mode'):
        sg.create_stereoimages(img, dep, 2.5, modes=['sideways'])
    with pytest.raises(AssertionError):
        sg.create_stereoimages(img, dep[:, :4], 2.5)
    out = sg.create_stereoimages(img, dep, 2.5, modes='left-only')       # non-list mode is wrapped
    assert len(out) == 1 and out[0].size == (8, 4)
    assert len(sg.create_stereoimages(img, dep, 2.5)) == 1               # default ['left-right']
    assert sg.apply_stereo_divergence(img, dep, 2.5, 0.0, 1.0, 'bogus') is None


def test_batch_equals_single(sg, gpu):
    """A batch normalises every image by its own min/max: N-batch == N single calls (byte identical)."""
    torch = gpu
    img, dep = util.survey_inputs(40, 200, 3, n=3)
    dep = dep.copy()
    dep[1] //= 2
    dep[2, :, :100] = 7
    it, dt = torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda()
    for fill in ('polylines_sharp', 'naive_interpolating'):
        b = sg.create_stereoimages_batch(it, dt, 3.0, 0.5, ['top-bottom', 'red-cyan-anaglyph', 'right-left'], 0.2, 1.0, fill)
        for i in range(3):
            s = sg.create_stereoimages_batch(it[i:i + 1], dt[i:i + 1], 3.0, 0.5, ['top-bottom', 'red-cyan-anaglyph', 'right-left'],
                                             0.2, 1.0, fill)
            for bb, ss in zip(b, s):
                assert torch.equal(bb[i], ss[0])


def test_normalmap_vs_oracle(gpu, oracle):
    import src.normalmap_generation as nm
    rng = np.random.default_rng(11)
    deps = [rng.integers(0, 65536, (33, 70), dtype=np.uint16),
            rng.integers(0, 65536, (37, 72), dtype=np.uint16),          # width % 4 == 0: the four-pixels-per-lane kernel
            rng.integers(0, 65536, (2, 4), dtype=np.uint16),
            (util.smooth_depth(50, 41, 2) * 20000 + 20000).clip(0, 65535).astype(np.uint16),
            util.survey_inputs(48, 64, 1)[1][0]]
    for dep in deps:
        for inv in (False, True):
            for args in ((None, 3, None), (None, None, None), (None, 5, None), (None, 1, None), (None, 7, None)):
                want = oracle.create_normalmap_array(dep, args[0], args[1], args[2], inv)
                got = np.asarray(nm.create_normalmap(dep, args[0], args[1], args[2], inv))
                assert np.array_equal(want, got), (dep.shape, inv, args, int((want != got).sum()))


def test_normalmap_reference_goldens(gpu):
    """The HIP normal-map kernels against outputs of the REFERENCE's own create_normalmap (tests/golden/
    make_golden_normalmap.py runs src/normalmap_generation.py:5-56 unmodified with an exact-integer cv2.Sobel stub):
    bit-exact for every Sobel aperture, np.gradient, both inversions and every input dtype; one LSB for the cases made
    with a stand-in Gaussian blur."""
    import src.normalmap_generation as nm
    z, index = util.load_normalmap_golden()
    exact = 0
    for c in index:
        d = z[c['depth'] + '__depth']
        got = np.asarray(nm.create_normalmap(d, c['pre_blur'], c['sobel'], c['post_blur'], c['invert']))
        want = z[c['key'] + '__out']
        if c['standin']:
            assert np.abs(got.astype(np.int16) - want.astype(np.int16)).max() <= 1, c['key']
        else:
            assert np.array_equal(got, want), (c['key'], int((got != want).sum()))
            exact += 1
    assert exact >= 60


def test_normalmap_blur_paths_vs_oracle(gpu, oracle):
    """Gaussian pre/post blur: parity with the oracle's restatement (OpenCV's own summation order is unpinned)."""
    import src.normalmap_generation as nm
    dep = (util.smooth_depth(40, 52, 8) * 20000 + 20000).clip(0, 65535).astype(np.uint16)
    for args in ((3, 3, None), (None, 3, 3), (5, 3, 5), (3, None, 3)):
        want = oracle.create_normalmap_array(dep, *args, False)
        got = np.asarray(nm.create_normalmap(dep, *args, False))
        assert np.array_equal(want, got), (args, int((want != got).sum()))


def test_depth_to_u16_vs_oracle(native, oracle, gpu):
    torch = gpu
    preds = np.stack([util.smooth_depth(64, 80, s) * (s + 1) - s for s in range(4)])
    preds[3] = 2.5      # flat prediction -> zeros
    t = torch.from_numpy(preds).cuda()
    for inv in (False, True):
        out, norm = native.depth_to_u16(t, inv, want_norm=True)
        for i in range(4):
            n_ref = oracle.depth_normalize01(preds[i], inv)
            assert np.array_equal(norm[i].cpu().numpy(), n_ref)
            assert np.array_equal(out[i].cpu().numpy(), oracle.convert_to_i16(n_ref))


def test_convert_to_i16_vs_oracle(gpu, oracle):
    import src.core as core
    rng = np.random.default_rng(2)
    a = np.concatenate([rng.random(5000), [0.0, 1.0, 0.99999999, 1e-12, -0.3, 1.7]])
    assert np.array_equal(core.convert_to_i16(a), oracle.convert_to_i16(a))
    a32 = a.astype(np.float32)
    assert np.array_equal(core.convert_to_i16(a32), oracle.convert_to_i16(a32))


def test_funnel_golden(gpu):
    """core_generation_funnel (custom-depth branch) against what the reference's own funnel yielded."""
    from PIL import Image
    import src.core as core
    z, index = util.load_funnel_golden()
    img = z['image']
    for case in index:
        d = z[f"{case['name']}__depth_in"]
        if case['name'].startswith('pil'):
            d = Image.fromarray(d)
        got = list(core.core_generation_funnel(None, [Image.fromarray(img)], [d], None, case['opts']))
        assert [[int(i), k] for i, k, _ in got] == case['kinds']
        for j, (idx, kind, res) in enumerate(got):
            want = z[f"{case['name']}__{j}__{kind}"]
            have = np.asarray(res)
            assert want.shape == have.shape and want.dtype == have.dtype, (case['name'], kind)
            assert np.array_equal(want, have), (case['name'], kind)
    assert core.run_depthmap is core.core_generation_funnel
    assert list(core.core_generation_funnel(None, [], None, None, {})) == []


def test_funnel_model_branch_with_registered_predictor(gpu, oracle):
    from PIL import Image
    import src.core as core
    torch = gpu
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (48, 72, 3), dtype=np.uint8)
    pred = util.smooth_depth(48, 72, 6)
    core.model_holder.register_predictor(4, lambda pil, nw, nh, dev: torch.from_numpy(pred).to(dev))
    core.model_holder.register_predictor(0, lambda pil, nw, nh, dev: torch.from_numpy(pred).to(dev))
    for mt, inv in ((4, False), (0, True)):
        got = list(core.core_generation_funnel(None, [Image.fromarray(img)], None, None,
                                               {'model_type': mt, 'gen_stereo': True, 'gen_normalmap': True,
                                                'stereo_modes': ['left-right']}))
        assert [k for _, k, _ in got] == ['depth', 'left-right', 'normalmap']
        d16 = oracle.convert_to_i16(oracle.depth_normalize01(pred, inv))
        assert np.array_equal(np.asarray(got[0][2]), d16)
        sbs = oracle.create_stereoimages_arrays(img, d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')[0]
        assert np.array_equal(np.asarray(got[1][2]), sbs)
        assert np.array_equal(np.asarray(got[2][2]), oracle.create_normalmap_array(d16))
    for opts in ({'clipdepth': True, 'clipdepth_mode': 'Range', 'clipdepth_far': 0.1, 'clipdepth_near': 0.8},
                 {'clipdepth': True, 'clipdepth_mode': 'Outliers', 'clipdepth_far': 0.02, 'clipdepth_near': 0.97},
                 {'clipdepth': True, 'clipdepth_mode': 'Outliers', 'clipdepth_far': 0.0, 'clipdepth_near': 0.6}):
        for mt, inv in ((4, False), (0, True)):                                                  # core.py:196-201
            got = list(core.core_generation_funnel(None, [Image.fromarray(img)], None, None, dict(opts, model_type=mt)))
            want = oracle.convert_to_i16(oracle.depth_postprocess(pred, inv, True, opts['clipdepth_mode'],
                                                                  opts['clipdepth_far'], opts['clipdepth_near']))
            assert [k for _, k, _ in got] == ['depth'] and np.array_equal(np.asarray(got[0][2]), want), opts
    with pytest.raises(NotImplementedError):        # midas_v21 (id 5): a family that is not built
        list(core.core_generation_funnel(None, [Image.fromarray(img)], None, None, {'model_type': 5}))
    with pytest.raises(FileNotFoundError):          # dpt_beit_large_512 (id 1) is built, but there is no checkpoint offline
        list(core.core_generation_funnel(None, [Image.fromarray(img)], None, None, {'model_type': 1}))
    with pytest.raises(NotImplementedError):
        list(core.core_generation_funnel(None, [Image.fromarray(img)], [pred], None, {'gen_rembg': True}))


def test_heatmap_vs_reference_and_oracle(gpu, oracle):
    """GEN_HEATMAP (core.py:271-274): ds_colorize_u16 + device percentiles against the reference's own colorize outputs
    (golden) and, on larger seeded inputs, against the oracle; then through the funnel."""
    import os
    from PIL import Image
    import src.core as core
    from src import heatmap
    torch = gpu
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'heatmap_cases.npz'))
    lut = heatmap.colormap_table('inferno')
    assert np.array_equal(lut, z['inferno_lut']), "this box's matplotlib builds a different inferno table than the golden run"
    for name in sorted(k[:-7] for k in z.files if k.endswith('__depth')):
        d = torch.from_numpy(z[f'{name}__depth'].copy()).cuda()
        got = heatmap.colorize_batch(d.unsqueeze(0))[0].cpu().numpy()
        assert np.array_equal(got, z[f'{name}__rgba']), name
    rng = np.random.default_rng(12)
    batch = np.stack([rng.integers(0, 65536, (301, 517), dtype=np.uint16),
                      (util.smooth_depth(301, 517, 3) * 65535).astype(np.uint16),
                      np.full((301, 517), 7, np.uint16)])
    got = heatmap.colorize_batch(torch.from_numpy(batch).cuda()).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], oracle.colorize_u16(batch[i], lut)), i
    img = rng.integers(0, 256, (48, 80, 3), dtype=np.uint8)
    dep = z['smooth__depth'].astype(np.float64) / 65536.0          # custom depth in [0,1): u16 = trunc(x*65536 + 1e-4)
    res = list(core.core_generation_funnel(None, [Image.fromarray(img)], [dep], None,
                                           {'gen_heatmap': True, 'do_output_depth': False}))
    assert [k for _, k, _ in res] == ['heatmap'] and res[0][2].mode == 'RGBA'
    assert np.array_equal(np.asarray(res[0][2]), oracle.colorize_u16(oracle.convert_to_i16(dep), lut))


def test_funnel_simple_mesh(gpu, tmp_path):
    """GEN_SIMPLE_MESH (core.py:277-306) through the funnel with a custom depth map: the geometry computed on the device
    equals the same tensor ops on the CPU, which tests/test_host_logic.py pins to the reference's own functions."""
    import os
    from PIL import Image
    import src.core as core
    from src import mesh_generation as mg
    torch = gpu
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mesh_cases.npz'))
    dep = z['c__depth'].astype(np.float64) / 4.0
    img = np.random.default_rng(2).integers(0, 256, (24, 24, 3), dtype=np.uint8)
    res = list(core.core_generation_funnel(str(tmp_path), [Image.fromarray(img)], [dep], None,
                                           {'gen_simple_mesh': True, 'do_output_depth': False}))
    assert [k for _, k, _ in res] == ['simple_mesh'] and os.path.exists(res[0][2])
    d = torch.from_numpy(dep)
    v, f, c = mg.create_mesh_arrays(torch.from_numpy(img), mg.mesh_depth(d, -1, False, True), keep_edges=False)
    want = mg.write_obj(str(tmp_path / 'cpu.obj'), v.numpy(), f.numpy(), c.numpy())
    assert open(res[0][2]).read() == open(want).read()
    v2 = mg.create_mesh_arrays(torch.from_numpy(img).cuda(), mg.mesh_depth(d.cuda(), -1, False, True), keep_edges=True, spherical=True)[0]
    v2c = mg.create_mesh_arrays(torch.from_numpy(img), mg.mesh_depth(d, -1, False, True), keep_edges=True, spherical=True)[0]
    assert np.allclose(v2.cpu().numpy(), v2c.numpy(), rtol=0, atol=1e-12)


def test_full_size_properties(sg, native, oracle, gpu):
    """BASELINE size (1024x1024): size-independent properties on the whole batch, plus the oracle on a subset of rows
    (row 500 is made to hold the image's min and max so the subset normalises like the full image)."""
    torch = gpu
    img, dep = util.survey_inputs(1024, 1024, 0, n=2)
    dep = dep.copy()
    dep[:, 500, 0] = 0
    dep[:, 500, 1] = 65535
    it, dt = torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda()
    sbs, ana = sg.create_stereoimages_batch(it, dt, 2.5, 0.0, ['left-right', 'red-cyan-anaglyph'], 0.0, 1.0, 'polylines_sharp')
    assert tuple(sbs.shape) == (2, 1024, 2048, 3) and tuple(ana.shape) == (2, 1024, 1024, 3)
    s = sbs.cpu().numpy()
    a = ana.cpu().numpy()
    # anaglyph = red of the left half + green/blue of the right half
    assert np.array_equal(a[..., 0], s[:, :, :1024, 0]) and np.array_equal(a[..., 1:], s[:, :, 1024:, 1:])
    # zero divergence: the scatter kernels move nothing, both eyes are the picture itself
    for fill in ('none', 'naive', 'naive_interpolating'):
        z = sg.create_stereoimages_batch(it, dt, 0.0, 0.0, ['left-right'], 0.0, 1.0, fill)[0].cpu().numpy()
        assert np.array_equal(z[:, :, :1024], img) and np.array_equal(z[:, :, 1024:], img)
    # balance -1 / +1: one eye is the untouched original, the other carries the whole divergence
    zl = sg.create_stereoimages_batch(it, dt, 2.5, 0.0, ['left-right'], -1.0, 1.0, 'polylines_sharp')[0].cpu().numpy()
    assert np.array_equal(zl[:, :, :1024], img) and not np.array_equal(zl[:, :, 1024:], img)
    rows = [0, 255, 256, 500, 511, 777, 1023]
    for fill in ('polylines_sharp', 'polylines_soft', 'naive', 'naive_interpolating', 'none'):
        got = sg.create_stereoimages_batch(it, dt, 2.5, 0.0, ['left-right'], 0.0, 1.0, fill)[0].cpu().numpy()
        for i in range(2):
            want = oracle.create_stereoimages_arrays(img[i, rows], dep[i, rows], 2.5, 0.0, ['left-right'], 0.0, 1.0, fill)[0]
            assert np.array_equal(want, got[i, rows]), (fill, i, int((want != got[i, rows]).sum()))
    # normal map at full size against the oracle (cheap on the CPU)
    import src.normalmap_generation as nm
    nmap = nm.create_normalmap_batch(dt).cpu().numpy()
    assert np.array_equal(nmap[0], oracle.create_normalmap_array(dep[0]))


def test_normalmap_accepts_any_real_dtype(gpu, oracle):
    """create_normalmap takes any real array like the reference (src/normalmap_generation.py:20-21 promote to float64):
    integer / float64 / float32 depth through ds_normalmap_f64 against the oracle's float64 restatement, every gradient
    mode; float32 + np.gradient is a float32 pipeline in the reference and has a kernel of its own (round 5): against the oracle
    run on the float32 array itself; float32 + np.gradient + a blur is refused, as are non-real / non-2-D inputs."""
    import src.normalmap_generation as nm
    import src._native as nat
    rng = np.random.default_rng(21)
    base = (util.smooth_depth(37, 45, 4) * 20000 + 20000).clip(0, 65535)
    cases = [base.astype(np.uint8), base.astype(np.int32), base.astype(np.int64), (base / 65535.0).astype(np.float64),
             (base * 1.7 - 300.0).astype(np.float32), rng.standard_normal((37, 45)) * 1000.0, base.astype(np.int16)]
    for dep in cases:
        for inv in (False, True):
            for args in ((None, 3, None), (None, 5, None), (3, 3, 3), (None, None, None)):
                if dep.dtype == np.float32 and args[1] is None:
                    want = oracle.create_normalmap_array(dep, args[0], args[1], args[2], inv)          # float32 throughout
                    assert np.array_equal(want, np.asarray(nm.create_normalmap(dep, *args, inv))), (inv, args)
                    continue
                want = oracle.create_normalmap_array(dep.astype(np.float64), args[0], args[1], args[2], inv)
                got = np.asarray(nm.create_normalmap(dep, args[0], args[1], args[2], inv))
                assert np.array_equal(want, got), (dep.dtype, inv, args, int((want != got).sum()))
    # a float64 array holding uint16 codes gives what the uint16 fused kernel gives (the exact-representability argument)
    d16 = base.astype(np.uint16)
    assert np.array_equal(np.asarray(nm.create_normalmap(d16.astype(np.float64))), np.asarray(nm.create_normalmap(d16)))
    for bad in (np.zeros((4, 4), np.complex64), np.zeros((4, 4, 3), np.uint16)):
        with pytest.raises(nat.DepthStereoError):
            nm.create_normalmap(bad)


def _full_frame_depths():
    """Three depth regimes at the benchmark's size, as uint16 [3, 1024, 1024]:
      0  the survey pattern (ramps, periodic steps, large occluders)
      1  noisy: a smooth field + per-pixel noise of a few hundred codes, what a random-weight network's prediction looks
         like to the stereo kernel (> 10^5 'general' pixels per image: backward segments everywhere)
      2  adversarial: white-noise columns, quantised plateaus (four levels) and single-pixel spikes; at a divergence whose
         pixel value is an integer the plateaus give coincident breakpoints -- rows whose sweep is history
         dependent and must go through the exact fallback (test_full_frame_1024_exact_fallback_rows)"""
    rng = np.random.default_rng(77)
    H = W = 1024
    d0 = util.survey_inputs(H, W, 0)[1][0]
    smooth = util.smooth_depth(H, W, 5).astype(np.float64)
    d1 = (smooth * 18000 + 20000 + rng.normal(0, 300, (H, W))).clip(0, 65535).astype(np.uint16)
    d2 = (rng.integers(0, 4, (H, W)) * 21845).astype(np.uint16)
    d2[:, 300:420] = rng.integers(0, 65536, (H, 120), dtype=np.uint16)
    d2[::7, ::13] = 65535
    d2[200:260] = (smooth[200:260] * 30000 + 20000).clip(0, 65535).astype(np.uint16)
    return np.stack([d0, d1, d2])


@pytest.mark.parametrize("fill", ["polylines_sharp", "polylines_soft"])
def test_full_frame_1024_all_rows_vs_oracle(sg, native, oracle, gpu, fill):
    """BASELINE's size, EVERY row, both eyes, bit-exact against the oracle (which reproduces the reference-made goldens),
    on the three regimes of _full_frame_depths -- including the bench's own regime (noisy network depth) and rows that
    take the exact fallback.  src/stereoimage_generation.py:162-283."""
    torch = gpu
    dep = _full_frame_depths()
    img = np.random.default_rng(78).integers(0, 256, (3, 1024, 1024, 3), dtype=np.uint8)
    it, dt = torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda()
    got = sg.create_stereoimages_batch(it, dt, 2.5, 0.0, ['left-right'], 0.0, 1.0, fill)[0].cpu().numpy()
    exact_rows, general = native.last_stats(it)
    for i in range(3):
        want = oracle.create_stereoimages_arrays(img[i], dep[i], 2.5, 0.0, ['left-right'], 0.0, 1.0, fill)[0]
        bad = int((want != got[i]).sum())
        assert bad == 0, (fill, i, bad, np.argwhere((want != got[i]).any(axis=2))[:5].tolist())
    # the batch as a whole produced general pixels; regime 1 alone is dominated by them.  (At 2.5 % = 25.6 px the quantised
    # plateaus of regime 2 never put two points on the same x -- its levels are 25.6 / 3 px apart -- so NO row of this call
    # needs the exact fallback: measured 0 of 6144.  The fallback at full size is the next test's subject.)
    assert general > 0 and exact_rows >= 0
    sg.create_stereoimages_batch(it[1:2], dt[1:2], 2.5, 0.0, ['left-right'], 0.0, 1.0, fill)
    _, g1 = native.last_stats(it)
    assert g1 > 0


@pytest.mark.parametrize("fill", ["polylines_sharp", "polylines_soft"])
def test_full_frame_1024_exact_fallback_rows(sg, native, oracle, gpu, fill):
    """Rows that MUST take the sequential exact sweep, at the benchmark's size: divergence 3.125 % of 1024 px is exactly
    32 px, and regime 2's depth has four levels 1/3 apart -- pixels 32 columns apart on levels 0 and 3 land on the SAME x
    (coincident breakpoints: zero-length sub-intervals, exact closeness ties), which is history dependent in the reference's
    sweep.  The call must flag rows (exact_rows > 0) and still be bit-exact on every row of both eyes."""
    torch = gpu
    dep = _full_frame_depths()[2:3]
    img = np.random.default_rng(80).integers(0, 256, (1, 1024, 1024, 3), dtype=np.uint8)
    it, dt = torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda()
    got = sg.create_stereoimages_batch(it, dt, 3.125, 0.0, ['left-right'], 0.0, 1.0, fill)[0].cpu().numpy()
    exact_rows, _ = native.last_stats(it)
    want = oracle.create_stereoimages_arrays(img[0], dep[0], 3.125, 0.0, ['left-right'], 0.0, 1.0, fill)[0]
    bad = int((want != got[0]).sum())
    assert bad == 0, (fill, bad, np.argwhere((want != got[0]).any(axis=2))[:5].tolist())
    assert exact_rows > 0, "no row took the exact fallback: the regime does not test what it claims"


@pytest.mark.parametrize("fill,div,sep,bal,exp", [("polylines_sharp", 4.0, 1.0, 0.3, 2.0), ("polylines_soft", 2.5, -1.5, -0.6, 0.5)])
def test_full_frame_1024_non_default_parameters(sg, native, oracle, gpu, fill, div, sep, bal, exp):
    """The same three regimes at 1024 x 1024 with every stereo parameter off its default: separation (a constant shift of
    both eyes), stereo balance (unequal divergence per eye), and an exponent != 1 (the pow LUT of :84-85 / :176), every row
    of both eyes bit-exact against the oracle.  src/stereoimage_generation.py:13-92,162-283."""
    torch = gpu
    dep = _full_frame_depths()
    img = np.random.default_rng(79).integers(0, 256, (3, 1024, 1024, 3), dtype=np.uint8)
    it, dt = torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda()
    got = sg.create_stereoimages_batch(it, dt, div, sep, ['left-right'], bal, exp, fill)[0].cpu().numpy()
    for i in range(3):
        want = oracle.create_stereoimages_arrays(img[i], dep[i], div, sep, ['left-right'], bal, exp, fill)[0]
        bad = int((want != got[i]).sum())
        assert bad == 0, (fill, i, bad, np.argwhere((want != got[i]).any(axis=2))[:5].tolist())


def test_funnel_batched_schedule_equals_image_by_image(gpu, oracle, monkeypatch):
    """The funnel groups consecutive same-size images into device batches and pipelines the groups; the yielded triples
    must be exactly what one funnel call per image yields (and what the oracle says), in the reference's order
    (src/core.py:133-306): mixed sizes and modes, a flat prediction in the middle of a batch (zeros, no
    'depth_prediction'), every post-processing branch, custom depth maps, and groups split by the pixel budget."""
    from PIL import Image
    import src.core as core
    torch = gpu
    rng = np.random.default_rng(31)
    sizes = [(48, 72), (48, 72), (48, 72), (40, 64), (40, 64), (48, 72), (48, 72)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    preds = [util.smooth_depth(h, w, 40 + i) * (i + 1) - i for i, (h, w) in enumerate(sizes)]
    preds[1][:] = 3.25                                       # flat: broken prediction inside a batch
    pils = [Image.fromarray(a) for a in imgs]
    pils[5] = pils[5].convert("RGBA")                        # another mode: its own group, 4-channel stereo
    lookup = {id(p): preds[i] for i, p in enumerate(pils)}

    class Pred:
        calls = []

        def __call__(self, pil, nw, nh, dev):
            return torch.from_numpy(lookup[id(pil)]).to(dev)

        def batch(self, plist, nw, nh, dev):
            Pred.calls.append(len(plist))
            return torch.stack([torch.from_numpy(lookup[id(p)]) for p in plist]).to(dev)

    core.model_holder.register_predictor(4, Pred())
    core.model_holder.register_predictor(0, Pred())
    option_sets = [
        {'gen_stereo': True, 'gen_normalmap': True, 'stereo_modes': ['left-right', 'red-cyan-anaglyph']},
        {'do_output_depth_prediction': True, 'gen_stereo': True, 'stereo_modes': ['top-bottom']},
        {'clipdepth': True, 'clipdepth_mode': 'Range', 'clipdepth_far': 0.1, 'clipdepth_near': 0.8, 'gen_normalmap': True},
        {'clipdepth': True, 'clipdepth_mode': 'Outliers', 'clipdepth_far': 0.02, 'clipdepth_near': 0.97, 'output_depth_invert': True},
        {'output_depth_combine': True, 'gen_heatmap': True},
    ]
    for mt in (4, 0):
        for opts in option_sets:
            o = dict(opts, model_type=mt)
            Pred.calls.clear()
            got = list(core.core_generation_funnel(None, list(pils), None, None, o))
            assert Pred.calls == [3, 2, 1, 1], Pred.calls           # 3 x 48x72 | 2 x 40x64 | RGBA | RGB
            want = []
            for i, p in enumerate(pils):
                want += [(i, k, r) for _, k, r in core.core_generation_funnel(None, [p], None, None, o)]
            assert [(i, k) for i, k, _ in got] == [(i, k) for i, k, _ in want], opts
            for (i, k, a), (_, _, b) in zip(got, want):
                assert np.array_equal(np.asarray(a), np.asarray(b)), (mt, opts, i, k)
            if 'do_output_depth_prediction' in opts:
                assert (1, 'depth_prediction') not in [(i, k) for i, k, _ in got]
                d1 = [r for i, k, r in got if i == 1 and k == 'depth'][0]
                assert not np.asarray(d1).any()
    # the oracle on one option set (the per-image runs above could share a mistake with the batched ones)
    got = list(core.core_generation_funnel(None, list(pils[:3]), None, None, dict(option_sets[0], model_type=0)))
    for i in (0, 2):
        d16 = oracle.convert_to_i16(oracle.depth_normalize01(preds[i], True))
        r = {k: np.asarray(x) for j, k, x in got if j == i}
        assert np.array_equal(r['depth'], d16)
        sbs, ana = oracle.create_stereoimages_arrays(imgs[i], d16, 2.5, 0.0, ['left-right', 'red-cyan-anaglyph'], 0.0, 1.0, 'polylines_sharp')
        assert np.array_equal(r['left-right'], sbs) and np.array_equal(r['red-cyan-anaglyph'], ana)
        assert np.array_equal(r['normalmap'], oracle.create_normalmap_array(d16))
    # the pixel budget splits a run of same-size images into several pipelined groups, results unchanged
    monkeypatch.setattr(core, "FUNNEL_BATCH_PIXELS", 48 * 72)
    Pred.calls.clear()
    one = list(core.core_generation_funnel(None, list(pils[:3]), None, None, dict(option_sets[0], model_type=0)))
    assert Pred.calls == [1, 1, 1]
    for (i, k, a), (j, l, b) in zip(one, got):
        assert (i, k) == (j, l) and np.array_equal(np.asarray(a), np.asarray(b))
    # custom depth maps are batched too
    deps = [rng.random((48, 72)) for _ in range(3)]
    res = list(core.core_generation_funnel(None, list(pils[:3]), deps, None, {'gen_stereo': True, 'stereo_modes': ['left-right']}))
    assert [(i, k) for i, k, _ in res] == [(0, 'depth'), (0, 'left-right'), (1, 'depth'), (1, 'left-right'), (2, 'depth'), (2, 'left-right')]
    for i in range(3):
        d16 = oracle.convert_to_i16(deps[i])
        assert np.array_equal(np.asarray(res[2 * i][2]), d16)
        assert np.array_equal(np.asarray(res[2 * i + 1][2]),
                              oracle.create_stereoimages_arrays(imgs[i], d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')[0])


def test_funnel_failure_in_a_later_group_keeps_earlier_results(gpu):
    """The pipelined funnel enqueues group k+1 before group k is handed out; a failure while preparing k+1 must not swallow
    group k, and a single-channel image with GEN_STEREO fails where the reference fails: inside its stereo step, AFTER the
    image's own 'depth' was yielded (src/core.py:133-259, src/stereoimage_generation.py:55)."""
    from PIL import Image
    import src.core as core
    rng = np.random.default_rng(41)
    rgb = [Image.fromarray(rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)) for _ in range(2)]
    gray = Image.fromarray(rng.integers(0, 256, (24, 40), dtype=np.uint8))            # mode 'L': np.asarray(...).ndim == 2
    deps = [Image.fromarray(rng.integers(0, 65536, (24, 40), dtype=np.uint16)) for _ in range(3)]
    got = []
    with pytest.raises(ValueError, match='not enough values to unpack'):
        for item in core.core_generation_funnel(None, rgb + [gray], deps, None, {'gen_stereo': True, 'stereo_modes': ['left-right']}):
            got.append((item[0], item[1]))
    assert got == [(0, 'depth'), (0, 'left-right'), (1, 'depth'), (1, 'left-right'), (2, 'depth')], got

    class Boom:                                                   # a depth source that fails when it is turned into an array
        size, mode, height, width = (40, 24), 'I;16', 24, 40

        def __array__(self, *a, **k):
            raise RuntimeError('decode failed')

    got = []
    with pytest.raises(Exception):
        for item in core.core_generation_funnel(None, rgb + [Image.fromarray(rng.integers(0, 256, (30, 44, 3), dtype=np.uint8))],
                                                deps[:2] + [Boom()], None, {'gen_stereo': True, 'stereo_modes': ['left-right']}):
            got.append((item[0], item[1]))
    assert got == [(0, 'depth'), (0, 'left-right'), (1, 'depth'), (1, 'left-right')], got


def test_exact_fallback_lds_kernel_equals_global_kernel(sg, native, oracle, gpu):
    """The two implementations of the exact row sweep -- one workgroup per flagged row with the row's arrays in LDS and a
    parallel stable sort (the default when the row fits), one lane per row against global scratch (wider rows,
    DS_PL_EXACT_GLOBAL=1) -- on inputs that flag rows: quantised depth at a power-of-two divergence (coincident breakpoints,
    exact ties; both fills, both eyes unbalanced), a constant depth map (NaN coordinates: the sort must leave them in place),
    float32 depth, and a row too wide for the LDS kernel.  Byte-identical to each other and to the oracle."""
    import os
    torch = gpu
    rng = np.random.default_rng(15)
    cases = []
    for (h, w, div, bal, fill) in [(24, 512, 6.25, 0.0, 'polylines_sharp'), (24, 512, 6.25, 0.4, 'polylines_soft'), (8, 2048, 1.5625, -0.5, 'polylines_sharp')]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        dep = (rng.integers(0, 4, (h, w)) * 21845).astype(np.uint16)
        cases.append((img, dep, div, bal, fill))
    img = rng.integers(0, 256, (6, 320, 3), dtype=np.uint8)
    cases.append((img, np.full((6, 320), 1234, np.uint16), 5.0, 0.0, 'polylines_sharp'))                    # constant: NaN
    cases.append((img, (rng.integers(0, 3, (6, 320)) * 0.5).astype(np.float32), 10.0, 0.0, 'polylines_soft'))   # float32, 32 px
    wide = rng.integers(0, 256, (2, 5200, 3), dtype=np.uint8)
    cases.append((wide, (rng.integers(0, 4, (2, 5200)) * 21845).astype(np.uint16), 100.0 * 32 / 5200, 0.0, 'polylines_sharp'))   # 187 KB of row arrays: global
    # round 5: the compact LDS image (36 bytes per column) takes a 3840-column row -- Boost on a 4K frame, BASELINE config 4, whose
    # flagged rows cost 143 ms per launch in the one-lane-per-row kernel -- and a wide divergence (128 px: an active set of hundreds)
    uhd = rng.integers(0, 256, (3, 3840, 3), dtype=np.uint8)
    cases.append((uhd, (rng.integers(0, 4, (3, 3840)) * 21845).astype(np.uint16), 100.0 * 64 / 3840, 0.3, 'polylines_sharp'))
    cases.append((uhd, (rng.integers(0, 5, (3, 3840)) * 16383).astype(np.uint16), 100.0 * 32 / 3840, 0.0, 'polylines_soft'))
    img1k = rng.integers(0, 256, (4, 1024, 3), dtype=np.uint8)
    cases.append((img1k, (rng.integers(0, 8, (4, 1024)) * 9362).astype(np.uint16), 12.5, -0.2, 'polylines_sharp'))
    # active sets of several 64-entry chunks with removals all over them (the cooperative sweep's permutation of the keep flags)
    img512 = rng.integers(0, 256, (6, 512, 3), dtype=np.uint8)
    cases.append((img512, (rng.integers(0, 16, (6, 512)) * 4369).astype(np.uint16), 25.0, 0.0, 'polylines_sharp'))
    cases.append((img512, (rng.integers(0, 16, (6, 512)) * 4369).astype(np.uint16), 25.0, 0.5, 'polylines_soft'))
    cases.append((img512[:, :333], (rng.integers(0, 6, (6, 333)) * 13107).astype(np.uint16), -18.0, 0.0, 'polylines_sharp'))     # odd width, negative divergence
    # round 6 (speculative chunks): rows that are smooth almost everywhere -- every chunk finds a column covered by one segment and its
    # assumption holds -- with a few quantised stretches that flag them; 1080p width at config 5's divergence, and a 1024-column frame
    for (h, w, div) in ((4, 1920, 2.5), (4, 1024, 3.125)):
        yy, xx = np.mgrid[0:h, 0:w]
        smooth = (30000 + 20000 * np.sin(xx / 97.0 + yy) + rng.integers(0, 40, (h, w))).astype(np.uint16)
        for x0 in (w // 7, w // 2, (5 * w) // 6):
            smooth[:, x0:x0 + 48] = (rng.integers(0, 4, (h, 48)) * 21845).astype(np.uint16)
        cases.append((rng.integers(0, 256, (h, w, 3), dtype=np.uint8), smooth, div, 0.0, 'polylines_sharp'))
    flagged = 0
    old = os.environ.get("DS_PL_EXACT_GLOBAL")
    try:
        for img, dep, div, bal, fill in cases:
            want = oracle.create_stereoimages_arrays(img, dep, div, 0.0, ['left-right'], bal, 1.0, fill)[0]
            it, dt = torch.from_numpy(img).cuda().unsqueeze(0), torch.from_numpy(dep).cuda().unsqueeze(0)
            outs = []
            # the LDS kernel with the sweep run by the whole wave (round 5, the default), with the sweep on one lane, the global kernel
            # (DS_PL_EXACT_COOP_MIN: the active-set size from which the cooperative sweep goes parallel -- default 2; 24 mixes the
            # sequential and the parallel scans inside one row)
            for force, coop, cmin in (("0", "1", None), ("0", "1", "24"), ("0", "0", None), ("1", "1", None)):
                os.environ["DS_PL_EXACT_GLOBAL"] = force
                os.environ["DS_PL_EXACT_COOP"] = coop
                if cmin is None:
                    os.environ.pop("DS_PL_EXACT_COOP_MIN", None)
                else:
                    os.environ["DS_PL_EXACT_COOP_MIN"] = cmin
                outs.append(sg.create_stereoimages_batch(it, dt, div, 0.0, ['left-right'], bal, 1.0, fill)[0][0].cpu().numpy())
                flagged += native.last_exact_rows(it)
            # round 6: the default runs the row as speculative chunks (k_polylines_exact_chunked); one chunk, three chunks, and every
            # assumption made wrong on purpose (DS_PL_EXACT_BREAK: each chunk is swept again from the true state) give the same bytes
            os.environ["DS_PL_EXACT_GLOBAL"], os.environ["DS_PL_EXACT_COOP"] = "0", "1"
            os.environ.pop("DS_PL_EXACT_COOP_MIN", None)
            for chunks, brk in (("1", None), ("3", None), ("8", "1")):
                os.environ["DS_PL_EXACT_CHUNKS"] = chunks
                if brk is None:
                    os.environ.pop("DS_PL_EXACT_BREAK", None)
                else:
                    os.environ["DS_PL_EXACT_BREAK"] = brk
                got = sg.create_stereoimages_batch(it, dt, div, 0.0, ['left-right'], bal, 1.0, fill)[0][0].cpu().numpy()
                assert np.array_equal(outs[0], got), (img.shape, fill, chunks, brk, 'chunked exact sweep differs', int((outs[0] != got).sum()))
            os.environ.pop("DS_PL_EXACT_CHUNKS", None)
            os.environ.pop("DS_PL_EXACT_BREAK", None)
            assert np.array_equal(outs[0], outs[1]), (img.shape, fill, 'cooperative sweeps with different parallel thresholds differ')
            assert np.array_equal(outs[0], outs[2]), (img.shape, fill, 'cooperative and one-lane sweeps of the LDS exact kernel differ')
            assert np.array_equal(outs[0], outs[3]), (img.shape, fill, 'LDS and global exact kernels differ')
            assert np.array_equal(outs[0], want), (img.shape, fill, int((outs[0] != want).sum()))
    finally:
        os.environ.pop("DS_PL_EXACT_COOP", None)
        os.environ.pop("DS_PL_EXACT_COOP_MIN", None)
        os.environ.pop("DS_PL_EXACT_CHUNKS", None)
        os.environ.pop("DS_PL_EXACT_BREAK", None)
        if old is None:
            os.environ.pop("DS_PL_EXACT_GLOBAL", None)
        else:
            os.environ["DS_PL_EXACT_GLOBAL"] = old
    assert flagged > 0


def test_general_pixels_one_window_per_wave_equals_private_windows(sg, native, oracle, gpu, monkeypatch):
    """The second pass renders 8 queued general pixels per wave; when they sit in one row it builds ONE source window for
    all of them (the default) instead of eight private ones (DS_PL_GEN_SHARED=0).  Both must give the same bytes, and the
    oracle's: noisy depth (general pixels everywhere, waves inside one row), narrow images (a wave's 8 entries straddle
    rows: the private-window path inside the default), both fills, unbalanced eyes, float32 depth, an image border in
    every window (w = 40), and a wide divergence (four 64-segment candidate words); the kernel specialised for one-word
    candidate masks (the default wherever the window has at most 64 segments) against the general one."""
    torch = gpu
    rng = np.random.default_rng(91)
    cases = []
    for (n, h, w, div, bal, fill, dt) in [(2, 96, 768, 4.0, 0.0, 'polylines_sharp', np.uint16), (2, 64, 640, 3.0, 0.35, 'polylines_soft', np.uint16),
                                          (3, 200, 40, 12.0, 0.0, 'polylines_sharp', np.uint16), (2, 120, 72, 9.0, -0.4, 'polylines_soft', np.float32),
                                          (1, 48, 1920, 2.5, 0.0, 'polylines_sharp', np.uint16), (1, 16, 2048, 9.0, 0.0, 'polylines_sharp', np.uint16)]:
        img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        base = rng.integers(0, 65536, (n, h, w))
        smooth = (np.linspace(0, 50000, w)[None, None, :] + rng.integers(0, 9000, (n, h, w))).clip(0, 65535)
        dep = np.where(rng.random((n, h, w)) < 0.5, base, smooth)
        dep = dep.astype(np.uint16) if dt is np.uint16 else (dep / 65535.0).astype(np.float32)
        cases.append((img, dep, div, bal, fill))
    total_general = 0
    for img, dep, div, bal, fill in cases:
        it, dt = torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda()
        outs = []
        for sw, kw in (("1", "1"), ("0", "1"), ("1", "4")):      # DS_PL_GEN_KW=4: the four-word candidate masks even where one word holds them
            monkeypatch.setenv("DS_PL_GEN_SHARED", sw)
            monkeypatch.setenv("DS_PL_GEN_KW", kw)
            outs.append(sg.create_stereoimages_batch(it, dt, div, 0.0, ['left-right'], bal, 1.0, fill)[0].cpu().numpy())
            total_general += native.last_stats(it)[1]
        assert np.array_equal(outs[0], outs[1]), (img.shape, fill, 'shared and private windows differ', int((outs[0] != outs[1]).sum()))
        assert np.array_equal(outs[0], outs[2]), (img.shape, fill, 'one-word and four-word candidate masks differ', int((outs[0] != outs[2]).sum()))
        for i in range(img.shape[0]):
            want = oracle.create_stereoimages_arrays(img[i], dep[i], div, 0.0, ['left-right'], bal, 1.0, fill)[0]
            assert np.array_equal(outs[0][i], want), (img.shape, fill, i, int((outs[0][i] != want).sum()))
    assert total_general > 10000


def test_funnel_reference_f16_postproc(gpu, oracle):
    """`reference_f16_postproc` (defined corners): the funnel's depth map of a float16 prediction under the
    reference's NumPy-1.x promotion rules (src/core.py:189-211, :44-50 fed by src/depthmap_generation.py:484-497), bit for
    bit against oracle.depth_postprocess_f16_numpy1 -- plain, inverted, 'Range' clip, a flat (broken) prediction in the batch,
    and the depth_prediction result (a float16 array, like the reference's).  Off (the default): float32 post-processing."""
    import torch
    from PIL import Image
    import src.core as core
    rng = np.random.default_rng(11)
    imgs = [rng.integers(0, 256, (48, 72, 3), dtype=np.uint8) for _ in range(3)]
    preds = [(util.smooth_depth(48, 72, 6 + i) * 41.0 + 2.0).astype(np.float16) for i in range(3)]
    preds[1][:] = np.float16(3.5)                                   # a flat prediction: zeros (:203-206)
    pils = [Image.fromarray(a) for a in imgs]
    lookup = {id(p): a for p, a in zip(pils, preds)}

    class Pred:
        returns_f16 = True

        def __call__(self, pil, nw, nh, dev):
            return self.batch([pil], nw, nh, dev)[0]

        def batch(self, plist, nw, nh, dev):
            return torch.stack([torch.from_numpy(lookup[id(p)].astype(np.float32)) for p in plist]).to(dev)

    for mt, inv in ((4, False), (0, True)):
        core.model_holder.register_predictor(mt, Pred())
        for opts in ({}, {'clipdepth': True, 'clipdepth_mode': 'Range', 'clipdepth_far': 0.1, 'clipdepth_near': 0.8},
                     {'do_output_depth_prediction': True}):
            o = dict(opts, model_type=mt)
            got = list(core.core_generation_funnel(None, list(pils), None, None, o, {'reference_f16_postproc': True}))
            depth = [np.asarray(r) for _, k, r in got if k == 'depth']
            assert len(depth) == 3
            for j in range(3):
                want = oracle.depth_postprocess_f16_numpy1(preds[j], inv, bool(opts.get('clipdepth')), opts.get('clipdepth_far', 0.0),
                                                           opts.get('clipdepth_near', 1.0))
                assert np.array_equal(depth[j], want), (mt, opts, j)
            if 'do_output_depth_prediction' in opts:
                pp = [(i, r) for i, k, r in got if k == 'depth_prediction']
                assert [i for i, _ in pp] == [0, 2] and all(r.dtype == np.float16 for _, r in pp)
                assert np.array_equal(pp[0][1], -preds[0] if inv else preds[0])
            # default (switch off): float32 post-processing of the same values
            got32 = list(core.core_generation_funnel(None, [pils[0]], None, None, o, {'reference_f16_postproc': False}))
            d32 = [np.asarray(r) for _, k, r in got32 if k == 'depth'][0]
            p32 = preds[0].astype(np.float32)
            want32 = oracle.convert_to_i16(oracle.depth_postprocess(p32, inv, True, 'Range', opts['clipdepth_far'], opts['clipdepth_near'])
                                           if opts.get('clipdepth') else oracle.depth_normalize01(p32, inv))
            assert np.array_equal(d32, want32), (mt, opts)
    core.model_holder.update_settings(reference_f16_postproc=None)


def test_normalmap_float32_gradient_is_the_reference_float32_evaluation(gpu, oracle):
    """create_normalmap(float32 depth, sobel_gradient=None): the reference evaluates this combination in FLOAT32 end to end
    (src/normalmap_generation.py:20-21,31,34-39,51-54); ds_normalmap_gradient_f32 must reproduce numpy's float32 arithmetic bit
    for bit -- against the oracle's restatement (itself pinned by the reference-made goldens, tests/golden/normalmap_cases.npz:
    the f32* cases) on larger frames than the goldens hold, both inversions, incl. flat areas and a 2 x 2 image."""
    import src.normalmap_generation as nm
    rng = np.random.default_rng(77)
    cases = [rng.normal(0.0, 900.0, (257, 381)).astype(np.float32), (rng.random((64, 1030)) * 37 + 5).astype(np.float32),
             np.zeros((5, 7), np.float32), rng.normal(0, 1e-3, (2, 2)).astype(np.float32),
             (rng.integers(0, 65536, (130, 90)).astype(np.float32))]
    cases[1][10:30, 100:600] += 11.5
    for d in cases:
        for inv in (False, True):
            for sob in (None, 0, -1):
                want = oracle.create_normalmap_array(d, None, sob, None, inv)
                got = np.asarray(nm.create_normalmap(d, None, sob, None, inv))
                assert got.dtype == np.uint8 and np.array_equal(got, want), (d.shape, inv, sob, int((got != want).sum()))
    # round 6: float32 + np.gradient + Gaussian blurs (cv2.GaussianBlur on CV_32F data: float32 coefficients and sums; the oracle's
    # restatement of the stand-in arithmetic, bit for bit) and float16 + np.gradient (numpy's float16 evaluation, pinned by the
    # reference-made f16* goldens) -- every input combination the reference accepts now has a device path
    for d in cases[:2]:
        for args in ((3, None, None), (None, 0, 5), (5, None, 3), (7, -1, 7)):
            for inv in (False, True):
                want = oracle.create_normalmap_array(d, args[0], args[1], args[2], inv)
                got = np.asarray(nm.create_normalmap(d, args[0], args[1], args[2], inv))
                assert np.array_equal(got, want), (d.shape, args, inv, int((got != want).sum()))
    with np.errstate(all='ignore'):
        for d in cases:
            d16 = d.astype(np.float16)
            for inv in (False, True):
                want = oracle.create_normalmap_array(d16, None, None, None, inv)
                got = np.asarray(nm.create_normalmap(d16, None, None, None, inv))
                assert np.array_equal(got, want), (d.shape, inv, int((got != want).sum()))
        d16 = rng.integers(0, 2048, (33, 47)).astype(np.float16)        # float16 through cv2.Sobel: float64 behind the float16 scaling
        for args in ((None, 3, None), (None, 5, 3)):
            want = oracle.create_normalmap_array(d16, *args, False)
            got = np.asarray(nm.create_normalmap(d16, *args, False))
            assert np.array_equal(got, want), args
    for args in ((3, None, None), (None, None, 3), (3, 3, None)):       # cv2.GaussianBlur rejects CV_16F: the reference raises too
        with pytest.raises(Exception):
            nm.create_normalmap(cases[0].astype(np.float16), args[0], args[1], args[2], False)


@pytest.mark.gpu
def test_integration_md_stub_runs_as_printed(gpu, oracle):
    """The ctypes stub INTEGRATION.md shows to a maintainer of the reference (its replacement of apply_stereo_divergence,
    src/stereoimage_generation.py:77-92) is executed exactly as printed -- only the library's file name is made absolute -- and
    must give the oracle's bytes for every fill technique."""
    import os
    import re
    import conftest
    import src._native as nat
    text = open(os.path.join(conftest.ROOT, "INTEGRATION.md")).read()
    block = re.findall(r"```python\n(.*?)```", text, flags=re.S)[0]
    assert 'ctypes.CDLL("libdepthstereo_hip.so")' in block
    ns = {}
    exec(compile(block.replace('"libdepthstereo_hip.so"', repr(nat.LIB_PATH)), "INTEGRATION.md", "exec"), ns)
    rng = np.random.default_rng(77)
    h, w = 48, 160
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    xs = np.arange(w)[None, :]
    ys = np.arange(h)[:, None]
    dep = ((xs * 30000) // (w - 1) + ((xs // 8 + ys // 8) % 2) * 8000).astype(np.uint16)
    dep[h // 4:h // 2, w // 3:2 * w // 3] = 60000
    for fill in ("none", "naive", "naive_interpolating", "polylines_soft", "polylines_sharp"):
        for div, sep in ((2.5, 0.0), (-4.0, 1.0)):
            got = ns["apply_stereo_divergence"](img, dep, div, sep, 1.0, fill)
            want = oracle.apply_stereo_divergence(img, dep, div, sep, 1.0, fill)
            assert got.dtype == np.uint8 and np.array_equal(got, want), (fill, div, sep)


@pytest.mark.gpu
def test_integration_md_linear_snippet_runs_as_printed(gpu):
    """The second block of INTEGRATION.md -- `self.act(self.fc1(x))` of the encoder's Mlp as one ds_linear call with the erf-GELU
    in the epilogue -- is executed as printed (the names it uses are a float16 nn.Linear, its input, an output buffer and the
    library handle of the first block) and held to x @ W.T + b -> GELU in float64 on the same rounded operands: 2e-3, the bar of the
    kernel's own test (tests/test_gpu_models.py::test_linear_kernel_matches_float32_and_is_race_free)."""
    import ctypes
    import os
    import re
    import torch
    import torch.nn.functional as F
    import conftest
    import src._native as nat
    text = open(os.path.join(conftest.ROOT, "INTEGRATION.md")).read()
    block = re.findall(r"```python\n(.*?)```", text, flags=re.S)[1]
    L = ctypes.CDLL(nat.LIB_PATH)
    ctx = ctypes.c_void_p()
    assert L.ds_ctx_create(ctypes.byref(ctx), 0) == 0
    torch.manual_seed(5)
    fc1 = torch.nn.Linear(1024, 4096).half().cuda()
    rows = 1032
    x = torch.randn(rows, 1024).half().cuda()
    y = torch.empty(rows, 4096, dtype=torch.float16, device="cuda")
    ns = {"ctypes": ctypes, "torch": torch, "_L": L, "_ctx": ctx, "fc1": fc1, "x": x, "y": y, "rows": rows}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    torch.cuda.synchronize()
    assert ns["rc"] == 0
    want = F.gelu(x.double() @ fc1.weight.double().T + fc1.bias.double())
    err = (y.double() - want).abs().max().item()
    assert err < 2e-3 * (1 + want.abs().max().item()), err
    L.ds_ctx_destroy(ctx)


def test_normalmap_fused_arithmetic_is_the_generic_one_on_its_domain(gpu):
    """Round 6: the fused normal-map kernels take their square root and reciprocal from the compiler's own float64 expansions WITHOUT
    the range scaling and special-case fix-ups (dead code for n^2 = zx^2 + zy^2 + 1 in [1, 2^22], src/normalmap_generation.py:34-39).
    Every n^2 they can form is K / 2^18 with an integer K in [2^18, 2^40): the device compares nm_sqrt / nm_rcp with sqrt() and
    1.0 / n on the 2^27 smallest K (flat and gently sloped surfaces: where real depth maps live), on the Sobel grid (K a multiple
    of 4) above them, and on strided sweeps of the whole range -- 0 differences of ~10^9 operands."""
    from src import _native
    total = 0
    for k0, stride, count in [(1 << 18, 1, 1 << 27), ((1 << 18) + (1 << 27), 4, 1 << 27), (1 << 18, 8191, 1 << 27), ((1 << 18) + 1, 7919, 1 << 27),
                              ((1 << 39) - (1 << 28), 1, 1 << 27), ((1 << 30) + 3, 3, 1 << 27), ((1 << 35) + 1, 29, 1 << 27)]:
        assert k0 + stride * (count - 1) < (1 << 40)
        assert _native.normalmap_selfcheck(k0, stride, count) == 0, (k0, stride, count)
        total += count
    assert total > 9e8
