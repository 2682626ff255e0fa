"""CPU: host-side logic of the drop-in (no GPU compute): option bundle, constants, the C ABI's exported symbols,
loud failure without a GPU, shard arithmetic."""
import ctypes
import os
import re

import numpy as np
import pytest

import conftest


def test_cabi_exports_every_declared_symbol():
    """include/depthstereo.h is the contract: every function it declares must be exported by the built library."""
    hdr = open(os.path.join(conftest.ROOT, "include", "depthstereo.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(ds_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 12 and "ds_stereo_warp" in declared and "ds_normalmap" in declared
    import src._native as nat
    assert os.path.exists(nat.LIB_PATH), "libdepthstereo_hip.so must be built in-tree (python __graft_entry__.py build)"
    L = ctypes.CDLL(nat.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in depthstereo.h but not exported"
    assert sorted(nat.EXPORTS) == declared, "src/_native.py binds a different set of symbols than the header declares"
    L.ds_version.restype = ctypes.c_int
    assert L.ds_version() == 100
    L.ds_last_error.restype = ctypes.c_char_p
    assert isinstance(L.ds_last_error(), bytes)


def test_cabi_argument_validation_without_gpu():
    """Entry points reject bad arguments with an error code and a message instead of crashing (no GPU needed)."""
    import src._native as nat
    L = nat.lib()
    assert L.ds_ctx_create(None, 0) != 0
    assert b"NULL" in L.ds_last_error() or b"null" in L.ds_last_error().lower()
    assert L.ds_ctx_destroy(None) == 0
    assert L.ds_stereo_warp(None, None, None, 0, 1, 1, 1, 3, 1.0, None, 4, None, 2, None) != 0
    assert L.ds_normalmap(None, None, 1, 1, 1, 0, 3, 0, 0, None, None) != 0
    assert L.ds_depth_to_u16(None, None, 1, 1, 1, 0, None, None, None) != 0


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import src._native as nat
    import src.stereoimage_generation as sg
    import src.normalmap_generation as nm
    img = np.zeros((4, 8, 3), np.uint8)
    dep = np.arange(32, dtype=np.uint16).reshape(4, 8)
    with pytest.raises(nat.DepthStereoError, match="no MI355X"):
        sg.create_stereoimages(img, dep, 2.5)
    with pytest.raises(nat.DepthStereoError, match="no MI355X"):
        nm.create_normalmap(dep)
    # cheap argument handling happens before the device is needed, like in the reference
    assert sg.create_stereoimages(img, dep, 2.5, modes=[]) == []


def test_product_code_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import or call it."""
    pkg = conftest.PKG
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), (root, f)


def test_generation_options_contract():
    """Names, order and defaults of the funnel options (the reference's GenerationOptions, common_constants.py:4-66)."""
    from src.common_constants import GenerationOptions as go
    names = [o.name for o in go]
    assert names[:7] == ['COMPUTE_DEVICE', 'MODEL_TYPE', 'BOOST', 'NET_SIZE_MATCH', 'NET_WIDTH', 'NET_HEIGHT', 'TILING_MODE']
    assert len(names) == 41 and [o.value for o in go] == list(range(1, 42))
    assert go.STEREO_MODES.df == ["left-right", "red-cyan-anaglyph"]
    assert go.STEREO_DIVERGENCE.df == 2.5 and go.STEREO_FILL_ALGO.df == "polylines_sharp" and go.STEREO_BALANCE.df == 0.0
    assert go.NORMALMAP_SOBEL.df is True and go.NORMALMAP_SOBEL_KERNEL.df == 3 and go.DO_OUTPUT_DEPTH.df is True
    assert go.NET_WIDTH.df == 448 and go.CLIPDEPTH_NEAR.df == 1.0 and go.REMBG_MODEL.df == "u2net"


def test_funnel_inp_bundle():
    from src.common_constants import GenerationOptions as go
    from src.core import CoreGenerationFunnelInp
    inp = CoreGenerationFunnelInp({'GEN_STEREO': True, go.STEREO_DIVERGENCE: 4.0, 'no_such_option': 1})
    assert inp[go.GEN_STEREO] is True and inp['gen_stereo'] is True and inp.gen_stereo is True
    assert inp[go.STEREO_DIVERGENCE] == 4.0
    assert inp[go.STEREO_MODES] == ["left-right", "red-cyan-anaglyph"]         # default filled in
    assert 'no_such_option' not in inp.values                                    # unknown keys dropped silently
    assert CoreGenerationFunnelInp(inp).values == inp.values                      # idempotent


def test_convert_i16_to_rgb_host_helper():
    from src.core import convert_i16_to_rgb
    img = np.array([[0, 255, 256, 65535]], dtype=np.uint16)
    like = np.zeros((1, 4, 3), np.uint8)
    out = convert_i16_to_rgb(img, like)
    assert out.dtype == np.uint8 and out[0, :, 0].tolist() == [0, 0, 1, 255] and np.array_equal(out[..., 0], out[..., 2])


def test_custom_depth_ingest_matches_reference_rules():
    """core.py:145-174: bit depth is guessed from the maximum; RGB depth uses channel 0 / 256."""
    from PIL import Image
    from src.core import _custom_depth_to_float
    image = Image.new('RGB', (6, 4))
    d8 = Image.fromarray(np.full((4, 6), 200, np.uint8))
    assert np.allclose(_custom_depth_to_float(d8, image), 200 / 256.0)
    d16 = Image.fromarray(np.full((4, 6), 40000, np.uint16))
    assert np.allclose(_custom_depth_to_float(d16, image), 40000 / 65536.0)
    drgb = Image.fromarray(np.full((4, 6, 3), 64, np.uint8))
    assert np.allclose(_custom_depth_to_float(drgb, image), 0.25)
    arr = np.random.default_rng(0).random((4, 6))
    assert np.array_equal(_custom_depth_to_float(arr, image), arr)
    with pytest.raises(AssertionError):
        _custom_depth_to_float(np.zeros((3, 6)), image)
    small = Image.fromarray(np.full((2, 3), 100, np.uint8))
    assert _custom_depth_to_float(small, image).shape == (4, 6)                  # resized to the image (LANCZOS)


def test_model_holder_refuses_missing_models():
    from src.depthmap_generation import ModelHolder
    mh = ModelHolder()
    assert mh.get_default_net_size(1) == [512, 512] and mh.get_default_net_size(14) == [518, 518]
    with pytest.raises(FileNotFoundError):          # built family, no checkpoint, no silent random init
        mh.ensure_models(1, 'cpu', False)
    with pytest.raises(NotImplementedError):        # family that is not built
        mh.ensure_models(5, 'cpu', False)
    with pytest.raises(NotImplementedError):        # Boost on a base model that is not built
        mh.ensure_models(5, 'cpu', True)
    with pytest.raises(FileNotFoundError):          # Boost on a MiDaS DPT base model is built; checkpoints are absent
        mh.ensure_models(1, 'cpu', True)
    assert mh.get_default_net_size(7) == [384, 512] and mh.get_default_net_size(8) == [384, 768]   # reference :323-339
    with pytest.raises(FileNotFoundError):          # Boost on LeReS is built, but the merge network's checkpoint is absent
        mh.ensure_models(0, 'cpu', True)
    mh.update_settings(boost_rmax=1600, no_half=True)
    assert mh.boost_rmax == 1600 and mh.no_half is True


def test_shard_bounds():
    from src.multigpu import shard_bounds
    assert shard_bounds(32, 8) == [(i * 4, i * 4 + 4) for i in range(8)]
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_bounds(0, 3) == [(0, 0)] * 3
    for n in range(0, 40):
        for w in range(1, 9):
            b = shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))


def test_u16_normalisation_identity_exhaustive(oracle):
    """The device computes (depth-min)/(max-min) for uint16 depth with one reciprocal and two FMAs per element; the C
    program enumerates all 2^32 operand pairs and compares with true float64 division (a few seconds on 8 cores)."""
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-C", here, "-s", "check_u16_division"])
    out = subprocess.run([os.path.join(here, "check_u16_division")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("bad=0 of 4294901760"), out.stdout


def test_normal_map_quotient_identity(oracle):
    """The fused normal-map kernels divide ONCE (1 / n, the z component) and get x / n and y / n from that reciprocal with a
    multiplication and two FMAs (Markstein's sequence); the C program compares with true float64 division on the operand set of
    the 3 x 3 Sobel path: exhaustively on a 6001 x 6001 block, and on 2 * 10^9 random pairs of the full range."""
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-C", here, "-s", "check_normal_division"])
    out = subprocess.run([os.path.join(here, "check_normal_division")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.startswith("bad=0 of "), out.stdout


def test_heatmap_table_and_gemm_tuning_host_side():
    """The colour table the heat map uploads is the one the golden run saw (same matplotlib in the image), and the GEMM
    tuning switch is a no-op without a GPU; its results file carries the validator header torch checks."""
    import os
    from src import gemm_tuning, heatmap
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'heatmap_cases.npz'))
    lut = heatmap.colormap_table('inferno')
    assert lut.shape == (256, 4) and lut.dtype == np.uint8 and np.array_equal(lut, z['inferno_lut'])
    import torch
    if not torch.cuda.is_available():
        assert gemm_tuning.enable() is False
    lines = open(gemm_tuning.RESULTS).read().splitlines()
    assert lines[0].startswith('Validator,PT_VERSION') and any(l.startswith('GemmAndBiasTunableOp_Half_TN') for l in lines)


def test_simple_mesh_geometry_matches_reference_functions(tmp_path):
    """src/mesh_generation.py (device tensor ops, run here on CPU tensors) against outputs of the reference's own
    depth_to_points / create_triangles / depth_edges_mask / pano_depth_to_world_points (tests/golden/make_golden_mesh.py):
    vertices bit for bit, faces and masks identical; then the .obj writer."""
    import torch
    from src import mesh_generation as mg
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mesh_cases.npz'))
    for name in "abc":
        d = torch.from_numpy(z[f"{name}__depth"].copy())
        h, w = d.shape
        pts = mg.depth_to_points(d)
        assert pts.dtype == torch.float64 and np.array_equal(pts.numpy(), z[f"{name}__points"]), name
        assert np.array_equal(mg.depth_edges_mask(d).numpy(), z[f"{name}__edges"]), name
        assert np.array_equal(mg.create_triangles(h, w).numpy(), z[f"{name}__tri_all"]), name
        masked = mg.create_triangles(h, w, mask=~mg.depth_edges_mask(d))
        assert np.array_equal(masked.numpy(), z[f"{name}__tri_masked"]), name
        pano = mg.pano_depth_to_world_points(d).numpy()
        assert np.allclose(pano, z[f"{name}__pano"], rtol=0, atol=1e-12 if d.dtype == torch.float64 else 1e-6), name
    d = torch.from_numpy(z["c__depth"].copy())
    img = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (24, 24, 3), dtype=np.uint8))
    v, f, c = mg.create_mesh_arrays(img, mg.mesh_depth(d, 1, False, False), keep_edges=False)
    assert v.shape == (576, 3) and c.shape == (576, 3) and f.shape[1] == 3 and 0 < f.shape[0] < 1058
    path = mg.write_obj(mg.unique_filename(str(tmp_path), 'depthmap', 'obj', 'simple'), v, f, c)
    lines = open(path).read().splitlines()
    assert path.endswith('depthmap-0000-simple.obj') and sum(l.startswith('v ') for l in lines) == 576
    assert sum(l.startswith('f ') for l in lines) == f.shape[0] and lines[-1].split()[0] == 'f'
    assert mg.unique_filename(str(tmp_path), 'depthmap', 'obj', 'simple').endswith('depthmap-0001-simple.obj')
    # mesh_depth: ZoeDepth predictions pass through untouched, everything else is inverted / shifted / offset (core.py:283-303)
    zd = torch.tensor([[2.0, 3.0]])
    assert torch.equal(mg.mesh_depth(zd, 7, False, False), zd)
    assert torch.equal(mg.mesh_depth(zd, 1, False, False), torch.tensor([[4.0, 3.0]]))
    # the rescale uses the min / max from BEFORE the shift (reference quirk): 4 * ([0, 21] + 1) / 21 + 1
    assert torch.allclose(mg.mesh_depth(torch.tensor([[-1.0, 20.0]]), 0, False, False), torch.tensor([[1.0 + 4 / 21, 1.0 + 88 / 21]]))


def test_funnel_group_plan():
    """core._plan_groups: consecutive images of one size / mode / depth-source kind form a device batch, capped by the
    pixel budget and at 64 images; Boost (batchable=False) goes image by image; order is preserved."""
    from PIL import Image
    import src.core as core
    a, b = Image.new("RGB", (64, 48)), Image.new("RGB", (32, 48))
    rgba = Image.new("RGBA", (64, 48))
    ims = [a, a, a, b, b, a, rgba, a]
    none = [None] * len(ims)
    assert core._plan_groups(ims, none, True) == [[0, 1, 2], [3, 4], [5], [6], [7]]
    assert core._plan_groups(ims, none, False) == [[i] for i in range(len(ims))]
    deps = [None, object(), object(), None, None, None, None, None]
    assert core._plan_groups(ims, deps, True) == [[0], [1, 2], [3, 4], [5], [6], [7]]
    many = [a] * 150
    g = core._plan_groups(many, [None] * 150, True)
    assert [len(x) for x in g] == [64, 64, 22] and sum(g, []) == list(range(150))
    old = core.FUNNEL_BATCH_PIXELS
    try:
        core.FUNNEL_BATCH_PIXELS = 64 * 48 * 2
        assert [len(x) for x in core._plan_groups([a] * 5, [None] * 5, True)] == [2, 2, 1]
        core.FUNNEL_BATCH_PIXELS = 1
        assert [len(x) for x in core._plan_groups([a] * 3, [None] * 3, True)] == [1, 1, 1]
    finally:
        core.FUNNEL_BATCH_PIXELS = old


def test_module_caches_keep_sizes_and_move_the_epoch_on_eviction():
    """src/vit_mi355x.cache_store / CACHE_EPOCH: the size-keyed module caches keep several entries, and every eviction moves
    the epoch that makes src/hip_graph.GraphedForward drop graphs that may hold pointers to the dropped tensors.  Pinned on
    the DINOv2 position-embedding cache (the one whose single entry once made a captured graph read freed memory)."""
    import torch
    from src import vit_mi355x as vm
    from ddepth_anything_v2.depth_anything_v2.dinov2 import DINOv2
    e0 = vm.CACHE_EPOCH[0]
    cache = {}
    for i in range(vm.SIZE_CACHE_ENTRIES):
        vm.cache_store(cache, i, i)
    assert len(cache) == vm.SIZE_CACHE_ENTRIES and vm.CACHE_EPOCH[0] == e0
    vm.cache_store(cache, "x", 1)
    assert list(cache) == ["x"] and vm.CACHE_EPOCH[0] == e0 + 1
    net = DINOv2('vits').eval()
    with torch.no_grad():
        a = net.interpolate_pos_encoding(25, 70, 70, torch.float32)
        e1 = vm.CACHE_EPOCH[0]
        b = net.interpolate_pos_encoding(30, 84, 70, torch.float32)
        a2 = net.interpolate_pos_encoding(25, 70, 70, torch.float32)
    assert a2 is a and b is not a and vm.CACHE_EPOCH[0] == e1          # both sizes stay cached: nothing was evicted


def test_linear_and_conv_dispatch_rules_host_side():
    """src/vit_mi355x.linear / conv2d / residual_conv_unit: float32 and CPU tensors take the plain torch definition (what the
    parity tests against the reference's modules run); the shape rules of the in-tree kernel are those of the C ABI."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from src import _native
    from src import vit_mi355x as vm
    g = torch.Generator().manual_seed(0)
    x = torch.randn((3, 7, 128), generator=g)
    lin = nn.Linear(128, 256)
    assert torch.equal(vm.linear(x, lin.weight, lin.bias), F.linear(x, lin.weight, lin.bias))
    assert torch.equal(vm.linear(x, lin.weight, lin.bias, gelu=True), F.gelu(F.linear(x, lin.weight, lin.bias)))
    mlp = vm.Mlp(128, 512)
    assert torch.allclose(mlp(x), mlp.fc2(F.gelu(mlp.fc1(x))))
    # shapes the kernel takes: out % 256 == 0, in % 128 == 0
    assert _native.linear_supported(x, torch.empty(256, 128)) and _native.linear_supported(x, torch.empty(1536, 384))
    assert not _native.linear_supported(x, torch.empty(200, 128)) and not _native.linear_supported(x, torch.empty(256, 64))
    conv = nn.Conv2d(256, 256, 3, padding=1)
    img = torch.randn((1, 256, 16, 16), generator=g)
    assert _native.conv3x3_supported(conv, img)
    assert not vm.conv3x3_hip_ok(conv, img)                                   # CPU tensor / float32: the library
    assert torch.equal(vm.conv2d(conv, img), conv(img))
    for bad in (nn.Conv2d(256, 256, 3, padding=1, padding_mode='circular'), nn.Conv2d(256, 256, 3, padding=1, stride=2),
                nn.Conv2d(256, 64, 3, padding=1), nn.Conv2d(64, 256, 3, padding=1), nn.Conv2d(256, 256, 1)):
        xin = torch.randn((1, bad.in_channels, 16, 16), generator=g)
        assert not _native.conv3x3_supported(bad, xin)
    assert _native.conv3x3_supported(nn.Conv2d(256, 128, 3, padding=1), img)   # 256 x 128 tiles (the head's first convolution)
    assert not _native.conv3x3_supported(conv, torch.randn((1, 256, 8, 8)))   # fewer than 256 pixels: below one tile
    c1, c2 = nn.Conv2d(16, 16, 3, padding=1), nn.Conv2d(16, 16, 3, padding=1)
    xr, sk = torch.randn((2, 16, 9, 11), generator=g), torch.randn((2, 16, 9, 11), generator=g)
    want = sk + (c2(F.relu(c1(F.relu(xr)))) + xr)
    assert torch.allclose(vm.residual_conv_unit(c1, c2, xr, skip=sk), want)


def test_float16_prediction_corner():
    """DESIGN.md 'defined corners': the reference's default GPU path hands core.py a FLOAT16 prediction
    (src/depthmap_generation.py:484-497) and post-processes it with whatever promotion rules its NumPy has (src/core.py:189-211,
    :44-50).  Pinned here: (a) NumPy 1.x semantics, restated with explicit casts (oracle.depth_postprocess_f16_numpy1): float16
    normalisation, float64 scaling -- promote_types(float16, uint32) is float64 in every NumPy, which is what value-based
    casting made of `arr * 65536`; (b) what the SAME reference statements do under NumPy >= 2 (this image): the python int
    becomes float16(inf) and the depth map is zeros; (c) the product's default, float32 post-processing, differs from (a) by
    float16's quantisation only."""
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    pred = (rng.random((48, 64)) * 37.0 + 3.0).astype(np.float16)
    assert np.promote_types(np.float16, np.uint32) == np.float64
    want = orc.depth_postprocess_f16_numpy1(pred)
    # (a) spelled differently: the reference's statements with the NumPy-1.x result dtypes forced
    out = pred.copy()
    out = ((out - out.min()) / (out.max() - out.min()))
    assert out.dtype == np.float16                       # array (op) numpy-scalar of one dtype: float16 in every NumPy
    a = np.clip(out.astype(np.float64) * np.uint32(65536) + 0.0001, 0, 65536 - 0.1).astype("uint16")
    assert np.array_equal(a, want)
    assert len(np.unique(want)) < 3000 and want.max() == 65535 and want.min() == 0      # float16's levels, full range
    # (b) the reference's convert_to_i16 verbatim on the float16 array under this NumPy
    if int(np.__version__.split(".")[0]) >= 2:
        import warnings
        with np.errstate(all='ignore'), warnings.catch_warnings():
            warnings.simplefilter('ignore')
            b = np.clip(out * (2 ** 16) + 0.0001, 0, (2 ** 16) - 0.1).astype("uint16")
        assert int(b.max()) == 0
    # (c) the default here: float32 throughout
    c = orc.convert_to_i16(orc.depth_normalize01(pred.astype(np.float32)))
    assert np.abs(c.astype(np.int64) - want.astype(np.int64)).max() <= 64      # half a float16 ulp at the top octave x 65536
    # inverted models and the 'Range' clip
    w2 = orc.depth_postprocess_f16_numpy1(pred, invert=True, clipdepth=True, far=0.25, near=0.75)
    assert w2.dtype == np.uint16 and w2.min() == 0 and w2.max() == 65535


def test_funnel_rgbx_pixel_export_is_only_taken_where_it_is_safe():
    """core._rgbx_pixels (the GIL-free PIL -> pinned staging path of the funnel): for RGB images that own their pixel store the
    exported 4-byte pixels equal np.asarray's RGB; images that map foreign memory (readonly: Image.fromarray of L / RGBA / I;16,
    where Pillow 12.2's export crashes the process), other modes and multi-block images are refused -> the np.asarray path."""
    import ctypes
    from PIL import Image
    import src.core as core
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    cases = [Image.fromarray(rgb), Image.fromarray(rgb).crop((1, 2, 40, 30)), Image.new("RGB", (9, 5), (1, 2, 3)),
             Image.frombuffer("RGB", (53, 37), rgb.tobytes(), "raw", "RGB", 0, 1), Image.fromarray(rgb).convert("L").convert("RGB")]
    if not hasattr(cases[0], "__arrow_c_array__"):
        assert all(core._rgbx_pixels(im) is None for im in cases)          # an older Pillow: the ordinary path everywhere
        return
    for im in cases:
        got = core._rgbx_pixels(im)
        assert got is not None, im
        addr, keep = got
        px = np.frombuffer((ctypes.c_uint8 * (im.width * im.height * 4)).from_address(addr), np.uint8).reshape(im.height, im.width, 4)
        assert np.array_equal(px[:, :, :3], np.asarray(im))
        del px, keep
    refused = [Image.fromarray(rgb[:, :, 0].copy()), Image.fromarray(np.dstack([rgb, rgb[:, :, :1]])),
               Image.fromarray(rng.integers(0, 65536, (8, 8), dtype=np.uint16)), Image.new("L", (4, 4)), Image.new("RGBA", (4, 4)),
               Image.new("RGB", (3000, 2000))]                             # 24 MB of pixels: more than one 16 MB block
    for im in refused:
        assert core._rgbx_pixels(im) is None, (im.mode, im.size)


def test_funnel_rgbx_export_refuses_padded_lines_and_unverified_layouts(monkeypatch):
    """The advisor's scenario: a host application that raised Pillow's line alignment pads every image line, and the funnel's
    memmove of h * w * 4 bytes would copy the padding as pixels -- the export must be refused; so it must when the once-per-process
    layout probe (a known image exported and compared byte for byte) has not passed."""
    from PIL import Image
    import src.core as core
    im = Image.new("RGB", (9, 5), (1, 2, 3))
    if not hasattr(im, "__arrow_c_array__"):
        return
    assert core._ARROW_LAYOUT_OK and core._rgbx_pixels(im) is not None
    if hasattr(Image.core, "get_alignment"):
        monkeypatch.setattr(Image.core, "get_alignment", lambda: 4, raising=False)
        assert core._rgbx_pixels(im) is None
        monkeypatch.undo()
    monkeypatch.setattr(core, "_ARROW_LAYOUT_OK", False)
    assert core._rgbx_pixels(im) is None


def test_funnel_pil_allocator_scope_is_reference_counted():
    """Two interleaved funnel calls share ONE scoped change of Pillow's blocks_max: the first raises it, the LAST one to finish
    restores the caller's value (round 4: the first to finish restored it under the second), and a value the host application set
    in the meantime is left alone."""
    from PIL import Image
    import src.core as core
    if not hasattr(Image.core, "get_blocks_max"):
        return
    start = Image.core.get_blocks_max()
    try:
        Image.core.set_blocks_max(0)
        a = core._tune_pil_allocator()
        assert a and Image.core.get_blocks_max() == int(os.environ.get("DS_PIL_BLOCKS_MAX", 64))
        b = core._tune_pil_allocator()
        assert b
        core._restore_pil_allocator(a)                     # the first call finishes: the second is still running
        assert Image.core.get_blocks_max() == int(os.environ.get("DS_PIL_BLOCKS_MAX", 64))
        core._restore_pil_allocator(b)
        assert Image.core.get_blocks_max() == 0            # the last one out restores the caller's value
        c = core._tune_pil_allocator()
        Image.core.set_blocks_max(7)                       # the host application changes it while a call runs
        core._restore_pil_allocator(c)
        assert Image.core.get_blocks_max() == 7            # ... and keeps its choice
        Image.core.set_blocks_max(1000)                    # already larger than the funnel would ask for: nothing to scope
        assert core._tune_pil_allocator() is False and Image.core.get_blocks_max() == 1000
    finally:
        Image.core.set_blocks_max(start)


def test_tight_token_pad_rules():
    """vm.pad_len: without a batch the pad is a multiple of 64 (one key tile); with one it is the tightest multiple of 8 / 16 / 32
    whose batch x stride rows are whole 256-row panels (what ds_linear_vt needs for its columns), else 64."""
    from src import vit_mi355x as vm
    assert vm.pad_len(1025) == 1088 and vm.pad_len(64) == 64 and vm.pad_len(1) == 64
    assert vm.pad_len(1025, 32) == 1032 and (32 * 1032) % 256 == 0          # dpt_beit_large_512 at the benchmark's batch: 129 panels
    assert vm.pad_len(2443, 8) == 2464 and (8 * 2464) % 256 == 0            # Depth-Anything-V2 ViT-L, 1080p, batch 8
    assert vm.pad_len(1025, 16) == 1040 and vm.pad_len(577, 32) == 584
    assert vm.pad_len(4097, 8) == 4128                                       # net 1024 (NET_SIZE_MATCH), batch 8: 129 panels again
    for b, n in ((1, 1025), (2, 1025), (3, 577), (4, 2443)):                # no tight pad keeps whole panels: 64 as before
        assert vm.pad_len(n, b) == vm.pad_len(n)
    for b in range(1, 70):
        for n in (1, 63, 64, 65, 577, 1025, 1370, 2443, 4097):
            s = vm.pad_len(n, b)
            assert s >= n and s % 8 == 0 and s <= vm.pad_len(n)
            assert s == vm.pad_len(n) or (b * s) % 256 == 0


def test_committed_profiles_parse_and_tell_the_same_story_as_the_docs():
    """tools/step_anatomy.py on the committed rocprofv3 summaries: the families add up to the whole, the in-tree share the
    documents quote (> 90 % of the step's kernel time; round 5: 94.6 %, library kernels 2.1 %) is what the file says, and the stalled
    LayerNorm launch of round 4's final verification box is reported instead of silently inflating an average.  The bench line's live
    in-step averages must agree with the trace of the same command (the contract of `roofline`)."""
    import contextlib
    import io
    import sys
    ROOT = conftest.ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import step_anatomy
    for name, expect_outlier in (("round5_kernel_stats.csv", False), ("round4_kernel_stats.csv", True), ("round4_kernel_stats_before_pipelined_ragged.csv", False)):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            step_anatomy.main([os.path.join(ROOT, "profiles", name)])
        text = buf.getvalue()
        shares = [float(line.split("%")[0]) for line in text.splitlines()[1:] if "%" in line and "in all" not in line and "outlier" not in line]
        assert abs(sum(shares) - 100.0) < 0.2, text
        in_tree = float([line for line in text.splitlines() if "in-tree kernels (hand-written HIP) in all" in line][0].split("%")[0])
        assert in_tree > 90.0, text
        assert ("outlier:" in text) == expect_outlier, text
        if name.startswith("round5"):                      # round 5: the reassemble stage in-tree -- library kernels <= 2.5 % of the step
            lib = sum(float(line.split("%")[0]) for line in text.splitlines() if "MIOpen / CK" in line or "library GEMMs" in line)
            assert in_tree > 94.0 and lib < 2.5, text


def test_bench_line_in_step_figures_agree_with_the_committed_trace():
    """profiles/round5_bench_n1.json (the driver's line) against profiles/round5_kernel_stats.csv (rocprofv3 of the same command on the
    same box): the in-step average durations the C ABI's event timers measured inside the timed region agree with the trace's averages
    within 5 % for the three big kernels, `roofline` is the in-step figure of the dominant one and its frac = achieved / peak."""
    import csv
    import json
    line = json.loads(open(os.path.join(conftest.ROOT, "profiles", "round5_bench_n1.json")).read().strip().splitlines()[-1])
    rows = {r["Name"]: float(r["AverageNs"]) * 1e-6 for r in csv.DictReader(open(os.path.join(conftest.ROOT, "profiles", "round5_kernel_stats.csv")))}
    pairs = (("roofline_linear", "k_linear256<0, 1, 0, 0, 0"), ("roofline_linear_residual", "k_linear256<0, 3, 0, 1"), ("roofline_attention", "k_attention_fwd2"))
    for key, kern in pairs:
        live = line[key]["avg_kernel_ms"]
        trace = next(v for n, v in rows.items() if kern in n)
        assert "in-step" in line[key]["source"] and abs(live - trace) / trace < 0.05, (key, live, trace)
        assert abs(line[key]["frac"] - line[key]["achieved"] / line[key]["peak"]) < 1e-9
    assert line["roofline"]["kernel"] == line["roofline_linear_residual"]["kernel"] and "in-step" in line["roofline"]["source"]
    assert set(line["other_configs"]) == {"c5", "c2", "c4"} and all("error" not in v for v in line["other_configs"].values())
    assert line["cpu_baseline"]["python_fallback"]["kind"] == "port-of-fallback"


def test_standalone_harnesses_compile(tmp_path):
    """tools/att_harness.cpp and tools/stereo_harness.cpp (the Python-free A/B harnesses of the GPU rounds) keep compiling
    against the HIP runtime and use the C ABI's current signatures by name (dlsym): every symbol they ask for is declared in
    include/depthstereo.h, except the experiments-only profile reader."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    hdr = open(os.path.join(conftest.ROOT, "include", "depthstereo.h")).read()
    for name in ("att_harness", "stereo_harness"):
        src = os.path.join(conftest.ROOT, "tools", name + ".cpp")
        text = open(src).read()
        for sym in re.findall(r'dlsym\(lib, "(\w+)"\)', text):
            assert sym == "ds_experiments_attention_profile" or re.search(r"\b%s\(" % sym, hdr), f"{name}: {sym} is not in the header"
        subprocess.check_call([hipcc, "-O1", "-std=c++17", src, "-o", str(tmp_path / name), "-ldl"])


def test_gconv_weight_image_matches_the_grouped_convolution():
    """_native.gconv_weight_image regroups torch's grouped weight [out, in / groups, 3, 3] into the [group][tap][in][out] image the HIP
    kernel's scalar loads walk (csrc/ds_gconv.hip).  A numpy restatement of the kernel's index arithmetic -- input pixel (oy - 1 + tap / 3,
    ox - 1 + tap % 3), channel g * cpg + ci, weight image[g][tap][ci][co] -- on that image must equal torch's grouped convolution."""
    import torch
    import torch.nn.functional as F
    import src._native as nat
    g = torch.Generator().manual_seed(5)
    for cpg, groups, h, w in ((8, 4, 5, 7), (16, 2, 4, 6), (32, 1 + 1, 3, 5)):
        c = cpg * groups
        x = torch.randn((2, c, h, w), generator=g)
        wt = torch.randn((c, cpg, 3, 3), generator=g)
        b = torch.randn((c,), generator=g)
        ref = F.conv2d(x, wt, b, 1, 1, 1, groups).permute(0, 2, 3, 1).numpy()
        img = nat.gconv_weight_image(wt, groups).numpy()
        assert img.shape == (groups, 9, cpg, cpg)
        xp = np.pad(x.permute(0, 2, 3, 1).numpy(), ((0, 0), (1, 1), (1, 1), (0, 0)))
        out = np.tile(b.numpy()[None, None, None, :], (2, h, w, 1)).astype(np.float64)
        for gi in range(groups):
            for tap in range(9):
                dy, dx = tap // 3, tap % 3
                patch = xp[:, dy:dy + h, dx:dx + w, gi * cpg:(gi + 1) * cpg].astype(np.float64)          # [b, h, w, ci]
                out[..., gi * cpg:(gi + 1) * cpg] += patch @ img[gi, tap].astype(np.float64)              # [ci, co]
        assert np.abs(out - ref).max() < 1e-4 * (1 + np.abs(ref).max())


def test_exact_sweep_parallel_removal_is_the_sequential_scan():
    """The reference's active-set scan (src/stereoimage_generation.py:247-255: `if ...: csg[i] = csg[end - 1]; end -= 1 else: i += 1`)
    restated as the permutation the wave-cooperative exact sweep applies (csrc/ds_stereo_polylines.hip, k_polylines_exact_lds<.., COOP>):
    with K kept entries, the kept entries below position K stay, the removed positions below K (ascending) receive the kept entries
    from K up taken from the END downwards; when nothing is kept, slot 0 holds the old entry 1 (what the sequential scan leaves in
    the slot the reference later reads while the set is empty).  Everything the sweep can observe -- the active part in order, and
    slot 0 of an emptied set -- must agree on random keep patterns, including 0, 1 and several 64-entry chunks of entries."""
    import random

    def sequential(csg, removable):
        csg, end, i = list(csg), len(csg), 0
        while i < end:
            if csg[i] in removable:
                csg[i] = csg[end - 1]
                end -= 1
            else:
                i += 1
        return end, csg

    def permutation(csg, removable):
        n = len(csg)
        keep = [v not in removable for v in csg]
        k = sum(keep)
        out = list(csg)
        if k == 0:
            if n >= 2:
                out[0] = csg[1]
            return 0, out
        holes = [p for p in range(k) if not keep[p]]
        donors = [p for p in range(n - 1, k - 1, -1) if keep[p]]
        assert len(holes) == len(donors)
        for h, d in zip(holes, donors):
            out[h] = csg[d]
        return k, out

    rnd = random.Random(7)
    for _ in range(20000):
        n = rnd.choice((0, 1, 2, 3, 5, 17, 63, 64, 65, 130, 200))
        vals = rnd.sample(range(10000), n)
        p = rnd.choice((0.0, 0.05, 0.5, 0.95, 1.0))
        removable = {v for v in vals if rnd.random() < p}
        ea, fa = sequential(vals, removable)
        eb, fb = permutation(vals, removable)
        assert ea == eb and fa[:ea] == fb[:eb], (vals, removable)
        if ea == 0 and n > 0:
            assert fa[0] == fb[0], (vals, removable)


def test_integration_md_binding_snippets_match_the_header():
    """INTEGRATION.md shows the stub a maintainer of the reference would add: its Python blocks must be valid Python, every
    `_L.ds_*` they call must be a symbol the header declares, the `ds_eye` structure must have the header's fields in the header's
    order, and the argtypes it spells out for ds_linear must be the ones src/_native.py binds."""
    import ast
    text = open(os.path.join(conftest.ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    assert len(blocks) >= 2
    hdr = open(os.path.join(conftest.ROOT, "include", "depthstereo.h")).read()
    declared = set(re.findall(r"\b(ds_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)))
    called = set()
    for b in blocks:
        ast.parse(b)                                                  # a SyntaxError here is a broken document
        called |= set(re.findall(r"_L\.(ds_[a-z0-9_]+)", b))
    assert called and called <= declared, called - declared
    fields = re.findall(r'\("([a-z_]+)", ctypes\.(c_[a-z0-9_]+)\)', blocks[0])
    body = re.search(r"typedef struct ds_eye \{(.*?)\} ds_eye;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    want = [(n, {"double": "c_double", "uint8_t *": "c_void_p", "int64_t": "c_int64"}[t.strip()])
            for t, n in re.findall(r"(double|uint8_t \*|int64_t)\s*([a-z_]+);", body)]
    assert fields == want, (fields, want)
    # the argument list the stub passes to ds_stereo_warp has as many entries as the declaration has parameters
    decl = re.search(r"int ds_stereo_warp\((.*?)\);", hdr, flags=re.S).group(1)
    call = ast.parse(blocks[0])
    n_args = [len(c.args) for c in ast.walk(call) if isinstance(c, ast.Call) and isinstance(c.func, ast.Attribute) and c.func.attr == "ds_stereo_warp"]
    assert n_args == [len(decl.split(","))]
    m = re.search(r"_L\.ds_linear\.argtypes = \[(.*?)\]", blocks[1])
    spelled = [a.strip() for a in m.group(1).split(",")]
    decl = re.search(r"int ds_linear\((.*?)\);", hdr, flags=re.S).group(1)
    kinds = ["I64" if "int64_t" in p else ("P" if "*" in p else "ctypes.c_int") for p in decl.split(",")]
    assert spelled == kinds, (spelled, kinds)


def test_miopen_find_db_seeding_never_overwrites(tmp_path, monkeypatch):
    """src/miopen_db.seed: the shipped MIOpen find results are copied into the user db directory when no file of that name exists
    there, never over an existing one, atomically; DS_MIOPEN_SEED=0 switches it off."""
    from src import miopen_db
    shipped = sorted(os.listdir(os.path.join(conftest.PKG, "miopen_db")))
    assert shipped and all(f.endswith(".ufdb.txt") for f in shipped)
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", str(tmp_path / "db"))
    monkeypatch.setattr(miopen_db, "_done", [False])
    monkeypatch.setenv("DS_MIOPEN_SEED", "0")
    assert miopen_db.seed() == [] and not (tmp_path / "db").exists()
    monkeypatch.setenv("DS_MIOPEN_SEED", "1")
    monkeypatch.setattr(miopen_db, "_done", [False])
    copied = miopen_db.seed()
    assert sorted(os.path.basename(c) for c in copied) == shipped
    assert sorted(os.listdir(tmp_path / "db")) == shipped                        # no temporary left behind
    mine = tmp_path / "db" / shipped[0]
    mine.write_text("the user's own find results\n")
    monkeypatch.setattr(miopen_db, "_done", [False])
    assert miopen_db.seed() == [] and mine.read_text() == "the user's own find results\n"
