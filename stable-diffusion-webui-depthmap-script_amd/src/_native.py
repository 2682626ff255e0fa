"""ctypes binding of libdepthstereo_hip.so (C ABI: include/depthstereo.h) for torch tensors in HBM.

PyTorch is plumbing here: it owns device memory and streams; every per-pixel operation is a
hand-written HIP kernel reached through the C ABI.  There is NO CPU fallback: if the shared library is
missing, or no MI355X is visible, the calls raise.
"""
import collections
import ctypes
import math
import os
import threading

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# DS_NATIVE_LIB: another build of the same library (the -DDS_EXPERIMENTS one of build_native.py, for A/B runs on hardware)
LIB_PATH = os.environ.get("DS_NATIVE_LIB") or os.path.join(_PKG_ROOT, "libdepthstereo_hip.so")

DS_DEPTH_U16, DS_DEPTH_F32, DS_DEPTH_F64 = 0, 1, 2
FILL_IDS = {"none": 0, "naive": 1, "naive_interpolating": 2, "polylines_soft": 3, "polylines_sharp": 4}

EXPORTS = [
    "ds_version", "ds_last_error", "ds_normalmap_f64", "ds_normalmap_gradient_f32", "ds_reassemble_readout", "ds_bias_act_nhwc", "ds_linear", "ds_linear_residual", "ds_linear_vt", "ds_conv3x3_nhwc", "ds_ctx_create", "ds_ctx_destroy", "ds_stereo_warp", "ds_depth_minmax",
    "ds_stereo_last_exact_rows", "ds_copy_view", "ds_overlap_red_cyan", "ds_normalmap", "ds_depth_to_u16",
    "ds_convert_to_i16", "ds_profile_enable", "ds_profile_last_ms", "ds_stereo_last_stats", "ds_attention_fwd", "ds_attention_bias_pack", "ds_colorize_u16", "ds_residual_layernorm", "ds_boost_blend", "ds_upsample_bilinear_nhwc", "ds_dpt_head_tail", "ds_preprocess_bicubic", "ds_linear_reload_env", "ds_attention_reload_env", "ds_normalmap_selfcheck", "ds_normalmap_gradient_f16", "ds_normalmap_gradient_blur_f32",
    "ds_linear_shuffle", "ds_linear_readout", "ds_kernel_timer_enable", "ds_kernel_timer_read", "ds_kernel_timer_read_each", "ds_group_norm_nchw",
    "ds_row_stats", "ds_linear_ln", "ds_linear_vt_ln", "ds_gconv3x3_nhwc_f32", "ds_add_relu_f32", "ds_bias_act_f32", "ds_relu_cat_f32",
]


class DepthStereoError(RuntimeError):
    pass


class ds_eye(ctypes.Structure):
    _fields_ = [("divergence_px", ctypes.c_double), ("separation_px", ctypes.c_double),
                ("out", ctypes.c_void_p), ("out_row_stride", ctypes.c_int64), ("out_img_stride", ctypes.c_int64)]


_lib = None
_lib_lock = threading.Lock()

# How often each C-ABI entry point was reached through this binding (tests and bench.py's route_check assert that a forward
# really took the in-tree kernels it claims: a routing threshold that silently sends a shape to the library shows up here).
CALLS = collections.Counter()


def lib():
    """Load the shared library (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise DepthStereoError(
                    f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                    "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
            L = ctypes.CDLL(LIB_PATH)
            vp, ci, cd, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int64
            L.ds_version.restype = ci
            L.ds_last_error.restype = ctypes.c_char_p
            L.ds_ctx_create.argtypes = [ctypes.POINTER(vp), ci]
            L.ds_ctx_destroy.argtypes = [vp]
            L.ds_stereo_warp.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, cd, vp, ci, ctypes.POINTER(ds_eye), ci, vp]
            L.ds_depth_minmax.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp]
            L.ds_stereo_last_exact_rows.argtypes = [vp, ctypes.POINTER(i64), vp]
            L.ds_copy_view.argtypes = [vp, vp, i64, i64, vp, i64, i64, ci, ci, i64, vp]
            L.ds_overlap_red_cyan.argtypes = [vp, vp, i64, i64, vp, i64, i64, ci, ci, ci, ci, vp, vp]
            L.ds_normalmap.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]
            L.ds_normalmap_f64.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]
            L.ds_normalmap_gradient_f32.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp]
            L.ds_normalmap_gradient_f16.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp]
            L.ds_normalmap_gradient_blur_f32.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp, vp]
            L.ds_depth_to_u16.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp]
            L.ds_convert_to_i16.argtypes = [vp, vp, ci, i64, vp, vp]
            L.ds_stereo_last_stats.argtypes = [vp, ctypes.POINTER(i64), vp]
            L.ds_attention_fwd.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ctypes.c_float, ci, vp]
            L.ds_normalmap_selfcheck.argtypes = [vp, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.POINTER(ctypes.c_ulonglong), vp]
            L.ds_attention_bias_pack.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp]
            L.ds_colorize_u16.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, vp, vp]
            L.ds_residual_layernorm.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, ci, ctypes.c_float, ci, vp]
            L.ds_boost_blend.argtypes = [vp, vp, i64, ci, ci, vp, ci, vp, ci, vp, ci, vp]
            L.ds_upsample_bilinear_nhwc.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
            L.ds_dpt_head_tail.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, ctypes.c_float, ci, vp, ci, vp]
            L.ds_preprocess_bicubic.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_float),
                                                ctypes.POINTER(ctypes.c_float), ci, vp]
            L.ds_reassemble_readout.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
            L.ds_bias_act_nhwc.argtypes = [vp, vp, vp, vp, vp, vp, i64, ci, ci, ci, vp]
            L.ds_group_norm_nchw.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ctypes.c_float, ci, ci, vp]
            L.ds_linear.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, ci, ci, vp]
            L.ds_linear_residual.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, ci, vp]
            L.ds_linear_vt.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, ci, vp]
            L.ds_conv3x3_nhwc.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]
            L.ds_linear_shuffle.argtypes = [vp, vp, vp, vp, vp, i64, i64, ci, ci, ci, ci, vp]
            L.ds_linear_readout.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, ci, vp]
            L.ds_row_stats.argtypes = [vp, vp, vp, i64, ci, ctypes.c_float, ci, vp]
            L.ds_linear_ln.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, ci, ci, vp]
            L.ds_linear_vt_ln.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, ci, vp]
            L.ds_gconv3x3_nhwc_f32.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
            L.ds_add_relu_f32.argtypes = [vp, vp, vp, vp, i64, vp]
            L.ds_bias_act_f32.argtypes = [vp, vp, vp, vp, vp, i64, ci, ci, vp]
            L.ds_relu_cat_f32.argtypes = [vp, vp, vp, vp, ci, i64, ci, ci, ci, vp]
            L.ds_kernel_timer_enable.argtypes = [vp, ci]
            L.ds_kernel_timer_read.argtypes = [vp, ci, ctypes.POINTER(i64), ctypes.POINTER(cd)]
            L.ds_kernel_timer_read_each.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_float), i64, ctypes.POINTER(i64)]
            L.ds_profile_enable.argtypes = [vp, ci]
            L.ds_profile_last_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
            for name in EXPORTS:          # fail at load time, not at first use, if a symbol is missing
                getattr(L, name)
            _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise DepthStereoError(f"libdepthstereo_hip error {rc}: {lib().ds_last_error().decode(errors='replace')}")


_ctxs = {}


def ctx_for(device_index):
    """One ds_ctx per (thread, device)."""
    key = (threading.get_ident(), int(device_index))
    c = _ctxs.get(key)
    if c is None:
        h = ctypes.c_void_p()
        _check(lib().ds_ctx_create(ctypes.byref(h), int(device_index)))
        _ctxs[key] = c = h
    return c


def _torch():
    import torch
    return torch


def require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise DepthStereoError("no MI355X visible to torch (torch.cuda.is_available() is False); "
                               "this package has no CPU path")
    return torch


def _stream(t):
    torch = _torch()
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _dev_index(t):
    torch = _torch()
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _depth_dtype_id(t):
    torch = _torch()
    if t.dtype in (torch.uint16, torch.int16):
        return DS_DEPTH_U16
    if t.dtype == torch.float32:
        return DS_DEPTH_F32
    if t.dtype == torch.float64:
        return DS_DEPTH_F64
    raise DepthStereoError(f"unsupported device depth dtype {t.dtype}")


# ------------------------------------------------------------------------------------------------
def depth_minmax(depth):
    """Per-image {min,max} (float64, on device) of a [N,H,W] depth tensor."""
    torch = require_gpu()
    n, h, w = depth.shape
    out = torch.empty((n, 2), dtype=torch.float64, device=depth.device)
    CALLS["ds_depth_minmax"] += 1
    _check(lib().ds_depth_minmax(ctx_for(_dev_index(depth)), depth.data_ptr(), _depth_dtype_id(depth), n, h, w,
                                 out.data_ptr(), _stream(depth)))
    return out


def build_pow_lut(depth_u16, exponent):
    """norm**exponent for every uint16 code of every image, computed on the HOST with libm pow
    (what `normalized_depth[row][col] ** stereo_offset_exponent` does, stereoimage_generation.py:108,182)."""
    torch = require_gpu()
    mm = depth_minmax(depth_u16).cpu().numpy()
    n = mm.shape[0]
    lut = np.zeros((n, 65536), np.float64)
    codes = np.arange(65536, dtype=np.float64)
    for i in range(n):
        mn, mx = mm[i]
        if not mx > mn:
            lut[i, :] = np.nan
            continue
        norm = (codes - mn) / (mx - mn)
        lo, hi = int(mn), int(mx)
        vals = [math.pow(x, exponent) for x in norm[lo:hi + 1].tolist()]
        lut[i, lo:hi + 1] = vals
    return torch.from_numpy(lut).to(depth_u16.device)


def stereo_warp(image, depth, eyes, fill, exponent=1.0, pow_lut=None):
    """image [N,H,W,C] uint8, depth [N,H,W]; eyes = list of (divergence_px, separation_px, out_tensor_view_ptr,
    row_stride_bytes, img_stride_bytes)."""
    require_gpu()
    assert image.is_contiguous() and depth.is_contiguous()
    n, h, w, c = image.shape
    arr = (ds_eye * len(eyes))()
    for i, (dv, sp, ptr, rs, is_) in enumerate(eyes):
        arr[i].divergence_px = float(dv)
        arr[i].separation_px = float(sp)
        arr[i].out = ptr
        arr[i].out_row_stride = int(rs)
        arr[i].out_img_stride = int(is_)
    CALLS["ds_stereo_warp"] += 1
    _check(lib().ds_stereo_warp(ctx_for(_dev_index(image)), image.data_ptr(), depth.data_ptr(), _depth_dtype_id(depth),
                                n, h, w, c, float(exponent), pow_lut.data_ptr() if pow_lut is not None else None,
                                FILL_IDS[fill], arr, len(eyes), _stream(image)))


def last_exact_rows(image):
    v = ctypes.c_int64(0)
    _check(lib().ds_stereo_last_exact_rows(ctx_for(_dev_index(image)), ctypes.byref(v), _stream(image)))
    return int(v.value)


def last_stats(image):
    """(rows re-rendered by the exact sweep, queue chunks of general pixels) of the last polylines call."""
    v = (ctypes.c_int64 * 2)()
    _check(lib().ds_stereo_last_stats(ctx_for(_dev_index(image)), v, _stream(image)))
    return int(v[0]), int(v[1])


def profile_enable(device_index, enable=True):
    _check(lib().ds_profile_enable(ctx_for(device_index), 1 if enable else 0))


def profile_last_ms(device_index):
    """(render_kernel_ms, exact_fallback_ms) of the most recent ds_stereo_warp on this thread's ctx."""
    a, b = ctypes.c_float(0), ctypes.c_float(0)
    _check(lib().ds_profile_last_ms(ctx_for(device_index), ctypes.byref(a), ctypes.byref(b)))
    return float(a.value), float(b.value)


def copy_view(src_ptr, src_rs, src_is, dst_ptr, dst_rs, dst_is, n, h, row_bytes, like):
    CALLS["ds_copy_view"] += 1
    _check(lib().ds_copy_view(ctx_for(_dev_index(like)), src_ptr, src_rs, src_is, dst_ptr, dst_rs, dst_is, n, h, row_bytes,
                              _stream(like)))


def overlap_red_cyan(im1_ptr, r1, i1, im2_ptr, r2, i2, n, h, w, c, out):
    CALLS["ds_overlap_red_cyan"] += 1
    _check(lib().ds_overlap_red_cyan(ctx_for(_dev_index(out)), im1_ptr, r1, i1, im2_ptr, r2, i2, n, h, w, c, out.data_ptr(),
                                     _stream(out)))


def normalmap(depth, pre_blur, sobel_ksize, post_blur, invert):
    """depth [n,h,w]: uint16 (the funnel's depth maps; the fused kernel), float64 (any other dtype, cast by the caller), float32
    with np.gradient (the reference's float32 evaluation: ds_normalmap_gradient_f32 / ds_normalmap_gradient_blur_f32) or float16
    with np.gradient and no blur (numpy's float16 evaluation: ds_normalmap_gradient_f16)."""
    torch = require_gpu()
    n, h, w = depth.shape
    assert depth.dtype in (torch.uint16, torch.float64, torch.float32, torch.float16) and depth.is_contiguous()
    out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=depth.device)
    if depth.dtype == torch.float16:
        assert int(pre_blur) == 0 and int(post_blur) == 0 and int(sobel_ksize) == 0, "float16 depth: np.gradient without blurs only"
        CALLS["ds_normalmap_gradient_f16"] += 1
        _check(lib().ds_normalmap_gradient_f16(ctx_for(_dev_index(depth)), depth.data_ptr(), n, h, w, 1 if invert else 0,
                                               out.data_ptr(), _stream(depth)))
        return out
    if depth.dtype == torch.float32:
        assert int(sobel_ksize) == 0, "float32 depth: np.gradient only (cv2.Sobel is fed float64)"
        if int(pre_blur) == 0 and int(post_blur) == 0:
            CALLS["ds_normalmap_gradient_f32"] += 1
            _check(lib().ds_normalmap_gradient_f32(ctx_for(_dev_index(depth)), depth.data_ptr(), n, h, w, 1 if invert else 0,
                                                   out.data_ptr(), _stream(depth)))
        else:
            CALLS["ds_normalmap_gradient_blur_f32"] += 1
            _check(lib().ds_normalmap_gradient_blur_f32(ctx_for(_dev_index(depth)), depth.data_ptr(), n, h, w, int(pre_blur), int(post_blur),
                                                        1 if invert else 0, out.data_ptr(), _stream(depth)))
        return out
    fn = lib().ds_normalmap if depth.dtype == torch.uint16 else lib().ds_normalmap_f64
    CALLS["ds_normalmap" if depth.dtype == torch.uint16 else "ds_normalmap_f64"] += 1
    _check(fn(ctx_for(_dev_index(depth)), depth.data_ptr(), n, h, w, int(pre_blur), int(sobel_ksize),
              int(post_blur), 1 if invert else 0, out.data_ptr(), _stream(depth)))
    return out


def depth_to_u16(pred_f32, invert=False, want_norm=False):
    torch = require_gpu()
    n, h, w = pred_f32.shape
    out = torch.empty((n, h, w), dtype=torch.uint16, device=pred_f32.device)
    norm = torch.empty((n, h, w), dtype=torch.float32, device=pred_f32.device) if want_norm else None
    CALLS["ds_depth_to_u16"] += 1
    _check(lib().ds_depth_to_u16(ctx_for(_dev_index(pred_f32)), pred_f32.data_ptr(), n, h, w, 1 if invert else 0,
                                 out.data_ptr(), norm.data_ptr() if want_norm else None, _stream(pred_f32)))
    return (out, norm) if want_norm else out


def convert_to_i16(arr):
    torch = require_gpu()
    assert arr.dtype in (torch.float32, torch.float64) and arr.is_contiguous()
    out = torch.empty(arr.shape, dtype=torch.uint16, device=arr.device)
    CALLS["ds_convert_to_i16"] += 1
    _check(lib().ds_convert_to_i16(ctx_for(_dev_index(arr)), arr.data_ptr(), 1 if arr.dtype == torch.float64 else 0,
                                   arr.numel(), out.data_ptr(), _stream(arr)))
    return out


def colorize_u16(depth, vmin_vmax, lut_rgba):
    """Heat map (include/depthstereo.h: ds_colorize_u16).  depth [n,h,w] uint16, vmin_vmax [n,2] float64, lut_rgba
    [N<=256, 4] uint8, all CUDA tensors.  Returns [n,h,w,4] uint8."""
    torch = require_gpu()
    assert depth.is_cuda and depth.dtype == torch.uint16 and depth.dim() == 3 and depth.is_contiguous()
    n, h, w = depth.shape
    vmm = vmin_vmax.to(device=depth.device, dtype=torch.float64).contiguous()
    lut = lut_rgba.to(device=depth.device, dtype=torch.uint8).contiguous()
    assert tuple(vmm.shape) == (n, 2) and lut.dim() == 2 and lut.shape[1] == 4
    out = torch.empty((n, h, w, 4), dtype=torch.uint8, device=depth.device)
    CALLS["ds_colorize_u16"] += 1
    _check(lib().ds_colorize_u16(ctx_for(_dev_index(depth)), depth.data_ptr(), n, h, w, vmm.data_ptr(), lut.data_ptr(),
                                 int(lut.shape[0]), out.data_ptr(), _stream(depth)))
    return out


class PackedAttentionBias:
    """The bias operand of attention_fwd: made by attention_bias_pack, opaque to the host code."""

    def __init__(self, data, heads, n, npad):
        self.data, self.heads, self.n, self.npad = data, heads, n, npad

    @property
    def dtype(self):
        return self.data.dtype


def attention_bias_pack(bias, npad, dtype):
    """bias [H, n, n] (CUDA tensor, any float dtype, natural units) -> PackedAttentionBias for sequences whose token stride is
    npad (include/depthstereo.h: ds_attention_bias_pack; the operand itself is laid out on npad rounded up to 64)."""
    torch = require_gpu()
    npad = (int(npad) + 63) // 64 * 64
    assert bias.is_cuda and bias.dim() == 3 and bias.shape[1] == bias.shape[2] and dtype in (torch.float16, torch.bfloat16)
    h, n = int(bias.shape[0]), int(bias.shape[1])
    src = bias.detach().float().contiguous()
    out = torch.empty((h * npad * npad,), dtype=dtype, device=bias.device)
    CALLS["ds_attention_bias_pack"] += 1
    _check(lib().ds_attention_bias_pack(ctx_for(_dev_index(src)), src.data_ptr(), h, n, int(npad),
                                        1 if dtype == torch.float16 else 2, out.data_ptr(), _stream(src)))
    return PackedAttentionBias(out, h, n, int(npad))


def attention_fwd(qk, vt, n_valid, scale, bias=None):
    """Fused MFMA attention (include/depthstereo.h: ds_attention_fwd).  qk [B,Np,2,H,64], vt [B,H*64,Np], float16 or
    bfloat16 CUDA tensors; bias: None or a PackedAttentionBias (attention_bias_pack) for the same H, Np and dtype.
    Returns [B,Np,H*64]."""
    torch = require_gpu()
    assert qk.is_cuda and vt.is_cuda and qk.dtype == vt.dtype and qk.dtype in (torch.float16, torch.bfloat16)
    b, npad, two, h, d = qk.shape
    assert two == 2 and d == 64 and npad % 8 == 0 and tuple(vt.shape) == (b, h * 64, npad), (qk.shape, vt.shape)
    qk = qk.contiguous()
    vt = vt.contiguous()
    if bias is not None:
        assert isinstance(bias, PackedAttentionBias), "bias must come from attention_bias_pack"
        assert (bias.heads, bias.npad, bias.dtype) == (h, (npad + 63) // 64 * 64, qk.dtype) and bias.data.device == qk.device
    out = torch.empty((b, npad, h * 64), dtype=qk.dtype, device=qk.device)
    dt = 1 if qk.dtype == torch.float16 else 2
    CALLS["ds_attention_fwd"] += 1
    _check(lib().ds_attention_fwd(ctx_for(_dev_index(qk)), qk.data_ptr(), vt.data_ptr(),
                                  bias.data.data_ptr() if bias is not None else None, out.data_ptr(),
                                  b, npad, h, int(n_valid), float(scale), dt, _stream(qk)))
    return out


def residual_layernorm(x, branch, gamma, ln_weight, ln_bias, eps=1e-6):
    """x_out = x + gamma * branch; h = LayerNorm(x_out) (include/depthstereo.h: ds_residual_layernorm), one pass.
    x, branch [..., C] float16/bfloat16 CUDA tensors (branch may be None: plain LayerNorm).  Returns (x_out, h)."""
    torch = require_gpu()
    assert x.is_cuda and x.dtype in (torch.float16, torch.bfloat16)
    x = x.contiguous()
    c = x.shape[-1]
    rows = x.numel() // c
    h = torch.empty_like(x)
    dt = 1 if x.dtype == torch.float16 else 2
    if branch is None:
        x_out, bp = x, None
    else:
        branch = branch.contiguous()
        assert branch.shape == x.shape and branch.dtype == x.dtype
        x_out = torch.empty_like(x)
        bp = branch.data_ptr()
    g = None if gamma is None else gamma.to(x.dtype).contiguous()
    w, b = ln_weight.to(x.dtype).contiguous(), ln_bias.to(x.dtype).contiguous()
    CALLS["ds_residual_layernorm"] += 1
    _check(lib().ds_residual_layernorm(ctx_for(_dev_index(x)), x.data_ptr(), bp, None if g is None else g.data_ptr(),
                                       w.data_ptr(), b.data_ptr(), None if branch is None else x_out.data_ptr(), h.data_ptr(),
                                       rows, c, float(eps), dt, _stream(x)))
    return x_out, h


def reassemble_readout(proj, clsvec):
    """gelu(proj[:, 1:] + clsvec[:, None]) -> [B, N-1, C] (include/depthstereo.h: ds_reassemble_readout).
    proj [B, N, C], clsvec [B, C]: float16 / bfloat16 CUDA tensors."""
    torch = require_gpu()
    assert proj.is_cuda and proj.dtype in (torch.float16, torch.bfloat16) and proj.dim() == 3 and clsvec.dtype == proj.dtype
    b, n, c = proj.shape
    proj, clsvec = proj.contiguous(), clsvec.contiguous()
    assert tuple(clsvec.shape) == (b, c)
    out = torch.empty((b, n - 1, c), dtype=proj.dtype, device=proj.device)
    CALLS["ds_reassemble_readout"] += 1
    _check(lib().ds_reassemble_readout(ctx_for(_dev_index(proj)), proj.data_ptr(), clsvec.data_ptr(), out.data_ptr(), b, n, c,
                                       1 if proj.dtype == torch.float16 else 2, _stream(proj)))
    return out


def linear_env(**switches):
    """Set (a value) or remove (None) DS_LIN_* switches of the GEMM path in os.environ and make the library re-read them
    (include/depthstereo.h: ds_linear_reload_env -- the library reads them once, not per launch).  Tests and A/B runs only."""
    for k, v in switches.items():
        assert k.startswith("DS_LIN_")
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    _check(lib().ds_linear_reload_env())


def normalmap_selfcheck(k0, stride, count):
    """Number of operands n^2 = (k0 + stride * i) / 2^18, i < count, on which the fused normal-map kernels' square root / reciprocal
    differ from the generic float64 operations (include/depthstereo.h: ds_normalmap_selfcheck): must be 0."""
    torch = require_gpu()
    bad = ctypes.c_ulonglong(0)
    _check(lib().ds_normalmap_selfcheck(ctx_for(torch.cuda.current_device()), ctypes.c_ulonglong(k0), ctypes.c_ulonglong(stride),
                                        ctypes.c_ulonglong(count), ctypes.byref(bad), None))
    return int(bad.value)


def attention_env(**switches):
    """Set (a value) or remove (None) DS_ATT_* switches of ds_attention_fwd in os.environ and make the library re-read them
    (include/depthstereo.h: ds_attention_reload_env).  DS_ATT_GEN=2 / 4 selects the kernel generation.  Tests and A/B runs only."""
    for k, v in switches.items():
        assert k.startswith("DS_ATT_")
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    _check(lib().ds_attention_reload_env())


def linear_supported(x, weight):
    """Shapes ds_linear takes: out_features % 256 == 0, in_features % 128 == 0 (every ViT of the path qualifies)."""
    return weight.shape[0] % 256 == 0 and weight.shape[1] % 128 == 0 and 128 <= weight.shape[1] <= 16384 and x.numel() > 0


def linear(x, weight, bias=None, gelu=False):
    """[gelu](x @ weight.T + bias) by the in-tree MFMA GEMM (include/depthstereo.h: ds_linear).
    x [..., K] float16 / bfloat16 CUDA, weight [N, K], bias [N] or None -> [..., N]."""
    torch = require_gpu()
    assert x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and linear_supported(x, weight)
    k = x.shape[-1]
    n = weight.shape[0]
    assert weight.shape[1] == k
    x2 = x.reshape(-1, k)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    rows = x2.shape[0]
    if rows < 256:                       # the kernel's unit is a 256-row tile: tiny inputs are padded, not special-cased
        x2 = torch.cat([x2, x2.new_zeros((256 - rows, k))])
    w = weight.detach()
    w = w if (w.dtype == x.dtype and w.is_contiguous()) else w.to(x.dtype).contiguous()
    b = None
    if bias is not None:
        b = bias.detach()
        b = b if (b.dtype == x.dtype and b.is_contiguous()) else b.to(x.dtype).contiguous()
    out = torch.empty((x2.shape[0], n), dtype=x.dtype, device=x.device)
    CALLS["ds_linear"] += 1
    _check(lib().ds_linear(ctx_for(_dev_index(x)), x2.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(),
                           out.data_ptr(), x2.shape[0], n, k, n, 1 if gelu else 0,
                           1 if x.dtype == torch.float16 else 2, _stream(x)))
    return out[:rows].view(x.shape[:-1] + (n,))


def linear_residual(x, weight, bias, gamma, residual):
    """residual + [gamma *] (x @ weight.T + bias) (include/depthstereo.h: ds_linear_residual).  x [..., K], residual [..., N]
    float16 / bfloat16 CUDA, at least 256 rows; gamma [N] or None."""
    torch = require_gpu()
    assert x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and linear_supported(x, weight)
    k, n = x.shape[-1], weight.shape[0]
    x2 = x.reshape(-1, k).contiguous()
    r2 = residual.reshape(-1, n).contiguous()
    assert x2.shape[0] == r2.shape[0] >= 256 and r2.dtype == x.dtype

    def prep(t):
        if t is None:
            return None
        t = t.detach()
        return t if (t.dtype == x.dtype and t.is_contiguous()) else t.to(x.dtype).contiguous()
    w, b, g = prep(weight), prep(bias), prep(gamma)
    out = torch.empty_like(r2)
    CALLS["ds_linear_residual"] += 1
    _check(lib().ds_linear_residual(ctx_for(_dev_index(x)), x2.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(),
                                    None if g is None else g.data_ptr(), r2.data_ptr(), out.data_ptr(), x2.shape[0], n, k,
                                    1 if x.dtype == torch.float16 else 2, _stream(x)))
    return out.view(residual.shape)


def linear_readout_supported(x_padded, w_tok):
    """Shapes ds_linear_readout takes: x [B, Np, K] contiguous with B * Np >= 256, w_tok [N % 256 == 0, K % 128 == 0]."""
    return (x_padded.dim() == 3 and x_padded.is_contiguous() and x_padded.shape[0] * x_padded.shape[1] >= 256
            and w_tok.shape[0] % 256 == 0 and w_tok.shape[1] == x_padded.shape[2] and x_padded.shape[2] % 128 == 0
            and 128 <= x_padded.shape[2] <= 16384 and x_padded.shape[0] * x_padded.shape[1] * x_padded.shape[1] < (1 << 32))


def linear_readout(x_padded, n_tokens, w_tok, cls_vec):
    """GELU(x[b, t] @ w_tok.T + cls_vec[b]) for the tokens t = 1 .. n_tokens - 1 of every image, token-major = the NHWC patch grid
    (include/depthstereo.h: ds_linear_readout).  x_padded [B, Np, K] float16 / bfloat16 CUDA (the encoder's padded block output),
    w_tok [N, K], cls_vec [B, N] -> [B, n_tokens - 1, N] (a view of a buffer with one dummy row behind it)."""
    torch = require_gpu()
    assert x_padded.is_cuda and x_padded.dtype in (torch.float16, torch.bfloat16) and linear_readout_supported(x_padded, w_tok)
    b, npad, k = x_padded.shape
    n = w_tok.shape[0]
    assert 2 <= n_tokens <= npad and tuple(cls_vec.shape) == (b, n)
    w = w_tok.detach()
    w = w if (w.dtype == x_padded.dtype and w.is_contiguous()) else w.to(x_padded.dtype).contiguous()
    cv = cls_vec if (cls_vec.dtype == x_padded.dtype and cls_vec.is_contiguous()) else cls_vec.to(x_padded.dtype).contiguous()
    out = torch.empty((b * (n_tokens - 1) + 1, n), dtype=x_padded.dtype, device=x_padded.device)
    CALLS["ds_linear_readout"] += 1
    _check(lib().ds_linear_readout(ctx_for(_dev_index(x_padded)), x_padded.data_ptr(), w.data_ptr(), cv.data_ptr(), out.data_ptr(),
                                   b, npad, n_tokens, n, k, 1 if x_padded.dtype == torch.float16 else 2, _stream(x_padded)))
    return out[:b * (n_tokens - 1)].view(b, n_tokens - 1, n)


def conv_transpose_shuffle_supported(layer, x):
    """ConvTranspose2d layers ds_linear_shuffle takes: kernel == stride (square), no padding / output padding / groups / dilation,
    in_channels % 128 == 0, out_channels % 8 == 0, stride^2 * out_channels % 256 == 0, at least 256 input pixels."""
    ks, st = tuple(layer.kernel_size), tuple(layer.stride)
    return (ks == st and ks[0] == ks[1] and 1 <= ks[0] <= 8 and tuple(layer.padding) == (0, 0) and tuple(layer.output_padding) == (0, 0)
            and tuple(layer.dilation) == (1, 1) and layer.groups == 1 and layer.in_channels % 128 == 0 and 128 <= layer.in_channels <= 16384
            and layer.out_channels % 8 == 0 and (ks[0] * ks[0] * layer.out_channels) % 256 == 0 and x.dim() == 4
            and x.shape[1] == layer.in_channels and x.shape[0] * x.shape[2] * x.shape[3] >= 256)


def conv_transpose_shuffle(layer, x):
    """layer(x) for an nn.ConvTranspose2d with kernel == stride on a float16 / bfloat16 CUDA activation, channels_last in and out
    (include/depthstereo.h: ds_linear_shuffle).  The [(kH, kW, out), in] weight image and the per-tap bias are cached on the module."""
    torch = require_gpu()
    assert x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and conv_transpose_shuffle_supported(layer, x)
    x = x.contiguous(memory_format=torch.channels_last)
    b, c, h, w = x.shape
    s, co = layer.stride[0], layer.out_channels
    key = (layer.weight.data_ptr(), layer.weight._version, x.dtype, None if layer.bias is None else layer.bias._version)
    hit = getattr(layer, "_ds_shuffle", None)
    if hit is None or hit[0] != key:
        if hit is not None:
            from . import vit_mi355x as _vm
            _vm.cache_evicted()                 # a live hipGraph may still read the old weight image
        wg = layer.weight.detach().permute(2, 3, 1, 0).reshape(s * s * co, c).to(x.dtype).contiguous()      # [in, out, kH, kW] -> [(kH, kW, out), in]
        bg = None if layer.bias is None else layer.bias.detach().to(x.dtype).repeat(s * s).contiguous()
        hit = (key, wg, bg)
        if not torch.is_grad_enabled():
            layer._ds_shuffle = hit
    _, wg, bg = hit
    out = torch.empty((b, h * s, w * s, co), dtype=x.dtype, device=x.device)
    xr = x.permute(0, 2, 3, 1)                   # the NHWC memory as [b, h, w, c]: contiguous
    CALLS["ds_linear_shuffle"] += 1
    _check(lib().ds_linear_shuffle(ctx_for(_dev_index(x)), xr.data_ptr(), wg.data_ptr(), None if bg is None else bg.data_ptr(), out.data_ptr(),
                                   b * h * w, c, w, s, co, 1 if x.dtype == torch.float16 else 2, _stream(x)))
    return out.permute(0, 3, 1, 2)               # logical NCHW, channels_last memory


def gconv3x3_supported(x, weight, stride, padding, dilation, groups):
    """What ds_gconv3x3_nhwc_f32 takes: a float32 channels_last activation, 3x3, stride 1, padding 1, no dilation, channels == out
    channels, 8 / 16 / 32 channels per group, channels % 32 == 0."""
    torch = _torch()
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and groups > 1 and tuple(weight.shape[2:]) == (3, 3)):
        return False
    c = x.shape[1]
    cpg = c // groups
    return (weight.shape[0] == c and weight.shape[1] == cpg and cpg in (8, 16, 32) and c % 32 == 0 and tuple(stride) == (1, 1)
            and tuple(padding) == (1, 1) and tuple(dilation) == (1, 1) and x.is_contiguous(memory_format=torch.channels_last))


def gconv_weight_image(weight, groups):
    """torch's grouped weight [out, in / groups, 3, 3] -> [groups][9 taps][in / groups][out / groups] float32 (what the kernel's scalar
    loads walk: the outputs of one (group, tap, input channel) are consecutive)."""
    c, cpg = weight.shape[0], weight.shape[1]
    w = weight.detach().float().reshape(groups, c // groups, cpg, 9)          # [g][co][ci][tap]
    return w.permute(0, 3, 2, 1).contiguous()                                   # [g][tap][ci][co]


def gconv3x3(x, w_gtio, bias, relu, cpg):
    """[relu](grouped 3x3 convolution + bias) on a float32 channels_last activation (include/depthstereo.h: ds_gconv3x3_nhwc_f32)."""
    torch = require_gpu()
    b, c, h, w = x.shape
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    assert w_gtio.dtype == torch.float32 and w_gtio.is_contiguous() and w_gtio.numel() == (c // cpg) * 9 * cpg * cpg
    bb = None if bias is None else bias.detach().float().contiguous()
    out = torch.empty_like(x, memory_format=torch.channels_last)
    CALLS["ds_gconv3x3_nhwc_f32"] += 1
    _check(lib().ds_gconv3x3_nhwc_f32(ctx_for(_dev_index(x)), x.data_ptr(), w_gtio.data_ptr(), None if bb is None else bb.data_ptr(), out.data_ptr(),
                                      b, h, w, c, cpg, 1 if relu else 0, _stream(x)))
    return out


def add_relu(a, b):
    """relu(a + b) for two float32 CUDA tensors of one shape and one memory layout (include/depthstereo.h: ds_add_relu_f32)."""
    torch = require_gpu()
    assert a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape and a.stride() == b.stride()
    assert a.numel() % 4 == 0 and (a.is_contiguous() or a.is_contiguous(memory_format=torch.channels_last))
    out = torch.empty_like(a)                       # preserves the (dense) strides
    assert out.stride() == a.stride()
    CALLS["ds_add_relu_f32"] += 1
    _check(lib().ds_add_relu_f32(ctx_for(_dev_index(a)), a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream(a)))
    return out


def bias_act_f32_ok(x, bias, res=None):
    """What ds_bias_act_f32 takes: a float32 CUDA activation in channels_last memory (or a dense [rows, C] matrix), C % 4 == 0, a
    float32 bias of C values, res laid out like x."""
    torch = require_gpu()
    if not (x.is_cuda and x.dtype == torch.float32 and bias is not None and bias.dtype == torch.float32 and bias.is_cuda and bias.numel() == x.shape[1]):
        return False
    if not (x.dim() == 4 and x.shape[1] % 4 == 0 and x.is_contiguous(memory_format=torch.channels_last) and x.numel() > 0):
        return False
    return res is None or (res.dtype == torch.float32 and res.shape == x.shape and res.is_cuda and res.is_contiguous(memory_format=torch.channels_last))


def bias_act_f32(x, bias, relu=False, res=None):
    """[relu]((x + bias[c]) [+ res]) IN PLACE on a float32 channels_last activation (include/depthstereo.h: ds_bias_act_f32)."""
    assert bias_act_f32_ok(x, bias, res)
    b, c, h, w = x.shape
    bias = bias.contiguous()
    CALLS["ds_bias_act_f32"] += 1
    _check(lib().ds_bias_act_f32(ctx_for(_dev_index(x)), x.data_ptr(), bias.data_ptr(), None if res is None else res.data_ptr(), x.data_ptr(),
                                 b * h * w, c, 1 if relu else 0, _stream(x)))
    return x


def _relu_cat_layout(a, b):
    """0: both channels_last; 1: a channels_last, b NCHW-contiguous; 2: both NCHW-contiguous; None: not taken."""
    torch = require_gpu()
    if not (a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 4 and b.dim() == 4
            and a.shape[0] == b.shape[0] and a.shape[2:] == b.shape[2:] and a.numel() > 0 and b.numel() > 0):
        return None
    cl = torch.channels_last
    # (a tensor with one channel or a 1 x 1 plane is contiguous in both senses: NCHW is tested first for b, channels_last first for a)
    if b.is_contiguous():
        lay = 2 if (a.is_contiguous() and not a.is_contiguous(memory_format=cl)) else (1 if a.is_contiguous(memory_format=cl) else None)
        if lay is not None and a.shape[1] % 32 == 0 and b.shape[1] % 32 == 0:
            return lay
    if a.is_contiguous(memory_format=cl) and b.is_contiguous(memory_format=cl) and a.shape[1] % 4 == 0 and b.shape[1] % 4 == 0:
        return 0
    return None


def relu_cat_f32_ok(a, b):
    return _relu_cat_layout(a, b) is not None


def relu_cat_f32(a, b):
    """relu(torch.cat([a, b], 1)) for float32 activations, one pass (include/depthstereo.h: ds_relu_cat_f32): channels_last out of two
    channels_last inputs; NCHW when b is NCHW (a channels_last or NCHW) -- the memory format torch.cat itself would produce."""
    torch = require_gpu()
    lay = _relu_cat_layout(a, b)
    assert lay is not None
    n, ca, h, w = a.shape
    cb = b.shape[1]
    out = torch.empty((n, ca + cb, h, w), dtype=a.dtype, device=a.device, memory_format=torch.channels_last if lay == 0 else torch.contiguous_format)
    CALLS["ds_relu_cat_f32"] += 1
    _check(lib().ds_relu_cat_f32(ctx_for(_dev_index(a)), a.data_ptr(), b.data_ptr(), out.data_ptr(), n, h * w, ca, cb, lay, _stream(a)))
    return out


KT_KINDS = {"linear_gelu": 0, "attention": 1, "linear_residual": 2, "linear": 3, "linear_vt": 4, "conv3x3": 5, "linear_readout": 6,
            "linear_shuffle": 7, "normalmap": 8}
KT_RAGGED = 12


def kernel_timer_enable(device_index, enable=True):
    """In-step kernel timers of the C ABI (include/depthstereo.h: ds_kernel_timer_enable): event pairs around the launches."""
    _check(lib().ds_kernel_timer_enable(ctx_for(device_index), 1 if enable else 0))


def kernel_timer_read(device_index, kind):
    """(launches timed, sum of their durations in ms) of one kind: a name of KT_KINDS, or "<name>+ragged" for its ragged round."""
    base, _, rag = kind.partition("+")
    k = KT_KINDS[base] + (KT_RAGGED if rag == "ragged" else 0)
    n, ms = ctypes.c_int64(), ctypes.c_double()
    _check(lib().ds_kernel_timer_read(ctx_for(device_index), k, ctypes.byref(n), ctypes.byref(ms)))
    return int(n.value), float(ms.value)


def kernel_timer_read_each(device_index, kind):
    """Durations (ms) of the timed launches of one kind in launch order (include/depthstereo.h: ds_kernel_timer_read_each)."""
    base, _, rag = kind.partition("+")
    k = KT_KINDS[base] + (KT_RAGGED if rag == "ragged" else 0)
    buf = (ctypes.c_float * 1024)()
    n = ctypes.c_int64()
    _check(lib().ds_kernel_timer_read_each(ctx_for(device_index), k, buf, 1024, ctypes.byref(n)))
    return [float(buf[i]) for i in range(min(int(n.value), 1024))]


def row_stats(x, eps):
    """Per row {rstd, -mean * rstd} (float32 [rows, 2]) of x [..., C]: the statistics half of a LayerNorm, for the GEMMs that fold
    the LayerNorm into their epilogue (include/depthstereo.h: ds_row_stats).  C in 384 / 768 / 1024 / 1536."""
    torch = require_gpu()
    assert x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.is_contiguous()
    c = x.shape[-1]
    rows = x.numel() // c
    out = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    CALLS["ds_row_stats"] += 1
    _check(lib().ds_row_stats(ctx_for(_dev_index(x)), x.data_ptr(), out.data_ptr(), rows, c, float(eps),
                              1 if x.dtype == torch.float16 else 2, _stream(x)))
    return out


def linear_ln(x, w_scaled, colsum, bias, stats, gelu=False):
    """[gelu](LN(x) @ W.T + b) with the LayerNorm folded into the GEMM (include/depthstereo.h: ds_linear_ln): x [..., K] is the
    UN-normalised input, w_scaled = W * ln_weight [N, K], colsum = w_scaled.sum(1) float32 [N], bias = b + W @ ln_bias [N] or None,
    stats = row_stats(x)."""
    torch = require_gpu()
    assert x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and linear_supported(x, w_scaled) and x.is_contiguous()
    k, n = x.shape[-1], w_scaled.shape[0]
    rows = x.numel() // k
    assert rows >= 256 and w_scaled.dtype == x.dtype and w_scaled.is_contiguous() and colsum.dtype == torch.float32 and colsum.numel() == n
    assert stats.dtype == torch.float32 and stats.numel() == 2 * rows and (bias is None or (bias.dtype == x.dtype and bias.is_contiguous()))
    out = torch.empty((rows, n), dtype=x.dtype, device=x.device)
    CALLS["ds_linear_ln"] += 1
    _check(lib().ds_linear_ln(ctx_for(_dev_index(x)), x.data_ptr(), w_scaled.data_ptr(), colsum.data_ptr(), None if bias is None else bias.data_ptr(),
                              stats.data_ptr(), out.data_ptr(), rows, n, k, n, 1 if gelu else 0, 1 if x.dtype == torch.float16 else 2, _stream(x)))
    return out.view(x.shape[:-1] + (n,))


def linear_vt_ln(w_v_scaled, colsum, x, stats):
    """V^T with the LayerNorm folded in (include/depthstereo.h: ds_linear_vt_ln): x [B, Np, K] un-normalised -> [B, C, Np]."""
    torch = require_gpu()
    assert x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and linear_vt_supported(w_v_scaled, x) and x.is_contiguous()
    b, npad, k = x.shape
    c = w_v_scaled.shape[0]
    assert w_v_scaled.dtype == x.dtype and w_v_scaled.is_contiguous() and colsum.dtype == torch.float32 and colsum.numel() == c
    assert stats.dtype == torch.float32 and stats.numel() == 2 * b * npad
    out = torch.empty((b, c, npad), dtype=x.dtype, device=x.device)
    CALLS["ds_linear_vt_ln"] += 1
    _check(lib().ds_linear_vt_ln(ctx_for(_dev_index(x)), w_v_scaled.data_ptr(), colsum.data_ptr(), x.data_ptr(), stats.data_ptr(), out.data_ptr(),
                                 c, b, npad, k, 1 if x.dtype == torch.float16 else 2, _stream(x)))
    return out


def linear_vt_supported(w_v, h):
    """Shapes ds_linear_vt takes: h [B, Np, K] with Np % 8 == 0 and (B * Np) % 256 == 0, w_v [C >= 256, K], K % 128 == 0."""
    return (h.dim() == 3 and h.shape[1] % 8 == 0 and (h.shape[0] * h.shape[1]) % 256 == 0 and w_v.shape[0] >= 256
            and w_v.shape[1] == h.shape[2] and h.shape[2] % 128 == 0 and 128 <= h.shape[2] <= 16384
            and h.shape[0] * h.shape[1] * h.shape[1] < (1 << 32))


def linear_vt(w_v, h):
    """V^T = W_v . h^T per batch element, written transposed by the GEMM's epilogue (include/depthstereo.h: ds_linear_vt).
    w_v [C, K], h [B, Np, K] float16 / bfloat16 CUDA -> [B, C, Np]."""
    torch = require_gpu()
    assert h.is_cuda and h.dtype in (torch.float16, torch.bfloat16) and linear_vt_supported(w_v, h)
    b, npad, k = h.shape
    c = w_v.shape[0]
    h2 = h if h.is_contiguous() else h.contiguous()
    w = w_v.detach()
    w = w if (w.dtype == h.dtype and w.is_contiguous()) else w.to(h.dtype).contiguous()
    out = torch.empty((b, c, npad), dtype=h.dtype, device=h.device)
    CALLS["ds_linear_vt"] += 1
    _check(lib().ds_linear_vt(ctx_for(_dev_index(h)), w.data_ptr(), h2.data_ptr(), out.data_ptr(), c, b, npad, k,
                              1 if h.dtype == torch.float16 else 2, _stream(h)))
    return out


def conv3x3_supported(conv, x):
    """What ds_conv3x3_nhwc takes: 3x3, stride 1, zero padding 1, no groups / dilation, in % 128 == 0, out % 128 == 0 (residual
    operands only with out % 256 == 0)."""
    return (tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros'
            and conv.in_channels % 128 == 0 and conv.out_channels % 128 == 0 and x.dim() == 4 and x.shape[1] == conv.in_channels
            and x.shape[0] * x.shape[2] * x.shape[3] >= 256)


def conv3x3(conv, x, relu=False, res1=None, res2=None, relu_in=False):
    """[relu](conv(x) + bias [+ res1] [+ res2]) for a 3x3 nn.Conv2d on a float16 / bfloat16 CUDA activation, channels_last
    in and out (include/depthstereo.h: ds_conv3x3_nhwc).  The [out, 3, 3, in] weight image is cached on the module.
    relu_in: conv(relu(x)) with the maximum taken inside the kernel (needs relu, no residual operands, out % 256 == 0)."""
    torch = require_gpu()
    assert x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and conv3x3_supported(conv, x)
    x = x.contiguous(memory_format=torch.channels_last)
    b, c, h, w = x.shape
    key = (conv.weight.data_ptr(), conv.weight._version, x.dtype, None if conv.bias is None else conv.bias._version)
    hit = getattr(conv, "_ds_ohwi", None)
    if hit is None or hit[0] != key:
        if hit is not None:
            from . import vit_mi355x as _vm
            _vm.cache_evicted()                 # a live hipGraph may still read the old weight image
        wk = conv.weight.detach().to(x.dtype).permute(0, 2, 3, 1).contiguous()
        bk = None if conv.bias is None else conv.bias.detach().to(x.dtype).contiguous()
        hit = conv._ds_ohwi = (key, wk, bk)
    _, wk, bk = hit
    out = torch.empty((b, conv.out_channels, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    for r in (res1, res2):
        assert r is None or (r.shape == out.shape and r.dtype == x.dtype and r.is_contiguous(memory_format=torch.channels_last))
    CALLS["ds_conv3x3_nhwc"] += 1
    _check(lib().ds_conv3x3_nhwc(ctx_for(_dev_index(x)), x.data_ptr(), wk.data_ptr(), None if bk is None else bk.data_ptr(),
                                 None if res1 is None else res1.data_ptr(), None if res2 is None else res2.data_ptr(),
                                 out.data_ptr(), b, h, w, c, conv.out_channels, (2 if relu else 0) | (4 if relu_in else 0),
                                 1 if x.dtype == torch.float16 else 2, _stream(x)))
    return out


def bias_act(x, bias, relu=False, res1=None, res2=None, inplace=True):
    """[relu](x + bias[c] [+ res1] [+ res2]) for an NCHW-shaped float16/bfloat16 CUDA tensor in channels_last memory format
    (include/depthstereo.h: ds_bias_act_nhwc); res1 / res2 must have x's shape and memory format.  In place by default."""
    torch = require_gpu()
    assert x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.dim() == 4 and x.shape[1] % 8 == 0
    assert x.is_contiguous(memory_format=torch.channels_last)
    for r in (res1, res2):
        assert r is None or (r.shape == x.shape and r.dtype == x.dtype and r.is_contiguous(memory_format=torch.channels_last))
    out = x if inplace else torch.empty_like(x)
    b = bias.detach().to(x.dtype).contiguous()
    CALLS["ds_bias_act_nhwc"] += 1
    _check(lib().ds_bias_act_nhwc(ctx_for(_dev_index(x)), x.data_ptr(), b.data_ptr(), None if res1 is None else res1.data_ptr(),
                                  None if res2 is None else res2.data_ptr(), out.data_ptr(), x.numel(), x.shape[1], 1 if relu else 0,
                                  1 if x.dtype == torch.float16 else 2, _stream(x)))
    return out


def group_norm_supported(x, groups):
    """What ds_group_norm_nchw takes: a contiguous NCHW float16 / bfloat16 CUDA activation whose planes hold a multiple of 8 pixels."""
    torch = _torch()
    return (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.dim() == 4 and x.is_contiguous() and x.shape[1] % groups == 0
            and (x.shape[2] * x.shape[3]) % 8 == 0 and x.shape[0] * groups <= 4096 and x.data_ptr() % 16 == 0)


def group_norm(x, groups, weight, bias, eps, relu=False, res=None):
    """[relu](group_norm(x) [+ res]) (include/depthstereo.h: ds_group_norm_nchw); weight / bias in any float type (cast to x's)."""
    torch = require_gpu()
    assert group_norm_supported(x, groups)
    assert res is None or (res.shape == x.shape and res.dtype == x.dtype and res.is_contiguous() and res.data_ptr() % 16 == 0)
    out = torch.empty_like(x)
    g, b = weight.detach().to(x.dtype).contiguous(), bias.detach().to(x.dtype).contiguous()
    CALLS["ds_group_norm_nchw"] += 1
    _check(lib().ds_group_norm_nchw(ctx_for(_dev_index(x)), x.data_ptr(), g.data_ptr(), b.data_ptr(), None if res is None else res.data_ptr(),
                                    out.data_ptr(), x.shape[0], x.shape[1], x.shape[2] * x.shape[3], int(groups), float(eps), 1 if relu else 0,
                                    1 if x.dtype == torch.float16 else 2, _stream(x)))
    return out


def boost_blend(dst, rects, coefs, preds, mask_template):
    """Blend all Boost patches into dst (float32 [H, W] CUDA, in place), in list order (include/depthstereo.h:
    ds_boost_blend).  rects: list of (x0, y0, w, h); coefs: list of (p0, p1); preds float32 [P, S, S]; mask_template
    float32 [M, M]."""
    torch = require_gpu()
    assert dst.is_cuda and dst.dtype == torch.float32 and dst.dim() == 2 and dst.stride(1) == 1
    n = len(rects)
    if n == 0:
        return dst
    preds = preds.contiguous()
    mask_template = mask_template.contiguous()
    assert preds.dtype == torch.float32 and preds.shape[0] == n and preds.shape[1] == preds.shape[2]
    assert mask_template.dtype == torch.float32 and mask_template.shape[0] == mask_template.shape[1]
    rec = np.zeros(n, dtype=[('x0', '<i4'), ('y0', '<i4'), ('w', '<i4'), ('h', '<i4'), ('p0', '<f8'), ('p1', '<f8')])
    for i, (r, c) in enumerate(zip(rects, coefs)):
        rec[i] = (int(r[0]), int(r[1]), int(r[2]), int(r[3]), float(c[0]), float(c[1]))
    buf = torch.from_numpy(rec.view(np.uint8).copy()).to(dst.device)
    CALLS["ds_boost_blend"] += 1
    _check(lib().ds_boost_blend(ctx_for(_dev_index(dst)), dst.data_ptr(), dst.stride(0), dst.shape[0], dst.shape[1],
                                buf.data_ptr(), n, preds.data_ptr(), preds.shape[1], mask_template.data_ptr(),
                                mask_template.shape[0], _stream(dst)))
    return dst


def preprocess_bicubic(images_u8, size_hw, mean, std, flip=True, dtype=None):
    """uint8 [B,H,W,3] CUDA -> normalised network input [B,3,h,w] (channels_last memory) in one pass
    (include/depthstereo.h: ds_preprocess_bicubic): channel flip, / 255, bicubic resize, (x - mean) / std, cast."""
    torch = require_gpu()
    assert images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[3] == 3
    dtype = torch.float32 if dtype is None else dtype
    code = {torch.float16: 1, torch.bfloat16: 2, torch.float32: 3}[dtype]
    x = images_u8.contiguous()
    b, h, w, _ = x.shape
    oh, ow = int(size_hw[0]), int(size_hw[1])
    out = torch.empty((b, 3, oh, ow), dtype=dtype, device=x.device, memory_format=torch.channels_last)
    m = (ctypes.c_float * 3)(*[float(v) for v in (mean if hasattr(mean, "__len__") else (mean,) * 3)])
    s = (ctypes.c_float * 3)(*[float(v) for v in (std if hasattr(std, "__len__") else (std,) * 3)])
    CALLS["ds_preprocess_bicubic"] += 1
    _check(lib().ds_preprocess_bicubic(ctx_for(_dev_index(x)), x.data_ptr(), out.data_ptr(), b, h, w, oh, ow, 1 if flip else 0, m, s, code,
                                       _stream(x)))
    return out


def upsample_bilinear(x, size=None, scale_factor=None, align_corners=True):
    """F.interpolate(x, mode="bilinear") for an NCHW-shaped float16/bfloat16 CUDA tensor in channels_last memory format
    (include/depthstereo.h: ds_upsample_bilinear_nhwc).  Returns a channels_last tensor."""
    torch = require_gpu()
    b, c, ih, iw = x.shape
    if size is None:
        size = (int(ih * scale_factor), int(iw * scale_factor))
    oh, ow = int(size[0]), int(size[1])
    assert x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and c % 8 == 0
    xin = x.contiguous(memory_format=torch.channels_last)
    out = torch.empty((b, c, oh, ow), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    CALLS["ds_upsample_bilinear_nhwc"] += 1
    _check(lib().ds_upsample_bilinear_nhwc(ctx_for(_dev_index(x)), xin.data_ptr(), out.data_ptr(), b, c, ih, iw, oh, ow,
                                           1 if align_corners else 0, 1 if x.dtype == torch.float16 else 2, _stream(x)))
    return out


def dpt_head_tail(x, size, conv3, conv1, relu_out=True):
    """upsample(x, size, bilinear, align_corners=True) -> conv3 (3x3, 128->32) -> ReLU -> conv1 (1x1, 32->1) -> ReLU, fused
    (include/depthstereo.h: ds_dpt_head_tail).  x: NCHW-shaped [B,128,h,w] float16/bfloat16 CUDA tensor (any memory format;
    channels_last is zero-copy).  conv3 / conv1: the nn.Conv2d modules.  Returns [B, 1, H, W]."""
    torch = require_gpu()
    b, c, ih, iw = x.shape
    assert c == 128 and tuple(conv3.weight.shape) == (32, 128, 3, 3) and tuple(conv1.weight.shape) == (1, 32, 1, 1)
    xin = x.contiguous(memory_format=torch.channels_last)
    # the repacked weights live ON the module (they die with it; a recycled id() of another model's head can never hit),
    # keyed by the storage and version of every parameter that went into them
    key = tuple((p.data_ptr(), p._version) for p in (conv3.weight, conv3.bias, conv1.weight, conv1.bias)) + (x.dtype, x.device)
    hit = getattr(conv3, "_ds_head_cache", None)
    if hit is None or hit[0] != key:
        if hit is not None:
            from . import vit_mi355x as _vm
            _vm.cache_evicted()
        w = conv3.weight.detach().to(x.dtype)                                   # [co, ci, ky, kx]
        # fragment order: [tap = ky*3+kx][slice s][half][co][j] with ci = 16 s + 8 half + j
        wf = w.permute(2, 3, 1, 0).reshape(9, 8, 2, 8, 32).permute(0, 1, 2, 4, 3).contiguous()
        hit = (key, wf, conv3.bias.detach().float().contiguous(), conv1.weight.detach().float().reshape(32).contiguous(),
               float(conv1.bias.detach().float().item()))
        conv3._ds_head_cache = hit
    _, wf, b2, w3, b3 = hit
    oh, ow = int(size[0]), int(size[1])
    out = torch.empty((b, 1, oh, ow), dtype=x.dtype, device=x.device)
    CALLS["ds_dpt_head_tail"] += 1
    _check(lib().ds_dpt_head_tail(ctx_for(_dev_index(x)), xin.data_ptr(), b, ih, iw, oh, ow, wf.data_ptr(), b2.data_ptr(),
                                  w3.data_ptr(), b3, 1 if relu_out else 0, out.data_ptr(),
                                  1 if x.dtype == torch.float16 else 2, _stream(x)))
    return out
