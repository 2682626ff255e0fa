"""Drop-in for the hot path of the reference's ``src/core.py``: ``core_generation_funnel`` (alias
``run_depthmap``), ``CoreGenerationFunnelInp``, ``convert_to_i16``, ``convert_i16_to_rgb``.

Reference: src/core.py:44-58 (i16 helpers), :61-80 (option bundle), :83-349 (the funnel generator).
Same signature, same yielded ``(input_index, kind, result)`` triples in the same order, same option
names/defaults (``common_constants.GenerationOptions``).  What differs is where the pixels are made:
depth post-processing, uint16 quantisation, stereo views and normal maps run as HIP kernels on an
MI355X (``libdepthstereo_hip.so``) and stay in HBM between stages; host code only converts to PIL at
the moment a result is yielded.

Out of scope here: background removal, the inpainted (3D-photo) mesh and
video assembly.  Asking for them raises NotImplementedError instead of silently doing nothing.
"""
import gc

import numpy as np
from contextlib import nullcontext as _nullcontext
from PIL import Image

from . import _native
from .common_constants import GenerationOptions
from .common_constants import GenerationOptions as go
from .depthmap_generation import ModelHolder
from .normalmap_generation import create_normalmap_batch
from .stereoimage_generation import create_stereoimages_batch

SCRIPT_NAME = "DepthMap (MI355X-native hot path)"

model_holder = ModelHolder()

_OUT_OF_SCOPE = (go.GEN_REMBG, go.GEN_INPAINTED_MESH)


def convert_to_i16(arr):
    """reference: src/core.py:44-50 -- clip(arr*65536 + 1e-4, 0, 65535.9) truncated to uint16, on the device.
    ndarray in -> ndarray out; float32 input is processed in float32, anything else in float64 (numpy rules)."""
    torch = _native.require_gpu()
    a = np.asarray(arr)
    if a.dtype != np.float32:
        a = a.astype(np.float64)
    t = torch.from_numpy(np.array(a, order='C')).cuda()
    return _native.convert_to_i16(t).cpu().numpy()


def convert_i16_to_rgb(image, like):
    """reference: src/core.py:52-58 -- three 8-bit channels holding image/256 (truncated)."""
    output = np.zeros_like(like)
    v = (np.asarray(image) / 256.0)
    output[:, :, 0] = v
    output[:, :, 1] = v
    output[:, :, 2] = v
    return output


class CoreGenerationFunnelInp:
    """Option bundle of the funnel (reference: src/core.py:61-80).  Keys are lower-cased option names or
    GenerationOptions members; unknown keys are dropped silently, missing ones take the enum default."""

    def __init__(self, values):
        if isinstance(values, CoreGenerationFunnelInp):
            values = values.values
        lowered = {}
        for k, v in values.items():
            lowered[(k.name if isinstance(k, GenerationOptions) else k).lower()] = v
        self.values = {}
        for setting in GenerationOptions:
            name = setting.name.lower()
            self.values[name] = lowered[name] if name in lowered else setting.df

    def __getitem__(self, item):
        if isinstance(item, GenerationOptions):
            return self.values[item.name.lower()]
        return self.values[item]

    def __getattr__(self, item):
        if item == 'values':
            raise AttributeError(item)
        return self[item]


def _custom_depth_to_float(dp, image):
    """Custom depthmap ingest (reference: src/core.py:145-174).  Host code: PIL resampling must be PIL's."""
    if isinstance(dp, Image.Image):
        if dp.width != image.width or dp.height != image.height:
            try:
                dp = dp.resize((image.width, image.height), Image.Resampling.LANCZOS)
            except Exception:
                dp = dp.resize((image.width, image.height))
        if len(dp.getbands()) == 1:
            out = np.asarray(dp, dtype="float")
            out_max = out.max()
            if out_max < 256:
                bit_depth = 8
            elif out_max < 65536:
                bit_depth = 16
            else:
                bit_depth = 32
            out = out / (2.0 ** bit_depth)
        else:
            out = np.asarray(dp, dtype="float")[:, :, 0]
            out = out / 256.0
        return out
    out = np.asarray(dp, dtype="float")
    assert image.height == out.shape[0], "Custom depthmap height mismatch"
    assert image.width == out.shape[1], "Custom depthmap width mismatch"
    return out


import ctypes as _ctypes
import os as _os
import threading as _threading
import time as _time

# device batch of the funnel: up to this many pixels (16 x 1024^2 by default), at most 64 images.  Measured on 32 x 1024^2
# through dpt_beit_large_512: one group of 32 -> 176 pairs/s, two pipelined groups of 16 -> 289 pairs/s (the PIL conversion
# of a group overlaps the next group's device work)
FUNNEL_BATCH_PIXELS = int(_os.environ.get("DS_FUNNEL_BATCH_PIXELS", 16 << 20))


# ---- PIL -> pinned staging without the interpreter lock ----------------------------------------------------------------------
# np.asarray(PIL image) is Image.tobytes(): Pillow's "raw" encoder packs the image's 32-bit pixel store (RGBX) to 3 bytes per
# pixel in 64 KB pieces under the GIL, the pieces are joined (a second copy) and then copied into the staging buffer (a third):
# ~2.5 ms per 1024^2 image, and a thread pool cannot share it -- 80 of the 105 ms a 32-image call took in round 3.  Pillow >= 11.2
# exports an image's pixel store through the Arrow C data interface (a stable C ABI: struct ArrowArray) without a copy; the
# funnel takes the pointer, copies the 4-byte pixels into pinned memory with memmove (which releases the GIL, so the host pool
# really runs in parallel), uploads them as they are and drops the fourth byte on the device.
# The export is only SAFE for images that own their pixel store: Pillow 12.2 crashes the process when asked to export an image
# that maps foreign memory (Image.fromarray of an L / RGBA / I;16 array, Image.frombuffer: `readonly` is set).  Mode "RGB" can
# never be mapped (the packed 3-byte layout is not Pillow's storage layout: Image._MAPMODES), so RGB and not readonly it is;
# an image stored in several blocks (larger than Pillow's 16 MB block) refuses the export with a ValueError; everything else --
# other modes, older Pillow -- takes np.asarray as before.  DS_FUNNEL_ARROW=0 switches the path off.
class _ArrowArray(_ctypes.Structure):
    pass


_ArrowArray._fields_ = [("length", _ctypes.c_int64), ("null_count", _ctypes.c_int64), ("offset", _ctypes.c_int64),
                        ("n_buffers", _ctypes.c_int64), ("n_children", _ctypes.c_int64), ("buffers", _ctypes.POINTER(_ctypes.c_void_p)),
                        ("children", _ctypes.POINTER(_ctypes.POINTER(_ArrowArray))), ("dictionary", _ctypes.c_void_p),
                        ("release", _ctypes.c_void_p), ("private_data", _ctypes.c_void_p)]
_capsule_pointer = _ctypes.pythonapi.PyCapsule_GetPointer
_capsule_pointer.restype = _ctypes.c_void_p
_capsule_pointer.argtypes = [_ctypes.py_object, _ctypes.c_char_p]
FUNNEL_ARROW = _os.environ.get("DS_FUNNEL_ARROW", "1") != "0"


def _arrow_layout_verified():
    """Once per process: export a small RGB image whose pixels are known and compare the exported bytes with them (the layout the
    funnel's memmove assumes -- RGBX, 4 bytes per pixel, line 0 first, no padding -- was validated on Pillow 12.2; any other
    Pillow has to show it here or takes np.asarray)."""
    try:
        probe = Image.new("RGB", (5, 3))
        probe.putdata([(10 * i + 1, 10 * i + 2, 10 * i + 3) for i in range(15)])
        if not hasattr(probe, "__arrow_c_array__") or getattr(probe, "readonly", 1):
            return False
        caps = probe.__arrow_c_array__()
        arr = _ArrowArray.from_address(_capsule_pointer(caps[1], b"arrow_array"))
        if arr.n_children != 1 or arr.length != 15 or arr.offset != 0:
            return False
        pix = arr.children[0].contents
        if pix.n_buffers != 2 or pix.length != 60 or pix.offset != 0 or not pix.buffers[1]:
            return False
        raw = _ctypes.string_at(int(pix.buffers[1]), 60)
        return all(raw[4 * i:4 * i + 3] == bytes((10 * i + 1, 10 * i + 2, 10 * i + 3)) for i in range(15))
    except Exception:
        return False


_ARROW_LAYOUT_OK = FUNNEL_ARROW and _arrow_layout_verified()


def _rgbx_pixels(im):
    """(address of the image's H*W 4-byte RGBX pixels, keep-alive object) for an RGB image that owns its pixel store, else None.
    The address is valid while the keep-alive object (the Arrow capsules, which hold the image's blocks) is referenced."""
    if not FUNNEL_ARROW or im.mode != "RGB" or getattr(im, "readonly", 1) or not hasattr(im, "__arrow_c_array__") or not _ARROW_LAYOUT_OK:
        return None
    try:
        # the memmove below assumes ONE block of h lines of exactly 4 * w bytes starting at line 0: true for Pillow's default line
        # alignment of 1; a host application that raised Image.core.set_alignment pads the lines
        if hasattr(Image.core, "get_alignment") and Image.core.get_alignment() != 1:
            return None
        capsules = im.__arrow_c_array__()
        arr = _ArrowArray.from_address(_capsule_pointer(capsules[1], b"arrow_array"))
        if arr.n_children != 1 or arr.length != im.width * im.height or arr.offset != 0:
            return None
        pix = arr.children[0].contents                         # fixed_size_list<uint8>[4]: the child holds the bytes
        if pix.n_buffers != 2 or pix.length != 4 * arr.length or pix.offset != 0 or not pix.buffers[1]:
            return None
        return int(pix.buffers[1]), capsules
    except Exception:           # multi-block images (ValueError), a Pillow whose export differs: the ordinary path
        return None


def _postprocess_batch(pred, invert, inp):
    """Depth post-processing of src/core.py:189-206 for a batch of same-size predictions, on the device.
    pred: float32 cuda tensor [B,H,W].  Returns (out [B,H,W] float32 or float64 in [0,1] ready for convert_to_i16 -- None
    for the fused no-clip branch --, prediction_copy [B,H,W] or None, broken: list of bool, one host sync per BATCH)."""
    torch = _native._torch()
    pred = pred.contiguous()
    flat = pred.flatten(1)
    pmin, pmax = flat.min(1).values, flat.max(1).values
    broken = [not bool(v) for v in ((pmax - pmin).abs() > np.finfo("float").eps).cpu().tolist()]      # :190
    out = pred * -1 if invert else pred.clone()                                                  # :192-195
    prediction_copy = out
    if inp[go.CLIPDEPTH]:
        if inp[go.CLIPDEPTH_MODE] == 'Range':                                                    # :197-198
            omin = out.flatten(1).min(1).values.view(-1, 1, 1)
            omax = out.flatten(1).max(1).values.view(-1, 1, 1)
            out = (out - omin) / (omax - omin)
            out = torch.clamp(out, min=float(inp[go.CLIPDEPTH_FAR]), max=float(inp[go.CLIPDEPTH_NEAR]))
        elif inp[go.CLIPDEPTH_MODE] == 'Outliers':                                               # :199-201
            # np.percentile on the device: exact order statistics by bisection on the float32 bit pattern, numpy's
            # linear interpolation; np.clip(float32 array, float64 bounds) promotes to float64 (NumPy >= 2)
            from .video_mode import _LOCAL, _global_percentiles
            rows = []
            for i in range(out.shape[0]):
                if broken[i]:
                    rows.append(out[i].double())
                    continue
                fb, nb = _global_percentiles(out[i], [float(inp[go.CLIPDEPTH_FAR]) * 100.0, float(inp[go.CLIPDEPTH_NEAR]) * 100.0], _LOCAL)   # per image: never a collective
                rows.append(torch.clamp(out[i].double(), min=fb, max=nb))
            out = torch.stack(rows)
    return out, prediction_copy, broken


def _wants_reference_f16():
    """`reference_f16_postproc` (ops / ModelHolder.update_settings; DS_REFERENCE_F16_POSTPROC=1): reproduce the reference's
    float16 depth post-processing for the networks whose prediction it receives as a float16 array (MiDaS ids 1-4 in half
    precision: src/depthmap_generation.py:268-275, :484-497).  DEFAULT OFF -- a defined corner, see DESIGN.md: by default a
    half-precision network's prediction is post-processed in float32 here (65 536 depth levels instead of float16's ~2 k per
    octave)."""
    want = getattr(model_holder, "reference_f16_postproc", None)
    if want is None:
        want = _os.environ.get("DS_REFERENCE_F16_POSTPROC", "0") != "0"
    if not want:
        return False
    dm = model_holder.depth_model
    if model_holder.pix2pix_model is not None:               # Boost never runs the base network in half (:271)
        return False
    if getattr(dm, "returns_f16", False):                    # a registered predictor that says so
        return True
    return getattr(dm, "model_type", None) in (1, 2, 3, 4) and not getattr(dm, "no_half", True)


def _postprocess_f16(pred, invert, inp):
    """src/core.py:189-211 on a float16 prediction with NumPy 1.x's promotion (defined corners, states the
    rules): min / max / subtract / divide as float16 operations (torch computes half arithmetic in float32 and rounds once:
    the correctly rounded float16 result), then convert_to_i16 in float64.  pred: float32 [B,H,W] holding float16 values.
    Returns (uint16 [B,H,W], broken)."""
    torch = _native._torch()
    if inp[go.CLIPDEPTH] and inp[go.CLIPDEPTH_MODE] != 'Range':
        raise NotImplementedError("reference_f16_postproc reproduces the float16 path for CLIPDEPTH off and 'Range' only "
                                  "(np.percentile of a float16 array is not restated)")
    out = pred.to(torch.float16)
    flat = out.flatten(1)
    pmin, pmax = flat.min(1).values, flat.max(1).values
    broken = [not bool(v) for v in ((pmax - pmin).abs().double() > np.finfo("float").eps).cpu().tolist()]      # :190
    if invert:
        out = -out

    def norm01(t):
        lo = t.flatten(1).min(1).values.view(-1, 1, 1)
        hi = t.flatten(1).max(1).values.view(-1, 1, 1)
        return (t - lo) / (hi - lo)
    if inp[go.CLIPDEPTH]:
        far = torch.tensor(float(inp[go.CLIPDEPTH_FAR]), dtype=torch.float16, device=out.device)
        near = torch.tensor(float(inp[go.CLIPDEPTH_NEAR]), dtype=torch.float16, device=out.device)
        out = torch.minimum(torch.maximum(norm01(out), far), near)
    d16 = _native.convert_to_i16(norm01(out).double().contiguous())
    return d16, broken


class _HostStaging:
    """Pinned host buffers of the funnel, two generations (the results of group k are converted to PIL while group k+1
    runs on the device).  A buffer is (re)allocated only when a larger one is needed."""

    def __init__(self):
        self.buf = {}

    def get(self, gen, tag, shape, dtype):
        torch = _native._torch()
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        key = (gen, tag)
        b = self.buf.get(key)
        if b is None or b.numel() < n:
            b = self.buf[key] = torch.empty((max(n, 1),), dtype=torch.uint8, pin_memory=True)
        return b[:n].view(dtype).view(shape)


_staging = _HostStaging()


def _split_run(n, limit):
    """Group sizes for a run of n same-size images with a device batch of at most `limit`.  DS_FUNNEL_PLAN="24,8" (measurements):
    the listed sizes first (each capped by 64), `limit` for what is left."""
    plan = [int(x) for x in _os.environ.get("DS_FUNNEL_PLAN", "").split(",") if x.strip().isdigit() and int(x) > 0]
    sizes = []
    for p in plan:
        if n <= 0:
            break
        sizes.append(min(p, n, 64))
        n -= sizes[-1]
    while n > 0:
        sizes.append(min(limit, n))
        n -= sizes[-1]
    return sizes


def _plan_groups(inputimages, inputdepthmaps, batchable):
    """Consecutive images of one size and mode (and one kind of depth source) form a run; a run is cut into device batches."""
    runs, cur, key = [], [], None
    for i, im in enumerate(inputimages):
        k = (im.size, im.mode, inputdepthmaps[i] is not None)
        if cur and k != key:
            runs.append(cur)
            cur = []
        key = k
        cur.append(i)
    if cur:
        runs.append(cur)
    groups = []
    for run in runs:
        im = inputimages[run[0]]
        limit = max(1, min(64, FUNNEL_BATCH_PIXELS // max(1, im.size[0] * im.size[1]))) if batchable else 1
        at = 0
        for sz in _split_run(len(run), limit):
            groups.append(run[at:at + sz])
            at += sz
    return groups


# DS_FUNNEL_TRACE=1 (tools/funnel_timeline.py): device timestamps of every group's stages (events with timing on the streams that
# carry them) and host timestamps of every unit's conversion, collected per funnel call in FUNNEL_TRACE.  Off: plain events.
_TRACE = bool(_os.environ.get("DS_FUNNEL_TRACE"))
FUNNEL_TRACE = {}


def _mark(g, name, stream=None):
    if not _TRACE:
        return
    torch = _native._torch()
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(stream) if stream is not None else ev.record()
    g.setdefault("trace", []).append((name, ev, _time.perf_counter()))


def _launch_group(gen, idxs, inputimages, inputdepthmaps, inp, device, stats=None):
    """Enqueue everything a group needs on the current stream -- H2D, network, post-processing, stereo, normal map, heat
    map, D2H into pinned buffers -- and return the handles; nothing here waits for the device except the post-processing
    branches that must know which predictions are flat (one sync per batch) and the percentile bisection of 'Outliers'."""
    torch = _native._torch()
    b = len(idxs)
    images = [inputimages[i] for i in idxs]
    w, h = images[0].size
    g = {"idxs": idxs, "images": images, "host": {}, "pred_host": None, "broken": [False] * b}
    _tm = [_time.perf_counter()]

    def lap(name):               # host seconds of this stage of the enqueue path (decode / forward / post), per funnel call
        if stats is not None:
            now = _time.perf_counter()
            stats[name] = stats.get(name, 0.0) + (now - _tm[0])
            _tm[0] = now
    custom = inputdepthmaps[idxs[0]] is not None
    want_stereo = inp[go.GEN_STEREO]

    def h2d(st):
        """Pinned staging -> device on the UPLOAD streams (round 6): the inputs of group k+1 cross the link beside the kernels of
        group k instead of in front of its own forward on the compute stream, which waits for the copies' events only.  Two halves
        on two streams: one hipMemcpyAsync runs on one SDMA engine at about half the link's rate."""
        main = torch.cuda.current_stream(device)
        up0 = _copy_stream(device, "up0")
        with torch.cuda.stream(up0):
            t = torch.empty(st.shape, dtype=st.dtype, device=device)
        n = st.shape[0]
        cuts = [0, n] if n < 2 else [0, n // 2, n]
        for i in range(len(cuts) - 1):
            up = _copy_stream(device, "up%d" % i)
            with torch.cuda.stream(up):
                t[cuts[i]:cuts[i + 1]].copy_(st[cuts[i]:cuts[i + 1]], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            main.wait_event(ev)
            if i > 0:
                t.record_stream(up)
        t.record_stream(main)
        return t

    def upload(arrays, tag, dtype):
        st = _staging.get(gen, tag, (b,) + arrays[0].shape, dtype)
        stn = st.numpy()
        for j, a in enumerate(arrays):
            np.copyto(stn[j], a)
        return h2d(st)

    def upload_pixels(tag, to_array, rgb=False):
        """PIL -> uint8 array -> pinned staging (decoded and copied on the host thread pool) -> device.
        rgb: the result is the images' RGB pixels [b,h,w,3]; RGB images that own their pixel store go through _rgbx_pixels."""
        if rgb:
            srcs = [_rgbx_pixels(im) for im in images]
            if all(x is not None for x in srcs):
                st = _staging.get(gen, tag + "x", (b, h, w, 4), torch.uint8)
                base, nbytes = st.data_ptr(), h * w * 4
                if b > 1:
                    list(_host_pool().map(lambda j: _ctypes.memmove(base + j * nbytes, srcs[j][0], nbytes), range(b)))
                else:
                    _ctypes.memmove(base, srcs[0][0], nbytes)
                del srcs
                return h2d(st)[..., :3].contiguous()
        first = to_array(images[0])
        st = _staging.get(gen, tag, (b,) + first.shape, torch.uint8)
        stn = st.numpy()
        np.copyto(stn[0], first)
        if b > 1:
            list(_host_pool().map(lambda j: np.copyto(stn[j], to_array(images[j])), range(1, b)))
        return h2d(st)

    img_t = None
    ndim0 = 3 if len(images[0].getbands()) > 1 else 2                                            # np.array(image).ndim
    if want_stereo and ndim0 != 3:                                                               # :252 np.array(image)
        # a single-channel image fails in the reference's stereo step (:55 `h, w, c = original_image.shape`), i.e. AFTER the
        # image's depth outputs were yielded: the group is rendered without stereo and _emit_group raises at that point
        g["stereo_error"] = ValueError('not enough values to unpack (expected 3, got %d)' % ndim0)
        want_stereo = False
    if want_stereo:
        # (RGB images that own their pixel store go through Pillow's Arrow export + memmove: _rgbx_pixels above; everything
        # else takes np.asarray)
        img_t = upload_pixels("img", lambda im: np.asarray(im, dtype=np.uint8), rgb=images[0].mode == "RGB")
    mesh_source = None
    if custom:
        outs = [_custom_depth_to_float(inputdepthmaps[i], inputimages[i]) for i in idxs]         # :145-174 (host: PIL)
        out_t = upload([np.asarray(o, dtype=np.float64) for o in outs], "cdepth", torch.float64)
        d16 = _native.convert_to_i16(out_t)                                                      # :211
        mesh_source = out_t
    else:
        if inp[go.NET_SIZE_MATCH]:                                                               # :177-184
            net_width, net_height = (w + 31) // 32 * 32, (h + 31) // 32 * 32
        else:
            net_width, net_height = inp[go.NET_WIDTH], inp[go.NET_HEIGHT]
        if images[0].mode == "RGB" and img_t is not None:
            rgb_t = img_t
        else:
            rgb_t = upload_pixels("rgb", lambda im: np.asarray(im.convert("RGB"), dtype=np.uint8), rgb=images[0].mode == "RGB")
        lap("launch_decode")
        _mark(g, "inputs on the device")
        pred, invert = model_holder.get_raw_prediction_batch(images, rgb_t, net_width, net_height)
        _mark(g, "forward done")
        lap("launch_forward")
        if pred is None:                 # Boost sharded over ranks (ModelHolder.boost_group): only the group's rank 0 renders
            g["skip"] = True
            return g
        pred = pred.to(device=device, dtype=torch.float32)
        mesh_source = pred
        if _wants_reference_f16():
            d16, broken = _postprocess_f16(pred, invert, inp)
            g["broken"] = broken
            if inp[go.DO_OUTPUT_DEPTH_PREDICTION]:            # the reference yields its float16 array (:194-195)
                pc = (pred * -1 if invert else pred).to(torch.float16)
                g["pred_host"] = (pc, _staging.get(gen, "pred16", tuple(pc.shape), torch.float16))
            if any(broken):
                keep = torch.tensor([0 if x else 1 for x in broken], device=device, dtype=torch.int16).view(-1, 1, 1)
                d16 = (d16.view(torch.int16) * keep).view(torch.uint16)
        elif not inp[go.CLIPDEPTH] and not inp[go.DO_OUTPUT_DEPTH_PREDICTION]:
            d16 = _native.depth_to_u16(pred, invert)         # fused: per-image min/max -> normalise -> uint16 (:189-211)
        else:
            out_t, prediction_copy, broken = _postprocess_batch(pred, invert, inp)
            g["broken"] = broken
            if inp[go.DO_OUTPUT_DEPTH_PREDICTION]:
                ph = _staging.get(gen, "pred", tuple(prediction_copy.shape), torch.float32)
                g["pred_host"] = (prediction_copy, ph)        # copied with the other results, on the copy stream
            if out_t.dtype == torch.float64:                                                     # 'Outliers': numpy promoted
                omin = out_t.flatten(1).min(1).values.view(-1, 1, 1)
                omax = out_t.flatten(1).max(1).values.view(-1, 1, 1)
                d16 = _native.convert_to_i16(((out_t - omin) / (omax - omin)).contiguous())      # :202, :211
            else:
                d16 = _native.depth_to_u16(out_t.to(torch.float32).contiguous(), False)          # :202, :211
            if any(broken):                                                                      # :203-206 zeros
                keep = torch.tensor([0 if x else 1 for x in broken], device=device, dtype=torch.int16).view(-1, 1, 1)
                d16 = (d16.view(torch.int16) * keep).view(torch.uint16)
    g["d16"], g["mesh_source"], g["img_t"], g["custom"] = d16, mesh_source, img_t, custom

    # results go back on a COPY STREAM of their own: the device-to-host copies of this group (11 MB per 1024^2 unit) then run
    # beside the next group's kernels instead of in front of them.  The copies are enqueued once, at the end of the group.
    pending_downloads = []

    def download(t, tag):
        pending_downloads.append((t, tag))

    if inp[go.DO_OUTPUT_DEPTH]:
        download(d16, "depth")
    # DS_FUNNEL_POST_STREAM=1: the per-pixel kernels of the group (float64 VALU / LDS bound) on a stream of their own, beside the next
    # group's forward (MFMA bound) -- what bench.py --overlap does for the resident loop.  Off by default: see DESIGN.md (hardware queues)
    post = _copy_stream(device, "post") if _os.environ.get("DS_FUNNEL_POST_STREAM", "0") != "0" else None
    if post is not None:
        fwd_done = torch.cuda.Event()
        fwd_done.record()
        post.wait_event(fwd_done)
        for t in (d16, img_t):
            if t is not None:
                t.record_stream(post)
    with (torch.cuda.stream(post) if post is not None else _nullcontext()):
        if want_stereo:                                                                              # :251-259
            modes = inp[go.STEREO_MODES]
            stereo = create_stereoimages_batch(img_t, d16, inp[go.STEREO_DIVERGENCE], inp[go.STEREO_SEPARATION], modes,
                                               inp[go.STEREO_BALANCE], inp[go.STEREO_OFFSET_EXPONENT], inp[go.STEREO_FILL_ALGO])
            for c in range(len(stereo)):
                download(stereo[c], "stereo%d" % c)
            g["n_stereo"] = len(stereo)
        if inp[go.GEN_NORMALMAP]:                                                                    # :261-269
            download(create_normalmap_batch(
                d16,
                inp[go.NORMALMAP_PRE_BLUR_KERNEL] if inp[go.NORMALMAP_PRE_BLUR] else None,
                inp[go.NORMALMAP_SOBEL_KERNEL] if inp[go.NORMALMAP_SOBEL] else None,
                inp[go.NORMALMAP_POST_BLUR_KERNEL] if inp[go.NORMALMAP_POST_BLUR] else None,
                inp[go.NORMALMAP_INVERT]), "normalmap")
        if inp[go.GEN_HEATMAP]:                                                                      # :271-274
            from .heatmap import colorize_batch
            download(colorize_batch(d16), "heatmap")
        _mark(g, "per-pixel kernels done")
        ready = torch.cuda.Event()
        ready.record()                                        # everything the copies read has been enqueued (main or post stream)
    cs = _copy_stream(device)
    # round 6: the copies go out in CHUNKS of units with an event behind each, and the conversion of every unit to PIL is handed to
    # the render pool right here: a unit is converted as soon as ITS chunk has landed -- beside the copies of the later chunks and
    # the enqueueing of the next group -- instead of after the whole group (the last group's conversion used to be a tail of
    # ~10 ms per 16 units with the device idle)
    nchunks = min(b, max(1, int(_os.environ.get("DS_FUNNEL_CHUNKS", 4))))
    bounds = [(b * c) // nchunks for c in range(nchunks + 1)]
    g["chunk_of"] = [next(c for c in range(nchunks) if bounds[c] <= j < bounds[c + 1]) for j in range(b)]
    g["chunk_done"] = []
    # ONE copy stream: the device-to-host copies of a group already run at the link's rate on it (176 MB in 3.4 ms), and HIP maps its
    # streams onto four hardware queues -- with four copy streams the compute stream shared a queue with one of them and the next
    # group's forward waited 3.8 ms for copies it does not depend on (tools/funnel_timeline.py; DS_FUNNEL_COPY_STREAMS for A/B runs)
    nstreams = max(1, min(nchunks, int(_os.environ.get("DS_FUNNEL_COPY_STREAMS", 1))))
    with torch.cuda.stream(cs):
        cs.wait_event(ready)
        if g["pred_host"] is not None:
            src, ph = g["pred_host"]
            ph.copy_(src, non_blocking=True)
            src.record_stream(cs)
            g["pred_host"] = ph
        pred_done = torch.cuda.Event()
        pred_done.record()
    # largest results first inside a chunk, an event behind every (chunk, result): the pair's conversion (the longest) starts while
    # the chunk's smaller results are still crossing the link
    hbufs = sorted(((t, tag, _staging.get(gen, tag, tuple(t.shape), t.dtype)) for t, tag in pending_downloads),
                   key=lambda e: -e[0][0].numel() * e[0].element_size())
    g["landed"] = {}
    for c in range(nchunks):
        cs_c = _copy_stream(device, c % nstreams)
        with torch.cuda.stream(cs_c):
            if c < nstreams:
                cs_c.wait_event(ready)
                if c > 0 and g["pred_host"] is not None:
                    cs_c.wait_event(pred_done)                # (every chunk's event also covers the group's prediction copy)
            for t, tag, hbuf in hbufs:
                hbuf[bounds[c]:bounds[c + 1]].copy_(t[bounds[c]:bounds[c + 1]], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                g["landed"][(c, tag)] = ev
            if not hbufs:                                     # (nothing to copy: mesh-only calls)
                ev = torch.cuda.Event()
                ev.record()
            g["chunk_done"].append(ev)
            _mark(g, "chunk %d on the host" % c)
    for t, tag, hbuf in hbufs:
        for i in range(nstreams):
            t.record_stream(_copy_stream(device, i))          # the allocator must not hand t's memory out while the copies run
        g["host"][tag] = hbuf
    g["inp"] = inp
    g["futures"] = [[_render_pool().submit(_run_task, g, j, t) for t in _unit_tasks(g, j)] for j in range(b)] if b > 1 else None
    lap("launch_post")
    return g


_pool = None
_copy_streams = {}
# wall-clock seconds the host spent per stage of the last FINISHED funnel call (bench.py's funnel leg reports them): 'launch' =
# decoding + staging + enqueueing the groups, 'wait' = blocked on a group's results.  Every call accumulates into a dict of its
# own and publishes it here when its generator finishes, so interleaved generators cannot mix their numbers.  FUNNEL_STATS is the call
# that ENDED last ('call' = its serial number, 'finished' = whether its generator ran to completion); a call that starts does not
# touch it, so a generator that interleaves with another one cannot wipe the other's published numbers; the last few calls stay
# readable by serial number in FUNNEL_STATS_BY_CALL.
FUNNEL_STATS = {}
FUNNEL_STATS_BY_CALL = {}
_funnel_calls = [0]
_funnel_calls_lock = _threading.Lock()


def _copy_stream(device, index=0):
    """Copy streams of a device: 0 .. for the results (device to host), "up" for the inputs.  One hipMemcpyAsync runs on ONE SDMA
    engine (~25 GB/s of the link's ~55): the chunks of a group's results go out on several streams side by side."""
    torch = _native._torch()
    key = (str(device), index)
    st = _copy_streams.get(key)
    if st is None:
        st = _copy_streams[key] = torch.cuda.Stream(device=device)
    return st


_pil_alloc_lock = _threading.Lock()
_pil_alloc = {"active": 0, "before": None, "set": None}      # funnel calls in flight, the caller's blocks_max, the value the funnel set


def _tune_pil_allocator():
    """Pillow allocates every image from fresh 16 MB blocks and returns them to the OS on release (blocks_max = 0): each of the
    funnel's results (8 MB for a 1024 x 2048 pair) then pays a first-touch page fault per 4 KB page, serialised on the process's
    memory-map lock however many threads convert -- measured here 9.9 ms -> 4.1 ms per unit (depth + pair + normal map) once
    freed blocks are kept for reuse.  The setting is process wide, so it is SCOPED to the funnel calls in flight and REFERENCE
    COUNTED (interleaved generators share it): the first call raises it to DS_PIL_BLOCKS_MAX blocks (default 64 = at most 1 GB
    retained while calls run; 0 leaves Pillow alone), the last one to finish puts the caller's value back -- which also releases
    the retained blocks -- and only if the value is still the one the funnel set (a host application that changed it meanwhile
    keeps its choice).  Returns True when this call holds a reference to give back."""
    try:
        want = int(_os.environ.get("DS_PIL_BLOCKS_MAX", 64))
        if want <= 0:
            return False
        with _pil_alloc_lock:
            if _pil_alloc["active"] == 0:
                before = Image.core.get_blocks_max()
                if before >= want:
                    return False                           # the host application already keeps at least as many: nothing to scope
                Image.core.set_blocks_max(want)
                _pil_alloc["before"], _pil_alloc["set"] = before, want
            elif _pil_alloc["set"] is None:
                return False
            _pil_alloc["active"] += 1
            return True
    except Exception:           # an older Pillow without the arena controls: nothing to tune
        return False


def _restore_pil_allocator(held):
    if not held:
        return
    with _pil_alloc_lock:
        _pil_alloc["active"] -= 1
        if _pil_alloc["active"] > 0:
            return
        before, was_set = _pil_alloc["before"], _pil_alloc["set"]
        _pil_alloc["before"] = _pil_alloc["set"] = None
        try:
            if Image.core.get_blocks_max() == was_set:     # still ours: give the caller's value back
                Image.core.set_blocks_max(before)
        except Exception:
            pass


def _host_pool():
    """Threads that turn pinned result buffers into PIL images (numpy copies release the GIL)."""
    global _pool
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max_workers=min(32, (_os.cpu_count() or 4)), thread_name_prefix="ds-funnel")
    return _pool


def _to_pil(a):
    """A view of a pinned result buffer -> a PIL image that owns its pixels (the buffer is reused two groups later).
    Image.fromarray ALIASES the array for the modes PIL can map in place (L, RGBA, I;16 ...: Image._MAPMODES) and UNPACKS --
    i.e. already copies -- 3-channel RGB into its 32-bit pixel store: an explicit copy in front of it is needed for the first
    kind only.  (For a 1024 x 2048 stereo pair that second pass was 6 MB read + 6 MB written per image on the host.)"""
    if a.ndim == 3 and a.shape[2] == 3:
        return Image.fromarray(a)
    return Image.fromarray(a.copy())


_rpool = None


def _render_pool():
    """Threads that turn pinned result buffers into PIL images (a pool of its own: its workers also WAIT for their chunk's copy
    event, and must not starve the decode / upload work _host_pool does for the next group)."""
    global _rpool
    if _rpool is None:
        from concurrent.futures import ThreadPoolExecutor
        _rpool = ThreadPoolExecutor(max_workers=min(32, (_os.cpu_count() or 4)), thread_name_prefix="ds-funnel-render")
    return _rpool


def _unit_tasks(g, j):
    """The conversions of unit j of a launched group as independent tasks, in the reference's output order (:208-274): (result buffer
    the task reads, callable returning (kind, result)).  One task per OUTPUT, not per unit: a 1024 x 2048 pair takes ~2 ms to unpack
    into PIL's pixel store, the normal map ~1 ms -- side by side the last unit of a group is out ~3 ms after its bytes landed
    instead of ~5 (tools/funnel_timeline.py)."""
    inp = g["inp"]
    host = g["host"]
    image = g["images"][j]
    tasks = []
    if g["pred_host"] is not None and not g["broken"][j]:
        tasks.append((None, lambda: ('depth_prediction', g["pred_host"].numpy()[j].copy())))
    if inp[go.DO_OUTPUT_DEPTH]:                                                              # :240-249
        def depth():
            img_output = host["depth"].numpy()[j]
            img_depth = np.bitwise_not(img_output) if inp[go.OUTPUT_DEPTH_INVERT] else img_output   # cv2.bitwise_not
            if inp[go.OUTPUT_DEPTH_COMBINE]:
                axis = 1 if inp[go.OUTPUT_DEPTH_COMBINE_AXIS] == 'Horizontal' else 0
                return ('concat_depth', Image.fromarray(np.concatenate((image, convert_i16_to_rgb(img_depth, np.asarray(image))), axis=axis)))
            return ('depth', Image.fromarray(img_depth.copy()))
        tasks.append(("depth", depth))
    if g.get("stereo_error") is not None:
        return tasks                                     # the reference raised here, inside create_stereoimages
    if inp[go.GEN_STEREO]:
        for c in range(g["n_stereo"]):
            tasks.append(("stereo%d" % c, lambda c=c: (inp[go.STEREO_MODES][c], _to_pil(host["stereo%d" % c].numpy()[j]))))
    if inp[go.GEN_NORMALMAP]:
        tasks.append(("normalmap", lambda: ('normalmap', _to_pil(host["normalmap"].numpy()[j]))))
    if inp[go.GEN_HEATMAP]:
        tasks.append(("heatmap", lambda: ('heatmap', _to_pil(host["heatmap"].numpy()[j]))))
    return tasks


def _run_task(g, j, task):
    tag, fn = task
    c = g["chunk_of"][j]
    # the task's own result of the unit's chunk has landed (the prediction buffer, one copy for the whole group, went first)
    (g["landed"].get((c, tag)) or g["chunk_done"][c]).synchronize()
    r = fn()
    if _TRACE:
        g.setdefault("converted", {})[j] = _time.perf_counter()
    return r


def _render_unit(g, j):
    return [_run_task(g, j, t) for t in _unit_tasks(g, j)]


def _emit_group(g, outpath, inp, device, stats):
    """Yield a finished group's results image by image in the reference's order (:208-306).  The pinned buffers are reused
    two groups later, so every result is copied out of them (PIL owns its pixels); the conversions were handed to the render
    pool when the group was launched (_unit_tasks: each waits for its own chunk of copies), the generator hands out what is done."""
    torch = _native._torch()
    if g.get("skip"):
        return
    futures = g.get("futures")

    def result_of(j):
        _t0 = _time.perf_counter()
        r = [f.result() for f in futures[j]] if futures is not None else _render_unit(g, j)
        stats["wait"] = stats.get("wait", 0.0) + (_time.perf_counter() - _t0)
        return r

    if _TRACE:
        FUNNEL_TRACE.setdefault("groups", []).append(g)
    try:
        for j, count in enumerate(g["idxs"]):
            image = g["images"][j]
            for kind, res in result_of(j):
                yield count, kind, res
            if g.get("stereo_error") is not None:
                raise g["stereo_error"]
            if inp[go.GEN_SIMPLE_MESH]:                                                              # :277-306
                from . import mesh_generation as mg
                mt = inp[go.MODEL_TYPE]              # callers pass the numeric id; the option's default is a display name
                depthi = mg.mesh_depth(g["mesh_source"][j], mt if isinstance(mt, int) else -1, bool(inp[go.BOOST]), g["custom"])
                rgb_t = torch.from_numpy(np.array(image.convert('RGB'), dtype=np.uint8, order='C')).to(device)
                verts, faces, colors = mg.create_mesh_arrays(rgb_t, depthi, keep_edges=not inp[go.SIMPLE_MESH_OCCLUDE],
                                                             spherical=bool(inp[go.SIMPLE_MESH_SPHERICAL]))
                fn = mg.unique_filename(outpath, 'depthmap', 'obj', 'simple')
                yield count, 'simple_mesh', mg.write_obj(fn, verts.cpu().numpy(), faces.cpu().numpy(), colors.cpu().numpy())
    finally:
        if futures is not None:
            for f in (f_ for fs in futures for f_ in fs):    # (also when the consumer stops early) the buffers must be
                f.result()                                   # free before the next group reuses them


def core_generation_funnel(outpath, inputimages, inputdepthmaps, inputnames, inp, ops=None):
    """Generator yielding ``(input_index, kind, result)`` (reference: src/core.py:83-349).

    Same signature, same triples in the same order as the reference's per-image loop (:133); what differs is the schedule:
    consecutive images of one size form a device batch (H2D of the uint8 pixels through pinned memory, ONE batched network
    forward, batched depth -> uint16 -> stereo / normal map / heat map kernels, asynchronous D2H into pinned buffers), and
    the next batch is enqueued before the results of the current one are converted to PIL, so host conversion and device
    work overlap.  COMPUTE_DEVICE is accepted and ignored: the hot path exists on the GPU only."""
    if len(inputimages) == 0 or inputimages[0] is None:
        return
    if inputdepthmaps is None or len(inputdepthmaps) == 0:
        inputdepthmaps = [None for _ in range(len(inputimages))]
    inputdepthmaps_complete = all([x is not None for x in inputdepthmaps])

    inp = CoreGenerationFunnelInp(inp)
    for opt in _OUT_OF_SCOPE:
        if inp[opt]:
            raise NotImplementedError(f"{opt.name} belongs to a subsystem that is out of scope of the MI355X hot path "
                                      "(SURVEY.md section 8); use the reference for it")
    if ops is None:
        ops = {}
    model_holder.update_settings(**ops)

    torch = _native.require_gpu()        # the hot path has no CPU implementation, whatever COMPUTE_DEVICE says
    device = torch.device('cuda', torch.cuda.current_device())
    pil_blocks_held = _tune_pil_allocator()
    with _funnel_calls_lock:
        _funnel_calls[0] += 1
        stats = {"finished": False, "call": _funnel_calls[0]}
    _t_start = _time.perf_counter()
    pending = launched = None
    if _TRACE:
        FUNNEL_TRACE.clear()
        _mark(FUNNEL_TRACE, "call")

    try:
        if not inputdepthmaps_complete:
            model_holder.ensure_models(inp[go.MODEL_TYPE], device, inp[go.BOOST], inp[go.TILING_MODE])
        for count in range(0, len(inputimages)):
            if inputimages[count].mode == 'I':                      # :135-137
                inputimages[count].point(lambda p: p * 0.0039063096, mode='RGB')
                inputimages[count] = inputimages[count].convert('RGB')
        # Boost renders one image at a time by construction (its own patch batches inside estimateboost)
        groups = _plan_groups(inputimages, inputdepthmaps, batchable=not inp[go.BOOST])
        pending = launched = None
        for gi, idxs in enumerate(groups):
            # group k+1 is enqueued before group k is handed out (host conversion overlaps device work); whatever goes wrong
            # while enqueueing it (a bad image, out of memory) must not swallow the finished results of group k: the
            # reference's per-image loop (:133-329) would have yielded them before reaching the failing image
            try:
                _t0 = _time.perf_counter()
                launched, failure = _launch_group(gi & 1, idxs, inputimages, inputdepthmaps, inp, device, stats), None
                stats["launch"] = stats.get("launch", 0.0) + (_time.perf_counter() - _t0)
            except Exception as e:          # noqa: BLE001 -- re-raised below, after the results that precede it
                launched, failure = None, e
            if pending is not None:
                yield from _emit_group(pending, outpath, inp, device, stats)
            if failure is not None:
                raise failure
            pending = launched
        if pending is not None:
            yield from _emit_group(pending, outpath, inp, device, stats)
        stats["groups"] = len(groups)
        stats["finished"] = True
    except Exception as e:
        if 'out of memory' in str(e).lower():                                                  # :308-326
            suggestion = "out of GPU memory, could not generate depthmap! " \
                         "Here are some suggestions to work around this issue:\n"
            if inp[go.MODEL_TYPE] != 6:
                suggestion += " * Use a different model (generally, more memory-consuming models produce better depthmaps)\n"
            if not inp[go.BOOST]:
                suggestion += " * Reduce net size (this could reduce quality)\n"
            raise Exception(suggestion)
        raise e
    finally:
        # published whether the generator ran to completion, was closed early or raised ('finished' says which)
        for g_ in (pending, launched):       # closed early or raised: conversions already handed to the render pool must be done with
            for f in (f_ for fs in ((g_ or {}).get("futures") or []) for f_ in fs):      # the pinned buffers before a later call reuses them
                try:
                    f.result()
                except Exception:            # noqa: BLE001
                    pass
        stats["total"] = _time.perf_counter() - _t_start
        with _funnel_calls_lock:
            FUNNEL_STATS.clear()
            FUNNEL_STATS.update(stats)
            FUNNEL_STATS_BY_CALL[stats["call"]] = dict(stats)
            for old in sorted(FUNNEL_STATS_BY_CALL)[:-8]:
                del FUNNEL_STATS_BY_CALL[old]
        _restore_pil_allocator(pil_blocks_held)
        if ops.get('keepmodels', True):
            model_holder.offload()
        else:
            model_holder.unload_models()
            gc.collect()         # (the reference collects after every run; with resident models there is nothing to free)


# the name BASELINE.json's north_star uses (a pre-0.4.0 name of the same function, see SURVEY.md naming note)
run_depthmap = core_generation_funnel


def unload_models():
    """reference: src/core.py:669."""
    model_holder.unload_models()
