"""Drop-in for the reference's ``src/stereoimage_generation.py`` backed by hand-written HIP kernels.

``create_stereoimages`` keeps the reference signature and semantics (src/stereoimage_generation.py:13-74):
PIL/ndarray in, list of PIL images out, same mode names, same error behaviour
(AssertionError on shape mismatch :78, Exception('Unknown mode') :73, unknown fill -> None eye).
The pixels are produced on an MI355X by ``libdepthstereo_hip.so`` (C ABI: include/depthstereo.h) and are
bit-identical to the reference's numba kernels.  ``create_stereoimages_batch`` is the same operation
for a batch that already lives in HBM (torch tensors in, torch tensors out, no host round trip).
"""
import numpy as np
from PIL import Image

from . import _native

PAIR_MODES = ('left-right', 'right-left', 'top-bottom', 'bottom-top')
KNOWN_MODES = PAIR_MODES + ('red-cyan-anaglyph', 'left-only', 'only-right', 'cyan-red-reverseanaglyph')
FILL_TECHNIQUES = ('none', 'naive', 'naive_interpolating', 'polylines_soft', 'polylines_sharp')


def create_stereoimages(original_image, depthmap, divergence, separation=0.0, modes=None,
                        stereo_balance=0.0, stereo_offset_exponent=1.0, fill_technique='polylines_sharp'):
    """Creates stereoscopic images (reference: src/stereoimage_generation.py:13-74).

    :param original_image: PIL image or HxWxC uint8 ndarray
    :param depthmap: HxW depthmap, white = near.  uint16 from the funnel; any real dtype is accepted.
    :param float divergence: 3D effect, in percent of the image width
    :param float separation: shift of the two halves, in percent
    :param list modes: 'left-right', 'right-left', 'top-bottom', 'bottom-top', 'red-cyan-anaglyph',
      'left-only', 'only-right', 'cyan-red-reverseanaglyph'
    :param float stereo_balance: [-1, 1], how divergence is split between the eyes
    :param float stereo_offset_exponent: see reference
    :param str fill_technique: 'none', 'naive', 'naive_interpolating', 'polylines_soft', 'polylines_sharp'
    """
    if modes is None:
        modes = ['left-right']
    if not isinstance(modes, list):
        modes = [modes]
    if len(modes) == 0:
        return []

    torch = _native.require_gpu()
    original_image = np.asarray(original_image)
    depth = np.asarray(depthmap)
    # apply_stereo_divergence asserts this before anything else (:78); at least one eye always moves
    assert original_image.shape[:2] == depth.shape, 'Depthmap and the image must have the same size'
    if original_image.ndim != 3:
        raise ValueError('not enough values to unpack (expected 3, got %d)' % original_image.ndim)  # h, w, c = shape (:99,:172)

    dev = torch.device('cuda', torch.cuda.current_device())
    img_t = torch.from_numpy(np.array(original_image, dtype=np.uint8, order='C')).to(dev).unsqueeze(0)
    depth_t = _depth_to_device(depth, stereo_offset_exponent, dev)
    exponent = stereo_offset_exponent
    if depth_t[1]:          # exponent already folded into the depth on the host
        exponent = 1.0
    results = create_stereoimages_batch(img_t, depth_t[0].unsqueeze(0), divergence, separation, modes, stereo_balance,
                                        exponent, fill_technique)
    return [None if r is None else Image.fromarray(r[0].cpu().numpy()) for r in results]


def _depth_to_device(depth, exponent, dev):
    """Pick the device representation whose normalisation reproduces numpy's (:79-81).

    uint16 / float32 / float64 are normalised on the device exactly as numpy does for that dtype.  Any
    other dtype is normalised on the host in numpy itself and shipped as float64 in [0,1]; the device
    re-normalisation (x-0)/(1-0) is then the identity.  Same for float inputs with exponent != 1, where
    pow() must be the host libm's: returns (tensor, exponent_folded)."""
    torch = _native._torch()
    if depth.dtype == np.uint16:
        return torch.from_numpy(np.array(depth, order='C')).to(dev), False
    if depth.dtype in (np.float32, np.float64) and exponent == 1.0:
        return torch.from_numpy(np.array(depth, order='C')).to(dev), False
    with np.errstate(all='ignore'):
        dmin, dmax = depth.min(), depth.max()
        norm = np.asarray((depth - dmin) / (dmax - dmin), dtype=np.float64)
        folded = False
        if exponent != 1.0:
            norm = np.frompyfunc(lambda x: float(x) ** exponent, 1, 1)(norm).astype(np.float64)
            folded = True
    return torch.from_numpy(np.ascontiguousarray(norm)).to(dev), folded


def create_stereoimages_batch(images, depth, divergence, separation=0.0, modes=None, stereo_balance=0.0,
                              stereo_offset_exponent=1.0, fill_technique='polylines_sharp'):
    """Batched, device-resident create_stereoimages.

    images: uint8 cuda tensor [N,H,W,C]; depth: cuda tensor [N,H,W] (uint16 / float32 / float64).
    Returns one uint8 cuda tensor per mode: [N,H,2W,C] (left-right/right-left), [N,2H,W,C] (top-bottom/
    bottom-top), [N,H,W,3] (anaglyphs), [N,H,W,C] (single eye).  Every image of the batch is normalised
    by its own depth min/max, exactly like N separate reference calls.
    """
    torch = _native.require_gpu()
    if modes is None:
        modes = ['left-right']
    if not isinstance(modes, list):
        modes = [modes]
    if len(modes) == 0:
        return []
    for mode in modes:
        if mode not in KNOWN_MODES:
            raise Exception('Unknown mode')
    assert images.dtype == torch.uint8 and images.dim() == 4, 'images must be uint8 [N,H,W,C]'
    assert tuple(images.shape[:3]) == tuple(depth.shape), 'Depthmap and the image must have the same size'
    images = images.contiguous()
    depth = depth.contiguous()
    n, h, w, c = images.shape
    rowb = w * c

    balance = (stereo_balance + 1) / 2
    left_is_original = balance < 0.001
    right_is_original = balance > 0.999
    unknown_fill = fill_technique not in FILL_TECHNIQUES      # reference: both moving eyes become None (:85-92)

    # eye parameters exactly as the reference computes them (:46-51, :82-83)
    l_div_px = ((+1 * divergence * balance) / 100.0) * w
    l_sep_px = ((-1 * separation) / 100.0) * w
    r_div_px = ((-1 * divergence * (1 - balance)) / 100.0) * w
    r_sep_px = (separation / 100.0) * w

    # render both eyes straight into the first pair-layout output; otherwise into lone eye buffers
    outs = {}
    primary = next((m for m in modes if m in PAIR_MODES), None)
    if primary in ('left-right', 'right-left'):
        buf = torch.empty((n, h, 2 * w, c), dtype=torch.uint8, device=images.device)
        lo, ro = (0, rowb) if primary == 'left-right' else (rowb, 0)
        left = (buf.data_ptr() + lo, 2 * rowb, h * 2 * rowb)
        right = (buf.data_ptr() + ro, 2 * rowb, h * 2 * rowb)
        outs[primary] = buf
    elif primary in ('top-bottom', 'bottom-top'):
        buf = torch.empty((n, 2 * h, w, c), dtype=torch.uint8, device=images.device)
        lo, ro = (0, h * rowb) if primary == 'top-bottom' else (h * rowb, 0)
        left = (buf.data_ptr() + lo, rowb, 2 * h * rowb)
        right = (buf.data_ptr() + ro, rowb, 2 * h * rowb)
        outs[primary] = buf
    else:
        lbuf = torch.empty((n, h, w, c), dtype=torch.uint8, device=images.device)
        rbuf = torch.empty((n, h, w, c), dtype=torch.uint8, device=images.device)
        left = (lbuf.data_ptr(), rowb, h * rowb)
        right = (rbuf.data_ptr(), rowb, h * rowb)
        outs['__left'], outs['__right'] = lbuf, rbuf

    if unknown_fill and not (left_is_original and right_is_original):
        # reference: apply_stereo_divergence returns None for an unknown fill (:85-92); np.hstack/vstack of None
        # gives an object array and Image.fromarray then raises TypeError for every mode.
        raise TypeError("Cannot handle this data type: fill_technique %r is not one of %s"
                        % (fill_technique, list(FILL_TECHNIQUES)))

    pow_lut = None
    exponent = float(stereo_offset_exponent)
    if exponent != 1.0:
        if depth.dtype not in (torch.uint16, torch.int16):
            raise _native.DepthStereoError('stereo_offset_exponent != 1 on device-resident depth needs uint16 depth')
        pow_lut = _native.build_pow_lut(depth, exponent)

    eyes = []
    if left_is_original:
        _native.copy_view(images.data_ptr(), rowb, h * rowb, left[0], left[1], left[2], n, h, rowb, images)
    else:
        eyes.append((l_div_px, l_sep_px) + left)
    if right_is_original:
        _native.copy_view(images.data_ptr(), rowb, h * rowb, right[0], right[1], right[2], n, h, rowb, images)
    else:
        eyes.append((r_div_px, r_sep_px) + right)
    if eyes:
        _native.stereo_warp(images, depth, eyes, fill_technique, exponent, pow_lut)

    def eye_copy(src):
        t = torch.empty((n, h, w, c), dtype=torch.uint8, device=images.device)
        _native.copy_view(src[0], src[1], src[2], t.data_ptr(), rowb, h * rowb, n, h, rowb, images)
        return t

    results = []
    for mode in modes:
        if mode in outs:
            results.append(outs[mode])
            continue
        if mode in ('left-right', 'right-left'):
            t = torch.empty((n, h, 2 * w, c), dtype=torch.uint8, device=images.device)
            a, b = (left, right) if mode == 'left-right' else (right, left)
            _native.copy_view(a[0], a[1], a[2], t.data_ptr(), 2 * rowb, h * 2 * rowb, n, h, rowb, images)
            _native.copy_view(b[0], b[1], b[2], t.data_ptr() + rowb, 2 * rowb, h * 2 * rowb, n, h, rowb, images)
        elif mode in ('top-bottom', 'bottom-top'):
            t = torch.empty((n, 2 * h, w, c), dtype=torch.uint8, device=images.device)
            a, b = (left, right) if mode == 'top-bottom' else (right, left)
            _native.copy_view(a[0], a[1], a[2], t.data_ptr(), rowb, 2 * h * rowb, n, h, rowb, images)
            _native.copy_view(b[0], b[1], b[2], t.data_ptr() + h * rowb, rowb, 2 * h * rowb, n, h, rowb, images)
        elif mode in ('red-cyan-anaglyph', 'cyan-red-reverseanaglyph'):
            if c < 3:
                raise IndexError('index 1 is out of bounds for axis 2 with size %d' % c)   # overlap_red_cyan reads channel 1,2
            t = torch.empty((n, h, w, 3), dtype=torch.uint8, device=images.device)
            a, b = (left, right) if mode == 'red-cyan-anaglyph' else (right, left)
            _native.overlap_red_cyan(a[0], a[1], a[2], b[0], b[1], b[2], n, h, w, c, t)
        elif mode == 'left-only':
            t = outs.get('__left')
            if t is None:
                t = eye_copy(left)
        elif mode == 'only-right':
            t = outs.get('__right')
            if t is None:
                t = eye_copy(right)
        else:
            raise Exception('Unknown mode')
        outs.setdefault(mode, t)
        results.append(t)
    return results


def apply_stereo_divergence(original_image, depth, divergence, separation, stereo_offset_exponent, fill_technique):
    """One eye, ndarray in / ndarray out (reference: src/stereoimage_generation.py:77-92)."""
    original_image = np.asarray(original_image)
    depth = np.asarray(depth)
    assert original_image.shape[:2] == depth.shape, 'Depthmap and the image must have the same size'
    if fill_technique not in FILL_TECHNIQUES:
        return None
    torch = _native.require_gpu()
    dev = torch.device('cuda', torch.cuda.current_device())
    img_t = torch.from_numpy(np.array(original_image, dtype=np.uint8, order='C')).to(dev).unsqueeze(0)
    depth_t, folded = _depth_to_device(depth, stereo_offset_exponent, dev)
    depth_t = depth_t.unsqueeze(0)
    exponent = 1.0 if folded else float(stereo_offset_exponent)
    n, h, w, c = img_t.shape
    out = torch.empty_like(img_t)
    pow_lut = _native.build_pow_lut(depth_t, exponent) if exponent != 1.0 else None
    divergence_px = (divergence / 100.0) * w
    separation_px = (separation / 100.0) * w
    _native.stereo_warp(img_t, depth_t, [(divergence_px, separation_px, out.data_ptr(), w * c, h * w * c)], fill_technique,
                        exponent, pow_lut)
    return out[0].cpu().numpy()


def overlap_red_cyan(im1, im2):
    """reference: src/stereoimage_generation.py:286-307."""
    torch = _native.require_gpu()
    dev = torch.device('cuda', torch.cuda.current_device())
    a = torch.from_numpy(np.ascontiguousarray(im1, dtype=np.uint8)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(im2, dtype=np.uint8)).to(dev)
    h, w, c = b.shape
    out = torch.empty((1, h, w, 3), dtype=torch.uint8, device=dev)
    _native.overlap_red_cyan(a.data_ptr(), a.shape[1] * a.shape[2], 0, b.data_ptr(), w * c, 0, 1, h, w, c, out)
    return out[0].cpu().numpy()
