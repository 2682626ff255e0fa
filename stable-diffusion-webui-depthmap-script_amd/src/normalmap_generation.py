"""Drop-in for the reference's ``src/normalmap_generation.py`` (create_normalmap, :5-56) on HIP kernels."""
import numpy as np
from PIL import Image

from . import _native


def _ksize(v):
    return int(v) if v is not None and v > 0 else 0


def create_normalmap(depthmap, pre_blur=None, sobel_gradient=3, post_blur=None, invert=False):
    """Generates normalmaps (reference: src/normalmap_generation.py:5-56).
    :param depthmap: HxW depthmap: uint16 from the funnel (core.py:262), or any other real array like the reference
    :param pre_blur: Gaussian blur before the gradient, None/<=0 to disable, otherwise kernel size
    :param sobel_gradient: Sobel kernel size, None/<=0 for np.gradient
    :param post_blur: Gaussian blur after the gradient, None/<=0 to disable, otherwise kernel size
    :param invert: depthmap will be inverted before calculating normalmap
    """
    torch = _native.require_gpu()
    depth = np.asarray(depthmap)
    if depth.ndim != 2 or depth.dtype.kind not in 'buif':
        raise _native.DepthStereoError('create_normalmap: depthmap must be a 2-D real array, got %s %s' % (depth.dtype, depth.shape))
    if depth.dtype != np.uint16:
        # :20-21: `depthmap * (-1.0) / 256.0` promotes integers to float64 and keeps float32 as float32; cv2.Sobel is fed
        # np.float64(normalmap) (:28-29), which equals the float64 evaluation (negation and /256 are exact), but np.gradient
        # (:31) and everything after it run in the array's own precision: float32 has a kernel of its own (every operation
        # in binary32, numpy's order); a blur in front of / behind it would be cv2's float32 GaussianBlur (unpinned: not built),
        # and float16 arithmetic (numpy rounds every operation to half) is not built either.
        f32_gradient = depth.dtype == np.float32 and _ksize(sobel_gradient) == 0
        if f32_gradient and (_ksize(pre_blur) or _ksize(post_blur)):
            raise _native.DepthStereoError('create_normalmap: float32 depth with np.gradient AND a Gaussian blur runs through cv2\'s '
                                           'float32 GaussianBlur in the reference and is not built; pass float64 or use a Sobel size')
        if depth.dtype == np.float16 and _ksize(sobel_gradient) == 0:
            raise _native.DepthStereoError('create_normalmap: float16 depth with np.gradient (sobel_gradient None) runs in float16 '
                                           'in the reference and is not built; pass float32 / float64 or use a Sobel size')
        if not f32_gradient:
            depth = depth.astype(np.float64)
    dev = torch.device('cuda', torch.cuda.current_device())
    d = torch.from_numpy(np.array(depth, order='C')).to(dev).unsqueeze(0)
    out = create_normalmap_batch(d, pre_blur, sobel_gradient, post_blur, invert)
    return Image.fromarray(out[0].cpu().numpy())


def create_normalmap_batch(depth_u16, pre_blur=None, sobel_gradient=3, post_blur=None, invert=False):
    """Device-resident batch: uint16 (or float64) cuda tensor [N,H,W] -> uint8 cuda tensor [N,H,W,3]."""
    return _native.normalmap(depth_u16.contiguous(), _ksize(pre_blur), _ksize(sobel_gradient), _ksize(post_blur), bool(invert))
