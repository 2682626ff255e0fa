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
        # :20-21: `depthmap * (-1.0) / 256.0` promotes integers to float64 and keeps float32 / float16; cv2.Sobel is fed
        # np.float64(normalmap) (:28-29), but np.gradient (:31) and everything after it run in the array's own precision:
        # float32 and float16 have kernels of their own (every operation rounded like numpy's).  float16 cannot be blurred:
        # cv2.GaussianBlur rejects CV_16F arrays in the reference too.
        gradient = _ksize(sobel_gradient) == 0
        if depth.dtype == np.float16 and (_ksize(pre_blur) or (gradient and _ksize(post_blur))):
            raise _native.DepthStereoError('create_normalmap: float16 data cannot be blurred (cv2.GaussianBlur does not take CV_16F '
                                           'arrays, the reference raises cv2.error here); pass float32 / float64')
        if depth.dtype in (np.float16, np.float32) and not gradient:
            # :20-21 scale in the array's own precision BEFORE the promotion to float64 at :28 (the quotient rounds when it is
            # subnormal): hand the kernel 256 times that value, which its float64 `/ 256.0` undoes exactly
            depth = (depth / 256.0).astype(np.float64) * 256.0
        elif not (depth.dtype in (np.float16, np.float32) and gradient):
            depth = depth.astype(np.float64)
    dev = torch.device('cuda', torch.cuda.current_device())
    d = torch.from_numpy(np.array(depth, order='C')).to(dev).unsqueeze(0)
    out = create_normalmap_batch(d, pre_blur, sobel_gradient, post_blur, invert)
    return Image.fromarray(out[0].cpu().numpy())


def create_normalmap_batch(depth_u16, pre_blur=None, sobel_gradient=3, post_blur=None, invert=False):
    """Device-resident batch: uint16 (or float64 / float32 / float16, see _native.normalmap) cuda tensor [N,H,W] -> uint8 cuda tensor [N,H,W,3]."""
    return _native.normalmap(depth_u16.contiguous(), _ksize(pre_blur), _ksize(sobel_gradient), _ksize(post_blur), bool(invert))
