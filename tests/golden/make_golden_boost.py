#!/usr/bin/env python3
"""End-to-end golden of Boost: the REFERENCE's own ``estimateboost`` (src/depthmap_generation.py:774-941 with
calculateprocessingres :969-1025, doubleestimate :1028-1049, generatepatchs / adaptiveselection :1070-1167, ImageandPatchs
:562-608, the np.polyfit merge and the Gaussian-mask blend :879-937), run UNMODIFIED from /root/reference on the CPU in float32
with the reference's own LeReS network (lib/) and pix2pix merge network (pix2pix/), name-seeded weights.

Build container only (a few minutes of CPU):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_boost.py  ->  boost_cases.npz

Stand-ins for what is absent here, and only that:
* ``cv2``: resize INTER_CUBIC / INTER_LINEAR = the numpy restatements of OpenCV's documented kernels (oracle._cv_cubic_resize,
  oracle._cv_linear_resize; the reference passes INTER_AREA / INTER_NEAREST in the ``dst`` slot at :988 and :1008, i.e. those two
  calls ARE plain bilinear resizes); Sobel ksize 3 = the documented 3 x 3 kernels with BORDER_REFLECT_101 as a direct 2-D sum;
  dilate = a maximum filter over the all-ones kernel (anchor at the centre, border ignored, like cv2's default);
  GaussianBlur = getGaussianKernel's documented formula, separable, REFLECT_101; integral = the (h+1) x (w+1) summed-area table;
  cvtColor = the channel swap.  OpenCV's own arithmetic stays unpinned; the control flow, every threshold, the order of the
  blend, the polynomial fit and the networks are the reference's code.
* ``skimage.measure.block_reduce`` = documented behaviour (pad with 0 to a multiple of the block, reduce with the function).
* ``torchvision.transforms`` Compose / ToTensor / Normalize for estimateleres' scale_torch (:425-440): the float32 HWC -> CHW
  conversion and (x - mean) / std they are.
Output: the image, the final depth (every second pixel of every second row, float32) and a few full-resolution statistics.
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, HERE)
import model_weights as mw  # noqa: E402

H, W, RMAX = 480, 640, 1600


def image():
    rng = np.random.default_rng(51)
    yy, xx = np.mgrid[0:H, 0:W]
    img = 127 + 60 * np.sin(xx / 37.0)[..., None] * np.cos(yy / 23.0)[..., None] + rng.normal(0, 12, (H, W, 3))
    img[120:330, 200:470] += 70 * np.sign(np.sin(xx[120:330, 200:470] / 3.0))[..., None]      # a textured region: patches get selected
    img[380:, :150] -= 50
    return img.clip(0, 255).astype(np.uint8)


def _corr2d_reflect101(a, k2):
    kh, kw = k2.shape
    p = np.pad(a, ((kh // 2, kh // 2), (kw // 2, kw // 2)), mode='reflect')
    out = np.zeros(a.shape, np.float64)
    for i in range(kh):
        for j in range(kw):
            if k2[i, j] != 0.0:
                out += k2[i, j] * p[i:i + a.shape[0], j:j + a.shape[1]]
    return out


def _sep_reflect101(a, g):
    r = len(g) // 2
    p = np.pad(a, ((0, 0), (r, r)), mode='reflect')
    rows = np.zeros(a.shape, np.float64)
    for j, c in enumerate(g):
        rows += c * p[:, j:j + a.shape[1]]
    p = np.pad(rows, ((r, r), (0, 0)), mode='reflect')
    out = np.zeros(a.shape, np.float64)
    for i, c in enumerate(g):
        out += c * p[i:i + a.shape[0], :]
    return out


def install_stubs():
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.INTER_CUBIC, cv2.INTER_AREA, cv2.COLOR_BGR2RGB, cv2.CV_64F = 0, 1, 2, 3, 4, 6

    # float32 images are resized in FLOAT32, as OpenCV does for CV_32F (resize.cpp: float coefficients from interpolateCubic /
    # the linear weights, float products and sums, the horizontal pass first, then the vertical one); float64 images and integer
    # images (promoted) keep the float64 restatement of the oracle.  Round 5: the first golden used the float64 restatement for
    # float32 images too (one rounding at the end) -- a summation order no float32 implementation has, and what put the product's
    # float32 CPU twin at 9.5e-5 of this golden through the networks' conditioning.
    def _taps_cubic32(n_out, n_in):
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5).astype(np.float32)      # fx = (float)((dx + 0.5) * scale - 0.5)
        i = np.floor(f).astype(np.int64)
        t = (f - i.astype(np.float32)).astype(np.float32)
        A, one = np.float32(-0.75), np.float32(1)
        w = np.empty((n_out, 4), np.float32)
        w[:, 0] = ((A * (t + one) - np.float32(5) * A) * (t + one) + np.float32(8) * A) * (t + one) - np.float32(4) * A
        w[:, 1] = ((A + np.float32(2)) * t - (A + np.float32(3))) * t * t + one
        w[:, 2] = ((A + np.float32(2)) * (one - t) - (A + np.float32(3))) * (one - t) * (one - t) + one
        w[:, 3] = one - w[:, 0] - w[:, 1] - w[:, 2]
        return np.clip(i[:, None] - 1 + np.arange(4)[None, :], 0, n_in - 1), w

    def _cubic32(src, hw):
        src = np.asarray(src, np.float32)
        iy, wy = _taps_cubic32(hw[0], src.shape[0])
        ix, wx = _taps_cubic32(hw[1], src.shape[1])
        rows = src[:, ix[:, 0]] * wx[:, 0][None, :]
        for k in range(1, 4):
            rows = rows + src[:, ix[:, k]] * wx[:, k][None, :]
        out = rows[iy[:, 0], :] * wy[:, 0][:, None]
        for k in range(1, 4):
            out = out + rows[iy[:, k], :] * wy[:, k][:, None]
        assert out.dtype == np.float32
        return out

    def _taps_linear32(n_out, n_in):
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        t = (f - i.astype(np.float32)).astype(np.float32)
        t = np.where(i < 0, np.float32(0), t)
        i = np.where(i < 0, 0, i)
        t = np.where(i >= n_in - 1, np.float32(1), t).astype(np.float32)
        i = np.where(i >= n_in - 1, n_in - 2, i)
        return i, t

    def _linear32(src, hw):
        src = np.asarray(src, np.float32)
        iy, ty = _taps_linear32(hw[0], src.shape[0])
        ix, tx = _taps_linear32(hw[1], src.shape[1])
        one = np.float32(1)
        rows = src[:, ix] * (one - tx)[None, :] + src[:, ix + 1] * tx[None, :]
        out = rows[iy, :] * (one - ty)[:, None] + rows[iy + 1, :] * ty[:, None]
        assert out.dtype == np.float32
        return out

    def resize(img, size, dst=None, fx=None, fy=None, interpolation=1):
        f = orc._cv_cubic_resize if interpolation == cv2.INTER_CUBIC else orc._cv_linear_resize
        img = np.asarray(img)
        if img.dtype == np.float32 and os.environ.get("DS_GOLDEN_RESIZE", "f32") == "f32":
            f = _cubic32 if interpolation == cv2.INTER_CUBIC else _linear32
        hw = (size[1], size[0])
        if img.shape[:2] == hw:
            return img.copy()
        if img.ndim == 2:
            return np.asarray(f(img, hw), dtype=img.dtype if img.dtype.kind == 'f' else np.float64)
        return np.stack([f(img[..., k], hw) for k in range(img.shape[-1])], axis=-1).astype(img.dtype if img.dtype.kind == 'f' else np.float64)

    def Sobel(src, ddepth, dx, dy, ksize=3):
        assert ksize == 3
        kx = np.array([-1.0, 0.0, 1.0]) if dx else np.array([1.0, 2.0, 1.0])
        ky = np.array([-1.0, 0.0, 1.0]) if dy else np.array([1.0, 2.0, 1.0])
        return _corr2d_reflect101(np.asarray(src, np.float64), np.outer(ky, kx))

    def dilate(src, kernel, iterations=1):
        assert iterations == 1 and kernel.min() == 1
        kh, kw = kernel.shape
        ay, ax = kh // 2, kw // 2                                   # anchor: the kernel centre (cv2 default)
        p = np.pad(np.asarray(src), ((ay, kh - 1 - ay), (ax, kw - 1 - ax)), mode='constant', constant_values=-np.inf)
        out = np.full(src.shape, -np.inf)
        for i in range(kh):
            for j in range(kw):
                out = np.maximum(out, p[i:i + src.shape[0], j:j + src.shape[1]])
        return out.astype(src.dtype)

    def GaussianBlur(src, ksize, sigma):
        k = ksize[0]
        x = np.arange(k, dtype=np.float64) - (k - 1) / 2.0
        g = np.exp(-(x * x) / (2.0 * float(sigma) ** 2))
        g /= g.sum()
        return _sep_reflect101(np.asarray(src, np.float64), g).astype(src.dtype)

    def integral(src):
        out = np.zeros((src.shape[0] + 1, src.shape[1] + 1), np.float64)
        out[1:, 1:] = np.cumsum(np.cumsum(np.asarray(src, np.float64), axis=0), axis=1)
        return out

    cv2.resize, cv2.Sobel, cv2.dilate, cv2.GaussianBlur, cv2.integral = resize, Sobel, dilate, GaussianBlur, integral
    cv2.cvtColor = lambda a, code: np.ascontiguousarray(np.asarray(a)[..., ::-1])
    sys.modules["cv2"] = cv2

    sk, skm = types.ModuleType("skimage"), types.ModuleType("skimage.measure")

    def block_reduce(img, block, func):
        n0, n1 = block
        h, w = img.shape
        ph, pw = (-h) % n0, (-w) % n1
        p = np.pad(img, ((0, ph), (0, pw)), mode='constant', constant_values=0)
        return func(func(p.reshape(p.shape[0] // n0, n0, p.shape[1] // n1, n1), axis=3), axis=1)

    skm.block_reduce = block_reduce
    sk.measure = skm
    sys.modules["skimage"], sys.modules["skimage.measure"] = sk, skm

    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts):
            self.transforms = ts

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    class ToTensor:
        def __call__(self, a):                                      # float32 HWC ndarray: no rescaling (only uint8 is divided)
            assert a.dtype == np.float32 and a.ndim == 3
            return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.mean) / self.std

    tvt.Compose, tvt.ToTensor, tvt.Normalize, tvt.transforms = Compose, ToTensor, Normalize, tvt
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    for name in ("diffusers", "transformers", "dmarigold", "dmarigold.marigold", "dzoedepth", "dzoedepth.models",
                 "dzoedepth.models.builder", "dzoedepth.utils", "dzoedepth.utils.config", "modules", "modules.shared", "modules.devices"):
        sys.modules.setdefault(name, mock.MagicMock())
    import fake_timm
    fake_timm.install()


def main():
    install_stubs()
    sys.path.insert(0, REF)
    argv, sys.argv = sys.argv, [sys.argv[0]]
    import src.depthmap_generation as ref_dg
    from lib.multi_depth_model_woauxi import RelDepthModel
    from pix2pix.models.pix2pix4depth_model import Pix2Pix4DepthModel
    from pix2pix.options.test_options import TestOptions
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    ref_dg.depthmap_device = torch.device("cpu")
    net = RelDepthModel(backbone='resnext101').eval()
    net.load_state_dict(mw.fill_state_dict(net.state_dict()), strict=True)
    opt = TestOptions().parse()                                      # :292-296
    opt.gpu_ids = []
    p2p = Pix2Pix4DepthModel(opt)
    p2p.netG.load_state_dict(mw.fill_state_dict(p2p.netG.state_dict()), strict=True)
    p2p.eval()
    sys.argv = argv
    img_u8 = image()
    img = np.ascontiguousarray(img_u8[..., ::-1]) / 255.0           # get_raw_prediction :381: cvtColor(BGR2RGB) / 255
    with torch.no_grad():
        out = ref_dg.estimateboost(img, net, 0, p2p, RMAX)
    out = np.asarray(out, dtype=np.float32)
    assert out.shape == (H, W)
    np.savez_compressed(os.path.join(HERE, "boost_cases.npz"), image=img_u8, depth_s2=out[::2, ::2],
                        stats=np.array([out.min(), out.max(), out.mean(), out.std(), out[H // 2, W // 2]], np.float64),
                        rmax=np.array([RMAX]))
    print("boost golden:", out.shape, float(out.min()), float(out.max()), float(out.mean()))


if __name__ == "__main__":
    main()
