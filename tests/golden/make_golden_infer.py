#!/usr/bin/env python3
"""End-to-end goldens of the reference's image -> raw prediction path: ``ModelHolder.get_raw_prediction``
(src/depthmap_generation.py:375-403) -> ``estimatemidas`` (:455-499, with dmidas/transforms.py:105-160) and
``estimatedepthanything_v2`` (:548-559, with ddepth_anything_v2/depth_anything_v2/dpt.py:196-221), run UNMODIFIED from
/root/reference on CPU in float32: PIL image in, float32 prediction at image size out (channel swaps, /255, the resize
rule, cubic resize in, normalisation, network forward, bicubic / bilinear resize back).

Build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_infer.py   ->  infer_cases.npz

Stand-ins for what is absent here (and only that):
* ``cv2``: ``cvtColor(BGR2RGB)`` = the channel swap it is; ``resize(INTER_CUBIC)`` = the numpy restatement of OpenCV's
  documented cubic kernel (oracle._cv_cubic_resize).  OpenCV's own arithmetic stays unpinned; everything around it is the
  reference's code.
* ``torchvision.transforms.Compose`` = "apply in order"; ``timm`` = tests/golden/fake_timm.py (containers only, every forward
  on the path is the reference's); the other imports of src/depthmap_generation.py that these two functions never touch
  (skimage, diffusers, marigold, zoedepth builders, the webui modules) are MagicMocks.
Weights: name-seeded (model_weights.py), the same the product's tests load.
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

# (name, H, W, net_w, net_h, seed)
MIDAS_CASES = [("beitb_landscape", 150, 210, 160, 160, 21), ("beitb_match", 130, 100, 96, 128, 22)]
DAV2_CASES = [("vits_landscape", 100, 150, 70, 23), ("vits_portrait", 120, 90, 84, 24)]


def image(h, w, seed):
    """A smooth-ish RGB image (low-pass noise): a cubic resize of white noise would make the comparison about nothing but
    the resampling kernel's rounding."""
    rng = np.random.default_rng(seed)
    base = rng.random((h // 8 + 2, w // 8 + 2, 3))
    yy = np.linspace(0, base.shape[0] - 1.001, h)
    xx = np.linspace(0, base.shape[1] - 1.001, w)
    y0, x0 = yy.astype(int), xx.astype(int)
    fy, fx = (yy - y0)[:, None, None], (xx - x0)[None, :, None]
    a = base[y0][:, x0] * (1 - fy) * (1 - fx) + base[y0 + 1][:, x0] * fy * (1 - fx) + base[y0][:, x0 + 1] * (1 - fy) * fx + base[y0 + 1][:, x0 + 1] * fy * fx
    a = a + 0.08 * rng.random((h, w, 3))
    return (255 * (a - a.min()) / (a.max() - a.min())).astype(np.uint8)


def install_stubs():
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    cv2 = types.ModuleType("cv2")
    cv2.INTER_CUBIC, cv2.INTER_AREA, cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.COLOR_BGR2RGB = 2, 3, 0, 1, 4

    def resize(img, size, interpolation=None):
        assert interpolation == cv2.INTER_CUBIC
        img = np.asarray(img)
        if img.ndim == 2:
            return orc._cv_cubic_resize(img, (size[1], size[0]))
        return np.stack([orc._cv_cubic_resize(img[..., k], (size[1], size[0])) for k in range(img.shape[-1])], axis=-1)

    cv2.cvtColor = lambda a, code: np.ascontiguousarray(np.asarray(a)[..., ::-1])
    cv2.resize = resize
    sys.modules["cv2"] = cv2
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts):
            self.transforms = ts

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    tvt.Compose = Compose
    tvt.transforms = tvt
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    for name in ("skimage", "skimage.measure", "diffusers", "transformers", "dmarigold", "dmarigold.marigold", "dzoedepth",
                 "dzoedepth.models", "dzoedepth.models.builder", "dzoedepth.utils", "dzoedepth.utils.config", "modules",
                 "modules.shared", "modules.devices"):
        sys.modules.setdefault(name, mock.MagicMock())
    import fake_timm
    fake_timm.install()


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import src.depthmap_generation as ref_dg
    from dmidas.dpt_depth import DPTDepthModel
    from dmidas.transforms import NormalizeImage
    from ddepth_anything_v2.depth_anything_v2.dpt import DepthAnythingV2
    from PIL import Image
    out = {}

    holder = ref_dg.ModelHolder()
    net = DPTDepthModel(path=None, backbone="beitb16_384", non_negative=True).eval()
    net.load_state_dict(mw.fill_state_dict_beit(net.state_dict()), strict=True)
    holder.depth_model, holder.depth_model_type, holder.device = net, 2, torch.device("cpu")
    holder.resize_mode = "minimal"                                                       # :141
    holder.normalization = NormalizeImage(mean=[0.5, 0.5, 0.5], std=[0.5, 0.5, 0.5])     # :142
    holder.no_half, holder.precision = True, "full"
    for name, h, w, nw, nh, seed in MIDAS_CASES:
        img = image(h, w, seed)
        pred, inv = holder.get_raw_prediction(Image.fromarray(img), nw, nh)
        assert pred.shape == (h, w) and inv is False
        out[f"midas__{name}__image"] = img
        out[f"midas__{name}__pred"] = np.asarray(pred, dtype=np.float32)
        print(name, pred.shape, float(pred.min()), float(pred.max()))

    holder = ref_dg.ModelHolder()
    net = DepthAnythingV2('vits', features=64, out_channels=[48, 96, 192, 384]).eval()
    net.load_state_dict(mw.fill_state_dict(net.state_dict()), strict=True)
    holder.depth_model, holder.depth_model_type, holder.device = net, 12, torch.device("cpu")
    for name, h, w, size, seed in DAV2_CASES:
        img = image(h, w, seed)
        pred, inv = holder.get_raw_prediction(Image.fromarray(img), size, size)
        assert pred.shape == (h, w) and inv is False
        out[f"dav2__{name}__image"] = img
        out[f"dav2__{name}__pred"] = np.asarray(pred, dtype=np.float32)
        print(name, pred.shape, float(pred.min()), float(pred.max()))
    np.savez_compressed(os.path.join(HERE, "infer_cases.npz"), **out)


if __name__ == "__main__":
    main()
