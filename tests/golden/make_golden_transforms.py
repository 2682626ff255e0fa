#!/usr/bin/env python3
"""Golden network inputs made by the REFERENCE's own transform chain (dmidas/transforms.py Resize / NormalizeImage /
PrepareForNet as estimatemidas composes them, src/depthmap_generation.py:457-476, after get_raw_prediction's
``cv2.cvtColor(np.asarray(input), cv2.COLOR_BGR2RGB) / 255.0``, :381).

cv2 and torchvision are absent in the build container: ``cv2.resize(..., INTER_CUBIC)`` is replaced by the numpy
restatement of OpenCV's documented cubic kernel (oracle._cv_cubic_resize, per channel), ``cv2.cvtColor`` by the channel
swap it is, and ``Compose`` by "apply in order".  What this pins: the size rule, the order of operations, the channel
order the network sees, the normalisation, the layout.  What it does not pin: OpenCV's own arithmetic.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_transforms.py     ->  transform_cases.npz
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'

# (name, H, W, net_w, net_h, resize_method, seed)
CASES = [("sq_minimal", 70, 100, 96, 96, "minimal", 1), ("match_490x640", 64, 49, 64, 64, "minimal", 2),
         ("wide_upper", 50, 120, 96, 96, "upper_bound", 3), ("tall_lower", 90, 40, 64, 64, "lower_bound", 4)]


def image(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def main():
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    cv2 = types.ModuleType("cv2")
    cv2.INTER_CUBIC, cv2.INTER_AREA, cv2.INTER_NEAREST, cv2.COLOR_BGR2RGB = 2, 3, 0, 4
    cv2.cvtColor = lambda a, code: np.ascontiguousarray(a[..., ::-1])
    cv2.resize = lambda img, size, interpolation=None: np.stack(
        [orc._cv_cubic_resize(img[..., k], (size[1], size[0])) for k in range(img.shape[-1])], axis=-1)
    sys.modules["cv2"] = cv2
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_transforms", os.path.join(REF, "dmidas", "transforms.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    out = {}
    for name, h, w, nw, nh, method, seed in CASES:
        pil_like = image(h, w, seed)
        img = cv2.cvtColor(np.asarray(pil_like), cv2.COLOR_BGR2RGB) / 255.0                     # depthmap_generation.py:381
        chain = [tr.Resize(nw, nh, resize_target=None, keep_aspect_ratio=True, ensure_multiple_of=32, resize_method=method,
                           image_interpolation_method=cv2.INTER_CUBIC),
                 tr.NormalizeImage(mean=[0.5, 0.5, 0.5], std=[0.5, 0.5, 0.5]), tr.PrepareForNet()]     # :457-470, :123-127
        sample = {"image": img}
        for t in chain:                                                                         # torchvision's Compose
            sample = t(sample)
        out[name] = sample["image"]
        print(name, sample["image"].shape, sample["image"].dtype)
    np.savez_compressed(os.path.join(HERE, "transform_cases.npz"), **out)


if __name__ == "__main__":
    main()
