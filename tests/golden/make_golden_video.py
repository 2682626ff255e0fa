#!/usr/bin/env python3
"""Golden vectors of video mode's normalisation, made by the REFERENCE's own ``process_predicitons``
(/root/reference/src/video_mode.py:103-128).

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_video.py

``src/video_mode.py`` imports ``src.core`` and ``src.backbone`` at module level (cv2, gradio-side modules, torch hub
models ...): both are pre-seeded as stubs -- process_predicitons touches neither -- and the reference function itself
runs unmodified on seeded float32 predictions.  Output: video_cases.npz (inputs are regenerated from the seeds by the
tests; stored: the reference outputs per case).
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'

# (name, frames, H, W, seed, kind)
CASES = [("n9", 9, 9, 14, 11, "normal"), ("n12", 12, 9, 14, 11, "normal"), ("n5_wide", 5, 17, 33, 3, "wide"),
         ("n2", 2, 6, 7, 5, "normal"), ("n7_ties", 7, 8, 8, 9, "ties"), ("n1", 1, 5, 6, 2, "normal")]


def make_predictions(frames, h, w, seed, kind):
    rng = np.random.default_rng(seed)
    p = rng.normal(3.0, 2.0, (frames, h, w)).astype(np.float32)
    if kind == "wide":
        p = (p * 1000.0 - 250.0).astype(np.float32)
    if kind == "ties":
        p = (np.round(p * 2) / 2).astype(np.float32)
    return p


def load_reference_video_mode():
    sys.path.insert(0, REF)
    import src                                               # the reference's package (namespace)
    for name in ("src.core", "src.backbone"):
        sys.modules[name] = MagicMock()
    cc = types.ModuleType("src.common_constants")
    cc.GenerationOptions = MagicMock()
    sys.modules["src.common_constants"] = cc
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_video_mode", os.path.join(REF, "src", "video_mode.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    vm = load_reference_video_mode()
    out = {}
    for name, frames, h, w, seed, kind in CASES:
        p = make_predictions(frames, h, w, seed, kind)
        for mode in ("none", "experimental"):
            res = vm.process_predicitons([x for x in p], mode)
            out[f"{name}/{mode}"] = np.stack(res)
            print(name, mode, out[f"{name}/{mode}"].dtype, out[f"{name}/{mode}"].shape)
    np.savez_compressed(os.path.join(HERE, "video_cases.npz"), **out)


if __name__ == "__main__":
    main()
