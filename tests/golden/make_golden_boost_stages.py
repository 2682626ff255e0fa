#!/usr/bin/env python3
"""Stage outputs of the PRODUCT's estimateboost on the CPU (float32 torch, the blend through the oracle) for the image of
boost_cases.npz -- the CPU twin that tests/test_models_cpu.py holds to 9.5e-5 against the reference's own estimateboost.
tests/test_gpu_models.py::test_boost_gpu_error_budget_per_stage compares the device run with these stage by stage (it used to
re-run the CPU twin inside the GPU suite: ~90 s of every run).  Subsampled to keep the file small.

    python tests/golden/make_golden_boost_stages.py   ->  tests/golden/boost_stage_cases.npz   (a few minutes of CPU)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd"), HERE):
    sys.path.insert(0, p)
import model_weights as mw  # noqa: E402

STRIDES = {"whole_estimate": (4, 4), "base": (2, 2), "mapped": (2, 2), "coef": None, "blended": (2, 2), "out": (2, 2)}


def subsample(key, t):
    st = STRIDES[key]
    if st is None:
        return t.numpy()
    return t[..., ::st[0], ::st[1]].contiguous().numpy()


def main():
    from lib.multi_depth_model_woauxi import RelDepthModel
    from pix2pix.models.pix2pix4depth_model import Pix2Pix4DepthModel
    from oracle import oracle as orc
    from src import boost
    orc.build()
    z = np.load(os.path.join(HERE, "boost_cases.npz"))
    net = RelDepthModel('resnext101').eval()
    net.load_state_dict(mw.fill_state_dict(net.state_dict()), strict=True)
    p2p = Pix2Pix4DepthModel().eval()
    p2p.netG.load_state_dict(mw.fill_state_dict(p2p.netG.state_dict()), strict=True)

    def oracle_blend(dst, rects, coefs, preds, mask):
        dst.copy_(torch.from_numpy(np.asarray(orc.boost_blend(dst.numpy(), rects, coefs, preds.numpy(), mask.numpy()), dtype=np.float32)))
    trace = {}
    out = boost.estimateboost(torch.from_numpy(z["image"]), net, 0, p2p, whole_size_threshold=int(z["rmax"][0]), blend=oracle_blend, trace=trace)
    trace["out"] = out
    np.savez_compressed(os.path.join(HERE, "boost_stage_cases.npz"), **{k: subsample(k, trace[k]) for k in STRIDES})
    print({k: tuple(subsample(k, trace[k]).shape) for k in STRIDES})


if __name__ == "__main__":
    main()
