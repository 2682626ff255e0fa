#!/usr/bin/env python3
"""Which convolutions of a network's forward still go to the library (MIOpen)?  One forward with torch's convolution entry points
wrapped; prints every call's shapes, strides and time.     python tools/library_conv_census.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import model_weights as mw  # noqa: E402
from dmidas.dpt_depth import DPTDepthModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = DPTDepthModel(path=None, backbone="beitl16_512", non_negative=True).eval()
m.load_state_dict(mw.fill_state_dict_beit(m.state_dict()), strict=True)
m = m.cuda().half()
x = mw.synthetic_image((1, 3, 512, 512), seed=31).repeat(B, 1, 1, 1).cuda().half().contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    m(x)
torch.cuda.synchronize()
log = []
for name in ("conv2d", "conv_transpose2d"):
    orig = getattr(F, name)

    def wrap(inp, w, *a, _o=orig, _n=name, **k):
        torch.cuda.synchronize(); t0 = time.time()
        y = _o(inp, w, *a, **k)
        torch.cuda.synchronize()
        log.append((_n, tuple(inp.shape), tuple(inp.stride()), tuple(w.shape), a[1:] if len(a) > 1 else k, tuple(y.shape), (time.time() - t0) * 1e3))
        return y
    setattr(F, name, wrap)
    setattr(torch, name, wrap)
with torch.no_grad():
    m(x)
for r in log:
    print("%s in %s strides %s w %s args %s -> %s  %.3f ms" % r)
print(len(log), "library convolution calls per forward")
