#!/usr/bin/env python3
"""Is an image's depth independent of its position in the batch and of the launch?  (maintenance tool, GPU)
dpt_beit_large_512 at 512^2, batch 32, float16, name-seeded weights: the same image at units 0 / 13 / 31, the forward run three
times; the same with torch.backends.cudnn.deterministic (MIOpen: no non-deterministic solvers).
    python tools/batch_invariance_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import model_weights as mw  # noqa: E402
from dmidas.dpt_depth import DPTDepthModel  # noqa: E402

m = DPTDepthModel(path=None, backbone="beitl16_512", non_negative=True).eval()
m.load_state_dict(mw.fill_state_dict_beit(m.state_dict()), strict=True)
m = m.cuda().half()
base = mw.synthetic_image((1, 3, 512, 512), seed=31)
x = torch.cat([torch.roll(base, shifts=7 * i, dims=3) for i in range(32)])
x[13], x[31] = base[0], base[0]
x = x.cuda().half().contiguous(memory_format=torch.channels_last)
for det in (False, True):
    torch.backends.cudnn.deterministic = det
    with torch.no_grad():
        ys = [m(x).float() for _ in range(3)]
    scale = ys[0].abs().max().item()
    print(f"cudnn.deterministic={det}: units 13 / 31 vs 0: {(ys[0][13] - ys[0][0]).abs().max().item():.3e} / {(ys[0][31] - ys[0][0]).abs().max().item():.3e}"
          f"   run 1 / 2 vs run 0: {(ys[1] - ys[0]).abs().max().item():.3e} / {(ys[2] - ys[0]).abs().max().item():.3e}   (range {scale:.3f})", flush=True)
    with torch.no_grad():
        y8 = m(x[24:32].contiguous(memory_format=torch.channels_last)).float()
    print(f"   units 24..31 alone (batch 8) vs inside the batch of 32: {(y8 - ys[0][24:32]).abs().max().item():.3e}", flush=True)
