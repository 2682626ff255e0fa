#!/usr/bin/env python3
"""ds_upsample_bilinear_nhwc at the four shapes of the bench step's decoder (batch 32, 256 channels, x 2, align_corners): time per
launch and the sum of a forward.     python tools/upsample_ab.py   (DS_NATIVE_LIB=<other build> for the A side)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
    sys.path.insert(0, p)
import torch
from src import _native as nat
def timeit(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
tot = 0
for hw in (16, 32, 64, 128):
    x = torch.randn((32, 256, hw, hw), device="cuda").half().contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: nat.upsample_bilinear(x, scale_factor=2, align_corners=True))
    gb = x.numel() * 2 * 5 / 1e9
    print(f"upsample 32 x 256 x {hw}^2 -> {2 * hw}^2: {t * 1e3:7.1f} us  {gb / t / 1e3 * 1e3:7.2f} TB/s of input + output")
    tot += t
y = nat.upsample_bilinear(x, scale_factor=2, align_corners=True)
import hashlib
print("per forward %.3f ms; sha %s" % (tot, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]))
