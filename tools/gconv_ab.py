#!/usr/bin/env python3
"""ds_gconv3x3_nhwc_f32 at the shapes a batch of eight 896^2 patches sends through layers 1-3 of LeReS's ResNeXt: time per launch.
    python tools/gconv_ab.py      (DS_NATIVE_LIB=<other build> for the A side)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
    sys.path.insert(0, p)
import hashlib  # noqa: E402
import torch  # noqa: E402
from src import _native as nat  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


g = torch.Generator().manual_seed(1)
h = hashlib.sha256()
for (b, c, cpg, hh, ww) in [(8, 256, 8, 224, 224), (8, 512, 16, 112, 112), (8, 1024, 32, 56, 56)]:
    x = torch.randn((b, c, hh, ww), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn((c, cpg, 3, 3), generator=g) * (9 * cpg) ** -0.5).cuda()
    bias = torch.randn((c,), generator=g).cuda()
    img = nat.gconv_weight_image(wt, c // cpg)
    y = nat.gconv3x3(x, img, bias, True, cpg)
    h.update(y.cpu().numpy().tobytes())
    t = timeit(lambda: nat.gconv3x3(x, img, bias, True, cpg))
    fl = 2.0 * b * hh * ww * c * cpg * 9
    print(f"gconv {b} x {c} x {hh}^2, {cpg}-wide groups: {t * 1e3:7.1f} us  {fl / t / 1e9:6.1f} TFLOP/s  {x.numel() * 8 / t / 1e9:6.2f} TB/s of input + output")
print("sha256 of the three outputs:", h.hexdigest()[:16])
