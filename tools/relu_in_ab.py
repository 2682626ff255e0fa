#!/usr/bin/env python3
"""A/B of the residual unit's first convolution at the decoder shapes of the bench step (batch 32, 256 channels): the clamp pass
+ ds_conv3x3_nhwc(act 2) against ds_conv3x3_nhwc(act 6: the ReLU on the MFMA fragments inside the K loop).

    python tools/relu_in_ab.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402
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


@torch.no_grad()
def main():
    torch.manual_seed(0)
    conv = nn.Conv2d(256, 256, 3, padding=1).cuda().half()
    tot = [0.0, 0.0, 0.0]
    for hw, n in ((16, 1), (32, 2), (64, 2), (128, 2)):
        x = torch.randn((32, 256, hw, hw), device="cuda").half().contiguous(memory_format=torch.channels_last)
        assert torch.equal(nat.conv3x3(conv, F.relu(x), relu=True), nat.conv3x3(conv, x, relu=True, relu_in=True))
        t_plain = timeit(lambda: nat.conv3x3(conv, x, relu=True))
        t_a = timeit(lambda: nat.conv3x3(conv, F.relu(x), relu=True))
        t_b = timeit(lambda: nat.conv3x3(conv, x, relu=True, relu_in=True))
        print(f"32 x 256 x {hw}^2: conv alone {t_plain * 1e3:7.1f} us | clamp + conv {t_a * 1e3:7.1f} us | ReLU inside {t_b * 1e3:7.1f} us   (x {n} per forward)")
        tot[0] += n * t_plain; tot[1] += n * t_a; tot[2] += n * t_b
    print(f"per forward (7 units): conv alone {tot[0]:.3f} ms, clamp + conv {tot[1]:.3f} ms, ReLU inside {tot[2]:.3f} ms")


if __name__ == "__main__":
    main()
