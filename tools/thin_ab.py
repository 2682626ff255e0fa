#!/usr/bin/env python3
"""Round 6 A/B: the ragged round of one dpt_beit_large_512 block's token GEMMs at batch 32 (token stride 1032: 129 row panels) --
k_linear_thin (32 x 64 pieces, two K-tiles per step) against k_linear_ragged (128 x 64 pieces), and the persistent walk alone.
Interleaved rounds, HIP events around 10 launches; the outputs of the three are compared bit for bit.

    python tools/thin_ab.py            (on the GPU box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
from src import _native as nat  # noqa: E402


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev, dt = torch.device("cuda"), torch.float16
    g = torch.Generator(device="cpu").manual_seed(5)
    for npad in (1032, 4104):
        b = 32 if npad == 1032 else 8
        m = b * npad
        x = torch.randn(m, 1024, generator=g).to(dev, dt)
        x4 = torch.randn(m, 4096, generator=g).to(dev, dt)
        res = torch.randn(m, 1024, generator=g).to(dev, dt)
        gam = torch.randn(1024, generator=g).to(dev, dt)

        def w_(n, k):
            return (torch.randn(n, k, generator=g) * k ** -0.5).to(dev, dt)
        w_qk, w_p, w_1, w_2 = w_(2048, 1024), w_(1024, 1024), w_(4096, 1024), w_(1024, 4096)
        b_qk, b_p, b_1, b_2 = (torch.randn(n, generator=g).to(dev, dt) for n in (2048, 1024, 4096, 1024))
        cases = [
            (f"qk       {m}x2048x1024", lambda: nat.linear(x, w_qk, b_qk, False)),
            (f"proj+res {m}x1024x1024", lambda: nat.linear_residual(x, w_p, b_p, gam, res)),
            (f"fc1+gelu {m}x4096x1024", lambda: nat.linear(x, w_1, b_1, True)),
            (f"fc2+res  {m}x1024x4096", lambda: nat.linear_residual(x4, w_2, b_2, gam, res)),
        ]
        for name, fn in cases:
            t = {"thin": [], "wide": [], "walk": []}
            outs = {}
            for _ in range(3):
                for key, env in (("thin", dict(DS_LIN_RAGGED="1", DS_LIN_RAGGED_THIN="1")), ("wide", dict(DS_LIN_RAGGED="1", DS_LIN_RAGGED_THIN="0")),
                                 ("walk", dict(DS_LIN_RAGGED="0"))):
                    nat.linear_env(**env)
                    t[key].append(timeit(fn))
                    outs[key] = fn()
            nat.linear_env(DS_LIN_RAGGED=None, DS_LIN_RAGGED_THIN=None)
            same = torch.equal(outs["thin"], outs["wide"]) and torch.equal(outs["thin"], outs["walk"])
            print(f"gemm {name}: thin {min(t['thin']) * 1e3:7.1f} us | k_linear_ragged {min(t['wide']) * 1e3:7.1f} us | walk alone {min(t['walk']) * 1e3:7.1f} us | "
                  f"bit-identical {same}", flush=True)


if __name__ == "__main__":
    main()
