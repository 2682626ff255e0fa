#!/usr/bin/env python3
"""Where one image of BASELINE config 4 (Boost: LeReS res101 + pix2pix on a 3840x2160 image) spends its device time, by aten op
and input shape (torch.profiler around one step after a priming step).

    python tools/c4_census.py [r_max]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


@torch.no_grad()
def main():
    rmax = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
    from src import boost, miopen_db
    miopen_db.seed()
    from lib.multi_depth_model_woauxi import RelDepthModel
    from pix2pix.models.pix2pix4depth_model import Pix2Pix4DepthModel
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = RelDepthModel('resnext101').eval().to(dev)
    p2p = Pix2Pix4DepthModel().eval().to(dev)
    H, W = 2160, 3840
    rng = np.random.default_rng(1000)
    yy, xx = np.mgrid[0:H, 0:W]
    img_np = (127 + 60 * np.sin(xx / 37.0)[..., None] * np.cos(yy / 23.0)[..., None] + 40 * (((xx // 240 + yy // 180) % 2)[..., None] - 0.5)
              + rng.normal(0, 25, (H, W, 3))).clip(0, 255).astype(np.uint8)
    img = torch.from_numpy(img_np).to(dev)
    boost.estimateboost(img, net, 0, p2p, whole_size_threshold=rmax)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        boost.estimateboost(img, net, 0, p2p, whole_size_threshold=rmax)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        t = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
        if t > 0:
            rows.append((t, e.count, e.key, str(e.input_shapes)[:120]))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"device time of one image: {tot / 1e3:.1f} ms (ops and kernels are both listed: an op row is its own kernels)")
    by_op = {}
    for t, c, k, s in rows:
        by_op[k] = by_op.get(k, 0) + t
    for k, t in sorted(by_op.items(), key=lambda kv: -kv[1])[:40]:
        print(f"{t / 1e3:9.2f} ms  {k[:120]}")
    print("---- by (op, shapes)")
    for t, c, k, s in rows[:90]:
        print(f"{t / 1e3:8.2f} ms x{c:<4d} {k[:40]:40s} {s}")


if __name__ == "__main__":
    main()
