#!/usr/bin/env python3
"""The per-pixel stage of BASELINE config 4 on Boost's own prediction (one 3840x2160 image): time of create_stereoimages_batch, rows
the exact sweep re-rendered, general-pixel queue chunks.     python tools/c4_stereo_probe.py
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
    from src import boost, miopen_db, _native as nat
    import src.stereoimage_generation as sg
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
    pred = boost.estimateboost(img, net, 0, p2p, whole_size_threshold=1600)
    d16 = nat.depth_to_u16(pred.unsqueeze(0), True)
    di = d16.int()
    print("depth codes: min %d max %d, distinct %d" % (int(di.min()), int(di.max()), int(torch.unique(di).numel())))
    dx = (di[0, :, 1:] - di[0, :, :-1]).abs().float()
    print("mean |d code / dx| %.1f, max %.0f; one pixel of divergence = %.0f codes" % (dx.mean().item(), dx.max().item(), 65535 / (0.025 * W)))

    def run():
        return sg.create_stereoimages_batch(img.unsqueeze(0), d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')[0]
    for env in (None, ("DS_PL_EXACT_CHUNKS", "1"), ("DS_PL_EXACT_CHUNKS", "3"), ("DS_PL_EXACT_CHUNKS", "8")):
        if env:
            os.environ[env[0]] = env[1]
        run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            run()
        b.record()
        torch.cuda.synchronize()
        st = nat.last_stats(img) if hasattr(nat, "last_stats") else None
        print(env, "create_stereoimages_batch %.2f ms per image; stats (exact rows, queue chunks): %s" % (a.elapsed_time(b) / 3, st))


if __name__ == "__main__":
    main()
