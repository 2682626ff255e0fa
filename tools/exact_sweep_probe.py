#!/usr/bin/env python3
"""Time of the exact row sweep (k_polylines_exact_lds) on 3840-column rows (PROBE_W / PROBE_H: another frame size) that the main kernel flags, with the sweep run by the whole
wave (default) and by one lane (DS_PL_EXACT_COOP=0), and by the global-scratch kernel (DS_PL_EXACT_GLOBAL=1): HIP events inside the C ABI
(ds_profile_last_ms).  gpurun -- 'python tools/exact_sweep_probe.py'"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")]
import src._native as nat  # noqa: E402
import src.stereoimage_generation as sg  # noqa: E402

rng = np.random.default_rng(4)
H, W = int(os.environ.get("PROBE_H", 2160)), int(os.environ.get("PROBE_W", 3840))      # PROBE_H / PROBE_W: another frame size (1080 x 1920: config 5's)
img = torch.from_numpy(rng.integers(0, 256, (1, H, W, 3), dtype=np.uint8)).cuda()
yy, xx = np.mgrid[0:H, 0:W]
smooth = 30000 + 20000 * np.sin(xx / 611.0) * np.cos(yy / 397.0)
dep = smooth.astype(np.uint16)
if os.environ.get("PROBE_KIND", "quantised") == "quantised":
    dep[100:140] = (rng.integers(0, 4, (40, W)) * 21845).astype(np.uint16)          # 40 rows of quantised depth: ties -> flagged rows
else:
    # PROBE_KIND=noisy: the regime of a network's prediction -- smooth with small noise, a few short quantised stretches per row that flag
    # it: the speculative chunks of round 6 find their restart columns (rows fully quantised leave them none)
    dep[100:140] = (dep[100:140].astype(np.int64) + rng.integers(0, 40, (40, W))).astype(np.uint16)
    for x0 in (W // 7, W // 2, (5 * W) // 6):
        dep[100:140, x0:x0 + 48] = (rng.integers(0, 4, (40, 48)) * 21845).astype(np.uint16)
dep_t = torch.from_numpy(dep).cuda().unsqueeze(0)
nat.profile_enable(0, True)
for name, env in (("wave (default)", {}), ("one lane", {"DS_PL_EXACT_COOP": "0"}), ("global scratch", {"DS_PL_EXACT_GLOBAL": "1"})):
    for k in ("DS_PL_EXACT_COOP", "DS_PL_EXACT_GLOBAL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    outs = []
    for _ in range(3):
        out = sg.create_stereoimages_batch(img, dep_t, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')[0]
        torch.cuda.synchronize()
        r, e = nat.profile_last_ms(0)
    rows = nat.last_exact_rows(img)
    outs.append(out.cpu())
    print(f"{name:16s} exact sweep {e:9.3f} ms for {rows} flagged rows ({e / max(rows, 1):.3f} ms per row, one workgroup each); main + general {r:.3f} ms")
