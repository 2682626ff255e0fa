#!/usr/bin/env python3
"""Run-to-run differences of the eager float16 forward of two small networks (maintenance tool): which component is not
bit-reproducible?  python tools/determinism_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ddepth_anything_v2 import DepthAnythingV2  # noqa: E402
from dmidas.dpt_depth import DPTDepthModel  # noqa: E402
from src import _native as nat  # noqa: E402
from src import vit_mi355x as vm  # noqa: E402

torch.manual_seed(0)
g = torch.Generator().manual_seed(4)
nets = (("dav2_vits", DepthAnythingV2('vits', features=64, out_channels=[48, 96, 192, 384]).eval().cuda().half(), lambda m, x: m.infer_batch(x, 70)),
        ("dpt_hybrid", DPTDepthModel(path=None, backbone="vitb_rn50_384", non_negative=True).eval().cuda().half(), lambda m, x: m.infer_batch(x, net_size=128, net_h=96)))
x = torch.randint(0, 256, (1, 96, 128, 3), generator=g, dtype=torch.uint8).cuda()
for name, net, call in nets:
    outs = [call(net, x).float() for _ in range(6)]
    d = max((o - outs[0]).abs().max().item() for o in outs[1:])
    print(f"{name}: max |run_i - run_0| = {d:.3e} of max {outs[0].abs().max().item():.3e}", flush=True)
# the fused attention alone, at the shapes those networks launch it with
for (b, n, h) in ((1, 36 + 1, 6), (1, 49 + 1, 12), (2, 201, 12)):
    npad = vm.pad_len(n)
    qk = torch.randn(b, npad, 2, h, 64, generator=g).half().cuda()
    vt = torch.randn(b, h * 64, npad, generator=g).half().cuda()
    outs = [nat.attention_fwd(qk, vt, n, 0.125) for _ in range(6)]
    print(f"attention b{b} n{n} h{h}: identical = {all(torch.equal(o[:, :n], outs[0][:, :n]) for o in outs[1:])}", flush=True)
