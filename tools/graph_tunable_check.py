#!/usr/bin/env python3
"""Does TunableOp (switched on by ModelHolder for the tuned library GEMMs) disturb hipGraph replay?  (maintenance tool)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ddepth_anything_v2 import DepthAnythingV2  # noqa: E402
from dmidas.dpt_depth import DPTDepthModel  # noqa: E402
from src import gemm_tuning  # noqa: E402
from src.hip_graph import GraphedForward  # noqa: E402

if os.environ.get("CHECK_TUNABLE", "1") == "1":
    print("TunableOp enabled:", gemm_tuning.enable(), flush=True)
torch.manual_seed(0)
g = torch.Generator().manual_seed(4)
for name, net, call in (("dav2", DepthAnythingV2('vits', features=64, out_channels=[48, 96, 192, 384]).eval().cuda().half(), lambda m, x: m.infer_batch(x, 70)),
                        ("hybrid", DPTDepthModel(path=None, backbone="vitb_rn50_384", non_negative=True).eval().cuda().half(), lambda m, x: m.infer_batch(x, net_size=128, net_h=96))):
    gf = GraphedForward(lambda x, net=net, call=call: call(net, x))
    for it, shape in enumerate(((1, 96, 128, 3), (1, 96, 128, 3), (2, 64, 96, 3), (1, 96, 128, 3))):
        x = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8).cuda()
        want = call(net, x)
        outs = [gf(x) for _ in range(3)]
        fin = [bool(torch.isfinite(o).all()) for o in outs]
        err = [(o - want).abs().max().item() for o in outs]
        print(name, it, shape, "finite", fin, "err", ["%.2e" % e for e in err], "of", "%.2e" % want.abs().max().item(), "graphs", len(gf.graphs), "failed", len(gf.failed), flush=True)
