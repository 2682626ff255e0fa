#!/usr/bin/env python3
"""Timeline of one core_generation_funnel call on BASELINE config c3 (32 x 1024^2 PIL images in, depth + pair + normal map out): when
each group's inputs, forward, per-pixel kernels and result chunks finish ON THE DEVICE (timing events on the carrying streams) and
when each unit's PIL conversion finishes on the host.  DS_FUNNEL_TRACE is set by this script.

    python tools/funnel_timeline.py [group sizes ...]      (GPU box)
"""
import os
import sys
import time

os.environ["DS_FUNNEL_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402
import bench  # noqa: E402
import src.core as core  # noqa: E402
from src.hip_graph import GraphedForward  # noqa: E402


def main():
    torch.cuda.set_device(0)
    model, minfo = bench.build_model("dpt_beit_large_512", 0)
    model = model.cuda().half().eval()
    img_np, _ = bench.synth_batch(32, 0)
    graphed = GraphedForward(lambda x: bench.run_forward(model, "dpt_beit_large_512", x, 512, None), lazy=2)

    class _Pred:
        def __call__(self, pil, nw, nh, device):
            return self.batch_tensor(torch.from_numpy(np.asarray(pil.convert("RGB"))[None]).to(device))[0]

        def batch_tensor(self, t, nw=None, nh=None):
            return graphed(t)

    core.model_holder.register_predictor(1, _Pred())
    pils = [Image.fromarray(a) for a in img_np]
    opts = {"model_type": 1, "gen_stereo": True, "stereo_modes": ["left-right"], "gen_normalmap": True, "net_width": 512, "net_height": 512}
    verbose = '-v' in sys.argv
    for rep in range(9):
        t0 = time.perf_counter()
        n = sum(1 for _ in core.core_generation_funnel(None, list(pils), None, None, opts))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tr = core.FUNNEL_TRACE
        print(f"call {rep}: {dt * 1e3:.1f} ms, {n} results, {len(graphed.graphs)} graph(s)")
        if rep < 3:
            continue
        gs = tr.get("groups", [])
        if len(gs) >= 2:
            d = {n: tr["trace"][0][1].elapsed_time(e) for n, e, _ in gs[0]["trace"]}
            d1 = {n: tr["trace"][0][1].elapsed_time(e) for n, e, _ in gs[1]["trace"]}
            last = max(v for k, v in d1.items() if k.startswith("chunk"))
            conv = 1e3 * (max(gs[1]["converted"].values()) - tr["trace"][0][2])
            print(f"   head {d['inputs on the device']:.2f} | forward 0 {d['forward done'] - d['inputs on the device']:.2f} | gap before group 1 "
                  f"{d1['inputs on the device'] - d['per-pixel kernels done']:.2f} | forward 1 {d1['forward done'] - d1['inputs on the device']:.2f} | "
                  f"last chunk lands +{last - d1['per-pixel kernels done']:.2f} | last conversion +{conv - last:.2f}")
        if not verbose:
            continue
        base_ev, base_t = tr["trace"][0][1], tr["trace"][0][2]
        for gi, g in enumerate(tr.get("groups", [])):
            for name, ev, th in g["trace"]:
                print(f"   group {gi} ({len(g['idxs'])} units) {name:28s} device {base_ev.elapsed_time(ev):7.2f} ms   (enqueued at host {1e3 * (th - base_t):6.2f} ms)")
            conv = sorted(g.get("converted", {}).items())
            if conv:
                print("   group %d conversions done at host ms: %s" % (gi, " ".join("%.1f" % (1e3 * (t - base_t)) for _, t in conv)))


if __name__ == "__main__":
    main()
