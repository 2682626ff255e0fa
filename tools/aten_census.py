#!/usr/bin/env python3
"""Which python lines launch the aten / runtime kernels that are left in a forward of a bench network: torch.profiler with stacks,
only the ops that are not in-tree C-ABI calls, grouped by (op, input shapes, innermost repo frames).

    python tools/aten_census.py [model] [batch]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import bench  # noqa: E402


@torch.no_grad()
def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "dpt_beit_large_512"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda")
    model, _ = bench.build_model(name)
    model = model.to(dev).half()
    img, _ = bench.synth_batch(batch, 0)
    img = torch.from_numpy(img).to(dev)
    size = bench.default_net_size(name)
    for _ in range(2):
        model.infer_batch(img, size)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        model.infer_batch(img, size)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
        t = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
        if t <= 0:
            continue
        frames = [s for s in e.stack if "stable-diffusion-webui-depthmap-script_amd" in s or "bench" in s][:3]
        rows.append((t, e.count, e.key, str(e.input_shapes)[:90], " <- ".join(f.split("stable-diffusion-webui-depthmap-script_amd/")[-1][:60] for f in frames)))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"device time attributed to aten ops in one forward: {tot / 1e3:.3f} ms")
    for t, c, k, s, f in rows[:60]:
        print(f"{t / 1e3:8.3f} ms x{c:<3d} {k[:34]:34s} {s:90s} {f}")


if __name__ == "__main__":
    main()
