#!/usr/bin/env python3
"""torch.profiler over ONE forward of the default bench network (after warm-up), grouped by aten op and input shape:
attributes every device kernel (library ones included) to the op that launched it.

    python tools/model_profile.py [model] [batch]
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        model.infer_batch(img, size)
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=70,
                                                             max_name_column_width=60, max_shapes_column_width=90))


if __name__ == "__main__":
    main()
