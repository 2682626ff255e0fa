#!/usr/bin/env python3
"""Where does one encoder block of the default bench spend its time?  Runs BEiT-L blocks (batch 32, 1025 tokens padded
to 1088, float16) under torch.profiler and prints, per aten op, the device kernels it launched; then times the
candidate formulations of the GEMMs of the block side by side.

    python tools/block_profile.py [profile] [gemms] [tune]      (tune: the same GEMM table under torch's TunableOp)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from src import vit_mi355x as vm  # noqa: E402
from dmidas.backbones import beit  # noqa: E402


def timeit(fn, reps=20, warm=3):
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


@torch.no_grad()
def main():
    which = set(sys.argv[1:]) or {"profile", "gemms"}
    dev = torch.device("cuda")
    B, N, C = 32, 1025, 1024
    npad = vm.pad_len(N)
    torch.manual_seed(0)
    if "profile" in which:
        blocks = torch.nn.ModuleList([beit.Block(C, 16, (32, 32)) for _ in range(2)]).to(dev).half().eval()
        x = torch.randn(B, npad, C, device=dev, dtype=torch.float16)
        for _ in range(2):
            vm.run_blocks(blocks, x, N, (32, 32), ())
        torch.cuda.synchronize()
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            for _ in range(3):
                vm.run_blocks(blocks, x, N, (32, 32), ())
            torch.cuda.synchronize()
        print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40,
                                                                 max_name_column_width=70, max_shapes_column_width=60))
    if "tune" in which:
        import torch.cuda.tunable as tun
        tun.enable(True)
        tun.tuning_enable(True)
        tun.set_max_tuning_duration(15)
        tun.set_max_tuning_iterations(20)
        tun.set_filename(os.path.join(ROOT, "gpurun_out", "tunableop_results.csv"))
        which.add("gemms")
    if "gemms" in which:
        h = torch.randn(B, npad, C, device=dev, dtype=torch.float16)
        w = torch.randn(3 * C, C, device=dev, dtype=torch.float16) * 0.02
        bq = torch.randn(2 * C, device=dev, dtype=torch.float16)
        b3 = torch.randn(3 * C, device=dev, dtype=torch.float16)
        w_qk, w_v = w[:2 * C], w[2 * C:]
        w_v_c = w_v.contiguous()
        h2 = h.view(B * npad, C)
        fl = 2.0 * B * npad * C * C
        rows = [
            ("qk   F.linear 3d view-weight", lambda: F.linear(h, w_qk, bq), 2 * fl),
            ("qk   F.linear 2d", lambda: F.linear(h2, w_qk, bq), 2 * fl),
            ("qkv  F.linear 2d (one GEMM)", lambda: F.linear(h2, w, b3), 3 * fl),
            ("v    F.linear 2d (not transposed)", lambda: F.linear(h2, w_v), fl),
            ("v^T  bmm(expand(w_v), h^T)", lambda: torch.bmm(w_v.unsqueeze(0).expand(B, -1, -1), h.transpose(1, 2)), fl),
            ("v^T  bmm(expand(w_v contiguous), h^T)", lambda: torch.bmm(w_v_c.unsqueeze(0).expand(B, -1, -1), h.transpose(1, 2)), fl),
            ("v^T  matmul(w_v, h^T)", lambda: torch.matmul(w_v_c, h.transpose(1, 2)), fl),
            ("v^T  (h @ w_v^T).transpose.contiguous", lambda: F.linear(h, w_v).transpose(1, 2).contiguous(), fl),
            ("proj F.linear 3d", lambda: F.linear(h, w_v_c, bq[:C]), fl),
        ]
        hid = torch.randn(B * npad, 4 * C, device=dev, dtype=torch.float16)
        w1 = torch.randn(4 * C, C, device=dev, dtype=torch.float16) * 0.02
        w2 = torch.randn(C, 4 * C, device=dev, dtype=torch.float16) * 0.02
        b1 = torch.randn(4 * C, device=dev, dtype=torch.float16)
        rows += [
            ("fc1  F.linear", lambda: F.linear(h2, w1, b1), 4 * fl),
            ("gelu", lambda: F.gelu(hid), 0.0),
            ("fc1+gelu", lambda: F.gelu(F.linear(h2, w1, b1)), 4 * fl),
            ("fc2  F.linear", lambda: F.linear(hid, w2, bq[:C]), 4 * fl),
        ]
        import time
        for name, fn, flops in rows:
            t0 = time.time()
            fn()
            torch.cuda.synchronize()
            first = time.time() - t0
            ms = timeit(fn)
            if "tune" in which:
                name = f"[tuned, first call {first:5.1f} s] " + name
            print(f"{name:42s} {ms * 1e3:8.1f} us  {flops / ms / 1e9:8.1f} TF/s")


if __name__ == "__main__":
    main()
