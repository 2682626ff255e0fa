#!/usr/bin/env python3
"""Shader clock and board power while the in-tree GEMM runs on random, zero and half-chip launches (the evidence behind
DESIGN.md's "the GEMMs are power limited" paragraph).  Samples the amdgpu hwmon files (freq1_input = sclk in Hz, power1_average /
power1_input in microwatts; rocm-smi as a fallback) from a thread while the main thread keeps the device busy.

    python tools/clock_probe.py
"""
import glob
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
from src import _native as nat  # noqa: E402


def hwmon_files():
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        f = os.path.join(d, "freq1_input")
        pw = [os.path.join(d, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(d, n))]
        if os.path.exists(f):
            return f, (pw[0] if pw else None), os.path.join(d, "power1_cap") if os.path.exists(os.path.join(d, "power1_cap")) else None
    return None, None, None


def read_num(path):
    try:
        with open(path) as f:
            return float(f.read().strip())
    except Exception:
        return float("nan")


class Sampler(threading.Thread):
    def __init__(self, fclk, fpow):
        super().__init__(daemon=True)
        self.fclk, self.fpow, self.rows, self.stop = fclk, fpow, [], False

    def run(self):
        while not self.stop:
            self.rows.append((read_num(self.fclk) / 1e6 if self.fclk else float("nan"), read_num(self.fpow) / 1e6 if self.fpow else float("nan")))
            time.sleep(0.02)


def main():
    dev = torch.device("cuda")
    fclk, fpow, fcap = hwmon_files()
    print("hwmon:", fclk, fpow, "cap W:", read_num(fcap) / 1e6 if fcap else None, flush=True)
    if fclk is None:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower"], capture_output=True, text=True).stdout
        print(out[-1500:])
    m, n, k = 34816, 4096, 4096
    xs = torch.randn(m, k, device=dev, dtype=torch.float16)
    ws = torch.randn(n, k, device=dev, dtype=torch.float16) * k ** -0.5
    bs = torch.randn(n, device=dev, dtype=torch.float16)
    zx, zw = torch.zeros_like(xs), torch.zeros_like(ws)
    for name, a, b, grid in (("idle", None, None, None), ("random operands, 256 workgroups", xs, ws, None), ("zero operands, 256 workgroups", zx, zw, None),
                             ("random operands, 128 workgroups", xs, ws, 128), ("random operands, 256 workgroups (again)", xs, ws, None)):
        nat.linear_env(DS_LIN_GRID=grid)
        s = Sampler(fclk, fpow)
        s.start()
        t0 = time.perf_counter()
        reps = 0
        if a is None:
            time.sleep(1.0)
        else:
            while time.perf_counter() - t0 < 2.0:
                for _ in range(20):
                    nat.linear(a, b, bs, False)
                torch.cuda.synchronize()
                reps += 20
        dt = time.perf_counter() - t0
        s.stop = True
        s.join()
        rows = s.rows[len(s.rows) // 3:]                     # the steady part
        clk = sorted(r[0] for r in rows)
        pw = sorted(r[1] for r in rows)
        tf = 2.0 * m * n * k * reps / dt / 1e12 if reps else 0.0
        print(f"{name:42s}: {tf:7.0f} TF/s   sclk MHz median {clk[len(clk) // 2]:6.0f} (min {clk[0]:6.0f} max {clk[-1]:6.0f})   "
              f"power W median {pw[len(pw) // 2]:6.0f} (max {pw[-1]:6.0f})   samples {len(rows)}", flush=True)
    nat.linear_env(DS_LIN_GRID=None)


if __name__ == "__main__":
    main()
