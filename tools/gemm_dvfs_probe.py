#!/usr/bin/env python3
"""Workload of the GEMM clock probe: the in-tree GEMM at 33024 x 4096 x 4096 (f16) on random operands, on zero operands and on
half the chip (128 workgroups), ten launches each, to be run under

    rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d <dir> -o p -- python tools/gemm_dvfs_probe.py
    python tools/gemm_dvfs_probe.py --parse <dir>

GRBM_GUI_ACTIVE counts the cycles the graphics engine was busy: divided by the kernel's duration it is the EFFECTIVE shader clock
of that launch (MI355X_MICROARCH.md, "DVFS give-back").  --parse prints, per phase, duration, busy cycles and clock.
"""
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(d):
    dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    for path in dbs:
        db = sqlite3.connect(path)
        names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        print(path, "tables/views:", names)
        for t in names:
            if "counters_collection" in t or "kernels" == t or "kernel" in t.lower():
                cols = [r[1] for r in db.execute(f"pragma table_info('{t}')")]
                print(" ", t, cols)
        try:
            cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
            q = "select * from counters_collection where kernel_name like '%k_linear256%' order by dispatch_id limit 400"
            rows = list(db.execute(q))
            print(cols)
            for r in rows[:6]:
                print(r)
            i_name, i_val, i_disp = cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
            st = cols.index("start") if "start" in cols else None
            en = cols.index("end") if "end" in cols else None
            per = {}
            for r in rows:
                if r[i_name] != "GRBM_GUI_ACTIVE":
                    continue
                e = per.setdefault(r[i_disp], {"vals": [], "t": None})
                e["vals"].append(float(r[i_val]))
                if st is not None:
                    e["t"] = (r[st], r[en])
            for dsp in sorted(per):
                e = per[dsp]
                dur = (e["t"][1] - e["t"][0]) if e["t"] else None
                mx = max(e["vals"])
                print(f"dispatch {dsp}: instances {len(e['vals'])} max busy cycles {mx:.0f} sum {sum(e['vals']):.0f}"
                      + (f" duration {dur / 1e3:.1f} us -> {mx / dur:.3f} GHz" if dur else ""))
        except Exception as ex:  # noqa: BLE001
            print("parse failed:", ex)


def main():
    for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
        sys.path.insert(0, p)
    import torch
    from src import _native as nat
    dev = torch.device("cuda")
    m, n, k = 33024, 4096, 4096
    xs = torch.randn(m, k, device=dev, dtype=torch.float16)
    ws = torch.randn(n, k, device=dev, dtype=torch.float16) * k ** -0.5
    bs = torch.randn(n, device=dev, dtype=torch.float16)
    zx, zw = torch.zeros_like(xs), torch.zeros_like(ws)
    for _ in range(3):
        nat.linear(xs, ws, bs, False)
    torch.cuda.synchronize()
    for name, a, b, grid in (("random", xs, ws, None), ("zeros", zx, zw, None), ("random grid 128", xs, ws, 128), ("random again", xs, ws, None)):
        nat.linear_env(DS_LIN_GRID=grid)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            nat.linear(a, b, bs, False)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:18s}: {ms * 1e3:8.1f} us per launch  {2.0 * m * n * k / ms / 1e9:6.0f} TF/s", flush=True)
    nat.linear_env(DS_LIN_GRID=None)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        main()
