#!/usr/bin/env python3
"""bench.py -- depth + stereo pairs/sec @1024x1024 on MI355X (BASELINE.json's metric).

One *unit* = one input image -> one uint16 depth map + one side-by-side stereo pair (both eyes),
SURVEY.md 8(d).  One *step* = one pass of the hot path over a batch of `--batch` units already resident
in HBM:

    float32 depth prediction  --ds_depth_to_u16-->  uint16 depth      (core.py:189-211)
    RGB + uint16 depth        --ds_stereo_warp-->   left-right pair   (stereoimage_generation.py:13-92,
                                                                       polylines_sharp, divergence 2.5 %)

Round-1 scope note: the neural depth forward is not built yet, so the
float32 prediction is synthetic and is an INPUT of the timed region, not produced inside it.  The
timed region is the reference's per-pixel path (its numba/numpy part) for the whole batch.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per
GPU with torch.distributed.run.  Units are sharded across ranks (weak scaling: every rank renders its
own batch); with --gather the collated outputs are gathered to rank 0 with one RCCL gather per step,
overlapped with the next step's kernels.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

H = W = 1024
HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
ALGO_BYTES_PER_UNIT = 11 * H * W  # read RGB 3HW + depth u16 2HW, write two eyes 6HW


DEPTH_KIND = "steps"


def synth_batch(batch, seed):
    """Synthetic inputs of SURVEY.md 8(d): seeded RGB noise; depth prediction = smooth field with ramps, periodic steps
    and large occluders (float32, arbitrary scale, like a MiDaS output)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (batch, H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    pred = np.empty((batch, H, W), np.float32)
    for i in range(batch):
        ph = rng.uniform(0, 6.28, 4).astype(np.float32)
        f = 0.5 * xx / W + 0.25 * np.sin(xx / 97.0 + ph[0]) * np.cos(yy / 61.0 + ph[1]) + 0.05 * np.sin(xx / 9.0 + ph[2])
        if DEPTH_KIND != "smooth":
            f += 0.1 * (((xx // 64 + yy // 64) % 2) == 0)
            x0, y0 = int(rng.integers(0, W // 2)), int(rng.integers(0, H // 2))
            f[y0:y0 + H // 4, x0:x0 + W // 3] += 0.8
            f[(3 * H) // 4:, : W // 5] -= 0.4
        pred[i] = f * 37.0 + 5.0
    return img, pred


def cpu_baseline(distinct_units, seed, min_seconds=12.0):
    """The CPU oracle (C restatement of the reference's numba path, OpenMP over rows like numba's prange) on a bounded
    sample of the same workload, timed on this host's cores: `distinct_units` synthetic units are rendered round-robin
    until at least `min_seconds` of wall time have been spent (one untimed pass first, like the numba JIT warm-up)."""
    from oracle import oracle as orc
    orc.build()
    img, pred = synth_batch(distinct_units, seed)

    def one(i):
        d16 = orc.convert_to_i16(orc.depth_normalize01(pred[i], False))
        orc.create_stereoimages_arrays(img[i], d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')

    one(0)
    done = 0
    t0 = time.perf_counter()
    while True:
        one(done % distinct_units)
        done += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds and done >= distinct_units:
            break
    return {"value": done / dt, "unit": "pairs/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"{done} units of 1024x1024 ({distinct_units} distinct; depth->u16 + polylines_sharp left-right), "
                      f"gcc -O2 -fopenmp restatement of the reference's numba kernels, {dt:.2f} s"}


def pmc_traffic(batch):
    """HBM bytes per k_polylines launch from the committed rocprofv3 PMC summary (separate --pmc passes, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950): profiles/round1_pmc_summary.json.  None when the summary is
    missing or was taken at another batch size -- bench.py cannot read PMC counters itself."""
    try:
        with open(os.path.join(ROOT, "profiles", "round1_pmc_summary.json")) as f:
            j = json.load(f)
        if int(j.get("batch", -1)) != int(batch):
            return None
        return float(j["k_polylines"]["hbm_bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="units per GPU per step")
    ap.add_argument("--fill", default="polylines_sharp")
    ap.add_argument("--gather", action="store_true", help="gather the collated outputs to rank 0 (N > 1)")
    ap.add_argument("--depth", default="steps", choices=["steps", "smooth"],
                    help="synthetic prediction: smooth field + periodic steps + occluders (default), or smooth only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8, help="distinct units of the CPU baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="minimum wall time of the CPU baseline leg")
    args = ap.parse_args()
    global DEPTH_KIND
    DEPTH_KIND = args.depth

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import src._native as nat
    import src.stereoimage_generation as sg

    img_np, pred_np = synth_batch(args.batch, seed=1000 + rank)
    img = torch.from_numpy(img_np).to(dev)
    pred = torch.from_numpy(pred_np).to(dev)
    nat.profile_enable(local_rank, True)

    gather_ok = args.gather and world > 1
    side = torch.cuda.Stream(device=dev) if gather_ok else None
    gathered = None
    if gather_ok and rank == 0:
        gathered = [torch.empty((args.batch, H, 2 * W, 3), dtype=torch.uint8, device=dev) for _ in range(world)]

    render_ms, exact_ms, exact_rows = [], [], 0

    def step(timed):
        nonlocal exact_rows
        d16 = nat.depth_to_u16(pred, False)
        sbs = sg.create_stereoimages_batch(img, d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, args.fill)[0]
        if timed:
            pass
        if gather_ok:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                dist.gather(sbs, gathered if rank == 0 else None, dst=0)
                sbs.record_stream(side)
        return sbs

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # per-kernel timing (HIP events on the launch stream, recorded inside the C ABI): separate untimed passes so
    # the event synchronisation does not perturb the throughput measurement above
    for _ in range(min(args.steps, 10)):
        step(False)
        r, e = nat.profile_last_ms(local_rank)
        render_ms.append(r)
        exact_ms.append(e)
    exact_rows, queue_chunks = nat.last_stats(img)
    torch.cuda.synchronize()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        units = args.batch * world * args.steps
        avg_render_s = float(np.mean(render_ms)) * 1e-3
        achieved = args.batch * ALGO_BYTES_PER_UNIT / avg_render_s / 1e9
        out = {
            "metric": "depth+stereo pairs/sec @1024x1024",
            "value": units / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"depth->u16 + create_stereoimages({args.fill}, left-right, divergence 2.5%) on "
                                   f"{args.batch} x 1024x1024 RGB per GPU, inputs resident in HBM; float32 depth prediction "
                                   "is a synthetic input (model forward not built in round 1)",
                       "global_batch": args.batch * world, "height": H, "width": W,
                       "parallelism": f"units sharded over {world} GPU(s), no data-path collective"
                                      + (", RCCL gather to rank 0 overlapped" if gather_ok else "")},
            "roofline": {"bound": "hbm", "kernel": "k_polylines", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(args.batch),
                         "algorithmic_bytes_per_launch": args.batch * ALGO_BYTES_PER_UNIT,
                         "avg_kernel_ms": float(np.mean(render_ms)), "exact_fallback_ms": float(np.mean(exact_ms)),
                         "exact_fallback_rows": exact_rows, "general_pixels": queue_chunks},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, seed=1000, min_seconds=args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
