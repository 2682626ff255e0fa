#!/usr/bin/env python3
"""bench.py -- depth + stereo pairs/sec @1024x1024 on MI355X (BASELINE.json's metric).

One *unit* = one RGB image -> one uint16 depth map + one side-by-side stereo pair (both eyes) + one normal map
(BASELINE config 3 is "depth + normalmap" at the metric's 1024x1024).  One *step* = one pass of the hot
path over a batch of `--batch` units already resident in HBM:

    uint8 RGB batch    --model forward (fp16, MFMA)-->  float32 depth prediction   (depthmap_generation.py:375-403 +
                                                                                    the model family's estimate*())
    float32 prediction --ds_depth_to_u16-------------->  uint16 depth              (core.py:189-211)
    RGB + uint16 depth --ds_stereo_warp--------------->  left-right pair           (stereoimage_generation.py:13-92,
                                                                                    polylines_sharp, divergence 2.5 %)
    uint16 depth       --ds_normalmap----------------->  normal map                (normalmap_generation.py:5-56)

fp16 is the reference's own GPU default for these networks (src/depthmap_generation.py:268-275); the GPU tests hold the
fp16 forward to 2e-2 of the reference's float32 output and the float32 forward to 1e-4 (tests/test_gpu_models.py).

`--model` picks the depth network (random-init weights of the named architecture -- there are no checkpoints offline):
    dpt_beit_large_512   MiDaS 3.1 DPT BEiT-L/16, net 512   (reference model id 1;  BASELINE config 3) [default]
    dav2_vitl            Depth-Anything-V2 ViT-L/14, net 518 (reference model id 14; BASELINE config 5)
    dpt_hybrid_384       MiDaS 3.0 ViT-B/16 + ResNetV2-50, net 384 (reference model id 4; BASELINE config 2)
    none                 no network: the float32 prediction is a synthetic INPUT, only the per-pixel path is timed
`--config` presets: c2 (dpt_hybrid_384, batch 1, 512x512: latency), c3 (default), c3match (config 3 with NET_SIZE_MATCH:
net 1024, 4097 tokens), c5 (dav2_vitl on 1920x1080 frames, 2443 tokens).  Only c3 is the metric's line; the others are
kept profile lines (profiles/round2_*).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU with
torch.distributed.run.  Units are sharded across ranks (weak scaling: every rank renders its own batch, no data-path
collective) and the collated outputs (stereo pair + uint16 depth + normal map, packed into one byte buffer) are gathered
to rank 0 with ONE RCCL gather per step, overlapped with the next step's kernels (--no-gather to leave it out).  Rank 0
prints ONE JSON line; on one GPU the metric's config also carries `funnel`: the same batch through the drop-in boundary
(core_generation_funnel, PIL in -> PIL out), reported beside `value`, never as it.
"""
import argparse
import contextlib
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
HBM_COPY_GBPS = 6290.0            # what a float4 copy kernel reaches on this chip (same guide: 79 % of the spec peak)
MFMA_PEAK_TFLOPS = 2500.0         # dense f16/bf16 MFMA peak (same guide)
PMC_SUMMARY = "profiles/round4_pmc_summary.json"      # tools/pmc_summary.py over rocprofv3 --pmc passes of THIS command
KERNEL_STATS = "profiles/round4_kernel_stats.csv"     # rocprofv3 --kernel-trace --stats of THIS command

DEPTH_KIND = "steps"


def algo_bytes_stereo():
    return 11 * H * W             # read RGB 3HW + depth u16 2HW, write two eyes 6HW


def algo_bytes_normalmap():
    return 5 * H * W              # uint16 in 2HW + RGB out 3HW


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


def default_net_size(model_name):
    return {"dav2_vitl": 518, "dpt_beit_large_512": 512, "dpt_hybrid_384": 384}.get(model_name, 0)


def build_model(name, seed=0):
    """Random-init network of the named architecture (torch.manual_seed(seed); no checkpoints offline)."""
    import torch
    torch.manual_seed(seed)
    if name == "dav2_vitl":
        from ddepth_anything_v2 import DepthAnythingV2
        m = DepthAnythingV2(encoder='vitl', features=256, out_channels=[256, 512, 1024, 1024])
        info = {"name": "Depth-Anything-V2 ViT-L/14", "patch": 14, "dim": 1024, "depth": 24, "heads": 16, "bias": False}
    elif name == "dpt_beit_large_512":
        from dmidas.dpt_depth import DPTDepthModel
        m = DPTDepthModel(path=None, backbone="beitl16_512", non_negative=True)
        info = {"name": "MiDaS 3.1 DPT BEiT-L/16 512", "patch": 16, "dim": 1024, "depth": 24, "heads": 16, "bias": True}
    elif name == "dpt_hybrid_384":
        from dmidas.dpt_depth import DPTDepthModel
        m = DPTDepthModel(path=None, backbone="vitb_rn50_384", non_negative=True)
        info = {"name": "MiDaS 3.0 dpt_hybrid_384 (ViT-B/16 + ResNetV2-50)", "patch": 16, "dim": 768, "depth": 12, "heads": 12, "bias": False}
    else:
        raise SystemExit(f"unknown --model {name}")
    return m.eval(), info


def net_grid(model_name, net_size, net_h=None):
    """(rows, cols) of the token grid the network runs at for an H x W image -- the same size rules the product applies."""
    if model_name == "dav2_vitl":
        from ddepth_anything_v2.depth_anything_v2.dpt import lower_bound_size
        nw, nh = lower_bound_size(W, H, net_size)
        return nh // 14, nw // 14
    from dmidas.dpt_depth import midas_net_size
    nw, nh = midas_net_size(W, H, net_size, net_size if net_h is None else net_h, "minimal")
    return nh // 16, nw // 16


def run_forward(model, model_name, img, net_size, net_h=None):
    # (Round 4 tried the batch as 2 / 4 micro-batches on as many streams, GEMMs on half the CUs, so that one micro-batch's
    # attention / LayerNorm / decoder kernels run beside another's GEMMs: 792.6 / 778.5 / 688.2 pairs/s against 791.6 on the same
    # box -- the chip is power limited during the GEMMs, concurrency moves work around without adding any; removed.)
    return _forward_one(model, model_name, img, net_size, net_h)


def _forward_one(model, model_name, img, net_size, net_h=None):
    if model_name == "dav2_vitl":
        return model.infer_batch(img, net_size)
    return model.infer_batch(img, net_size=net_size, resize_mode="minimal", net_h=net_h)


# ---- CPU baseline -------------------------------------------------------------------------------------------------------
def cpu_baseline(model_name, net_size, net_h, distinct_units, seed, min_seconds, normalmap, python_unit, init_seed=0):
    """The same workload on this host's cores, bounded: the float32 torch-eager forward of the same network (what the
    reference runs on a CPU device) + the CPU oracle (C restatement of the reference's numba kernels, OpenMP over rows
    like numba's prange) + the numpy normal map.  `distinct_units` units are processed round-robin until `min_seconds`
    have been spent.  Beside it: one unit of `python_unit`^2 through the pure-Python restatement of the reference's
    numba-less fallback (what the reference runs when numba is missing, src/stereoimage_generation.py:1-8), 1 core."""
    import torch
    from oracle import oracle as orc
    orc.build()
    img, pred = synth_batch(distinct_units, seed)
    model = None
    if model_name != "none":
        model, _ = build_model(model_name, init_seed)
        model = model.float()

    def one(i):
        if model is not None:
            with torch.no_grad():
                p = run_forward(model, model_name, torch.from_numpy(img[i:i + 1]), net_size, net_h).numpy()[0]
        else:
            p = pred[i]
        d16 = orc.convert_to_i16(orc.depth_normalize01(p, False))
        orc.create_stereoimages_arrays(img[i], d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')
        if normalmap:
            orc.create_normalmap_array(d16)

    one(0)
    done = 0
    t0 = time.perf_counter()
    while True:
        one(done % distinct_units)
        done += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds and done >= min(distinct_units, 4):
            break
    what = "torch-eager float32 forward of the same network on the CPU + " if model is not None else ""
    out = {"value": done / dt, "unit": "pairs/s", "cores": max(orc.num_threads(), torch.get_num_threads()), "kind": "port",
           "sample": f"{done} units of {H}x{W} ({distinct_units} distinct): {what}depth->u16 + polylines_sharp left-right "
                     f"with the gcc -O2 -fopenmp restatement of the reference's numba kernels"
                     + (" + the numpy normal map" if normalmap else "") + f", {dt:.2f} s"}
    if python_unit > 0:
        from oracle import oracle_py
        s = int(python_unit)
        sub, sd = img[0, :s, :s], orc.convert_to_i16(orc.depth_normalize01(pred[0, :s, :s], False))
        t1 = time.perf_counter()
        got = oracle_py.create_stereoimages_arrays(sub, sd, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')[0]
        dt1 = time.perf_counter() - t1
        same = bool(np.array_equal(got, orc.create_stereoimages_arrays(sub, sd, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')[0]))
        out["python_fallback"] = {"value": 1.0 / dt1, "unit": "pairs/s", "cores": 1, "kind": "port",
                                  "sample": f"1 unit of {s}x{s} (BASELINE config 1's size), polylines_sharp left-right, pure-Python "
                                            f"restatement of the reference's numba-less fallback, {dt1:.2f} s; stereo stage only",
                                  "identical_to_c_port": same}
    return out


def traffic_from_profile(kernel, batch):
    """HBM bytes per launch of the kernel whose name contains `kernel`, from the committed rocprofv3 PMC summary of the default
    bench command (separate --pmc passes for FETCH_SIZE and WRITE_SIZE, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950).  bench.py cannot read PMC counters itself: this is a figure FROM A PROFILE of the same command, labelled as
    such; None when the summary is missing, was taken at another batch size, or lacks the kernel."""
    try:
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            j = json.load(f)
        if int(j.get("batch", -1)) != int(batch):
            return None
        for name, row in j.items():
            if isinstance(row, dict) and kernel in name and "hbm_read_bytes" in row and "hbm_write_bytes" in row:
                return {"hbm_bytes_per_launch": float(row["hbm_read_bytes"]) + float(row["hbm_write_bytes"]),
                        "hbm_read_bytes": float(row["hbm_read_bytes"]), "hbm_write_bytes": float(row["hbm_write_bytes"]),
                        "kernel": name, "dispatches": row.get("dispatches"), "source": PMC_SUMMARY}
    except Exception:
        pass
    return None


def clock_from_profile(kernel, flops, batch):
    """Effective shader clock and MFMA cycle fraction of the kernel whose name contains `kernel` inside the step, from the committed
    profiles: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs by rocprofv3) of the PMC summary / 8 = cycles of one launch;
    / the average duration of the kernel-trace summary = the clock the power management granted (MI355X_MICROARCH.md, "DVFS
    give-back"); the MFMA work of `flops` is flops / (1024 SIMDs x 1024 flop per cycle) cycles.  None when a profile lacks it."""
    try:
        import csv
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            j = json.load(f)
        if int(j.get("batch", -1)) != int(batch):
            return None
        cyc = next(float(r["GRBM_GUI_ACTIVE"]) / 8.0 for n, r in j.items() if isinstance(r, dict) and kernel in n and "GRBM_GUI_ACTIVE" in r)
        with open(os.path.join(ROOT, KERNEL_STATS)) as f:
            ns = next(float(r["AverageNs"]) for r in csv.DictReader(f) if kernel in r["Name"])
        mfma = flops / (1024.0 * 1024.0)
        return {"busy_cycles_per_launch": cyc, "effective_clock_ghz": cyc / ns, "mfma_cycles": mfma, "mfma_cycle_frac": mfma / cyc,
                "note": "the chip clocks down under dense MFMA work on random operands (DESIGN.md 3.10): `frac` is against the 2.4 GHz peak, "
                        "`mfma_cycle_frac` is the share of the launch's cycles that are MFMA issue cycles", "source": PMC_SUMMARY + " + " + KERNEL_STATS}
    except Exception:
        return None


def in_step_from_profile(kernel, work, peak, unit_scale):
    """Average duration of the kernel whose name contains `kernel` INSIDE the timed step, from the committed rocprofv3
    --kernel-trace --stats summary of the default bench command, and what that duration means for `work` (flops or bytes per
    launch): the microbenchmark figures beside it run the same launch shape on randn operands."""
    try:
        import csv
        with open(os.path.join(ROOT, KERNEL_STATS)) as f:
            for row in csv.DictReader(f):
                if kernel in row["Name"]:
                    ms = float(row["AverageNs"]) * 1e-6
                    ach = work / (ms * 1e-3) / unit_scale
                    return {"avg_kernel_ms": ms, "calls": int(row["Calls"]), "achieved": ach, "frac": ach / peak, "kernel": row["Name"],
                            "source": KERNEL_STATS}
    except Exception:
        pass
    return None


def main():
    global H, W, DEPTH_KIND
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c3match", "c4", "c5"], help="BASELINE.json config preset (see the header)")
    ap.add_argument("--batch", type=int, default=None, help="units per GPU per step")
    ap.add_argument("--model", default=None, choices=["dav2_vitl", "dpt_beit_large_512", "dpt_hybrid_384", "none"])
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--net-size", type=int, default=None, help="network input size (default: the model's; 0 = NET_SIZE_MATCH)")
    ap.add_argument("--fill", default="polylines_sharp")
    ap.add_argument("--no-normalmap", action="store_true", help="leave the normal map out of the step (round-1 workload)")
    ap.add_argument("--overlap", action="store_true", help="run the per-pixel kernels of step k on a second stream beside the forward of "
                                                         "step k+1 (measured: +0.5 %, inside the run-to-run noise; off by default)")
    ap.add_argument("--graph", action="store_true", help="replay the network forward as a hipGraph (default for the batch-1 config c2)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: do not gather the collated outputs to rank 0")
    ap.add_argument("--funnel", action="store_true", help="also time the drop-in funnel (PIL in -> PIL out) on the same batch "
                                                        "(default for the metric's config c3 on one GPU; reported beside `value`, never as it)")
    ap.add_argument("--no-funnel", action="store_true")
    ap.add_argument("--depth", default="steps", choices=["steps", "smooth"],
                    help="--model none only: synthetic prediction with steps + occluders (default), or smooth only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-route-check", action="store_true", help="leave out the untimed route check (profile runs: its library-routed "
                                                                  "forward would show up in the kernel statistics of the step)")
    ap.add_argument("--tune-gemms", default=None, metavar="CSV",
                    help="run with TunableOp tuning ON and accumulate the winners in CSV (maintenance: regenerates "
                         "src/tunableop_gfx950.csv); the default run only READS the shipped file")
    ap.add_argument("--cpu-sample", type=int, default=4, help="distinct units of the CPU baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="minimum wall time of the CPU baseline leg")
    ap.add_argument("--cpu-python-unit", type=int, default=512, help="side of the one unit timed through the pure-Python port (0 = skip)")
    ap.add_argument("--boost-rmax", type=int, default=1600, help="c4: Boost's whole-image size limit (standalone default 1600, paper 3000)")
    args = ap.parse_args()
    if args.config == "c4":
        return run_c4(args)
    preset = {"c2": ("dpt_hybrid_384", 1, 512, 512, None), "c3": ("dpt_beit_large_512", 32, 1024, 1024, None),
              "c3match": ("dpt_beit_large_512", 8, 1024, 1024, 0), "c5": ("dav2_vitl", 8, 1080, 1920, None)}[args.config]
    model_name = args.model or preset[0]
    batch = args.batch or preset[1]
    H = args.height or preset[2]
    W = args.width or preset[3]
    net_size = args.net_size if args.net_size is not None else preset[4]
    net_h = None
    if net_size is None:
        net_size = default_net_size(model_name)
    elif net_size == 0:                                            # NET_SIZE_MATCH (core.py:177-181)
        net_size, net_h = (W + 31) // 32 * 32, (H + 31) // 32 * 32
    DEPTH_KIND = args.depth
    normalmap = not args.no_normalmap

    import torch
    import torch.distributed as dist
    # MIOpen's search over its solvers for the ~15 library convolution shapes of a forward (the 256 -> 128 head convolution gets a
    # CK kernel at 880 us instead of the heuristic's 2.0 ms igemm): +2.4-3 % on the step, ~40 s of the untimed priming pass on a
    # fresh box (measured: 9 s -> 49 s wall for the whole command).  DS_CUDNN_BENCHMARK=0 leaves the heuristic choice.
    # (not for the batch-1 latency line c2: its forward is replayed from a hipGraph, where MIOpen cannot be given a workspace,
    # and the searched choices measured slower there: 4.99 vs 4.58 ms)
    # MIOpen's solver search (torch.backends.cudnn.benchmark): off by default since the end of round 3.  It bought 1.3-3 % while the
    # library convolutions still carried their bias; since they run without it (vit_mi355x.conv_module) the heuristic choice is
    # the same CK kernels: 806.6 (search) vs 809.7 (heuristic) pairs/s on the same box, and the search costs ~40 s of the untimed
    # priming pass on a fresh box.  DS_CUDNN_BENCHMARK=1 switches it on.
    miopen_find = os.environ.get("DS_CUDNN_BENCHMARK", "0") != "0"
    if miopen_find:
        torch.backends.cudnn.benchmark = True
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
    import src.normalmap_generation as nmg
    import src.stereoimage_generation as sg
    from src import multigpu
    from src import vit_mi355x as vm

    img_np, pred_np = synth_batch(batch, seed=1000 + rank)
    img = torch.from_numpy(img_np).to(dev)
    pred_in = torch.from_numpy(pred_np).to(dev)
    nat.profile_enable(local_rank, True)
    model, minfo = (None, None)
    if model_name != "none":
        from src import gemm_tuning
        if args.tune_gemms:
            gemm_tuning.enable(tune=True, results=os.path.abspath(args.tune_gemms))
        else:
            gemm_tuning.enable()
        # a random-init network may emit a (near-)constant map (dead final ReLUs): such a step would render a degenerate
        # stereo pair and time nothing real -- take the first seed whose prediction varies on every unit, else stop
        for seed in range(8):
            model, minfo = build_model(model_name, seed)
            model = model.to(dev).half()                   # the reference's default on a GPU (depthmap_generation.py:268-275)
            p = run_forward(model, model_name, img, net_size, net_h)
            lo, hi = p.flatten(1).min(1).values, p.flatten(1).max(1).values
            if bool(((hi - lo) > 1e-3 * hi.abs().clamp_min(1e-6)).all()):
                minfo["init_seed"] = seed
                break
            model = None
        if model is None:
            raise SystemExit(f"{model_name}: 8 random initialisations all gave a constant depth map on some unit; refusing to "
                             "time a degenerate stereo workload")
        gh, gw = net_grid(model_name, net_size, net_h)
        minfo["tokens"] = gh * gw + 1
        minfo["net"] = f"{gw * minfo['patch']}x{gh * minfo['patch']}"

    gather_ok = world > 1 and not args.no_gather
    side = torch.cuda.Stream(device=dev) if gather_ok else None
    gathered = None
    # the collated output of a unit is its stereo pair, its uint16 depth map and its normal map (north_star: "a single RCCL
    # gather ... for the collated output"): packed into ONE byte buffer per rank, ONE gather per step
    unit_bytes = H * 2 * W * 3 + H * W * 2 + (H * W * 3 if normalmap else 0)
    if gather_ok and rank == 0:
        gathered = [torch.empty((batch, unit_bytes), dtype=torch.uint8, device=dev) for _ in range(world)]

    # --overlap: the per-pixel kernels (float64 VALU / LDS bound) of step k run on their own stream beside the network forward
    # of step k+1 (MFMA bound); the units of every step are still complete inside the timed region.  Default: one stream.
    post = torch.cuda.Stream(device=dev) if (model is not None and args.overlap) else None

    # launch-bound shapes (batch 1): the forward's ~600 launches are captured once into a hipGraph and replayed
    use_graph = model is not None and (args.graph or (args.config == "c2" and not args.no_graph))
    fwd = None
    if use_graph:
        from src.hip_graph import GraphedForward
        fwd = GraphedForward(lambda x: run_forward(model, model_name, x, net_size, net_h))

    def step(check=False):
        if model is not None:
            pred = fwd(img) if fwd is not None else run_forward(model, model_name, img, net_size, net_h)
        else:
            pred = pred_in
        if check:                                            # outside the timed region: every unit's prediction varies
            lo, hi = pred.flatten(1).min(1).values, pred.flatten(1).max(1).values
            assert bool((hi > lo).all()), "degenerate (constant) depth prediction: the stereo leg would be meaningless"
        if post is not None:
            ready = torch.cuda.Event()
            ready.record()
            post.wait_event(ready)
            pred.record_stream(post)
        with torch.cuda.stream(post) if post is not None else contextlib.nullcontext():
            d16 = nat.depth_to_u16(pred, False)
            sbs = sg.create_stereoimages_batch(img, d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, args.fill)[0]
            nmap = nmg.create_normalmap_batch(d16) if normalmap else None
            if gather_ok:
                packed, _ = multigpu.pack_collated([sbs, d16] + ([nmap] if normalmap else []))
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    dist.gather(packed, gathered if rank == 0 else None, dst=0)
                    packed.record_stream(side)
        return sbs, nmap, d16

    step(check=True)            # priming pass, never timed: library kernel selection (MIOpen find), bias operands, allocator
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # per-kernel timing in separate untimed passes (event synchronisation must not perturb the throughput measurement):
    # k_polylines via HIP events recorded inside the C ABI on the launch stream, ds_normalmap via events on torch's current
    # stream (which IS the launch stream of the ctypes call) ...
    render_ms, exact_ms, nm_ms = [], [], []
    for _ in range(min(args.steps, 5)):
        _, _, d16 = step()
        torch.cuda.synchronize()
        r, e = nat.profile_last_ms(local_rank)
        render_ms.append(r)
        exact_ms.append(e)
        if normalmap:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            nmg.create_normalmap_batch(d16)
            e1.record()
            e1.synchronize()
            nm_ms.append(e0.elapsed_time(e1))
    exact_rows, general_px = nat.last_stats(img)
    # ... and the fused attention kernel at exactly the shape one encoder block launches it with
    attn = None
    if model is not None:
        n_tok = minfo["tokens"]
        npad = vm.pad_len(n_tok, batch)
        qk = torch.randn(batch, npad, 2, minfo["heads"], 64, device=dev, dtype=torch.float16)
        vt = torch.randn(batch, minfo["heads"] * 64, npad, device=dev, dtype=torch.float16)
        bias = None
        if minfo["bias"]:
            bias = nat.attention_bias_pack(torch.randn(minfo["heads"], n_tok, n_tok, device=dev), npad, torch.float16)
        for _ in range(3):
            nat.attention_fwd(qk, vt, n_tok, 0.125, bias)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            nat.attention_fwd(qk, vt, n_tok, 0.125, bias)
        e1.record()
        e1.synchronize()
        attn_ms = e0.elapsed_time(e1) / reps
        attn_flops = 4.0 * n_tok * n_tok * minfo["dim"] * batch            # QK^T + PV, 2 flops per MAC
        attn = {"bound": "mfma", "kernel": "k_attention_fwd", "achieved": attn_flops / (attn_ms * 1e-3) / 1e12,
                "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": attn_flops / (attn_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                "traffic": None,                          # PMC counters cannot be read from inside the run ...
                "traffic_from_profile": traffic_from_profile("k_attention_fwd2", batch) if args.config == "c3" else None,
                "in_step_from_profile": in_step_from_profile("k_attention_fwd2", attn_flops, MFMA_PEAK_TFLOPS, 1e12) if args.config == "c3" else None,
                "clock_from_profile": clock_from_profile("k_attention_fwd2", attn_flops, batch) if (args.config == "c3" and minfo["bias"]) else None,
                "algorithmic_flops_per_launch": attn_flops, "avg_kernel_ms": attn_ms, "operands": "random (randn)",
                "source": "microbenchmark: separate launches at the in-step shape on randn operands, HIP events on the launch stream",
                "launches_per_step": minfo["depth"], "shape": {"batch": batch, "tokens": n_tok, "heads": minfo["heads"], "bias": minfo["bias"]}}
    # ... and the in-tree MFMA GEMM (csrc/ds_linear.hip) at the shapes it runs at: fc1 + GELU of one encoder block, and the
    # 3x3 convolution of the decoder's last residual units (256 -> 256 at net/4 resolution), random operands
    lin = conv_roof = None
    if model is not None and vm.LINEAR_HIP != "0":
        m_rows, dim = batch * vm.pad_len(minfo["tokens"], batch), minfo["dim"]
        xw = torch.randn(m_rows, dim, device=dev, dtype=torch.float16)
        ww = torch.randn(4 * dim, dim, device=dev, dtype=torch.float16) * dim ** -0.5
        bw = torch.randn(4 * dim, device=dev, dtype=torch.float16)
        if nat.linear_supported(xw, ww):
            for _ in range(3):
                nat.linear(xw, ww, bw, True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                nat.linear(xw, ww, bw, True)
            e1.record()
            e1.synchronize()
            lin_ms = e0.elapsed_time(e1) / 20
            lin_flops = 2.0 * m_rows * 4 * dim * dim
            lin = {"bound": "mfma", "kernel": "k_linear256 (fc1 + erf-GELU)", "achieved": lin_flops / (lin_ms * 1e-3) / 1e12,
                   "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": lin_flops / (lin_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                   "traffic": None, "traffic_from_profile": traffic_from_profile("k_linear256<0, 1, 0, 0, 0", batch) if args.config == "c3" else None,
                   "in_step_from_profile": in_step_from_profile("k_linear256<0, 1, 0, 0, 0", lin_flops, MFMA_PEAK_TFLOPS, 1e12) if args.config == "c3" else None,
                   "clock_from_profile": clock_from_profile("k_linear256<0, 1, 0, 0, 0", lin_flops, batch) if args.config == "c3" else None,
                   "algorithmic_flops_per_launch": lin_flops, "avg_kernel_ms": lin_ms, "operands": "random (randn)",
                   "source": "microbenchmark: separate launches at the in-step shape on randn operands, HIP events on the launch stream",
                   "launches_per_step": minfo["depth"], "shape": {"rows": m_rows, "out_features": 4 * dim, "in_features": dim}}
        del xw, ww, bw
    if model is not None and vm.CONV_HIP and model_name.startswith("dpt_"):
        import torch.nn as nn
        hw = (net_h or net_size) // 4, net_size // 4
        cv = nn.Conv2d(256, 256, 3, padding=1).to(dev, torch.float16)
        xc = torch.randn(batch, 256, hw[0], hw[1], device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        if vm.conv3x3_hip_ok(cv, xc):
            for _ in range(3):
                nat.conv3x3(cv, xc, relu=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                nat.conv3x3(cv, xc, relu=True)
            e1.record()
            e1.synchronize()
            cv_ms = e0.elapsed_time(e1) / 10
            cv_flops = 2.0 * batch * hw[0] * hw[1] * 256 * 9 * 256
            conv_roof = {"bound": "mfma", "kernel": "k_linear256 (implicit 3x3 convolution + bias + ReLU)",
                         "achieved": cv_flops / (cv_ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": cv_flops / (cv_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None,
                         "algorithmic_flops_per_launch": cv_flops, "avg_kernel_ms": cv_ms, "operands": "random (randn)",
                         "source": "microbenchmark: separate launches at the in-step shape on randn operands, HIP events on the launch stream",
                         "shape": {"batch": batch, "height": hw[0], "width": hw[1], "in_channels": 256, "out_channels": 256}}
        del cv, xc
    torch.cuda.synchronize()

    # ---- route_check (untimed): the forward the step runs -- every block GEMM and the decoder's 3x3 convolutions in-tree, which
    # needs the batch -- against the SAME network on the same images with every GEMM / 3x3 convolution sent to the ROCm
    # libraries (vm.library_routing), plus how often the fused entry points were reached in one forward of the step
    route = None
    if model is not None and not args.no_route_check:
        names = ("ds_linear", "ds_linear_residual", "ds_linear_vt", "ds_conv3x3_nhwc", "ds_attention_fwd", "ds_residual_layernorm",
                 "ds_dpt_head_tail", "ds_preprocess_bicubic")
        before = dict(nat.CALLS)
        with torch.no_grad():
            p_hip = run_forward(model, model_name, img, net_size, net_h).float()
        calls = {n: nat.CALLS[n] - before.get(n, 0) for n in names}
        nlib = min(batch, 4)
        with torch.no_grad(), vm.library_routing():
            p_lib = run_forward(model, model_name, img[:nlib], net_size, net_h).float()
        span = (p_lib.flatten(1).max(1).values - p_lib.flatten(1).min(1).values).clamp_min(1e-12)
        err = (p_hip[:nlib] - p_lib).abs().flatten(1).max(1).values / span
        route = {"max_abs_diff_over_prediction_range": float(err.max().item()), "units_compared": nlib,
                 "what": "prediction of the timed forward (in-tree GEMM / convolution routing at the step's batch) vs the same network on "
                         "the same images with every token GEMM and 3x3 convolution through hipBLASLt / MIOpen; fp16 both sides",
                 "c_abi_calls_per_forward": calls}
        del p_hip, p_lib

    funnel = None
    want_funnel = args.funnel or (args.config == "c3" and world == 1 and args.model is None and args.batch is None)
    if want_funnel and not args.no_funnel and rank == 0 and model is not None:
        funnel = funnel_leg(model, model_name, img_np, net_size, net_h, normalmap)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        units = batch * world * args.steps
        avg_render_s = float(np.mean(render_ms)) * 1e-3
        achieved = batch * algo_bytes_stereo() / avg_render_s / 1e9
        stereo_roof = {"bound": "hbm", "kernel": "k_polylines", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                       "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                       "traffic_from_profile": traffic_from_profile("k_polylines<", batch) if (H, W) == (1024, 1024) else None,
                       "traffic_general_pass_from_profile": traffic_from_profile("k_polylines_general", batch) if (H, W) == (1024, 1024) else None,
                       "source": "HIP events inside the C ABI around k_polylines + k_polylines_general, in the step, on the network's own depth",
                       "algorithmic_bytes_per_launch": batch * algo_bytes_stereo(),
                       "avg_kernel_ms": float(np.mean(render_ms)), "exact_fallback_ms": float(np.mean(exact_ms)),
                       "exact_fallback_rows": exact_rows, "general_pixels": general_px,
                       # what actually bounds it: float64 VALU issue (bit-exactness forces binary64 in the reference's order)
                       "fp64_valu": {"peak_tflops": 78.6, "note": "see DESIGN.md 3.1: instruction count per 64 pixel-eyes from "
                                                                  "the committed PMC profile"}}
        wl = (f"{minfo['name']} forward (fp16, random-init weights seed {minfo['init_seed']}, net {minfo['net']}, "
              f"{minfo['tokens']} tokens) + " if model is not None else "")
        out = {
            "metric": "depth+stereo pairs/sec @1024x1024" if (H, W) == (1024, 1024) else f"depth+stereo pairs/sec @{W}x{H}",
            "value": units / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16 (network) / f64 (stereo, normal map)" if model is not None else "f64",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.config}: {wl}depth->u16 + create_stereoimages({args.fill}, left-right, "
                                   f"divergence 2.5%)" + (" + create_normalmap (Sobel 3)" if normalmap else "")
                                   + f" on {batch} x {W}x{H} RGB per GPU, inputs resident in HBM"
                                   + ("" if model is not None else "; float32 depth prediction is a synthetic input (--model none)"),
                       "depth_network": model_name, "units_per_step": batch * world, "height": H, "width": W,
                       "network_precision_vs_reference": "fp16 = the reference's GPU default; held to 2e-2 of its float32 output "
                                                         "(float32 path: 1e-4), tests/test_gpu_models.py",
                       "forward_launch": ("hipGraph replay" if (fwd is not None and fwd.graphs) else "eager"),
                       "graph_capture_failures": (len(fwd.failed) if fwd is not None else None),
                       "library_convolutions": "MIOpen, solver search on (torch.backends.cudnn.benchmark)" if miopen_find else "MIOpen, heuristic solver choice",
                       "overlap": "per-pixel kernels of step k on a second stream beside the forward of step k+1" if post is not None else "single stream",
                       "parallelism": f"units sharded over {world} GPU(s), no data-path collective"
                                      + (", ONE RCCL gather of the collated outputs (stereo pair + uint16 depth"
                                         + (" + normal map" if normalmap else "") + f", {unit_bytes} bytes per unit) to rank 0 per step, overlapped"
                                         if gather_ok else "")},
            # the dominant hand-written kernel of the step (largest launches x average duration): the fc1 + GELU GEMM or the
            # fused attention when a network runs, else the stereo kernel; the others follow under their own keys
            "roofline": stereo_roof,
        }
        if attn is not None:
            cands = [r for r in (attn, lin) if r is not None]
            out["roofline"] = max(cands, key=lambda r: r["avg_kernel_ms"] * r["launches_per_step"])
            out["roofline_attention"] = attn
            if lin is not None:
                out["roofline_linear"] = lin
            if conv_roof is not None:
                out["roofline_conv3x3"] = conv_roof
            out["roofline_stereo"] = stereo_roof
            enc = vm.count_encoder_flops(minfo["depth"], minfo["tokens"], minfo["dim"]) * batch
            out["encoder_tflops_per_step"] = enc / 1e12
        if normalmap and nm_ms:
            a = batch * algo_bytes_normalmap() / (float(np.mean(nm_ms)) * 1e-3) / 1e9
            out["roofline_normalmap"] = {"bound": "hbm", "kernel": "k_normalmap_fused", "achieved": a, "peak": HBM_PEAK_GBPS,
                                         "unit": "GB/s", "frac": a / HBM_PEAK_GBPS, "traffic": None,
                                         "traffic_from_profile": traffic_from_profile("k_normalmap_fused4", batch) if (H, W) == (1024, 1024) else None,
                                         "algorithmic_bytes_per_launch": batch * algo_bytes_normalmap(),
                                         "avg_kernel_ms": float(np.mean(nm_ms))}
        # `traffic`: HBM bytes per launch from the PMC counters.  bench.py cannot read counters itself; the figure comes from the
        # committed rocprofv3 --pmc passes of THIS command (profiles/, FETCH_SIZE doubled as the guide prescribes for gfx950),
        # and stays null when that profile was taken at another batch size or lacks the kernel.  HBM-bound kernels also carry
        # their fraction of what a copy kernel reaches on this chip (6.29 TB/s), next to the fraction of the 8 TB/s spec peak.
        for key, r in out.items():
            if key.startswith("roofline") and isinstance(r, dict):
                tp = r.get("traffic_from_profile")
                if tp:
                    r["traffic"] = tp["hbm_bytes_per_launch"]
                if r.get("bound") == "hbm":
                    r["frac_of_measured_copy_peak"] = r["achieved"] / HBM_COPY_GBPS
        if route is not None:
            out["route_check"] = route
        if funnel is not None:
            out["funnel"] = funnel
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model_name, net_size, net_h, args.cpu_sample, seed=1000, min_seconds=args.cpu_seconds,
                                               normalmap=normalmap, python_unit=args.cpu_python_unit,
                                               init_seed=minfo["init_seed"] if minfo else 0)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def run_c4(args):
    """BASELINE config 4: Boost (LeReS res101 + the pix2pix merge network, random init) on ONE 3840x2160 image, the patches
    sharded over the ranks (src/boost.estimateboost: rank 0 runs the whole-image passes and the selection, every rank a
    contiguous run of patch chunks, ONE gather, the blend on rank 0), then depth -> uint16 -> stereo pair + normal map on
    rank 0.  One step = one image; strong scaling (the image is the fixed total work)."""
    global H, W
    # (float32 ResNeXt / U-Net convolutions at a dozen shapes: on a fresh box MIOpen's search makes the priming pass take
    # minutes; MIOPEN_FIND_MODE=FAST shortens it but picks slower kernels -- 4.5 s instead of 2.4 s per image at r_max 3000)
    import torch
    import torch.distributed as dist
    H, W = args.height or 2160, args.width or 3840
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
    import src.normalmap_generation as nmg
    import src.stereoimage_generation as sg
    from src import boost
    from lib.multi_depth_model_woauxi import RelDepthModel
    from pix2pix.models.pix2pix4depth_model import Pix2Pix4DepthModel
    torch.manual_seed(0)
    net = RelDepthModel('resnext101').eval().to(dev)                 # float32: Boost never runs LeReS in half (reference :271)
    p2p = Pix2Pix4DepthModel().eval().to(dev)
    rng = np.random.default_rng(1000)
    yy, xx = np.mgrid[0:H, 0:W]
    img_np = (127 + 60 * np.sin(xx / 37.0)[..., None] * np.cos(yy / 23.0)[..., None] + 40 * (((xx // 240 + yy // 180) % 2)[..., None] - 0.5)
              + rng.normal(0, 25, (H, W, 3))).clip(0, 255).astype(np.uint8)
    img = torch.from_numpy(img_np).to(dev)
    grp = dist.group.WORLD if world > 1 else None
    stats = {}

    def step():
        pred = boost.estimateboost(img, net, 0, p2p, whole_size_threshold=args.boost_rmax, stats=stats, group=grp)
        if pred is None:
            return None
        d16 = nat.depth_to_u16(pred.unsqueeze(0), True)                # LeReS is a near-is-dark model (inverted)
        sbs = sg.create_stereoimages_batch(img.unsqueeze(0), d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, args.fill)[0]
        nm = nmg.create_normalmap_batch(d16) if not args.no_normalmap else None
        return sbs, nm

    # algorithmic flops of one image: every convolution of the two networks, counted by hooks during the (untimed) priming step
    flops = [0.0]

    def count(mod, inp, outp):
        w = mod.weight
        if isinstance(mod, torch.nn.ConvTranspose2d):
            flops[0] += 2.0 * inp[0].numel() * w.shape[1] * w.shape[2] * w.shape[3]
        else:
            flops[0] += 2.0 * outp.numel() * w.shape[1] * w.shape[2] * w.shape[3]
    hooks = [m.register_forward_hook(count) for net_ in (net, p2p) for m in net_.modules()
             if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))]
    step()
    for hk in hooks:
        hk.remove()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    fl = torch.tensor([flops[0]], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(fl, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    if rank == 0:
        ach = float(fl.item()) / (elapsed / args.steps) / 1e12
        print(json.dumps({
            # the step is float32 convolutions of two library-backed networks (Boost never runs them in half: reference :271): the
            # roofline is the float32 MFMA peak (157.3 TF/s per GPU, MI355X_MICROARCH.md) against the convolutions' algorithmic flops
            "roofline": {"bound": "mfma", "kernel": "MIOpen float32 convolutions of LeReS res101 + the pix2pix U-Net (library), whole step",
                         "achieved": ach, "peak": 157.3 * world, "unit": "TFLOP/s", "frac": ach / (157.3 * world), "traffic": None,
                         "algorithmic_flops_per_image": float(fl.item()), "source": "forward hooks on every convolution (priming step) / wall time of the timed steps"},
            "metric": f"Boost depth+stereo images/sec @{W}x{H}", "value": args.steps / elapsed, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32 (networks) / f64 (stereo, normal map)",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config c4: Boost multi-resolution merge (LeReS res101 + pix2pix, random init, float32) on one "
                                   f"{W}x{H} image, r_max {args.boost_rmax}: {stats.get('patches')} patches at 896^2 + 448^2, whole image at "
                                   f"{stats.get('whole_image_optimal_size')}, then depth->u16 + create_stereoimages({args.fill}) + normal map",
                       "parallelism": f"patches sharded over {world} GPU(s) in whole chunks of 8, one RCCL gather of the merged patches, "
                                      "blend on rank 0", "boost": stats}}))
    if world > 1:
        dist.destroy_process_group()


def funnel_leg(model, model_name, img_np, net_size, net_h, normalmap):
    """The same batch through the DROP-IN boundary: core_generation_funnel(PIL images in -> PIL results out), host copies,
    PIL conversion and all (what a reference caller actually gets; never `value`)."""
    import torch
    from PIL import Image
    import src.core as core
    mt = {"dpt_beit_large_512": 1, "dpt_hybrid_384": 4, "dav2_vitl": 14}[model_name]

    class _Pred:                                             # the bench's random-init network behind the predictor hook
        def __call__(self, pil, nw, nh, device):
            return self.batch([pil], nw, nh, device)[0]

        def batch(self, pils, nw, nh, device):
            t = torch.from_numpy(np.stack([np.asarray(p.convert("RGB")) for p in pils])).to(device)
            return self.batch_tensor(t)

        def batch_tensor(self, t, nw=None, nh=None):   # what the funnel calls with the pixels it has already uploaded (like the
            if graphed is not None:                        # product's own _NetPredictor.predict_batch)
                return graphed(t)
            return run_forward(model, model_name, t, net_size, net_h)

    graphed = None
    if os.environ.get("DS_FUNNEL_GRAPH", "0") != "0":        # the group's forward as ONE hipGraph replay (src/hip_graph.py)
        from src.hip_graph import GraphedForward
        graphed = GraphedForward(lambda x: run_forward(model, model_name, x, net_size, net_h))

    core.model_holder.register_predictor(mt, _Pred())
    pils = [Image.fromarray(a) for a in img_np]
    opts = {"model_type": mt, "gen_stereo": True, "stereo_modes": ["left-right"], "gen_normalmap": normalmap,
            "net_width": net_size, "net_height": net_size if net_h is None else net_h}
    for _ in range(2):
        n_out = sum(1 for _ in core.core_generation_funnel(None, list(pils), None, None, opts))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_out = sum(1 for _ in core.core_generation_funnel(None, list(pils), None, None, opts))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = dict(core.FUNNEL_STATS)
    return {"value": len(pils) / dt, "unit": "pairs/s", "results": n_out, "seconds": dt,
            "host_seconds": {"enqueue (decode + stage + launch)": st.get("launch"), "enqueue: decode + upload": st.get("launch_decode"),
                             "enqueue: network forward": st.get("launch_forward"), "enqueue: post-processing + downloads": st.get("launch_post"),
                             "blocked on device results": st.get("wait"),
                             "groups": st.get("groups"), "rest (PIL conversion, generator overhead)":
                             None if not st else st.get("total", dt) - (st.get("launch") or 0.0) - (st.get("wait") or 0.0)},
            "what": "core_generation_funnel: PIL in -> uint16 depth, left-right pair" + (", normal map" if normalmap else "")
                    + " as PIL out (host<->device copies and PIL conversion included)"}


if __name__ == "__main__":
    main()
