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
net 1024, 4097 tokens), c4 (Boost on one 4K image), c5 (dav2_vitl on 1920x1080 frames, 2443 tokens).  c3 is the metric's line;
the default invocation ALSO runs short legs of c5, c2 and c4 (sub-processes, after the timed region) and carries their digests
under `other_configs`, so that the driver's one line holds four driver-timed configurations.

Contract: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the driver launches one rank per GPU with
torch.distributed.run; run WITHOUT a launcher (`python bench.py --gpus 8`), bench.py launches the N ranks itself (the same
torch.distributed.run command, rendezvous on 127.0.0.1) and rank 0's line is the output.  Units are sharded across ranks (weak
scaling: every rank renders its own batch, no data-path collective) and the collated outputs (stereo pair + uint16 depth + normal
map, packed into one byte buffer) are gathered to rank 0 with ONE RCCL gather per step, overlapped with the next step's kernels
(--no-gather to leave it out); after the timed region rank 0 re-renders the LAST rank's units itself and compares them with the
gathered bytes (`gather_check`).  Rank 0 prints ONE JSON line; on one GPU the metric's config also carries
`funnel`: the same batch through the drop-in boundary (core_generation_funnel, PIL in -> PIL out), reported beside `value`,
never as it.

`roofline` is the dominant in-tree kernel of the step by its time INSIDE the timed region: the C ABI brackets its launches with
HIP events on the launch stream while the steps are timed (ds_kernel_timer_enable, include/depthstereo.h), `achieved` = algorithmic
flops per launch / that average duration.  The microbenchmark of the same launch shape on randn operands is the side note
(`microbenchmark`), not the figure.
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

from benchlib import workload as wl  # noqa: E402
from benchlib.launch import compare_gathered, dist_setup, launch_command, max_over_ranks, self_launch, selftest_launch  # noqa: E402,F401
from benchlib.legs import funnel_leg, other_configs_leg, route_check_leg  # noqa: E402
from benchlib.profiles import traffic_from_profile, valu_from_profile  # noqa: E402
from benchlib.rooflines import (HBM_COPY_GBPS, HBM_PEAK_GBPS, IN_STEP, MICRO, encoder_rooflines, mfma_roofline,  # noqa: E402,F401
                                microbench_conv)
from benchlib.workload import (algo_bytes_normalmap, algo_bytes_stereo, build_model, cpu_baseline, default_net_size, net_grid,  # noqa: E402,F401
                               run_forward, synth_batch)

H = W = 1024                      # frame size of the run (mirrors benchlib.workload's state: run_pipeline / run_c4 set both)
DEPTH_KIND = "steps"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c3match", "c4", "c5"], help="BASELINE.json config preset (see the header)")
    ap.add_argument("--batch", type=int, default=None, help="units per GPU per step (--scaling strong: units per step of the whole job)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank renders its own --batch units; strong: the SAME --batch units (BASELINE.md 3: 'identical "
                         "batch run on 1/2/4/8 GPUs') are split over the ranks in contiguous shards -- the gathered bytes do not depend on "
                         "the number of ranks (outputs_sha256)")
    ap.add_argument("--model", default=None, choices=["dav2_vitl", "dpt_beit_large_512", "dpt_hybrid_384", "none"])
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--net-size", type=int, default=None, help="network input size (default: the model's; 0 = NET_SIZE_MATCH)")
    ap.add_argument("--fill", default="polylines_sharp")
    ap.add_argument("--no-normalmap", action="store_true", help="leave the normal map out of the step (round-1 workload)")
    ap.add_argument("--overlap", action="store_true", help="run the per-pixel kernels of step k on a second stream beside the forward of "
                                                         "step k+1 (c3: inside the run-to-run noise; c5: 190 -> 223 pairs per second; off by default)")
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
    ap.add_argument("--timers-in-region", action="store_true", help="record the kernel timers' events inside the timed region (rounds 3-5; costs ~2 ms per step)")
    ap.add_argument("--no-kernel-timers", action="store_true", help="do not bracket the kernels with events inside the timed region (A/B of the "
                                                                    "timers' own cost; the rooflines then fall back to the microbenchmarks)")
    ap.add_argument("--no-other-configs", action="store_true", help="default invocation only: skip the short c5 / c2 / c4 legs behind the metric's line")
    ap.add_argument("--other-configs-timeout", type=float, default=240.0, help="seconds per other-config leg (c4: twice that)")
    ap.add_argument("--no-micro", action="store_true", help="skip the microbenchmarks beside the in-step rooflines (profile runs)")
    ap.add_argument("--selftest-launch", action="store_true", help="CPU / gloo: exercise the launcher + gather plumbing without a GPU (tests)")
    ap.add_argument("--tune-gemms", default=None, metavar="CSV",
                    help="run with TunableOp tuning ON and accumulate the winners in CSV (maintenance: regenerates "
                         "src/tunableop_gfx950.csv); the default run only READS the shipped file")
    ap.add_argument("--cpu-sample", type=int, default=4, help="distinct units of the CPU baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="minimum wall time of the CPU baseline leg")
    ap.add_argument("--cpu-python-unit", type=int, default=512, help="side of the one unit timed through the pure-Python port (0 = skip)")
    ap.add_argument("--boost-rmax", type=int, default=1600, help="c4: Boost's whole-image size limit (standalone default 1600, paper 3000)")
    return ap.parse_args(argv)


def seed_miopen():
    """The shipped MIOpen find-db (src/miopen_db.py): without it the c4 legs spend 334 s of wall searching float32 solvers on a fresh box."""
    from src import miopen_db
    return miopen_db.seed()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # run as the driver runs N = 1 -- plain `python bench.py --gpus N` -- there is no launcher around us: be the launcher
        sys.exit(self_launch(args, argv))
    if args.selftest_launch:
        return selftest_launch(args)
    seed_miopen()
    if args.config == "c4":
        return run_c4(args)
    return run_pipeline(args)


def run_pipeline(args):
    global H, W, DEPTH_KIND
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
    wl.configure(H, W, DEPTH_KIND)
    normalmap = not args.no_normalmap
    default_invocation = (args.config == "c3" and args.model is None and args.batch is None and args.height is None and args.width is None
                          and args.net_size is None)

    import torch
    import torch.distributed as dist
    # MIOpen's solver search (torch.backends.cudnn.benchmark): off by default since the end of round 3.  It bought 1.3-3 % while the
    # library convolutions still carried their bias; since they run without it (vit_mi355x.conv_module) the heuristic choice is
    # the same CK kernels: 806.6 (search) vs 809.7 (heuristic) pairs/s on the same box, and the search costs ~40 s of the untimed
    # priming pass on a fresh box.  DS_CUDNN_BENCHMARK=1 switches it on.
    miopen_find = os.environ.get("DS_CUDNN_BENCHMARK", "0") != "0"
    if miopen_find:
        torch.backends.cudnn.benchmark = True
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists)")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank, world, _ = dist_setup("nccl", dev)

    import src._native as nat
    import src.normalmap_generation as nmg
    import src.stereoimage_generation as sg
    from src import multigpu
    from src import vit_mi355x as vm

    strong = args.scaling == "strong"
    global_batch = batch
    if strong:
        # the same units whatever the number of ranks: rank r renders the contiguous shard [r * batch, (r + 1) * batch) of them
        if global_batch % world:
            raise SystemExit(f"--scaling strong: {global_batch} units do not split evenly over {world} ranks")
        batch = global_batch // world
        g_img, g_pred = synth_batch(global_batch, seed=1000)
        shard = lambda r: (g_img[r * batch:(r + 1) * batch], g_pred[r * batch:(r + 1) * batch])      # noqa: E731
    else:
        shard = lambda r: synth_batch(batch, seed=1000 + r)                                          # noqa: E731
    img_np, pred_np = shard(rank)
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

    layout_box = []

    def render(images, pred_given=None, forward=None):
        """One pass of the hot path over `images`: (stereo pairs, normal maps, uint16 depth)."""
        if model is not None:
            pred = forward(images) if forward is not None else run_forward(model, model_name, images, net_size, net_h)
        else:
            pred = pred_given
        return pred

    def step(check=False):
        pred = render(img, pred_in, fwd)
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
                packed, layout = multigpu.pack_collated([sbs, d16] + ([nmap] if normalmap else []))
                if not layout_box:
                    layout_box.append(layout)
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
    # kernel timers: event pairs around the launches of the big in-tree kernels (ds_kernel_timer_enable).  Rounds 3-5 recorded them
    # INSIDE the timed region on the belief that a record is a free marker; round 6 measured it: ~250 pairs per step cost 2.2 ms of a
    # 39 ms step (eager 818.7 pairs/s with them, 868.8 without, 867.3 as a hipGraph replay, which skips them: profiles/
    # round6_timers_ab.txt) -- every record is a barrier packet between two kernels.  So the timed region runs WITHOUT them, and the
    # roofline durations come from an instrumented repeat of the same steps right behind it (same process, tensors, launches and
    # clocks; its own step time is reported as instrumented_ms_per_step).  --timers-in-region restores the old arrangement (A/B).
    timers_on = model is not None and not args.no_kernel_timers
    if timers_on and args.timers_in_region:
        nat.kernel_timer_enable(local_rank, True)
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
    instrumented_ms = None
    if timers_on and not args.timers_in_region:
        n_inst = min(args.steps, 20)                             # the event rings hold 1024 launches per kernel kind
        nat.kernel_timer_enable(local_rank, True)
        ti = time.perf_counter()
        for _ in range(n_inst):
            step()
        torch.cuda.synchronize()
        instrumented_ms = (time.perf_counter() - ti) / n_inst * 1e3
        if world > 1:
            dist.barrier()
    timed, timed_each = {}, {}
    timer_steps = args.steps if args.timers_in_region else min(args.steps, 20)
    if timers_on:
        for kind in ("linear_residual", "linear_residual+ragged"):
            timed_each[kind] = nat.kernel_timer_read_each(local_rank, kind)
        for kind in ("linear_gelu", "linear_residual", "attention", "linear", "linear_vt", "conv3x3", "linear_readout", "linear_shuffle", "normalmap"):
            timed[kind] = nat.kernel_timer_read(local_rank, kind)
            if kind.startswith("linear"):
                timed[kind + "+ragged"] = nat.kernel_timer_read(local_rank, kind + "+ragged")
        nat.kernel_timer_enable(local_rank, False)

    # per-kernel timing of the per-pixel path in separate untimed passes (event synchronisation must not perturb the throughput
    # measurement): k_polylines via HIP events recorded inside the C ABI on the launch stream, ds_normalmap via events on torch's
    # current stream (which IS the launch stream of the ctypes call)
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

    # gather_check (N > 1, untimed): rank 0 renders the LAST rank's units itself (its inputs are a function of the rank) and compares
    # them with the bytes that rank sent through the gather in the last step
    gather_check = None
    if gather_ok and rank == 0:
        torch.cuda.synchronize()
        side.synchronize()
        o_img_np, o_pred_np = shard(world - 1)
        o_img = torch.from_numpy(o_img_np).to(dev)
        with torch.no_grad():
            o_pred = render(o_img, torch.from_numpy(o_pred_np).to(dev), None)
        o_d16 = nat.depth_to_u16(o_pred, False)
        o_sbs = sg.create_stereoimages_batch(o_img, o_d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, args.fill)[0]
        o_parts = [o_sbs, o_d16] + ([nmg.create_normalmap_batch(o_d16)] if normalmap else [])
        want, layout = multigpu.pack_collated(o_parts)
        # round 6: the metric's network is bit-reproducible and batch-position / batch-size invariant (vit_mi355x.INVARIANT: every GEMM
        # and convolution of dpt_beit_large_512 in-tree, one accumulation chain per output element wherever its tile lands) -- identity
        # is asserted with that network as well; the other networks still call MIOpen (split-K atomics) unless DS_DETERMINISTIC=1
        from src import vit_mi355x as _vm
        expected = (model is None or _vm.DETERMINISTIC_LIBRARY
                    or (_vm.INVARIANT and _vm.LINEAR_HIP == "all" and model_name == "dpt_beit_large_512"))
        exact = model is None or _vm.DETERMINISTIC_LIBRARY      # asserted (the run fails); with the in-tree network: reported as `identical`
        gather_check = compare_gathered(gathered[world - 1], want, layout, exact=exact)
        gather_check["identity_expected"] = bool(expected)
        gather_check["rank"] = world - 1
        gather_check["what"] = ("rank 0's own render of the last rank's units vs the bytes gathered from that rank"
                                + ("; identity asserted" if exact else "; identity expected (every kernel of this network is in-tree and batch invariant), reported as `identical`" if expected else "; this network still calls MIOpen, whose split-K solvers are not "
                                   "bit-reproducible between launches (DS_DETERMINISTIC=1 makes them): fractions reported instead of identity"))
        del o_img, o_pred, o_d16, o_sbs, o_parts, want

    # outputs_sha256 (untimed): SHA-256 of the collated bytes (stereo pair | uint16 depth | normal map, unit by unit) of the last step
    # -- strong scaling: of ALL units in rank order, i.e. the same digest on 1 / 2 / 4 / 8 GPUs ("outputs must be byte-identical
    # across GPU counts"); weak scaling: of rank 0's units, whose inputs do not depend on the number of ranks either
    outputs_sha256 = None
    if rank == 0:
        import hashlib
        hsh = hashlib.sha256()
        if gather_ok and strong:
            for r in range(world):
                hsh.update(gathered[r].cpu().numpy().tobytes())
        else:
            o_sbs, o_nmap, o_d16 = step()
            torch.cuda.synchronize()
            if side is not None:
                side.synchronize()
            own, _ = multigpu.pack_collated([o_sbs, o_d16] + ([o_nmap] if normalmap else []))
            hsh.update(own.cpu().numpy().tobytes())
            del o_sbs, o_nmap, o_d16, own
        outputs_sha256 = {"sha256": hsh.hexdigest(), "of": ("all units of the job, rank order" if (gather_ok and strong) else "the units of rank 0"),
                          "units": (global_batch if (gather_ok and strong) else batch)}
    elif world > 1 and not (gather_ok and strong):
        step()                                                  # (the ranks stay in step with rank 0's extra render: its gather is collective)
        torch.cuda.synchronize()

    roofs, conv_roof = {}, None
    if model is not None:
        if args.no_micro:
            roofs = {}
            for kind, (n, ms) in timed.items():
                if n > 0 and "+" not in kind:
                    roofs[kind] = {"kernel": kind, "avg_kernel_ms": ms / n, "launches": n, "source": IN_STEP.format(n=n)}
        else:
            roofs = encoder_rooflines(nat, vm, dev, local_rank, batch, minfo, args.config, timed, timed_each)
            if vm.CONV_HIP and model_name.startswith("dpt_"):
                cv_ms, hw = microbench_conv(nat, vm, dev, batch, net_size, net_h)
                if cv_ms is not None:
                    conv_roof = mfma_roofline("k_linear256<CONV> (implicit 3x3 convolution + bias + ReLU)", 2.0 * batch * hw[0] * hw[1] * 256 * 9 * 256,
                                              cv_ms, None, MICRO, {"batch": batch, "height": hw[0], "width": hw[1], "in_channels": 256, "out_channels": 256},
                                              operands="random (randn)")
                    n, ms = timed.get("conv3x3", (0, 0.0))
                    if n > 0:
                        conv_roof["in_step_all_shapes"] = {"launches": n, "avg_kernel_ms": ms / n, "source": IN_STEP.format(n=n),
                                                           "note": "every ds_conv3x3_nhwc launch of the decoder (several shapes): a time share, not a rate"}
    torch.cuda.synchronize()

    route = None
    if model is not None and not args.no_route_check:
        route = route_check_leg(nat, vm, model, model_name, img, batch, net_size, net_h)

    funnel = None
    want_funnel = args.funnel or (default_invocation and world == 1)
    if want_funnel and not args.no_funnel and rank == 0 and model is not None:
        funnel = funnel_leg(model, model_name, img_np, net_size, net_h, normalmap)

    elapsed = max_over_ranks(elapsed, world, dev)

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
                                                                  "the committed PMC profile",
                                     "main_kernel": valu_from_profile("k_polylines<", batch) if (H, W) == (1024, 1024) else None,
                                     "general_pass": valu_from_profile("k_polylines_general", batch) if (H, W) == (1024, 1024) else None}}
        wl_text = (f"{minfo['name']} forward (fp16, random-init weights seed {minfo['init_seed']}, net {minfo['net']}, "
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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f16 (network) / f64 (stereo, normal map)" if model is not None else "f64",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.config}: {wl_text}depth->u16 + create_stereoimages({args.fill}, left-right, "
                                   f"divergence 2.5%)" + (" + create_normalmap (Sobel 3)" if normalmap else "")
                                   + (f" on the same {global_batch} x {W}x{H} RGB split over the GPUs ({batch} per GPU), inputs resident in HBM" if strong
                                      else f" on {batch} x {W}x{H} RGB per GPU, inputs resident in HBM")
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
            # the dominant hand-written kernel of the step by its time INSIDE the timed region (in-step launches x average duration);
            # the others follow under their own keys; with no network it is the stereo kernel
            "roofline": stereo_roof,
        }
        if roofs and not args.no_micro:
            out["roofline"] = max(roofs.values(), key=lambda r: r["avg_kernel_ms"] * r["launches_per_step"])
            for kind, key in (("attention", "roofline_attention"), ("linear_gelu", "roofline_linear"), ("linear_residual", "roofline_linear_residual")):
                if kind in roofs:
                    out[key] = roofs[kind]
            if conv_roof is not None:
                out["roofline_conv3x3"] = conv_roof
            out["roofline_stereo"] = stereo_roof
            out["encoder_tflops_per_step"] = vm.count_encoder_flops(minfo["depth"], minfo["tokens"], minfo["dim"]) * batch / 1e12
        elif roofs:
            out["in_step_kernel_ms"] = roofs
        if timed:
            out["in_step_kernel_time_ms_per_step"] = {k: v[1] / timer_steps for k, v in timed.items() if v[0] > 0}
            out["instrumented_ms_per_step"] = instrumented_ms
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
        if gather_check is not None:
            out["gather_check"] = gather_check
        out["outputs_sha256"] = outputs_sha256
        if route is not None:
            out["route_check"] = route
        if funnel is not None:
            out["funnel"] = funnel
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model_name, net_size, net_h, args.cpu_sample, seed=1000, min_seconds=args.cpu_seconds,
                                               normalmap=normalmap, python_unit=args.cpu_python_unit,
                                               init_seed=minfo["init_seed"] if minfo else 0)
        if default_invocation and world == 1 and not args.no_other_configs:
            # the other BASELINE configurations, driver-timed in the same command (sub-processes; the parent's GPU work is done)
            torch.cuda.empty_cache()
            out["other_configs"] = other_configs_leg(args.other_configs_timeout)
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
    wl.configure(H, W)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists)")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank, world, _ = dist_setup("nccl", dev)
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
    # (counted at the FUNCTION level: the networks call F.conv2d / nn.Conv2d._conv_forward with folded or bias-free weights and the
    # in-tree grouped convolution directly, so forward hooks on the modules miss most of them -- rounds 4-5 counted 18 TFLOP of the
    # image's ~40 that way and under-reported the rate)
    import torch.nn.functional as F_
    real_conv, real_convt, real_gconv = F_.conv2d, F_.conv_transpose2d, nat.gconv3x3

    def conv2d_counted(x, w, *a, **k):
        y = real_conv(x, w, *a, **k)
        flops[0] += 2.0 * y.numel() * w.shape[1] * w.shape[2] * w.shape[3]
        return y

    def convt_counted(x, w, *a, **k):
        flops[0] += 2.0 * x.numel() * w.shape[1] * w.shape[2] * w.shape[3]
        return real_convt(x, w, *a, **k)

    def gconv_counted(x, img, b, relu, cpg):
        flops[0] += 2.0 * x.numel() * cpg * 9
        return real_gconv(x, img, b, relu, cpg)
    F_.conv2d, F_.conv_transpose2d, nat.gconv3x3 = conv2d_counted, convt_counted, gconv_counted
    try:
        step()
    finally:
        F_.conv2d, F_.conv_transpose2d, nat.gconv3x3 = real_conv, real_convt, real_gconv
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
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    fl = torch.tensor([flops[0]], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(fl, op=dist.ReduceOp.SUM)
    if rank == 0:
        ach = float(fl.item()) / (elapsed / args.steps) / 1e12
        print(json.dumps({
            # the step is float32 convolutions of two library-backed networks (Boost never runs them in half: reference :271): the
            # roofline is the float32 MFMA peak (157.3 TF/s per GPU, MI355X_MICROARCH.md) against the convolutions' algorithmic flops
            "roofline": {"bound": "mfma", "kernel": "float32 convolutions of LeReS res101 + the pix2pix U-Net, whole step (dense ones: MIOpen; the grouped 3x3 of the "
                                                    "ResNeXt bottlenecks: ds_gconv3x3_nhwc_f32)",
                         "achieved": ach, "peak": 157.3 * world, "unit": "TFLOP/s", "frac": ach / (157.3 * world), "traffic": None,
                         "avg_kernel_ms": elapsed / args.steps * 1e3,
                         "algorithmic_flops_per_image": float(fl.item()), "source": "every F.conv2d / F.conv_transpose2d / ds_gconv3x3_nhwc_f32 call of the (untimed) priming step counted from its shapes / wall time of the timed steps"},
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


if __name__ == "__main__":
    main()
