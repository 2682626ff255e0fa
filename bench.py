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

H = W = 1024
HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0            # what a float4 copy kernel reaches on this chip (same guide: 79 % of the spec peak)
MFMA_PEAK_TFLOPS = 2500.0         # dense f16/bf16 MFMA peak (same guide)
PMC_SUMMARY = "profiles/round5_pmc_summary.json"      # tools/pmc_summary.py over rocprofv3 --pmc passes of THIS command
KERNEL_STATS = "profiles/round5_kernel_stats.csv"     # rocprofv3 --kernel-trace --stats of THIS command

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
    if model_name == "dav2_vitl":
        return model.infer_batch(img, net_size)
    return model.infer_batch(img, net_size=net_size, resize_mode="minimal", net_h=net_h)


# ---- launching the ranks ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(n, argv, port=None):
    """The command the driver itself uses for N > 1 (one rank per GPU of ONE node, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or _free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks here.  The ranks' output
    passes through; rank 0 prints the one JSON line.  Returns the launcher's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    return subprocess.call(launch_command(args.gpus, argv), env=env)


def dist_setup(backend, device=None):
    """(rank, world, local_rank) from the launcher's environment; the process group when world > 1."""
    import torch.distributed as dist
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if device is not None:
            dist.init_process_group(backend, device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def max_over_ranks(elapsed, world, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def compare_gathered(got_u8, want_u8, layout, exact):
    """gather_check: rank 0's own render of another rank's units against the bytes that rank sent through the gather.  `layout`
    = multigpu.pack_collated's; per part the fraction of equal bytes and, for the uint16 depth, the largest code difference.
    exact (the per-pixel path alone, --model none): everything is integer / IEEE float64 work, the bytes must be identical."""
    import torch
    from src import multigpu
    a, b = multigpu.unpack_collated(got_u8, layout), multigpu.unpack_collated(want_u8, layout)
    out = {"units": int(got_u8.shape[0]), "identical": bool(torch.equal(got_u8, want_u8)), "parts": []}
    for x, y in zip(a, b):
        part = {"dtype": str(x.dtype).replace("torch.", ""), "shape_per_unit": list(x.shape[1:]),
                "equal_fraction": float((x == y).float().mean().item())}
        if x.dtype == torch.uint16:
            part["max_code_difference"] = int((x.to(torch.int32) - y.to(torch.int32)).abs().max().item())
        out["parts"].append(part)
    if exact:
        assert out["identical"], f"gather_check: the gathered bytes differ from rank 0's own render of the same units: {out}"
    return out


def selftest_launch(args):
    """--selftest-launch (CPU, gloo; tests/test_multigpu_gloo.py): the launcher path of `--gpus N` end to end WITHOUT a GPU -- rank
    environment, process group, ONE gather of packed per-unit byte buffers to rank 0, gather_check against rank 0's own render of
    the last rank's units, barrier + max-over-ranks timing, one JSON line from rank 0.  The "render" is a seeded byte pattern: what
    is under test is the plumbing bench.py shares with the real path, not a kernel."""
    import torch
    import torch.distributed as dist
    from src import multigpu
    rank, world, _ = dist_setup("gloo")
    strong = args.scaling == "strong"
    global_batch = args.batch or (12 if strong else 2)
    if strong and global_batch % world:
        raise SystemExit(f"--scaling strong: {global_batch} units do not split evenly over {world} ranks")
    batch = global_batch // world if strong else global_batch

    def pattern(n, seed):
        rng = np.random.default_rng(seed)
        return (torch.from_numpy(rng.integers(0, 256, (n, 4, 16, 3), dtype=np.uint8)), torch.from_numpy(rng.integers(0, 65536, (n, 4, 8), dtype=np.uint16)))

    def render(r):                                               # strong: the shard of ONE seeded job; weak: the rank's own units
        if strong:
            sbs, d16 = pattern(global_batch, 1000)
            return multigpu.pack_collated([sbs[r * batch:(r + 1) * batch], d16[r * batch:(r + 1) * batch]])
        return multigpu.pack_collated(list(pattern(batch, 1000 + r)))
    packed, layout = render(rank)
    gathered = [torch.empty_like(packed) for _ in range(world)] if rank == 0 else None
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1:
            dist.gather(packed, gathered, dst=0)
    if world > 1:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, world)
    if rank == 0:
        import hashlib
        check = None
        if world > 1:
            check = compare_gathered(gathered[world - 1], render(world - 1)[0], layout, exact=True)
            check["rank"] = world - 1
        parts = gathered if (world > 1 and strong) else [packed]
        hsh = hashlib.sha256()
        for t in parts:
            hsh.update(t.numpy().tobytes())
        print(json.dumps({"metric": "selftest: launcher + gather plumbing (no kernel)", "value": batch * world * args.steps / max(elapsed, 1e-9),
                          "unit": "units/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "selftest": True,
                          "scaling": args.scaling, "gather_check": check,
                          "outputs_sha256": {"sha256": hsh.hexdigest(), "units": sum(int(t.shape[0]) for t in parts)}}))
    if world > 1:
        dist.destroy_process_group()


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
        # kind "port-of-fallback": oracle/oracle_py.py restates the reference's numba-LESS fallback (pure-Python loops); the
        # reference's own fallback file timed on the build box took 12.4 s for the same unit (profiles/round2_reference_fallback.json)
        out["python_fallback"] = {"value": 1.0 / dt1, "unit": "pairs/s", "cores": 1, "kind": "port-of-fallback",
                                  "sample": f"1 unit of {s}x{s} (BASELINE config 1's size), polylines_sharp left-right, pure-Python "
                                            f"restatement of the reference's numba-less fallback, {dt1:.2f} s; stereo stage only",
                                  "identical_to_c_port": same}
    return out


# ---- figures out of the committed profiles (labelled as such) --------------------------------------------------------------
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


def stats_from_profile(kernel):
    """Average duration of the kernel whose name contains `kernel` in the committed rocprofv3 --kernel-trace --stats summary of the
    default bench command: what the live in-step figure must agree with."""
    try:
        import csv
        with open(os.path.join(ROOT, KERNEL_STATS)) as f:
            for row in csv.DictReader(f):
                if kernel in row["Name"]:
                    return {"avg_kernel_ms": float(row["AverageNs"]) * 1e-6, "calls": int(row["Calls"]), "kernel": row["Name"], "source": KERNEL_STATS}
    except Exception:
        pass
    return None


# ---- microbenchmarks: one launch shape on randn operands (the side note of every roofline object) ----------------------------
def _event_ms(fn, reps, warm=3):
    import torch
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def microbench_attention(nat, vm, dev, batch, minfo):
    import torch
    n_tok = minfo["tokens"]
    npad = vm.pad_len(n_tok, batch)
    qk = torch.randn(batch, npad, 2, minfo["heads"], 64, device=dev, dtype=torch.float16)
    vt = torch.randn(batch, minfo["heads"] * 64, npad, device=dev, dtype=torch.float16)
    bias = None
    if minfo["bias"]:
        bias = nat.attention_bias_pack(torch.randn(minfo["heads"], n_tok, n_tok, device=dev), npad, torch.float16)
    return _event_ms(lambda: nat.attention_fwd(qk, vt, n_tok, 0.125, bias), 20)


def microbench_linear(nat, vm, dev, batch, minfo, kind):
    """fc1 + GELU ("linear_gelu") or the fc2 / projection pair with LayerScale + residual ("linear_residual": the average of the two
    shapes one block launches) at the step's row count, randn operands."""
    import torch
    m_rows, dim = batch * vm.pad_len(minfo["tokens"], batch), minfo["dim"]
    x1 = torch.randn(m_rows, dim, device=dev, dtype=torch.float16)
    if kind == "linear_gelu":
        w = torch.randn(4 * dim, dim, device=dev, dtype=torch.float16) * dim ** -0.5
        b = torch.randn(4 * dim, device=dev, dtype=torch.float16)
        if not nat.linear_supported(x1, w):
            return None
        return _event_ms(lambda: nat.linear(x1, w, b, True), 20)
    x4 = torch.randn(m_rows, 4 * dim, device=dev, dtype=torch.float16)
    wp = torch.randn(dim, dim, device=dev, dtype=torch.float16) * dim ** -0.5
    w2 = torch.randn(dim, 4 * dim, device=dev, dtype=torch.float16) * (4 * dim) ** -0.5
    b = torch.randn(dim, device=dev, dtype=torch.float16)
    g = torch.randn(dim, device=dev, dtype=torch.float16)
    if not (nat.linear_supported(x1, wp) and nat.linear_supported(x4, w2)):
        return None

    def both():
        nat.linear_residual(x1, wp, b, g, x1)
        nat.linear_residual(x4, w2, b, g, x1)
    return _event_ms(both, 10) / 2.0


def microbench_conv(nat, vm, dev, batch, net_size, net_h):
    import torch
    import torch.nn as nn
    hw = (net_h or net_size) // 4, net_size // 4
    cv = nn.Conv2d(256, 256, 3, padding=1).to(dev, torch.float16)
    xc = torch.randn(batch, 256, hw[0], hw[1], device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    if not vm.conv3x3_hip_ok(cv, xc):
        return None, hw
    return _event_ms(lambda: nat.conv3x3(cv, xc, relu=True), 10), hw


def mfma_roofline(kernel, flops, ms, launches_per_step, source, shape, **extra):
    ach = flops / (ms * 1e-3) / 1e12
    out = {"bound": "mfma", "kernel": kernel, "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
           "traffic": None, "algorithmic_flops_per_launch": flops, "avg_kernel_ms": ms, "launches_per_step": launches_per_step,
           "source": source, "shape": shape}
    out.update(extra)
    return out


IN_STEP = ("in-step: HIP events recorded by the C ABI around every launch of this kernel on the launch stream (ds_kernel_timer_enable, "
           "include/depthstereo.h) in an instrumented repeat of the timed steps right behind the timed region -- same process, tensors and "
           "launches; the event records cost ~2 ms per step, so the timed region runs without them (--timers-in-region: inside it); "
           "average over {n} launches")
MICRO = "microbenchmark: separate launches at the in-step shape on randn operands, HIP events on the launch stream"


def encoder_rooflines(nat, vm, dev, local_rank, batch, minfo, config, timed, timed_each=None):
    """Roofline objects of the encoder's three big kernels.  `timed`: {kind: (launches, total ms)} read from the C ABI's in-step
    timers after the timed region -- the figure of each object when present; the microbenchmark of the same launch shape on randn
    operands is the side note.  Algorithmic flops use the VALID tokens (batch x n), not the padded rows the kernels walk."""
    n_tok, dim, heads, depth = minfo["tokens"], minfo["dim"], minfo["heads"], minfo["depth"]
    rows = batch * n_tok
    c3 = config == "c3"
    specs = {
        "linear_gelu": ("k_linear256<EPI 1> (fc1 + erf-GELU)", "k_linear256<0, 1, 0, 0, 0", 2.0 * rows * 4 * dim * dim, depth,
                        {"rows_valid": rows, "rows_padded": batch * vm.pad_len(n_tok, batch), "out_features": 4 * dim, "in_features": dim}),
        "linear_residual": ("k_linear256<EPI 3, RES 1> (projection and fc2 + LayerScale + residual: the average of the two launches of a block)",
                            "k_linear256<0, 3, 0, 1", 2.0 * rows * dim * (dim + 4 * dim) / 2.0, 2 * depth,
                            {"rows_valid": rows, "rows_padded": batch * vm.pad_len(n_tok, batch), "out_features": dim, "in_features": [dim, 4 * dim]}),
        "attention": ("k_attention_fwd2 (fused attention" + (", relative-position bias through the MFMA pipe)" if minfo["bias"] else ")"),
                      "k_attention_fwd2", 4.0 * n_tok * n_tok * dim * batch, depth,
                      {"batch": batch, "tokens": n_tok, "heads": heads, "bias": minfo["bias"]}),
    }
    out = {}
    for kind, (label, prof_name, flops, per_step, shape) in specs.items():
        if kind == "attention":
            micro_ms = microbench_attention(nat, vm, dev, batch, minfo)
        else:
            micro_ms = microbench_linear(nat, vm, dev, batch, minfo, kind) if vm.LINEAR_HIP == "all" else None
        micro = None if micro_ms is None else {"avg_kernel_ms": micro_ms, "achieved": flops / (micro_ms * 1e-3) / 1e12,
                                               "frac": flops / (micro_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "operands": "random (randn)", "source": MICRO}
        n, ms = timed.get(kind, (0, 0.0))
        if n > 0:
            r = mfma_roofline(label, flops, ms / n, per_step, IN_STEP.format(n=n), shape, microbenchmark=micro)
            rag = timed.get(kind + "+ragged", (0, 0.0))
            if rag[0] > 0:                                   # the ragged round of the same GEMMs (k_linear_ragged), launched behind them
                r["ragged_round"] = {"launches": rag[0], "avg_kernel_ms": rag[1] / rag[0],
                                     "note": "k_linear_ragged renders the last, nearly empty round of tiles; its time is NOT in avg_kernel_ms, "
                                             "its flops are (the whole GEMM's algorithmic flops over the main kernel's time: an upper bound "
                                             "of a few percent)"}
                r["avg_gemm_ms_with_ragged_round"] = ms / n + rag[1] / n
                r["achieved_with_ragged_round"] = flops / ((ms / n + rag[1] / n) * 1e-3) / 1e12
        elif micro is not None:
            r = mfma_roofline(label, flops, micro_ms, per_step, MICRO, shape, operands="random (randn)")
        else:
            continue
        each = (timed_each or {}).get(kind)
        if kind == "linear_residual" and each and len(each) % 2 == 0:
            # the two launches of a block alternate: projection (K = dim) first, fc2 (K = 4 dim) second -- one roofline entry each
            rag_each = (timed_each or {}).get(kind + "+ragged") or []
            by_shape = {}
            for idx, (nm_, kk) in enumerate((("projection", dim), ("fc2", 4 * dim))):
                d = each[idx::2]
                fl = 2.0 * rows * dim * kk
                avg = sum(d) / len(d)
                e = {"kernel": "k_linear256<EPI 3, RES 1> (%s + LayerScale + residual)" % nm_, "in_features": kk, "launches": len(d),
                     "avg_kernel_ms": avg, "algorithmic_flops_per_launch": fl, "achieved": fl / (avg * 1e-3) / 1e12,
                     "frac": fl / (avg * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"}
                if len(rag_each) == len(each):
                    e["ragged_round_avg_ms"] = sum(rag_each[idx::2]) / len(d)
                by_shape[nm_] = e
            r["by_shape"] = by_shape
        if c3:
            r["traffic_from_profile"] = traffic_from_profile(prof_name, batch)
            r["profile_avg"] = stats_from_profile(prof_name)
            r["clock_from_profile"] = clock_from_profile(prof_name, flops, batch)
        out[kind] = r
    return out


def route_check_leg(nat, vm, model, model_name, img, batch, net_size, net_h):
    """route_check (untimed): the forward the step runs -- every block GEMM, the reassemble stage and the decoder's 3x3 convolutions
    in-tree, which needs the batch -- against the SAME network on the same images with every GEMM / convolution sent to the ROCm
    libraries (vm.library_routing), plus how often the fused entry points were reached in one forward of the step."""
    import torch
    names = ("ds_linear", "ds_linear_ln", "ds_linear_residual", "ds_linear_vt", "ds_linear_vt_ln", "ds_row_stats", "ds_linear_readout",
             "ds_linear_shuffle", "ds_conv3x3_nhwc", "ds_attention_fwd", "ds_residual_layernorm", "ds_dpt_head_tail", "ds_preprocess_bicubic")
    before = dict(nat.CALLS)
    with torch.no_grad():
        p_hip = run_forward(model, model_name, img, net_size, net_h).float()
    calls = {n: nat.CALLS[n] - before.get(n, 0) for n in names}
    nlib = min(batch, 4)
    with torch.no_grad(), vm.library_routing():
        p_lib = run_forward(model, model_name, img[:nlib], net_size, net_h).float()
    span = (p_lib.flatten(1).max(1).values - p_lib.flatten(1).min(1).values).clamp_min(1e-12)
    err = (p_hip[:nlib] - p_lib).abs().flatten(1).max(1).values / span
    return {"max_abs_diff_over_prediction_range": float(err.max().item()), "units_compared": nlib,
            "what": "prediction of the timed forward (in-tree GEMM / convolution routing at the step's batch) vs the same network on "
                    "the same images with every token GEMM and convolution through hipBLASLt / MIOpen; fp16 both sides",
            "c_abi_calls_per_forward": calls}


def other_configs_leg(timeout_s):
    """Short legs of the other BASELINE configurations, each in a sub-process of its own after the timed region (a leg that fails or
    hangs costs its timeout, never the line): c5 (8 steps = 64 frames), c2 (20 hipGraph replays), c3match (3 steps), c4 (1 image, at
    r_max 1600 and 3000).  A digest of each leg's own JSON
    line -- value, ms per step, workload, roofline -- goes under `other_configs`."""
    # c5 with --overlap: its per-pixel passes (the polylines fallbacks of a network's noisy 1080p prediction: one or two workgroups
    # sweeping flagged rows for ~5 ms, a VALU-bound general pass) run on a second stream beside the next frames' forward (round 6:
    # 190 -> 223 pairs/s on one box; on c3 the same switch buys nothing: 793 vs 781); 8 steps = 64 frames.  c3match = SURVEY 8(d)'s
    # second form of the metric's network (NET_SIZE_MATCH: net 1024, 4097 tokens); c4 also at the paper's r_max 3000.
    # The c4 legs run MIOpen's float32 convolutions on ~60 shapes nobody has searched on a fresh box: the default find mode costs
    # ~330 s of wall there (kernel compilation) for 0.6 s of timed work, MIOPEN_FIND_MODE=FAST 9 s -- but FAST's immediate-mode
    # choices run the image in 1156 ms instead of 607 (measured, round 6), so the search stays; the r_max 3000 leg comes second and
    # finds most of its shapes (fixed patch sizes, pix2pix at 1024^2) in the user find-db the first leg left.
    legs = [("c5", ["--config", "c5", "--steps", "8", "--warmup", "1", "--overlap"], timeout_s, None),
            ("c2", ["--config", "c2", "--steps", "20", "--warmup", "3"], timeout_s, None),
            ("c3match", ["--config", "c3match", "--steps", "3", "--warmup", "1"], timeout_s, None),
            ("c4", ["--config", "c4", "--steps", "1", "--warmup", "0"], 2 * timeout_s, None),
            ("c4_rmax3000", ["--config", "c4", "--steps", "1", "--warmup", "0", "--boost-rmax", "3000"], 2 * timeout_s, None)]
    out = {}
    for name, extra, limit, env_extra in legs:
        cmd = [sys.executable, os.path.abspath(__file__)] + extra + ["--no-cpu-baseline", "--no-route-check", "--no-funnel", "--no-other-configs"]
        env = dict(os.environ)
        env.update({k: v for k, v in (env_extra or {}).items() if k not in os.environ})          # (the caller's own setting wins)
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit, text=True, env=env)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not line:
                out[name] = {"error": f"exit code {p.returncode}", "stderr_tail": p.stderr[-400:], "seconds": time.perf_counter() - t0}
                continue
            j = json.loads(line[-1])
            roof = j.get("roofline") or {}
            out[name] = {"metric": j.get("metric"), "value": j.get("value"), "unit": j.get("unit"), "ms_per_step": j.get("ms_per_step"),
                         "steps": j.get("steps"), "warmup": j.get("warmup"), "n_gpus": j.get("n_gpus"), "dtype": j.get("dtype"),
                         "workload": (j.get("config") or {}).get("workload"), "forward_launch": (j.get("config") or {}).get("forward_launch"),
                         "roofline": {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_kernel_ms", "source")},
                         "overlap": (j.get("config") or {}).get("overlap"),
                         "command": " ".join(f"{k}={v}" for k, v in (env_extra or {}).items()) + (" " if env_extra else "") + "python bench.py " + " ".join(extra),
                         "seconds": time.perf_counter() - t0}
        except subprocess.TimeoutExpired:
            out[name] = {"error": f"timed out after {limit} s", "seconds": time.perf_counter() - t0}
        except Exception as e:                                # a leg must never take the line down
            out[name] = {"error": repr(e)[:300], "seconds": time.perf_counter() - t0}
    return out


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


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # run as the driver runs N = 1 -- plain `python bench.py --gpus N` -- there is no launcher around us: be the launcher
        sys.exit(self_launch(args, argv))
    if args.selftest_launch:
        return selftest_launch(args)
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
        # round 6: the network path is bit-reproducible too (in-tree kernels: one accumulation chain per output element wherever its
        # tile lands; library convolutions: MIOpen's deterministic solvers, vit_mi355x.deterministic_library) -- identity is asserted
        # with a network as well, unless DS_DETERMINISTIC=0 switched the library half off
        from src import vit_mi355x as _vm
        exact = model is None or _vm.DETERMINISTIC_LIBRARY
        gather_check = compare_gathered(gathered[world - 1], want, layout, exact=exact)
        gather_check["rank"] = world - 1
        gather_check["what"] = ("rank 0's own render of the last rank's units vs the bytes gathered from that rank"
                                + ("; identity asserted" if exact else "; DS_DETERMINISTIC=0: MIOpen's split-K solvers are not "
                                   "bit-reproducible between launches, fractions reported instead of asserting identity"))
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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f16 (network) / f64 (stereo, normal map)" if model is not None else "f64",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.config}: {wl}depth->u16 + create_stereoimages({args.fill}, left-right, "
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
    if os.environ.get("DS_FUNNEL_GRAPH", "1") != "0":        # the group's forward as ONE hipGraph replay (src/hip_graph.py), captured on
        from src.hip_graph import GraphedForward             # the third use of a shape like the product's own predictor ("auto")
        graphed = GraphedForward(lambda x: run_forward(model, model_name, x, net_size, net_h), lazy=2)

    core.model_holder.register_predictor(mt, _Pred())
    pils = [Image.fromarray(a) for a in img_np]
    opts = {"model_type": mt, "gen_stereo": True, "stereo_modes": ["left-right"], "gen_normalmap": normalmap,
            "net_width": net_size, "net_height": net_size if net_h is None else net_h}
    for _ in range(3):                                       # warm-up calls (the second one's groups are captured into hipGraphs)
        n_out = sum(1 for _ in core.core_generation_funnel(None, list(pils), None, None, opts))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_out = sum(1 for _ in core.core_generation_funnel(None, list(pils), None, None, opts))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = dict(core.FUNNEL_STATS)
    return {"value": len(pils) / dt, "unit": "pairs/s", "results": n_out, "seconds": dt,
            "forward_launch": ("hipGraph replay" if (graphed is not None and graphed.graphs) else "eager"),
            "host_seconds": {"enqueue (decode + stage + launch)": st.get("launch"), "enqueue: decode + upload": st.get("launch_decode"),
                             "enqueue: network forward": st.get("launch_forward"), "enqueue: post-processing + downloads": st.get("launch_post"),
                             "blocked on device results": st.get("wait"),
                             "groups": st.get("groups"), "rest (PIL conversion, generator overhead)":
                             None if not st else st.get("total", dt) - (st.get("launch") or 0.0) - (st.get("wait") or 0.0)},
            "what": "core_generation_funnel: PIL in -> uint16 depth, left-right pair" + (", normal map" if normalmap else "")
                    + " as PIL out (host<->device copies and PIL conversion included)"}


if __name__ == "__main__":
    main()
