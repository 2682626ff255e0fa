#!/usr/bin/env python3
"""bench.py -- depth + stereo pairs/sec @1024x1024 on MI355X (BASELINE.json's metric).

One *unit* = one 1024x1024 RGB image -> one uint16 depth map + one side-by-side stereo pair (both eyes), SURVEY.md 8(d).
One *step* = one pass of the hot path over a batch of `--batch` units already resident in HBM:

    uint8 RGB batch    --model forward (fp16, MFMA)-->  float32 depth prediction   (depthmap_generation.py:375-403 +
                                                                                    the model family's estimate*())
    float32 prediction --ds_depth_to_u16-------------->  uint16 depth              (core.py:189-211)
    RGB + uint16 depth --ds_stereo_warp--------------->  left-right pair           (stereoimage_generation.py:13-92,
                                                                                    polylines_sharp, divergence 2.5 %)

`--model` picks the depth network (random-init weights of the named architecture -- there are no checkpoints offline):
    dav2_vitl            Depth-Anything-V2 ViT-L/14, net 518 (reference model id 14; BASELINE config 5's network)
    dpt_beit_large_512   MiDaS 3.1 DPT BEiT-L/16, net 512   (reference model id 1;  BASELINE config 3's network) [default]
    dpt_hybrid_384       MiDaS 3.0 ViT-B/16 + ResNetV2-50, net 384 (reference model id 4; BASELINE config 2's network)
    none                 no network: the float32 prediction is a synthetic INPUT and only the per-pixel path is timed
                         (what round 1 measured first; kept to track the stereo kernels on their own)

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU with
torch.distributed.run.  Units are sharded across ranks (weak scaling: every rank renders its own batch, no data-path
collective); with --gather the collated outputs are gathered to rank 0 with one RCCL gather per step, overlapped with
the next step's kernels.  Rank 0 prints ONE JSON line.
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
MFMA_PEAK_TFLOPS = 2500.0         # dense f16/bf16 MFMA peak (same guide)
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


def model_input_size(model_name):
    return {"dav2_vitl": 518, "dpt_beit_large_512": 512, "dpt_hybrid_384": 384}.get(model_name, 0)


def build_model(name, seed=0):
    """Random-init network of the named architecture (torch.manual_seed(seed); no checkpoints offline)."""
    import torch
    torch.manual_seed(seed)
    if name == "dav2_vitl":
        from ddepth_anything_v2 import DepthAnythingV2
        m = DepthAnythingV2(encoder='vitl', features=256, out_channels=[256, 512, 1024, 1024])
        info = {"name": "Depth-Anything-V2 ViT-L/14", "net": 518, "tokens": 37 * 37 + 1, "dim": 1024, "depth": 24, "heads": 16}
    elif name == "dpt_beit_large_512":
        from dmidas.dpt_depth import DPTDepthModel
        m = DPTDepthModel(path=None, backbone="beitl16_512", non_negative=True)
        info = {"name": "MiDaS 3.1 DPT BEiT-L/16 512", "net": 512, "tokens": 32 * 32 + 1, "dim": 1024, "depth": 24, "heads": 16}
    elif name == "dpt_hybrid_384":
        from dmidas.dpt_depth import DPTDepthModel
        m = DPTDepthModel(path=None, backbone="vitb_rn50_384", non_negative=True)
        info = {"name": "MiDaS 3.0 dpt_hybrid_384 (ViT-B/16 + ResNetV2-50)", "net": 384, "tokens": 24 * 24 + 1, "dim": 768, "depth": 12, "heads": 12}
    else:
        raise SystemExit(f"unknown --model {name}")
    return m.eval(), info


def cpu_baseline(model_name, distinct_units, seed, min_seconds=12.0):
    """The same workload on this host's cores, bounded: the float32 torch-eager forward of the same network (what the
    reference runs on a CPU device) + the CPU oracle (C restatement of the reference's numba kernels, OpenMP over rows
    like numba's prange).  `distinct_units` units are processed round-robin until `min_seconds` have been spent."""
    import torch
    from oracle import oracle as orc
    orc.build()
    img, pred = synth_batch(distinct_units, seed)
    model = None
    if model_name != "none":
        model, _ = build_model(model_name)
        model = model.float()

    def one(i):
        if model is not None:
            with torch.no_grad():
                p = model.infer_batch(torch.from_numpy(img[i:i + 1]), model_input_size(model_name)).numpy()[0]
        else:
            p = pred[i]
        d16 = orc.convert_to_i16(orc.depth_normalize01(p, False))
        orc.create_stereoimages_arrays(img[i], d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')

    one(0)
    done = 0
    t0 = time.perf_counter()
    while True:
        one(done % distinct_units)
        done += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds and done >= min(distinct_units, 2):
            break
    what = "torch-eager float32 forward of the same network on the CPU + " if model is not None else ""
    return {"value": done / dt, "unit": "pairs/s", "cores": max(orc.num_threads(), torch.get_num_threads()), "kind": "port",
            "sample": f"{done} units of 1024x1024 ({distinct_units} distinct): {what}depth->u16 + polylines_sharp left-right "
                      f"with the gcc -O2 -fopenmp restatement of the reference's numba kernels, {dt:.2f} s"}


def pmc_traffic(batch, kernel="k_polylines"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (separate --pmc passes, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950): profiles/round1_pmc_summary.json.  None when the summary is
    missing or was taken at another batch size -- bench.py cannot read PMC counters itself."""
    try:
        with open(os.path.join(ROOT, "profiles", "round1_pmc_summary.json")) as f:
            j = json.load(f)
        if int(j.get("batch", -1)) != int(batch):
            return None
        return float(j[kernel]["hbm_bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="units per GPU per step")
    ap.add_argument("--model", default="dpt_beit_large_512", choices=["dav2_vitl", "dpt_beit_large_512", "dpt_hybrid_384", "none"])
    ap.add_argument("--fill", default="polylines_sharp")
    ap.add_argument("--gather", action="store_true", help="gather the collated outputs to rank 0 (N > 1)")
    ap.add_argument("--depth", default="steps", choices=["steps", "smooth"],
                    help="--model none only: synthetic prediction with steps + occluders (default), or smooth only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune-gemms", default=None, metavar="CSV",
                    help="run with TunableOp tuning ON and accumulate the winners in CSV (maintenance: regenerates "
                         "src/tunableop_gfx950.csv); the default run only READS the shipped file")
    ap.add_argument("--cpu-sample", type=int, default=4, help="distinct units of the CPU baseline sample")
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
    from src import vit_mi355x as vm

    img_np, pred_np = synth_batch(args.batch, seed=1000 + rank)
    img = torch.from_numpy(img_np).to(dev)
    pred_in = torch.from_numpy(pred_np).to(dev)
    nat.profile_enable(local_rank, True)
    model, minfo = (None, None)
    if args.model != "none":
        from src import gemm_tuning
        if args.tune_gemms:
            gemm_tuning.enable(tune=True, results=os.path.abspath(args.tune_gemms))
        else:
            gemm_tuning.enable()
        model, minfo = build_model(args.model)
        model = model.to(dev).half()                       # the reference's default on a GPU (depthmap_generation.py:268-275)

    gather_ok = args.gather and world > 1
    side = torch.cuda.Stream(device=dev) if gather_ok else None
    gathered = None
    if gather_ok and rank == 0:
        gathered = [torch.empty((args.batch, H, 2 * W, 3), dtype=torch.uint8, device=dev) for _ in range(world)]

    def step():
        if model is not None:
            pred = model.infer_batch(img, model_input_size(args.model))
        else:
            pred = pred_in
        d16 = nat.depth_to_u16(pred, False)
        sbs = sg.create_stereoimages_batch(img, d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, args.fill)[0]
        if gather_ok:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                dist.gather(sbs, gathered if rank == 0 else None, dst=0)
                sbs.record_stream(side)
        return sbs

    step()                      # priming pass, never timed: library kernel selection (MIOpen find), bias operands, allocator
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
    # k_polylines via HIP events recorded inside the C ABI on the launch stream ...
    render_ms, exact_ms = [], []
    for _ in range(min(args.steps, 5)):
        step()
        r, e = nat.profile_last_ms(local_rank)
        render_ms.append(r)
        exact_ms.append(e)
    exact_rows, general_px = nat.last_stats(img)
    # ... and the fused attention kernel at exactly the shape one encoder block launches it with (torch's current stream
    # IS the launch stream of the ctypes call, so torch events bracket it)
    attn = None
    if model is not None:
        n_tok = minfo["tokens"]
        npad = vm.pad_len(n_tok)
        qk = torch.randn(args.batch, npad, 2, minfo["heads"], 64, device=dev, dtype=torch.float16)
        vt = torch.randn(args.batch, minfo["heads"] * 64, npad, device=dev, dtype=torch.float16)
        bias = None
        if args.model == "dpt_beit_large_512":
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
        attn_flops = 4.0 * n_tok * n_tok * minfo["dim"] * args.batch          # QK^T + PV, 2 flops per MAC
        attn = {"bound": "mfma", "kernel": "k_attention_fwd", "achieved": attn_flops / (attn_ms * 1e-3) / 1e12,
                "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": attn_flops / (attn_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                # HBM bytes per launch (PMC, committed summary): only measured for the default network's shape
                "traffic": pmc_traffic(args.batch, "k_attention_fwd") if args.model == "dpt_beit_large_512" else None,
                "algorithmic_flops_per_launch": attn_flops, "avg_kernel_ms": attn_ms,
                "launches_per_step": minfo["depth"], "shape": {"batch": args.batch, "tokens": n_tok, "heads": minfo["heads"]}}
    torch.cuda.synchronize()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        units = args.batch * world * args.steps
        avg_render_s = float(np.mean(render_ms)) * 1e-3
        achieved = args.batch * ALGO_BYTES_PER_UNIT / avg_render_s / 1e9
        stereo_roof = {"bound": "hbm", "kernel": "k_polylines", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                       "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(args.batch),
                       "algorithmic_bytes_per_launch": args.batch * ALGO_BYTES_PER_UNIT,
                       "avg_kernel_ms": float(np.mean(render_ms)), "exact_fallback_ms": float(np.mean(exact_ms)),
                       "exact_fallback_rows": exact_rows, "general_pixels": general_px}
        wl = (f"{minfo['name']} forward (fp16, random-init weights, net {minfo['net']}) + " if model is not None else "")
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
            "dtype": "f16 (network) / f64 (stereo)" if model is not None else "f64",
            "data": "synthetic",
            "config": {"workload": f"{wl}depth->u16 + create_stereoimages({args.fill}, left-right, divergence 2.5%) on "
                                   f"{args.batch} x 1024x1024 RGB per GPU, inputs resident in HBM"
                                   + ("" if model is not None else "; float32 depth prediction is a synthetic input (--model none)"),
                       "depth_network": args.model, "units_per_step": args.batch * world, "height": H, "width": W,
                       "parallelism": f"units sharded over {world} GPU(s), no data-path collective"
                                      + (", RCCL gather to rank 0 overlapped" if gather_ok else "")},
            # the dominant hand-written kernel of the step: the fused attention when a network runs, else the stereo kernel
            "roofline": attn if attn is not None else stereo_roof,
        }
        if attn is not None:
            out["roofline_stereo"] = stereo_roof
            enc = vm.count_encoder_flops(minfo["depth"], minfo["tokens"], minfo["dim"]) * args.batch
            out["encoder_tflops_per_step"] = enc / 1e12
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.model, args.cpu_sample, seed=1000, min_seconds=args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
