"""The workload bench.py times: frame size and depth regime (module state, set by bench.run_pipeline through configure()), the synthetic
inputs of SURVEY.md 8(d), the random-init networks of the BASELINE configurations, and the CPU baseline leg."""
import time

import numpy as np

H = W = 1024
DEPTH_KIND = "steps"


def configure(height, width, depth_kind=None):
    """Frame size (and depth regime) of the run: every function below reads these."""
    global H, W, DEPTH_KIND
    H, W = int(height), int(width)
    if depth_kind is not None:
        DEPTH_KIND = depth_kind


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
