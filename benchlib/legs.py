"""Untimed legs of the default bench line: the route check (in-tree kernels vs library routing), the other BASELINE configurations
as sub-processes, and the same batch through the drop-in boundary (core_generation_funnel)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

from .launch import BENCH_PY
from .workload import run_forward


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
        cmd = [sys.executable, BENCH_PY] + extra + ["--no-cpu-baseline", "--no-route-check", "--no-funnel", "--no-other-configs"]
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
    # the same call on four times the images: what the funnel sustains once the first group's decode + upload and the last group's
    # copy + conversion (a fixed ~9 ms per call) are spread over eight groups instead of two
    many = list(pils) * 4
    sum(1 for _ in core.core_generation_funnel(None, list(many), None, None, opts))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n_many = sum(1 for _ in core.core_generation_funnel(None, list(many), None, None, opts))
    torch.cuda.synchronize()
    dt_many = time.perf_counter() - t1
    return {"value": len(pils) / dt, "unit": "pairs/s", "results": n_out, "seconds": dt,
            "forward_launch": ("hipGraph replay" if (graphed is not None and graphed.graphs) else "eager"),
            "sustained": {"value": len(many) / dt_many, "unit": "pairs/s", "images": len(many), "results": n_many, "seconds": dt_many,
                          "what": "one call on 4 x the batch (8 groups of 16): the call's fixed head and tail spread over more groups"},
            "host_seconds": {"enqueue (decode + stage + launch)": st.get("launch"), "enqueue: decode + upload": st.get("launch_decode"),
                             "enqueue: network forward": st.get("launch_forward"), "enqueue: post-processing + downloads": st.get("launch_post"),
                             "blocked on device results": st.get("wait"),
                             "groups": st.get("groups"), "rest (PIL conversion, generator overhead)":
                             None if not st else st.get("total", dt) - (st.get("launch") or 0.0) - (st.get("wait") or 0.0)},
            "what": "core_generation_funnel: PIL in -> uint16 depth, left-right pair" + (", normal map" if normalmap else "")
                    + " as PIL out (host<->device copies and PIL conversion included)"}
