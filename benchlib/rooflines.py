"""Roofline objects of the MFMA kernels: algorithmic flops per launch over the average launch duration the C ABI's kernel timers
measured in the instrumented repeat of the timed steps; microbenchmarks of the same launch shapes as side notes."""
from .profiles import clock_from_profile, stats_from_profile, traffic_from_profile

HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0            # what a float4 copy kernel reaches on this chip (same guide: 79 % of the spec peak)
MFMA_PEAK_TFLOPS = 2500.0         # dense f16/bf16 MFMA peak (same guide)


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
