#!/usr/bin/env python3
"""Standalone timings of the hand-written HIP kernels at the shapes the default bench launches them with
(dpt_beit_large_512, batch 32).  No library convolutions are involved, so it runs in seconds on a fresh box.

    python tools/microbench.py [attention] [head] [rln] [upsample] [stereo]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from src import _native as nat  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    which = (set(sys.argv[1:]) - {"lin1", "conv1"}) or {"attention", "head", "rln", "upsample", "stereo", "normalmap", "readout"}
    B = 32
    dev = torch.device("cuda")
    if "attention" in which:
        from src import vit_mi355x as vm
        tag = "v1" if os.environ.get("DS_ATT_V1") else ("v2 split" if os.environ.get("DS_ATT_SPLIT") else "v2")
        for name, n, bias, bb in (("beit-l 1025 +bias", 1025, True, B), ("dinov2-l 1370", 1370, False, B),
                                  ("dinov2-l 2443 (1080p)", 2443, False, 8), ("beit-l 4097 +bias (net 1024)", 4097, True, 8),
                                  ("vit-b 577", 577, False, B)):
            npad = (n + 63) // 64 * 64
            hh = 12 if n == 577 else 16
            qk = torch.randn(bb, npad, 2, hh, 64, device=dev, dtype=torch.float16)
            vt = torch.randn(bb, hh * 64, npad, device=dev, dtype=torch.float16)
            braw = torch.randn(hh, n, n, device=dev) if bias else None
            bt = nat.attention_bias_pack(braw, npad, torch.float16) if bias else None
            ms = timeit(lambda: nat.attention_fwd(qk, vt, n, 0.125, bt))
            # value check of the timed configuration on its first two batch elements (float32 definition)
            padded = None
            if bias:
                padded = torch.zeros((hh, npad, npad), device=dev)
                padded[:, :n, :n] = braw
            got = nat.attention_fwd(qk[:2], vt[:2], n, 0.125, bt)
            want = vm.attention_reference(qk[:2].float(), vt[:2].float(), n, 0.125, padded)
            err = (got.float()[:, :n] - want[:, :n]).abs().max().item()
            print(f"attention [{tag}] {name} x{bb}: {ms:.3f} ms  {4.0 * n * n * hh * 64 * bb / ms / 1e9:.1f} TF/s  max|err| {err:.2e}")
    if "head" in which:
        conv3 = nn.Conv2d(128, 32, 3, padding=1).to(dev).half()
        conv1 = nn.Conv2d(32, 1, 1).to(dev).half()
        x = torch.randn(B, 128, 256, 256, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        ref = None
        for mode in ("tile", "persist", "stream"):
            os.environ["DS_HEAD_MODE"] = mode
            out = nat.dpt_head_tail(x, (512, 512), conv3, conv1, True).float()
            ref = out if ref is None else ref
            ms = timeit(lambda: nat.dpt_head_tail(x, (512, 512), conv3, conv1, True))
            print(f"dpt_head_tail [{mode}] 256^2 -> 512^2 x{B}: {ms:.3f} ms  {2.0 * 9 * 128 * 32 * 512 * 512 * B / ms / 1e9:.1f} TF/s"
                  f"  max|diff to tile| {(out - ref).abs().max().item():.2e}")
        os.environ.pop("DS_HEAD_MODE", None)
    if "rln" in which:
        x = torch.randn(B * 1088, 1024, device=dev, dtype=torch.float16)
        o = torch.randn_like(x)
        g = torch.randn(1024, device=dev, dtype=torch.float16)
        ms = timeit(lambda: nat.residual_layernorm(x, o, g, g, g, 1e-6))
        print(f"residual_layernorm {B}x1088x1024: {ms * 1e3:.1f} us  {4 * x.numel() * 2 / ms / 1e6:.0f} GB/s")
    if "upsample" in which:
        x = torch.randn(B, 256, 128, 128, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        ms = timeit(lambda: nat.upsample_bilinear(x, scale_factor=2, align_corners=True))
        print(f"upsample 128^2 -> 256^2 x256ch x{B}: {ms * 1e3:.1f} us  {5 * x.numel() * 2 / ms / 1e6:.0f} GB/s")
    if "stereo" in which:
        import numpy as np
        import src.stereoimage_generation as sg
        sys.path.insert(0, ROOT)
        import bench
        img_np, pred_np = bench.synth_batch(B, 1000)
        img = torch.from_numpy(img_np).to(dev)
        d16 = nat.depth_to_u16(torch.from_numpy(pred_np).to(dev), False)
        ms = timeit(lambda: sg.create_stereoimages_batch(img, d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp'), reps=10)
        print(f"create_stereoimages_batch polylines_sharp x{B} (1024^2): {ms:.3f} ms  {B / ms * 1e3:.0f} pairs/s")


def extra(which):
    B = 32
    dev = torch.device("cuda")
    if "normalmap" in which:
        import numpy as np
        import src.normalmap_generation as nmg
        d16 = torch.from_numpy(np.random.default_rng(0).integers(0, 65536, (B, 1024, 1024), dtype=np.uint16)).to(dev)
        ms = timeit(lambda: nmg.create_normalmap_batch(d16))
        print(f"create_normalmap_batch x{B} (1024^2): {ms * 1e3:.1f} us  {5 * d16.numel() / ms / 1e6:.0f} GB/s")
    if "readout" in which:
        proj = torch.randn(B, 1025, 1024, device=dev, dtype=torch.float16)
        cls = torch.randn(B, 1024, device=dev, dtype=torch.float16)
        ms = timeit(lambda: nat.reassemble_readout(proj, cls))
        print(f"reassemble_readout {B}x1025x1024: {ms * 1e3:.1f} us  {2 * proj.numel() * 2 / ms / 1e6:.0f} GB/s")
        x = torch.randn(B, 256, 128, 128, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        bias = torch.randn(256, device=dev, dtype=torch.float16)
        ms = timeit(lambda: nat.bias_act(x, bias, relu=True, res1=x, inplace=False))
        print(f"bias_act {B}x256x128x128 (+residual, relu): {ms * 1e3:.1f} us  {3 * x.numel() * 2 / ms / 1e6:.0f} GB/s")


def linear_bench():
    """ds_linear against the library GEMM (+ aten GELU) at the encoder's shapes: values vs float32, a race screen
    (repeated launches must be bit-identical), interleaved timing rounds."""
    import torch.nn.functional as F
    dev = torch.device("cuda")
    shapes = [(34816, 4096, 1024, True, "fc1+gelu L/512 b32"), (34816, 1024, 4096, False, "fc2"),
              (34816, 2048, 1024, False, "qk"), (34816, 1024, 1024, False, "proj"),
              (3264, 4096, 1024, True, "fc1+gelu ragged rows"), (9772, 4096, 1024, True, "fc1+gelu DA-V2 1080p b4"),
              (1088, 1536, 384, True, "fc1+gelu vits b1")]
    only = os.environ.get("DS_LIN_SHAPES")
    for dt in (torch.float16, torch.bfloat16):
        for (m, n, k, gelu, name) in shapes:
            if only and name.split()[0] not in only.split(","):
                continue
            g = torch.Generator(device="cpu").manual_seed(m + n + k)
            x = (torch.randn(m, k, generator=g)).to(dev, dt)
            w = (torch.randn(n, k, generator=g) * k ** -0.5).to(dev, dt)
            b = torch.randn(n, generator=g).to(dev, dt)
            got = nat.linear(x, w, b, gelu)
            rows = torch.randint(0, m, (512,), generator=g).to(dev)
            rows[-1] = m - 1
            ref = x[rows].float() @ w.float().T + b.float()
            ref = F.gelu(ref) if gelu else ref
            err = (got[rows].float() - ref).abs().max().item()
            lib = F.linear(x, w, b)
            lib = F.gelu(lib) if gelu else lib
            err_lib = (lib[rows].float() - ref).abs().max().item()
            same = all(torch.equal(nat.linear(x, w, b, gelu), got) for _ in range(4))
            full = (got.float() - lib.float()).abs().max().item()
            t_hip, t_lib = [], []
            for _ in range(3):
                t_hip.append(timeit(lambda: nat.linear(x, w, b, gelu), reps=10, warm=2))
                t_lib.append(timeit((lambda: F.gelu(F.linear(x, w, b))) if gelu else (lambda: F.linear(x, w, b)), reps=10, warm=2))
            fl = 2.0 * m * n * k
            print(f"linear {name:26s} {str(dt)[6:]:8s} M{m} N{n} K{k}: hip {min(t_hip) * 1e3:7.1f} us {fl / min(t_hip) / 1e9:6.0f} TF | "
                  f"library {min(t_lib) * 1e3:7.1f} us {fl / min(t_lib) / 1e9:6.0f} TF | err {err:.2e} (library {err_lib:.2e}) "
                  f"max|hip-lib| {full:.2e} repeat-identical {same}", flush=True)


def linear_ab():
    """Round 3: the token GEMMs of one dpt_beit_large_512 block at batch 32 -- in-tree with the ragged round, in-tree without
    it (DS_LIN_RAGGED=0: the persistent walk alone), and the library call the host used before -- interleaved rounds."""
    import torch.nn.functional as F
    dev = torch.device("cuda")
    dt = torch.float16
    g = torch.Generator(device="cpu").manual_seed(5)
    m = 32 * int(os.environ.get("DS_MB_NPAD", "1032"))        # token stride of dpt_beit_large_512 at batch 32 (1088 until round 4)
    x = torch.randn(m, 1024, generator=g).to(dev, dt)
    x4 = torch.randn(m, 4096, generator=g).to(dev, dt)
    res = torch.randn(m, 1024, generator=g).to(dev, dt)
    gam = torch.randn(1024, generator=g).to(dev, dt)

    def w_(n, k):
        return (torch.randn(n, k, generator=g) * k ** -0.5).to(dev, dt)
    w_qk, w_v, w_p, w_1, w_2 = w_(2048, 1024), w_(1024, 1024), w_(1024, 1024), w_(4096, 1024), w_(1024, 4096)
    b_qk, b_p, b_1, b_2 = (torch.randn(n, generator=g).to(dev, dt) for n in (2048, 1024, 4096, 1024))
    h3 = x.view(32, m // 32, 1024)
    cases = [
        (f"qk       {m}x2048x1024", 2.0 * m * 2048 * 1024, lambda: nat.linear(x, w_qk, b_qk, False), lambda: F.linear(x, w_qk, b_qk)),
        (f"v^T      1024x{m}x1024", 2.0 * m * 1024 * 1024, lambda: nat.linear_vt(w_v, h3), lambda: torch.bmm(w_v.unsqueeze(0).expand(32, -1, -1), h3.transpose(1, 2))),
        (f"proj+res {m}x1024x1024", 2.0 * m * 1024 * 1024, lambda: nat.linear_residual(x, w_p, b_p, gam, res), lambda: F.linear(x, w_p, b_p)),
        (f"fc1+gelu {m}x4096x1024", 2.0 * m * 4096 * 1024, lambda: nat.linear(x, w_1, b_1, True), lambda: F.gelu(F.linear(x, w_1, b_1))),
        (f"fc2+res  {m}x1024x4096", 2.0 * m * 1024 * 4096, lambda: nat.linear_residual(x4, w_2, b_2, gam, res), lambda: F.linear(x4, w_2, b_2)),
    ]
    for name, fl, hip, libf in cases:
        t = {"ragged": [], "walk": [], "lib": [], "late": [], "nosplit": [], "nopipe": []}
        for _ in range(3):
            nat.linear_env(DS_LIN_RAGGED="1", DS_LIN_EARLY="1")
            t["ragged"].append(timeit(hip, reps=10, warm=2))
            nat.linear_env(DS_LIN_RAGGED_KSPLIT="1")
            t["nosplit"].append(timeit(hip, reps=10, warm=2))
            nat.linear_env(DS_LIN_RAGGED_KSPLIT=None, DS_LIN_RAGGED_PIPE="0")
            t["nopipe"].append(timeit(hip, reps=10, warm=2))
            nat.linear_env(DS_LIN_RAGGED_PIPE=None)
            nat.linear_env(DS_LIN_EARLY="0")
            t["late"].append(timeit(hip, reps=10, warm=2))
            nat.linear_env(DS_LIN_RAGGED="0", DS_LIN_EARLY="1")
            t["walk"].append(timeit(hip, reps=10, warm=2))
            t["lib"].append(timeit(libf, reps=10, warm=2))
        nat.linear_env(DS_LIN_RAGGED="1")
        r, wk, lb, lt = min(t["ragged"]), min(t["walk"]), min(t["lib"]), min(t["late"])
        print(f"gemm {name}: in-tree {r * 1e3:7.1f} us {fl / r / 1e9:6.0f} TF | ragged round without its K split {min(t['nosplit']) * 1e3:7.1f} us, without pipelined reads {min(t['nopipe']) * 1e3:7.1f} us | "
              f"prologue after the epilogue (round 2 order) {lt * 1e3:7.1f} us | "
              f"without ragged round {wk * 1e3:7.1f} us | library (GEMM only) {lb * 1e3:7.1f} us {fl / lb / 1e9:6.0f} TF", flush=True)
    # the LayerNorm pass behind the fused epilogues reads one operand instead of two
    xs, br = torch.randn(m, 1024, device=dev, dtype=dt), torch.randn(m, 1024, device=dev, dtype=dt)
    t2 = timeit(lambda: nat.residual_layernorm(xs, br, gam, gam, gam, 1e-6))
    t1 = timeit(lambda: nat.residual_layernorm(xs, None, None, gam, gam, 1e-6))
    print(f"residual_layernorm {m}x1024: with branch {t2 * 1e3:.1f} us, LayerNorm only {t1 * 1e3:.1f} us")


def single_shapes(which):
    """One shape per kernel instantiation, a few launches each: what the hardware-counter passes profile."""
    dev = torch.device("cuda")
    if "lin1" in which:
        x = torch.randn(34816, 1024, device=dev, dtype=torch.float16)
        w = torch.randn(4096, 1024, device=dev, dtype=torch.float16) / 32
        b = torch.randn(4096, device=dev, dtype=torch.float16)
        ms = timeit(lambda: nat.linear(x, w, b, True), reps=10, warm=2)
        print(f"linear fc1+gelu 34816x4096x1024: {ms * 1e3:.1f} us  {2.0 * 34816 * 4096 * 1024 / ms / 1e9:.0f} TF/s")
    if "conv1" in which:
        conv = nn.Conv2d(256, 256, 3, padding=1).to(dev, torch.float16)
        x = torch.randn(32, 256, 128, 128, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        ms = timeit(lambda: nat.conv3x3(conv, x, relu=True), reps=10, warm=2)
        print(f"conv3x3 32x128x128 256->256 (+bias, ReLU): {ms * 1e3:.1f} us  {2.0 * 32 * 128 * 128 * 256 * 2304 / ms / 1e9:.0f} TF/s")


def linear_sweep():
    """Per-round time of ds_linear as a function of K (fixed cost per tile vs cost per K-tile), one full round of tiles
    (256) and eight rounds; and the cost of the GELU / residual epilogues."""
    dev = torch.device("cuda")
    dt = torch.float16
    ks = [int(v) for v in os.environ.get("DS_SWEEP_K", "128,256,512,1024,2048,4096,8192").split(",")]
    for rounds in ((8,) if os.environ.get("DS_SWEEP_K") else (1, 8)):
        m = 256 * 16 * rounds
        for k in ks:
            x = torch.randn(m, k, device=dev, dtype=dt)
            w = torch.randn(4096, k, device=dev, dtype=dt) * k ** -0.5
            b = torch.randn(4096, device=dev, dtype=dt)
            t0 = min(timeit(lambda: nat.linear(x, w, b, False), reps=10, warm=2) for _ in range(3))
            t1 = min(timeit(lambda: nat.linear(x, w, b, True), reps=10, warm=2) for _ in range(3))
            print(f"sweep rounds={rounds} K={k:5d}: plain {t0 * 1e3 / rounds:7.2f} us/round  gelu {t1 * 1e3 / rounds:7.2f} us/round  "
                  f"{2.0 * m * 4096 * k / t0 / 1e9:6.0f} TF", flush=True)


def conv_bench():
    """ds_conv3x3_nhwc against the library convolution (+ the element-wise tail it replaces) at the decoder's shapes."""
    import torch.nn.functional as F
    dev = torch.device("cuda")
    torch.backends.cudnn.benchmark = True
    shapes = [(32, 256, 256, 256, 128, "head conv 256->128"), (32, 128, 128, 256, 256, "refinenet1 RCU conv"), (32, 64, 64, 256, 256, "refinenet2 RCU conv"),
              (32, 32, 32, 256, 256, "refinenet3 RCU conv"), (32, 64, 64, 512, 256, "layer2_rn"), (32, 32, 32, 1024, 256, "layer3_rn"),
              (4, 148, 264, 256, 256, "DA-V2 1080p path_1 RCU conv")]
    for dt in (torch.float16,):
        for (b, h, w, cin, cout, name) in shapes:
            conv = nn.Conv2d(cin, cout, 3, padding=1).to(dev, dt).to(memory_format=torch.channels_last)
            x = torch.randn(b, cin, h, w, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
            r = torch.randn(b, cout, h, w, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
            if cout % 256 != 0:
                r = None                                   # the 256 x 128 tiles have no residual epilogue
            got = nat.conv3x3(conv, x, relu=False, res1=r)
            ref = conv._conv_forward(x, conv.weight, None)
            ref = nat.bias_act(ref, conv.bias, relu=False, res1=r)
            err = (got.float() - ref.float()).abs().max().item()
            t_hip, t_lib, t_libc = [], [], []
            for _ in range(3):
                t_hip.append(timeit(lambda: nat.conv3x3(conv, x, relu=False, res1=r), reps=10, warm=2))
                t_lib.append(timeit(lambda: nat.bias_act(conv._conv_forward(x, conv.weight, None), conv.bias, relu=False, res1=r), reps=10, warm=2))
                t_libc.append(timeit(lambda: conv._conv_forward(x, conv.weight, None), reps=10, warm=2))
            fl = 2.0 * b * h * w * cout * 9 * cin
            print(f"conv3x3 {name:28s} {b}x{h}x{w} {cin}->{cout}: hip(+tail) {min(t_hip) * 1e3:7.1f} us {fl / min(t_hip) / 1e9:6.0f} TF | "
                  f"library+tail {min(t_lib) * 1e3:7.1f} us (conv alone {min(t_libc) * 1e3:7.1f} us {fl / min(t_libc) / 1e9:6.0f} TF) | max|hip-lib| {err:.2e}",
                  flush=True)


if __name__ == "__main__":
    if "lin1" in sys.argv[1:] or "conv1" in sys.argv[1:]:
        single_shapes(set(sys.argv[1:]))
        if not (set(sys.argv[1:]) - {"lin1", "conv1"}):
            sys.exit(0)
    if "sweep" in sys.argv[1:]:
        linear_sweep()
        sys.exit(0)
    if "conv" in sys.argv[1:]:
        conv_bench()
        sys.exit(0)
    if "linear" in sys.argv[1:]:
        linear_bench()
        sys.exit(0)
    if "gemms" in sys.argv[1:]:
        linear_ab()
        sys.exit(0)
    main()
    extra(set(sys.argv[1:]) or {"normalmap", "readout"})
