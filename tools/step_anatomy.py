#!/usr/bin/env python3
"""Shares of one bench step by kernel family, from a `rocprofv3 --kernel-trace --stats` summary (the figures DESIGN.md §5 quotes).

    python tools/step_anatomy.py [profiles/round5_kernel_stats.csv] [--forwards 19]

`--forwards` = network forwards inside the profiled command (`bench.py --steps 10 --warmup 2` runs 19: seed probing, priming,
warm-up, the timed steps and the per-kernel timing passes); it only scales the per-forward column."""
import csv
import sys

def _lin_conv(n):
    """k_linear256<BF16, EPI, CONV, RES, VT, NH>: the third template argument"""
    return n.startswith("void k_linear256<") and n.split("<")[1].split(">")[0].split(",")[2].strip() == "1"


FAMILIES = [
    ("token GEMMs, main kernel (k_linear256, dense)", lambda n: n.startswith("void k_linear256<") and not _lin_conv(n)),
    ("3x3 convolutions in-tree (k_linear256, implicit GEMM)", _lin_conv),
    ("ragged rounds (k_linear_ragged)", lambda n: "k_linear_ragged" in n),
    ("attention (k_attention_fwd2)", lambda n: "k_attention_fwd" in n),
    ("LayerNorm (k_residual_layernorm)", lambda n: "k_residual_layernorm" in n),
    ("polylines warp (all three passes)", lambda n: "k_polylines" in n),
    ("other in-tree kernels", lambda n: n.startswith("void k_")),
    ("MIOpen / CK convolutions", lambda n: n.startswith("igemm") or "conv" in n.lower() or n.startswith("SubTensorOp") or "naive_conv" in n),
    ("library GEMMs (hipBLASLt / rocBLAS)", lambda n: n.startswith("Cijk") or "rocblas" in n.lower()),
    ("aten element-wise / resize / copies", lambda n: True),
]


def main(argv):
    path = "profiles/round5_kernel_stats.csv"
    forwards = 19
    args = list(argv)
    while args:
        a = args.pop(0)
        if a == "--forwards":
            forwards = int(args.pop(0))
        else:
            path = a
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    acc = {name: 0.0 for name, _ in FAMILIES}
    for r in rows:
        for name, pred in FAMILIES:
            if pred(r["Name"]):
                acc[name] += float(r["TotalDurationNs"])
                break
    in_tree = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith("void k_"))
    print(f"{path}: {total / 1e6:.1f} ms of kernel time, {total / 1e6 / forwards:.2f} ms per forward ({forwards} forwards)")
    for name, _ in FAMILIES:
        print(f"  {acc[name] / total * 100:6.2f} %  {acc[name] / 1e6 / forwards:7.3f} ms  {name}")
    print(f"  {in_tree / total * 100:6.2f} %  in-tree kernels (hand-written HIP) in all")
    # a launch that stalled (a code object paged in, a hiccup of the box) distorts an average: say so
    for r in rows:
        if int(r["Calls"]) >= 8 and float(r["MaxNs"]) > 20.0 * float(r["AverageNs"]) and float(r["MaxNs"]) > 1e6:
            clean = (float(r["TotalDurationNs"]) - float(r["MaxNs"])) / (int(r["Calls"]) - 1)
            print(f"  outlier: {r['Name'][:60]} has one launch of {float(r['MaxNs']) / 1e6:.1f} ms "
                  f"(average {float(r['AverageNs']) / 1e3:.1f} us with it, {clean / 1e3:.1f} us without)")


if __name__ == "__main__":
    main(sys.argv[1:])
