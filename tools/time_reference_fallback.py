#!/usr/bin/env python3
"""Time the REFERENCE's own numba-less fallback (src/stereoimage_generation.py executed by CPython, :1-8) on one
512x512 unit -- BASELINE config 1 -- in the BUILD container (the only place /root/reference exists), with the two outside
shims of SURVEY.md Appendix A.  Writes profiles/round2_reference_fallback.json; bench.py's cpu_baseline.python_fallback
leg (a pure-Python port timed on the GPU box's host) is the figure that travels.

    PYTHONDONTWRITEBYTECODE=1 python tools/time_reference_fallback.py
"""
import json
import os
import platform
import sys
import time

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    import make_golden as mg
    sg = mg.load_reference_stereo()      # FIRST: the reference's `src` is a namespace package and must win over ours
    import bench
    from oracle import oracle as orc
    from oracle import oracle_py
    bench.H = bench.W = 512
    bench.wl.configure(512, 512)
    img, pred = bench.synth_batch(1, 1000)
    d16 = orc.convert_to_i16(orc.depth_normalize01(pred[0], False))
    out = {"host": platform.processor() or platform.machine(), "cores_used": 1, "unit": "512x512, polylines_sharp, left-right, divergence 2.5"}
    t = time.perf_counter()
    ref = np.asarray(sg.create_stereoimages(img[0], d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')[0])
    out["reference_fallback_seconds"] = time.perf_counter() - t
    t = time.perf_counter()
    port = oracle_py.create_stereoimages_arrays(img[0], d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')[0]
    out["python_port_seconds"] = time.perf_counter() - t
    t = time.perf_counter()
    cport = orc.create_stereoimages_arrays(img[0], d16, 2.5, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')[0]
    out["c_port_seconds"] = time.perf_counter() - t
    out["c_port_threads"] = orc.num_threads()
    out["all_identical"] = bool(np.array_equal(ref, port) and np.array_equal(ref, cport))
    with open(os.path.join(ROOT, "profiles", "round2_reference_fallback.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
