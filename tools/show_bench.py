#!/usr/bin/env python3
"""One-screen digest of a bench.py JSON line (used by tools/gpu_call.sh)."""
import json
import sys

try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    print("no bench line:", e)
    sys.exit(0)
print(f"{j['value']:.1f} {j['unit']}  {j['ms_per_step']:.2f} ms/step")
for k, v in j.items():
    if k.startswith("roofline") and isinstance(v, dict):
        print(f"  {k}: {v.get('kernel')}: {v.get('avg_kernel_ms', 0):.4f} ms  {v.get('achieved', 0):.1f} {v.get('unit')}  frac {v.get('frac', 0):.3f}")
for k in ("route_check", "funnel", "cpu_baseline"):
    if k in j:
        v = dict(j[k])
        v.pop("what", None), v.pop("sample", None), v.pop("python_fallback", None)
        print(f"  {k}: {json.dumps(v)[:400]}")
