#!/bin/bash
# round 5, GPU call 10: the cooperative exact sweep on 1080p frames (config 5) -- is it a win there, and from which active-set size?
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5j; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "exact" > $O/pytest.log 2>&1; grep -v MIOpen $O/pytest.log | tail -3
B="--config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-route-check --no-micro"
for v in "COOP=0" "COOP_MIN=0" "COOP_MIN=8" "COOP_MIN=24" "COOP_MIN=64" "COOP=0"; do
  env DS_PL_EXACT_$v timeout 200 python bench.py $B > $O/c5_$v.json 2> $O/c5_$v.log
  python - <<PY
import json
j=json.loads(open("$O/c5_$v.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("$v", round(j["value"],1), "pairs/s", round(j["ms_per_step"],2), "ms/step; polylines main+general", round(r.get("avg_kernel_ms",0),3), "ms, exact", round(r.get("exact_fallback_ms",0),3), "ms, rows", r.get("exact_fallback_rows"))
PY
done
timeout 200 python tools/exact_sweep_probe.py 2>&1 | grep -v "MIOpen\|amdgpu" > $O/exact4k.txt; cat $O/exact4k.txt
