#!/bin/bash
# round 5, last GPU seconds: the exact sweep on 1080p rows (config 5's width), and the funnel with its forward replayed from a hipGraph
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5final5; rm -rf $O; mkdir -p $O
PROBE_H=1080 PROBE_W=1920 timeout 100 python tools/exact_sweep_probe.py 2>&1 | grep -v "MIOpen\|amdgpu" > $O/exact1080.txt; cat $O/exact1080.txt
DS_FUNNEL_GRAPH=1 timeout 150 python bench.py --no-cpu-baseline --no-other-configs --no-route-check --no-micro --steps 5 --warmup 2 > $O/funnel_graph.json 2> $O/funnel_graph.log
python - <<PY
import json
j=json.loads(open("$O/funnel_graph.json").read().strip().splitlines()[-1])
print("value", round(j["value"],1), "funnel with a hipGraph forward", round(j["funnel"]["value"],1), "pairs/s", j["funnel"]["host_seconds"])
PY
