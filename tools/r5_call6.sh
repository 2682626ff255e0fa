#!/bin/bash
# round 5, GPU call 6: the compact exact-fallback kernel (tests + config 4 before / after) and the Boost parity value
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5f; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "exact or full_frame or reference_goldens or stereo or funnel or video" > $O/pytest.log 2>&1; grep -v MIOpen $O/pytest.log | tail -6
for v in lds global; do
  if [ $v = global ]; then export DS_PL_EXACT_GLOBAL=1; else unset DS_PL_EXACT_GLOBAL; fi
  timeout 800 python bench.py --config c4 --steps 3 --warmup 1 > $O/c4_$v.json 2> $O/c4_$v.log
  echo "c4 $v: $(python tools/show_bench.py $O/c4_$v.json | head -1)"
done
unset DS_PL_EXACT_GLOBAL
timeout 600 python tools/parity_probe.py boost > $O/probe.log 2>&1; cp gpurun_out/parity_probe.json $O/parity_probe_boost.json; grep -A4 '"boost"' $O/parity_probe_boost.json
