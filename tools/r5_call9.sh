#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5i; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "exact or full_frame or video or funnel" > $O/pytest.log 2>&1; grep -v MIOpen $O/pytest.log | tail -6
timeout 300 python tools/exact_sweep_probe.py > $O/exact_sweep_probe.txt 2>&1; grep -v "MIOpen\|amdgpu" $O/exact_sweep_probe.txt
timeout 300 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-route-check --no-micro > $O/c5.json 2> $O/c5.log; python tools/show_bench.py $O/c5.json | head -3
