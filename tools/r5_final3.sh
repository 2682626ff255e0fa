#!/bin/bash
# round 5: the whole GPU suite + smoke on the final library, and the exact sweep probe with the final default
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5final3; rm -rf $O; mkdir -p $O
timeout 200 python tools/exact_sweep_probe.py 2>&1 | grep -v "MIOpen\|amdgpu" > $O/exact4k.txt; cat $O/exact4k.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -v MIOpen $O/pytest.log | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.log; python tools/show_bench.py $O/bench.json | head -3
