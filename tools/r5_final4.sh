#!/bin/bash
# round 5: the remaining profile lines on the final library -- per-pixel path alone, NET_SIZE_MATCH, config 4
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5final4; rm -rf $O; mkdir -p $O
timeout 200 python bench.py --model none --steps 20 --warmup 3 --no-cpu-baseline > $O/stereo_only.json 2> $O/stereo_only.log; python tools/show_bench.py $O/stereo_only.json | head -2
timeout 300 python bench.py --config c3match --steps 10 --warmup 2 --no-cpu-baseline --no-route-check > $O/c3match.json 2> $O/c3match.log; python tools/show_bench.py $O/c3match.json | head -3
timeout 800 python bench.py --config c4 --steps 3 --warmup 1 > $O/c4.json 2> $O/c4.log; python tools/show_bench.py $O/c4.json | head -2
