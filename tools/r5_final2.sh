#!/bin/bash
# round 5: kernel trace of the default command on a box whose MIOpen find db has seen the shapes (the first trace of tools/r5_final.sh ran
# as the first command of a fresh box and holds MIOpen's one-time search kernels: 40 naive_conv launches of 68 ms)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r5final2; rm -rf $O; mkdir -p $O
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-funnel --no-route-check --no-micro --no-other-configs > $O/warm.json 2> $O/warm.log
python tools/show_bench.py $O/warm.json | head -1
bash tools/gpu_call.sh r5final2b prof
cp gpurun_out/r5final2b/kernel_stats.csv $O/kernel_stats.csv
python tools/step_anatomy.py $O/kernel_stats.csv
