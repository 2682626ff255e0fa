#!/bin/bash
# rocprofv3 kernel stats of BASELINE config 4 (Boost on one 4K image): gpurun -- 'bash tools/c4_profile.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6c4
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6c4/prof -o k -- python $R/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-funnel --no-route-check --no-micro --no-other-configs > $R/gpurun_out/r6c4/prof.log 2>&1
find $R/gpurun_out/r6c4/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/r6c4/kernel_stats.csv
rm -rf $R/gpurun_out/r6c4/prof
tail -2 $R/gpurun_out/r6c4/prof.log | cut -c1-300
