#!/bin/bash
# round 5, GPU call 5: kernel anatomy of BASELINE config 4 (Boost) -- which convolutions carry the step
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --config c4 --steps 2 --warmup 0 > $GRAFT_REPO_ROOT/$O/c4.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c4_kernel_stats.csv; rm -rf $O/prof
head -30 $O/c4_kernel_stats.csv | cut -c1-200
tail -2 $O/c4.log | cut -c1-600
