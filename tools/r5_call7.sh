#!/bin/bash
# round 5, GPU call 7: which float32 convolutions carry config 4 -- shape probe + kernel trace of the (primed) step
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5g; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/c4_conv_probe.py > $O/c4_conv_probe.txt 2>&1; grep -v MIOpen $O/c4_conv_probe.txt | tail -60
timeout 700 python bench.py --config c4 --steps 1 --warmup 0 > $O/c4_prime.json 2> $O/c4_prime.log      # primes MIOpen's find db for this box
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --config c4 --steps 2 --warmup 0 > $GRAFT_REPO_ROOT/$O/c4.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c4_kernel_stats.csv; rm -rf $O/prof
head -16 $O/c4_kernel_stats.csv | cut -c1-180
