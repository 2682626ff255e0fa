cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6c2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6c2/prof -o k -- python $R/bench.py --config c2 --steps 50 --warmup 3 --no-cpu-baseline --no-funnel --no-route-check --no-micro --no-other-configs > $R/gpurun_out/r6c2/prof.log 2>&1
find $R/gpurun_out/r6c2/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/r6c2/kernel_stats.csv
rm -rf $R/gpurun_out/r6c2/prof
head -40 $R/gpurun_out/r6c2/kernel_stats.csv | cut -c1-170
tail -3 $R/gpurun_out/r6c2/prof.log | cut -c1-300
