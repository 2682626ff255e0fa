cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6c5p
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6c5p/prof -o k -- python $R/bench.py --config c5 --steps 8 --warmup 1 --overlap --no-cpu-baseline --no-funnel --no-route-check --no-micro --no-other-configs > $R/gpurun_out/r6c5p/prof.log 2>&1
find $R/gpurun_out/r6c5p/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/r6c5p/kernel_stats.csv
rm -rf $R/gpurun_out/r6c5p/prof
