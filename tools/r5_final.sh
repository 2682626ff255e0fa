#!/bin/bash
# round 5, final evidence on ONE box: kernel trace + counters of the default command, then (with those summaries in profiles/, so that
# the line's traffic_from_profile / profile_avg fields are filled) smoke, the whole GPU suite, the driver's line, config 5's anatomy
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/gpu_call.sh r5final prof pmc
O=gpurun_out/r5final
cp $O/kernel_stats.csv profiles/round5_kernel_stats.csv
cp $O/pmc_summary.json profiles/round5_pmc_summary.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -v MIOpen $O/pytest.log | tail -4
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.log; tail -2 $O/bench.log; python tools/show_bench.py $O/bench.json
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof5 -o k -- python $R/bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-route-check --no-micro > $R/$O/c5_prof.log 2>&1)
find $O/prof5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c5_kernel_stats.csv; rm -rf $O/prof5
head -12 $O/c5_kernel_stats.csv | cut -c1-160
