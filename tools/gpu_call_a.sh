#!/bin/bash
# Round-2 call A: new parity tests at the benchmarked shapes + the reworked bench (all configs, funnel leg).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/a
rm -rf $O; mkdir -p $O
cd $R
timeout 700 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
timeout 300 python bench.py --funnel > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-600 $O/bench_c3.json; tail -3 $O/bench_c3.err
DS_FUNNEL_BATCH_PIXELS=16777216 timeout 200 python bench.py --funnel --no-cpu-baseline --steps 5 > $O/bench_c3_f16.json 2> $O/bench_c3_f16.err; python -c "import json;print(json.load(open('$O/bench_c3_f16.json')).get('funnel'))"
timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 20 --funnel > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-300 $O/bench_c2.json
timeout 200 python bench.py --config c5 --no-cpu-baseline --steps 5 > $O/bench_c5.json 2> $O/bench_c5.err; cut -c1-300 $O/bench_c5.json; tail -2 $O/bench_c5.err
timeout 200 python bench.py --config c3match --no-cpu-baseline --steps 5 > $O/bench_c3match.json 2> $O/bench_c3match.err; cut -c1-300 $O/bench_c3match.json; tail -2 $O/bench_c3match.err
python tools/microbench.py > $O/microbench.txt 2>&1; cat $O/microbench.txt
ls $O
