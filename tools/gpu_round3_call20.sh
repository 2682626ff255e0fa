#!/bin/bash
# Round 3, GPU call 20: two-word candidate-mask specialisation of the general-pixel pass (1080p windows): parity + c5 kernel times.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call20
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
timeout 300 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "funnel or video" > $O/pytest_funnel.log 2>&1; tail -2 $O/pytest_funnel.log
cd /tmp && export TMPDIR=/tmp
DS_CUDNN_BENCHMARK=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python $R/bench.py --config c5 --no-cpu-baseline --steps 10 > $O/c5.json 2> $O/prof_c5.log
f=$(ls $O/prof_c5/*/*kernel_stats.csv 2>/dev/null | head -1); grep -i "polylines" $f | cut -c1-160 | tee $O/c5_polylines_kernels.txt; rm -rf $O/prof_c5
