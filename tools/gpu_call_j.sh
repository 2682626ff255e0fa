#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j
rm -rf $O; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_models.py -m gpu -q -k "conv3x3 or linear" > $O/pytest_sel.log 2>&1; grep -v MIOpen $O/pytest_sel.log | tail -8
timeout 200 python tools/microbench.py sweep > $O/sweep.txt 2>&1; tail -16 $O/sweep.txt
DS_LIN_SHAPES=fc1+gelu,fc2,qk,proj timeout 200 python tools/microbench.py linear > $O/linear.txt 2>&1; grep float16 $O/linear.txt | head -8
timeout 300 python tools/microbench.py conv > $O/conv.txt 2>&1; grep -v MIOpen $O/conv.txt | tail -8
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o c3 -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/prof_bench.json 2> $O/prof_bench.err
ls $O/prof | head; find $O/prof -name "*kernel_stats*" | head -2
