#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/k
rm -rf $O; mkdir -p $O
cd $R
for sc in 1 0; do
DS_LIN_SCHED=$sc timeout 400 python -m pytest tests/test_gpu_models.py -m gpu -q -k "conv3x3 or linear" > $O/pytest_s$sc.log 2>&1; echo "sched $sc:"; grep -v MIOpen $O/pytest_s$sc.log | tail -4
done
for sc in 1 0; do
echo "== sweep sched $sc"; DS_LIN_SCHED=$sc timeout 200 python tools/microbench.py sweep > $O/sweep_s$sc.txt 2>&1; grep "rounds=8" $O/sweep_s$sc.txt
done
echo "== sweep sched 1, no stores"; DS_LIN_SCHED=1 DS_LIN_ABLATE=1 timeout 200 python tools/microbench.py sweep > $O/sweep_s1_nostore.txt 2>&1; grep "rounds=8" $O/sweep_s1_nostore.txt
echo "== sweep sched 1, no K loop"; DS_LIN_SCHED=1 DS_LIN_ABLATE=2 timeout 200 python tools/microbench.py sweep > $O/sweep_s1_noloop.txt 2>&1; grep "rounds=8" $O/sweep_s1_noloop.txt | head -3
for sc in 1 0; do
echo "== linear sched $sc"; DS_LIN_SCHED=$sc DS_LIN_SHAPES=fc1+gelu,fc2,qk,proj timeout 200 python tools/microbench.py linear > $O/linear_s$sc.txt 2>&1; grep float16 $O/linear_s$sc.txt | head -7
echo "== conv sched $sc"; DS_LIN_SCHED=$sc timeout 300 python tools/microbench.py conv > $O/conv_s$sc.txt 2>&1; grep -v MIOpen $O/conv_s$sc.txt | grep conv3x3
done
