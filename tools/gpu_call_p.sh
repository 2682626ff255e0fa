#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/p
rm -rf $O; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_models.py -m gpu -q -k "conv3x3 or linear" > $O/pytest_sel.log 2>&1; grep -v MIOpen $O/pytest_sel.log | tail -5
export DS_SWEEP_K=128,1024,4096
echo "== sweep"; timeout 100 python tools/microbench.py sweep 2>&1 | grep "rounds=8" | tee $O/sweep.txt
echo "== sweep nostore"; DS_LIN_ABLATE=1 timeout 100 python tools/microbench.py sweep 2>&1 | grep "rounds=8" | tee $O/sweep_nostore.txt
unset DS_SWEEP_K
DS_LIN_SHAPES=fc1+gelu timeout 200 python tools/microbench.py linear 2>&1 | grep float | head -4 | tee $O/linear.txt
