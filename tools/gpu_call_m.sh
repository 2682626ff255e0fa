#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/m
rm -rf $O; mkdir -p $O
cd $R
for v in 0 1 2 3 4 5 0; do
echo "== VAR $v"; DS_LIN_VAR=$v DS_SWEEP_K=1024,4096,8192 timeout 100 python tools/microbench.py sweep 2>&1 | grep "rounds=8" | tee -a $O/sweep_var$v.txt
done
for v in 1 3; do
DS_LIN_VAR=$v timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -q -k "linear" > $O/pytest_var$v.log 2>&1; echo "var $v:"; grep -v MIOpen $O/pytest_var$v.log | tail -3
done
