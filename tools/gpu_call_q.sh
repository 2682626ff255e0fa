#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/q
rm -rf $O; mkdir -p $O
cd $R
DS_ATT_VERSION=3 timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "attention" > $O/pytest_v3.log 2>&1; grep -v MIOpen $O/pytest_v3.log | tail -15
echo "== v3"; DS_ATT_VERSION=3 timeout 120 python tools/microbench.py attention 2>&1 | grep -i "attention" | tee $O/att_v3.txt
echo "== v2"; timeout 120 python tools/microbench.py attention 2>&1 | grep -i "attention" | tee $O/att_v2.txt
