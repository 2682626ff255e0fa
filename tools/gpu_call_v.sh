#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/v
rm -rf $O; mkdir -p $O
cd $R
for i in 1 2 3; do timeout 200 python -m pytest tests/test_gpu_models.py -m gpu -q -k "graph" > $O/pytest_graph$i.log 2>&1; grep -v MIOpen $O/pytest_graph$i.log | tail -3; done
