#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/u
rm -rf $O; mkdir -p $O
cd $R
echo "== default"; timeout 200 python tools/determinism_check.py 2>&1 | grep -v MIOpen | tail -6 | tee $O/default.txt
echo "== DS_ATT_LATE=0"; DS_ATT_LATE=0 timeout 200 python tools/determinism_check.py 2>&1 | grep -v MIOpen | tail -6 | tee $O/late0.txt
echo "== DS_LINEAR=0"; DS_LINEAR=0 timeout 200 python tools/determinism_check.py 2>&1 | grep -v MIOpen | tail -6 | tee $O/lin0.txt
