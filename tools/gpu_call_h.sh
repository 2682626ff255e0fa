#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h
rm -rf $O; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_models.py -m gpu -q -k "graph or linear" > $O/pytest_sel.log 2>&1; grep -v MIOpen $O/pytest_sel.log | tail -25
timeout 300 python tools/microbench.py linear > $O/linear.txt 2>&1; cat $O/linear.txt | tail -20
DS_ATT_NQB=1 timeout 120 python tools/microbench.py attention > $O/att_base.txt 2>&1; grep -i "attention" $O/att_base.txt | head -8
DS_ATT_NQB=1 DS_ATT_LATE=1 timeout 120 python tools/microbench.py attention > $O/att_late.txt 2>&1; grep -i "attention" $O/att_late.txt | head -8
ls $O
