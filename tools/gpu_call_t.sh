#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/t
rm -rf $O; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_models.py -m gpu -q -k "conv3x3 or linear" > $O/pytest_sel.log 2>&1; grep -v MIOpen $O/pytest_sel.log | tail -4
for a in 0 16 0 16; do echo "== ablate $a (16 = direct stores)"; DS_LIN_ABLATE=$a DS_LIN_SHAPES=fc1+gelu,fc2,qk,proj timeout 200 python tools/microbench.py linear 2>&1 | grep float16 | head -4 | cut -c1-120 | tee -a $O/linear_a$a.txt; DS_LIN_ABLATE=$a timeout 200 python tools/microbench.py conv 2>&1 | grep conv3x3 | head -2 | cut -c1-130 | tee -a $O/conv_a$a.txt; done
export DS_SWEEP_K=128,1024,4096
for a in 0 16; do echo "== sweep ablate $a"; DS_LIN_ABLATE=$a timeout 100 python tools/microbench.py sweep 2>&1 | grep "rounds=8" | tee $O/sweep_a$a.txt; done
