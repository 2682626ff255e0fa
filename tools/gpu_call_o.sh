#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/o
rm -rf $O; mkdir -p $O
cd $R
export DS_SWEEP_K=128,1024,4096
for a in 0 4 8 1 0; do
echo "== ablate $a"; DS_LIN_ABLATE=$a timeout 100 python tools/microbench.py sweep 2>&1 | grep "rounds=8" | tee -a $O/sweep_ablate$a.txt
done
