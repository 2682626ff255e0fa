#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/n
rm -rf $O; mkdir -p $O
cd $R
export DS_SWEEP_K=128,1024,4096
echo "== persistent"; timeout 100 python tools/microbench.py sweep 2>&1 | grep "rounds=8" | tee $O/sweep_persist.txt
echo "== one tile per workgroup"; DS_LIN_GRID=1000000 timeout 100 python tools/microbench.py sweep 2>&1 | grep "rounds=8" | tee $O/sweep_nopersist.txt
for us in 5 10 20 40; do
echo "== persistent, stagger $us us"; DS_LIN_STAGGER_US=$us timeout 100 python tools/microbench.py sweep 2>&1 | grep "rounds=8" | tee $O/sweep_stagger$us.txt
done
echo "== persistent"; timeout 100 python tools/microbench.py sweep 2>&1 | grep "rounds=8" | tee $O/sweep_persist2.txt
unset DS_SWEEP_K
for us in 0 15; do
echo "== linear stagger $us"; DS_LIN_STAGGER_US=$us DS_LIN_SHAPES=fc1+gelu,fc2,qk,proj timeout 200 python tools/microbench.py linear 2>&1 | grep float16 | head -4 | tee $O/linear_st$us.txt
echo "== conv stagger $us"; DS_LIN_STAGGER_US=$us timeout 300 python tools/microbench.py conv 2>&1 | grep conv3x3 | head -2 | tee $O/conv_st$us.txt
done
