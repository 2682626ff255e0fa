#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s
rm -rf $O; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_models.py -m gpu -q -k "conv3x3 or linear or graph" > $O/pytest_sel.log 2>&1; grep -v MIOpen $O/pytest_sel.log | tail -4
DS_LIN_ORDER=1 timeout 400 python -m pytest tests/test_gpu_models.py -m gpu -q -k "linear" > $O/pytest_o1.log 2>&1; grep -v MIOpen $O/pytest_o1.log | tail -2
for o in 0 1 0 1; do echo "== order $o"; DS_LIN_ORDER=$o DS_LIN_SHAPES=fc1+gelu,fc2,qk,proj timeout 200 python tools/microbench.py linear 2>&1 | grep float16 | head -4 | cut -c1-120 | tee -a $O/linear_o$o.txt; done
timeout 200 python tools/microbench.py conv 2>&1 | grep conv3x3 | tee $O/conv.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE; do timeout 150 rocprofv3 --pmc $c -d $O/pmc_$c -o a -- python $R/tools/microbench.py lin1 conv1 > $O/pmc_$c.log 2>&1; done
cd $R; python tools/pmc_summary.py $O/pmc_* --match k_linear256 2>&1 | grep -E "k_linear|hbm_|FETCH|WRITE"
find $O -name "*.db" -size +5M -delete
