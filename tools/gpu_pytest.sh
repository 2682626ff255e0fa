#!/bin/bash
# the gpu-marked suite on an MI355X box:  gpurun --timeout 900 -- 'bash tools/gpu_pytest.sh [-k expr]'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pytest
rm -rf $O; mkdir -p $O
cd $R
timeout 800 python -m pytest tests -m gpu -q "$@" > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
