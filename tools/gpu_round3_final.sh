#!/bin/bash
# Round 3, last GPU call: smoke + the whole GPU suite on the final state of the tree.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -v MIOpen $O/pytest_gpu.log | tail -4
