#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/l
rm -rf $O; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_models.py -m gpu -q -k "conv3x3 or linear" > $O/pytest_sel.log 2>&1; grep -v MIOpen $O/pytest_sel.log | tail -6
echo "== sweep"; timeout 200 python tools/microbench.py sweep > $O/sweep.txt 2>&1; grep "rounds=8" $O/sweep.txt
echo "== linear"; DS_LIN_SHAPES=fc1+gelu,fc2,qk,proj timeout 200 python tools/microbench.py linear > $O/linear.txt 2>&1; grep float16 $O/linear.txt | head -7
echo "== conv"; timeout 300 python tools/microbench.py conv > $O/conv.txt 2>&1; grep -v MIOpen $O/conv.txt | grep conv3x3
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j['value'],1), round(j['ms_per_step'],3), j['config'].get('forward_launch'))
PY
}
timeout 300 python bench.py --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; show $O/bench_c3.json
timeout 300 python bench.py --no-cpu-baseline --graph > $O/bench_c3_graph.json 2> $O/bench_c3_graph.err; show $O/bench_c3_graph.json
DS_CONV=0 DS_LINEAR=0 timeout 300 python bench.py --no-cpu-baseline > $O/bench_c3_old.json 2> $O/bench_c3_old.err; show $O/bench_c3_old.json
