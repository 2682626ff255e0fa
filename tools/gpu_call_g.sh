#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/g
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -q -k "graph or readout or bias_act" > $O/pytest_sel.log 2>&1; tail -5 $O/pytest_sel.log
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j['value'],1), round(j['ms_per_step'],3), j['config'].get('forward_launch'))
PY
}
timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 50 --warmup 5 > $O/bench_c2_graph.json 2> $O/bench_c2_graph.err; show $O/bench_c2_graph.json; tail -3 $O/bench_c2_graph.err
timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 50 --warmup 5 --no-graph > $O/bench_c2_eager.json 2> $O/bench_c2_eager.err; show $O/bench_c2_eager.json
timeout 200 python bench.py --no-cpu-baseline --graph > $O/bench_c3_graph.json 2> $O/bench_c3_graph.err; show $O/bench_c3_graph.json; tail -2 $O/bench_c3_graph.err
timeout 200 python bench.py --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; show $O/bench_c3.json
ls $O
