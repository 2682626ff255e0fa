#!/bin/bash
# Round 3, GPU call 11: decoder bias passes in-tree, funnel stage timing.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call11
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -k "beit or hybrid or zoedepth or funnel or hip_graph or infer_batch or leres" > $O/pytest.log 2>&1; grep -v MIOpen $O/pytest.log | tail -4
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "funnel" > $O/pytest2.log 2>&1; tail -2 $O/pytest2.log
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); f=j.get('funnel') or {}; print(sys.argv[1].split('/')[-1], round(j['value'],1), 'pairs/s', round(j['ms_per_step'],3), 'ms/step', 'funnel', f.get('value'), f.get('host_seconds'))
PY
}
for i in 1 2; do timeout 400 python bench.py --no-cpu-baseline > $O/bench_c3_$i.json 2> $O/bench_c3_$i.err; show $O/bench_c3_$i.json; done
timeout 300 python bench.py --config c2 --no-cpu-baseline --steps 20 > $O/bench_c2.json 2> $O/bench_c2.err; show $O/bench_c2.json
