#!/bin/bash
# Round 3, GPU call 6: funnel (Pillow block cache, copy stream), ragged threshold A/B.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call6
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q -k "funnel or linear" > $O/pytest.log 2>&1; grep -v MIOpen $O/pytest.log | tail -4
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j['value'],1), 'pairs/s', round(j['ms_per_step'],3), 'ms/step', 'funnel', (j.get('funnel') or {}).get('value'))
PY
}
timeout 400 python bench.py --no-cpu-baseline > $O/bench_funnel.json 2> $O/bench_funnel.err; show $O/bench_funnel.json
DS_PIL_BLOCKS_MAX=0 timeout 400 python bench.py --no-cpu-baseline > $O/bench_funnel_noblocks.json 2> $O/bench_funnel_noblocks.err; show $O/bench_funnel_noblocks.json
DS_FUNNEL_BATCH_PIXELS=8388608 timeout 400 python bench.py --no-cpu-baseline > $O/bench_funnel_g8.json 2> $O/bench_funnel_g8.err; show $O/bench_funnel_g8.json
for d in 4 2 4 2; do DS_LIN_RAGGED_DEN=$d timeout 300 python bench.py --no-cpu-baseline --no-funnel > $O/bench_den$d.json 2> $O/bench_den$d.err; show $O/bench_den$d.json; done
DS_LIN_RAGGED_DEN=2 timeout 300 python tools/microbench.py gemms 2>&1 | grep -E "^gemm fc1|^gemm qk" | cut -c1-200 | tee $O/microbench_den2.txt
