#!/bin/bash
# First GPU call of the next round: test and time the two variants prepared (but never run on hardware) at the end of round 2.
#   gpurun --timeout 1200 -- 'bash tools/gpu_ab_prepared.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prepared
rm -rf $O; mkdir -p $O
cd $R
echo "== persistent head tail: value test"; DS_HEAD_PERSIST=1 timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -q -k "head" > $O/pytest_head_persist.log 2>&1; grep -v MIOpen $O/pytest_head_persist.log | tail -3
for v in 0 1 0 1; do echo "== head tail, DS_HEAD_PERSIST=$v"; DS_HEAD_PERSIST=$v timeout 120 python tools/microbench.py head 2>&1 | grep -i "head" | tee -a $O/head_$v.txt; done
echo "== fused projection (DS_LINEAR=proj): model value tests"; DS_LINEAR=proj timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -k "beit or dav2 or hybrid or forward" > $O/pytest_proj.log 2>&1; grep -v MIOpen $O/pytest_proj.log | tail -3
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j['value'],1), round(j['ms_per_step'],3))
PY
}
for cfg in "gelu 0" "proj 0" "gelu 1" "proj 1"; do set -- $cfg; DS_LINEAR=$1 DS_HEAD_PERSIST=$2 timeout 300 python bench.py --no-cpu-baseline > $O/bench_lin$1_head$2.json 2> $O/bench_lin$1_head$2.err; show $O/bench_lin$1_head$2.json; done
