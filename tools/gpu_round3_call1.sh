#!/bin/bash
# Round 3, GPU call 1: the whole GPU suite (new parity tests included), the two variants prepared at the end of round 2
# (persistent head tail, projection GEMM with fused LayerScale + residual), and the default bench line.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round3_call1.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call1
rm -rf $O; mkdir -p $O
cd $R
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q -x --durations=12 > $O/pytest_gpu.log 2>&1; grep -v MIOpen $O/pytest_gpu.log | tail -22
echo "== persistent head tail: value test"; DS_HEAD_PERSIST=1 timeout 200 python -m pytest tests/test_gpu_models.py -m gpu -q -k "head_tail" > $O/pytest_head_persist.log 2>&1; grep -v MIOpen $O/pytest_head_persist.log | tail -3
for v in 0 1 0 1; do DS_HEAD_PERSIST=$v timeout 120 python tools/microbench.py head 2>&1 | grep -i "head" | sed "s/^/persist=$v /" | tee -a $O/head_ab.txt; done
echo "== fused projection (DS_LINEAR=proj): model value tests"; DS_LINEAR=proj timeout 400 python -m pytest tests/test_gpu_models.py -m gpu -q -k "beit_large_512 or dav2_vitl or dpt_beit_half or dav2_half" > $O/pytest_proj.log 2>&1; grep -v MIOpen $O/pytest_proj.log | tail -3
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j['value'],1), 'pairs/s', round(j['ms_per_step'],3), 'ms/step', 'funnel', (j.get('funnel') or {}).get('value'))
PY
}
for cfg in "gelu 0" "proj 0" "all 0" "gelu 1" "proj 1"; do set -- $cfg; DS_LINEAR=$1 DS_HEAD_PERSIST=$2 timeout 300 python bench.py --no-cpu-baseline --no-funnel > $O/bench_lin$1_head$2.json 2> $O/bench_lin$1_head$2.err; show $O/bench_lin$1_head$2.json; done
echo "== default bench line (funnel + cpu baseline)"; timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; show $O/bench_default.json
timeout 200 python tools/microbench.py linear 2>&1 | grep -v bfloat16 | tee $O/microbench_linear.txt | tail -8
