#!/bin/bash
# Round 3, GPU call 7: fused preprocess kernel; routing of the token GEMMs at an unfavourable tile count (c5).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call7
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -k "preprocess or infer_batch or beit or dav2 or funnel or hybrid" > $O/pytest.log 2>&1; grep -v MIOpen $O/pytest.log | tail -4
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j['value'],1), 'pairs/s', round(j['ms_per_step'],3), 'ms/step', 'funnel', (j.get('funnel') or {}).get('value'))
PY
}
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline > $O/bench_c3_$i.json 2> $O/bench_c3_$i.err; show $O/bench_c3_$i.json; done
for mt in 96 512; do DS_LINEAR_MIN_TILES=$mt timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 5 > $O/bench_c5_mt$mt.json 2> $O/bench_c5_mt$mt.err; show $O/bench_c5_mt$mt.json; done
DS_LINEAR=gelu timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 5 > $O/bench_c5_gelu.json 2> $O/bench_c5_gelu.err; show $O/bench_c5_gelu.json
DS_LINEAR=proj timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 5 > $O/bench_c5_proj.json 2> $O/bench_c5_proj.err; show $O/bench_c5_proj.json
/usr/bin/time -f "wall %e s (default)" timeout 300 python bench.py --no-cpu-baseline --no-funnel > $O/t0.json 2> $O/t0.err; tail -1 $O/t0.err
DS_CUDNN_BENCHMARK=1 /usr/bin/time -f "wall %e s (MIOpen find)" timeout 400 python bench.py --no-cpu-baseline --no-funnel > $O/t1.json 2> $O/t1.err; tail -1 $O/t1.err; show $O/t1.json
