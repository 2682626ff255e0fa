#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/i
rm -rf $O; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_models.py -m gpu -q -k "conv3x3 or linear" > $O/pytest_sel.log 2>&1; grep -v MIOpen $O/pytest_sel.log | tail -25
timeout 300 python tools/microbench.py conv > $O/conv.txt 2>&1; grep -v MIOpen $O/conv.txt | tail -12
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j['value'],1), round(j['ms_per_step'],3))
PY
}
timeout 300 python bench.py --no-cpu-baseline > $O/bench_c3_new.json 2> $O/bench_c3_new.err; show $O/bench_c3_new.json
DS_CONV=0 DS_LINEAR=0 timeout 300 python bench.py --no-cpu-baseline > $O/bench_c3_old.json 2> $O/bench_c3_old.err; show $O/bench_c3_old.json
DS_CONV=1 DS_LINEAR=all timeout 300 python bench.py --no-cpu-baseline > $O/bench_c3_all.json 2> $O/bench_c3_all.err; show $O/bench_c3_all.json
DS_CONV=0 DS_LINEAR=gelu timeout 300 python bench.py --no-cpu-baseline > $O/bench_c3_gelu.json 2> $O/bench_c3_gelu.err; show $O/bench_c3_gelu.json
ls $O
