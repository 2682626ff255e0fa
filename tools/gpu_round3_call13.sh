#!/bin/bash
# Round 3, GPU call 13: the streaming head-tail kernel (DS_HEAD_MODE=stream): value tests, microbenchmark against the tile and
# persistent kernels, and the c3 step with it.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call13
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "head_tail" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python tools/microbench.py head > $O/microbench_head.txt 2>&1; tail -4 $O/microbench_head.txt
show() { python - "$1" "$2" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[2], round(j['value'],1), j['unit'], round(j['ms_per_step'],3), 'ms/step')
PY
}
for m in persist stream; do
  DS_HEAD_MODE=$m DS_CUDNN_BENCHMARK=0 timeout 300 python bench.py --no-cpu-baseline --no-funnel --steps 20 > $O/c3_$m.json 2> $O/c3_$m.err; show $O/c3_$m.json "c3 head=$m"
done
DS_HEAD_MODE=stream DS_CUDNN_BENCHMARK=0 timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 20 > $O/c5_stream.json 2> $O/c5_stream.err; show $O/c5_stream.json "c5 head=stream"
