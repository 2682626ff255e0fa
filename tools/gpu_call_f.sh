#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/f
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j['value'],1), round(j['ms_per_step'],2), 'attn', round(j['roofline']['avg_kernel_ms'],4), 'stereo', round(j['roofline_stereo']['avg_kernel_ms'],3), j['config'].get('overlap'))
PY
}
timeout 200 python bench.py --no-cpu-baseline > $O/bench_overlap.json 2> $O/bench_overlap.err; show $O/bench_overlap.json
timeout 200 python bench.py --no-cpu-baseline --no-overlap > $O/bench_nooverlap.json 2> $O/bench_nooverlap.err; show $O/bench_nooverlap.json
DS_CUDNN_BENCHMARK=1 timeout 400 python bench.py --no-cpu-baseline > $O/bench_cudnnbench.json 2> $O/bench_cudnnbench.err; show $O/bench_cudnnbench.json; tail -2 $O/bench_cudnnbench.err
DS_HEAD_RPW=2 timeout 200 python bench.py --no-cpu-baseline > $O/bench_rpw2.json 2> $O/bench_rpw2.err; show $O/bench_rpw2.json
python tools/microbench.py stereo head 2>&1 | grep -v amdgpu
DS_HEAD_RPW=2 python tools/microbench.py head 2>&1 | grep -v amdgpu
ls $O
