#!/bin/bash
# Round 3, GPU call 19: general-pixel pass specialised for one-word candidate masks: parity, kernel times on c5 / c3, microbenchmarks.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call19
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
show() { python - "$1" "$2" <<PY
import json,sys
j=json.load(open(sys.argv[1])); r=j.get('roofline_stereo') or {}
print(sys.argv[2], round(j['value'],1), j['unit'], round(j['ms_per_step'],3), 'ms/step', 'polylines main+general', r.get('avg_kernel_ms'), 'exact', r.get('exact_fallback_ms'))
PY
}
DS_CUDNN_BENCHMARK=0 timeout 300 python bench.py --no-cpu-baseline --no-funnel --steps 20 > $O/c3.json 2> $O/c3.err; show $O/c3.json "c3"
timeout 100 python bench.py --model none --no-cpu-baseline > $O/stereo_only.json 2> $O/stereo_only.err; show $O/stereo_only.json "stereo only"
cd /tmp && export TMPDIR=/tmp
DS_CUDNN_BENCHMARK=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python $R/bench.py --config c5 --no-cpu-baseline --steps 10 > $O/c5.json 2> $O/prof_c5.log
f=$(ls $O/prof_c5/*/*kernel_stats.csv 2>/dev/null | head -1); grep -i "polylines" $f | cut -c1-160 | tee $O/c5_polylines_kernels.txt; rm -rf $O/prof_c5
cd $R
show $O/c5.json "c5 (under the profiler)"
timeout 200 python tools/microbench.py > $O/microbench.txt 2>&1; grep -v amdgpu.ids $O/microbench.txt | cut -c1-160
