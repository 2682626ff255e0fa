#!/bin/bash
# Round 3, GPU call 17: one source window per wave in the general-pixel pass of the polylines kernel: parity (every stereo test),
# then the c5 / c3 / stereo-only lines with the switch off and on.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call17
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.log 2>&1; tail -4 $O/pytest_parity.log
show() { python - "$1" "$2" <<PY
import json,sys
j=json.load(open(sys.argv[1])); r=j.get('roofline_stereo') or {}
print(sys.argv[2], round(j['value'],1), j['unit'], round(j['ms_per_step'],3), 'ms/step', 'stereo', r.get('ms_per_launch', r.get('avg_ms')))
PY
}
for sw in 0 1; do
  DS_PL_GEN_SHARED=$sw DS_CUDNN_BENCHMARK=0 timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 20 > $O/c5_$sw.json 2> $O/c5_$sw.err; show $O/c5_$sw.json "c5 shared=$sw"
  DS_PL_GEN_SHARED=$sw DS_CUDNN_BENCHMARK=0 timeout 300 python bench.py --no-cpu-baseline --no-funnel --steps 20 > $O/c3_$sw.json 2> $O/c3_$sw.err; show $O/c3_$sw.json "c3 shared=$sw"
done
cd /tmp && export TMPDIR=/tmp
for sw in 0 1; do
  DS_PL_GEN_SHARED=$sw DS_CUDNN_BENCHMARK=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5_$sw -- python $R/bench.py --config c5 --no-cpu-baseline --steps 10 > $O/prof_c5_$sw.log 2>&1
  f=$(ls $O/prof_c5_$sw/*/*kernel_stats.csv 2>/dev/null | head -1); echo "shared=$sw"; grep -i "polylines" $f | cut -c1-160
done
