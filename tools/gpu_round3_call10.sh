#!/bin/bash
# Round 3, GPU call 10: general-pixel pass at 6 waves per SIMD (A/B), default line with MIOpen's search on.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call10
rm -rf $O; mkdir -p $O
cd $R
DS_PL_GEN_WPE=6 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "full_frame or golden or sweep or exact" > $O/pytest_wpe6.log 2>&1; tail -2 $O/pytest_wpe6.log
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); rs=j.get('roofline_stereo', j['roofline']); print(sys.argv[1].split('/')[-1], round(j['value'],1), 'pairs/s', round(j['ms_per_step'],3), 'ms/step', 'funnel', (j.get('funnel') or {}).get('value'), 'stereo ms', round(rs['avg_kernel_ms'],4), 'exact ms', round(rs['exact_fallback_ms'],4))
PY
}
for w in 4 6 4 6; do DS_PL_GEN_WPE=$w timeout 100 python bench.py --model none --no-cpu-baseline > $O/none_wpe$w.json 2> $O/none.err; show $O/none_wpe$w.json; done
for w in 4 6; do DS_PL_GEN_WPE=$w DS_CUDNN_BENCHMARK=0 timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 5 > $O/c5_wpe$w.json 2> $O/c5.err; show $O/c5_wpe$w.json; done
for w in 4 6; do DS_PL_GEN_WPE=$w DS_CUDNN_BENCHMARK=0 timeout 300 python bench.py --no-cpu-baseline --no-funnel > $O/c3_wpe$w.json 2> $O/c3.err; show $O/c3_wpe$w.json; done
SECONDS=0; timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default command wall $SECONDS s"; show $O/bench_default.json
