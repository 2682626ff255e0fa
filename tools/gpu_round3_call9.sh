#!/bin/bash
# Round 3, GPU call 9: LDS-resident exact fallback (one workgroup per flagged row).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call9
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); rs=j.get('roofline_stereo', j['roofline']); print(sys.argv[1].split('/')[-1], round(j['value'],1), 'pairs/s', round(j['ms_per_step'],3), 'ms/step', 'stereo ms', round(rs['avg_kernel_ms'],4), 'exact ms', round(rs['exact_fallback_ms'],4), 'rows', rs['exact_fallback_rows'], 'general px', rs.get('general_pixels'))
PY
}
for i in 1 2 3; do timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 5 > $O/bench_c5_$i.json 2> $O/bench_c5_$i.err; show $O/bench_c5_$i.json; done
DS_PL_EXACT_GLOBAL=1 timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 5 > $O/bench_c5_global.json 2> $O/bench_c5_global.err; show $O/bench_c5_global.json
timeout 300 python bench.py --no-cpu-baseline --no-funnel > $O/bench_c3.json 2> $O/bench_c3.err; show $O/bench_c3.json
