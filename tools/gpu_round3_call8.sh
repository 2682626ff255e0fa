#!/bin/bash
# Round 3, GPU call 8: general-pixel pass at 4 waves per SIMD (+ queue-entry prefetch): every stereo parity test, stereo timings.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call8
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -q -k "preprocess" > $O/pytest_pre.log 2>&1; tail -2 $O/pytest_pre.log
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); rs=j.get('roofline_stereo', j['roofline']); print(sys.argv[1].split('/')[-1], round(j['value'],1), 'pairs/s', round(j['ms_per_step'],3), 'ms/step', 'funnel', (j.get('funnel') or {}).get('value'), 'stereo ms', round(rs['avg_kernel_ms'],4), 'general px', rs.get('general_pixels'))
PY
}
timeout 300 python bench.py --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; show $O/bench_c3.json
timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 5 > $O/bench_c5.json 2> $O/bench_c5.err; show $O/bench_c5.json
timeout 100 python bench.py --model none --no-cpu-baseline > $O/bench_none.json 2> $O/bench_none.err; show $O/bench_none.json
for ps in 4 8 16; do DS_PL_PER_SEG=$ps timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 5 > $O/bench_c5_ps$ps.json 2> $O/bench_c5_ps$ps.err; show $O/bench_c5_ps$ps.json; done
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $R/bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_c5.csv; rm -rf $O/prof; grep -E "polylines|Name" $O/kernel_stats_c5.csv | cut -c1-140
SECONDS=0; timeout 300 python bench.py --no-cpu-baseline --no-funnel > $O/t0.json 2> $O/t0.err; echo "default wall $SECONDS s"
SECONDS=0; DS_CUDNN_BENCHMARK=1 timeout 400 python bench.py --no-cpu-baseline --no-funnel > $O/t1.json 2> $O/t1.err; echo "MIOpen find wall $SECONDS s"; show $O/t1.json
