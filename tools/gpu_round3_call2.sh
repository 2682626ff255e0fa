#!/bin/bash
# Round 3, GPU call 2: the whole GPU suite with every block GEMM in-tree (ragged round, V^T epilogue, fused residuals),
# the GEMM A/B microbench, the bench line against round 2's routing, and a kernel trace of the step.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call2
rm -rf $O; mkdir -p $O
cd $R
echo "== pytest -m gpu"; timeout 1000 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; grep -v MIOpen $O/pytest_gpu.log | tail -16
echo "== gemm A/B"; timeout 300 python tools/microbench.py gemms 2>&1 | grep -E "^gemm|residual_layernorm" | tee $O/microbench_gemms.txt
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j['value'],1), 'pairs/s', round(j['ms_per_step'],3), 'ms/step', 'funnel', (j.get('funnel') or {}).get('value'))
PY
}
for cfg in all gelu all gelu; do DS_LINEAR=$cfg timeout 300 python bench.py --no-cpu-baseline --no-funnel > $O/bench_lin$cfg.json 2> $O/bench_lin$cfg.err; show $O/bench_lin$cfg.json; done
DS_LIN_RAGGED=0 timeout 300 python bench.py --no-cpu-baseline --no-funnel > $O/bench_noragged.json 2> $O/bench_noragged.err; show $O/bench_noragged.json
echo "== kernel trace"; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python $R/bench.py --no-cpu-baseline --no-funnel --steps 10 --warmup 2 > $O/prof_bench.json 2> $O/prof.err; cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; head -25 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/prof
