#!/bin/bash
# Round 3, GPU call 4: early prologue (next tile's DMAs before the epilogue), deeper ring of the ragged kernel.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call4
rm -rf $O; mkdir -p $O
cd $R
echo "== pytest (linear / conv / models)"; timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -k "linear or conv3x3 or beit_large_512 or dav2_vitl or dpt_beit_half or dav2_half or infer_batch or hip_graph or funnel" > $O/pytest_gpu.log 2>&1; grep -v MIOpen $O/pytest_gpu.log | tail -6
echo "== gemm A/B"; timeout 400 python tools/microbench.py gemms 2>&1 | grep -E "^gemm|residual_layernorm" | tee $O/microbench_gemms.txt
timeout 200 python tools/microbench.py conv 2>&1 | grep conv3x3 | cut -c1-200 | tee $O/microbench_conv.txt
DS_LIN_EARLY=0 timeout 200 python tools/microbench.py conv 2>&1 | grep conv3x3 | cut -c1-120 | sed 's/^/late /' | tee -a $O/microbench_conv.txt
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j['value'],1), 'pairs/s', round(j['ms_per_step'],3), 'ms/step', 'funnel', (j.get('funnel') or {}).get('value'))
PY
}
for e in 1 0 1 0; do DS_LIN_EARLY=$e timeout 300 python bench.py --no-cpu-baseline --no-funnel > $O/bench_early$e.json 2> $O/bench_early$e.err; show $O/bench_early$e.json; done
echo "== kernel trace"; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o step -- python $R/bench.py --no-cpu-baseline --no-funnel --steps 10 --warmup 2 > $O/prof_bench.json 2> $O/prof.err; cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; head -14 $O/kernel_stats.csv | cut -c1-130
rm -rf $O/prof
