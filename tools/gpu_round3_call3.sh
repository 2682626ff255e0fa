#!/bin/bash
# Round 3, GPU call 3: the LDS-staged ragged round -- GEMM tests, the GEMM A/B, the bench line, the kernel trace.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call3
rm -rf $O; mkdir -p $O
cd $R
echo "== pytest (linear / attention / models)"; timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -k "linear or conv3x3 or attention or beit_large_512 or dav2_vitl or dpt_beit_half or infer_batch or hip_graph" > $O/pytest_gpu.log 2>&1; grep -v MIOpen $O/pytest_gpu.log | tail -6
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "full_frame or funnel_failure or normalmap_reference" > $O/pytest_parity.log 2>&1; tail -4 $O/pytest_parity.log
echo "== gemm A/B"; timeout 300 python tools/microbench.py gemms 2>&1 | grep -E "^gemm|residual_layernorm" | tee $O/microbench_gemms.txt
echo "== attention"; timeout 200 python tools/microbench.py attention 2>&1 | grep "^attention" | tee $O/microbench_attention.txt
DS_ATT_ORDER=0 timeout 200 python tools/microbench.py attention 2>&1 | grep "4097" | sed 's/^/order0 /' | tee -a $O/microbench_attention.txt
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j['value'],1), 'pairs/s', round(j['ms_per_step'],3), 'ms/step', 'funnel', (j.get('funnel') or {}).get('value'))
PY
}
for cfg in all gelu all gelu; do DS_LINEAR=$cfg timeout 300 python bench.py --no-cpu-baseline --no-funnel > $O/bench_lin$cfg.json 2> $O/bench_lin$cfg.err; show $O/bench_lin$cfg.json; done
echo "== kernel trace"; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o step -- python $R/bench.py --no-cpu-baseline --no-funnel --steps 10 --warmup 2 > $O/prof_bench.json 2> $O/prof.err; cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; head -22 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/prof
