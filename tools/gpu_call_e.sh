#!/bin/bash
# Round-2 call E: attention rows-per-wave A/B, the whole gpu suite (new fused decoder kernels, read-out), bench c3 + funnel, c4.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/e
rm -rf $O; mkdir -p $O
cd $R
timeout 100 python tools/microbench.py attention 2>&1 | grep "attention \[" | sed "s/^/nqb=2 /" > $O/att_nqb.txt
DS_ATT_NQB=1 timeout 100 python tools/microbench.py attention 2>&1 | grep "attention \[" | sed "s/^/nqb=1 /" >> $O/att_nqb.txt; cat $O/att_nqb.txt
DS_ATT_NQB=1 timeout 200 python -m pytest tests/test_gpu_models.py -m gpu -q -k "attention" > $O/pytest_att_nqb1.log 2>&1; tail -3 $O/pytest_att_nqb1.log
timeout 900 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; tail -14 $O/pytest_gpu.log
timeout 300 python bench.py --funnel --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; python - <<PY
import json
j=json.load(open('$O/bench_c3.json')); print('c3', j['value'], j['ms_per_step']); print(j['roofline']['avg_kernel_ms'], j['roofline']['achieved']); print(j['roofline_stereo']['avg_kernel_ms'], j.get('funnel'))
PY
DS_FUNNEL_BATCH_PIXELS=16777216 timeout 200 python bench.py --funnel --no-cpu-baseline --steps 5 > $O/bench_c3_f16.json 2> $O/bench_c3_f16.err; python -c "import json;print(json.load(open('$O/bench_c3_f16.json')).get('funnel'))"
DS_ATT_NQB=1 timeout 200 python bench.py --no-cpu-baseline > $O/bench_c3_nqb1.json 2> $O/bench_c3_nqb1.err; cut -c1-160 $O/bench_c3_nqb1.json
timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 20 --funnel > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-200 $O/bench_c2.json; python -c "import json;print(json.load(open('$O/bench_c2.json')).get('funnel'))"
timeout 400 python bench.py --config c4 --steps 3 --warmup 1 > $O/bench_c4_1600.json 2> $O/bench_c4_1600.err; cut -c1-700 $O/bench_c4_1600.json; tail -2 $O/bench_c4_1600.err
ls $O
