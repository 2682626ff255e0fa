#!/bin/bash
# Round-2 call B: attention kernel generations A/B (value-checked), then the whole gpu suite on the new default.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/b
rm -rf $O; mkdir -p $O
cd $R
DS_ATT_V1=1 timeout 120 python tools/microbench.py attention > $O/att_v1.txt 2>&1; grep attention $O/att_v1.txt
timeout 120 python tools/microbench.py attention > $O/att_v2.txt 2>&1; grep -v amdgpu.ids $O/att_v2.txt | tail -8
DS_ATT_SPLIT=1 timeout 120 python tools/microbench.py attention > $O/att_v2s.txt 2>&1; grep -v amdgpu.ids $O/att_v2s.txt | tail -8
timeout 200 python -m pytest tests/test_gpu_models.py -m gpu -q -k "attention" > $O/pytest_att.log 2>&1; tail -5 $O/pytest_att.log
DS_ATT_SPLIT=1 timeout 200 python -m pytest tests/test_gpu_models.py -m gpu -q -k "attention" > $O/pytest_att_split.log 2>&1; tail -5 $O/pytest_att_split.log
timeout 900 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.log 2>&1; tail -30 $O/pytest_gpu.log
timeout 200 python bench.py --no-cpu-baseline --steps 10 > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-200 $O/bench_c3.json
ls $O
