#!/bin/bash
# Round 3, GPU call 15: streaming head-tail kernel, third version (producer reads prefetched, the finish under the MFMA stream): tests, microbenchmark,
# the c3 / c5 steps, then an -DDS_EXPERIMENTS build for the role ablations (1: no steady-state producer work, 2: no MFMAs).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call15
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "head_tail" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python tools/microbench.py head > $O/microbench_head.txt 2>&1; tail -3 $O/microbench_head.txt
show() { python - "$1" "$2" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[2], round(j['value'],1), j['unit'], round(j['ms_per_step'],3), 'ms/step')
PY
}
for m in persist stream; do
  DS_HEAD_MODE=$m DS_CUDNN_BENCHMARK=0 timeout 300 python bench.py --no-cpu-baseline --no-funnel --steps 20 > $O/c3_$m.json 2> $O/c3_$m.err; show $O/c3_$m.json "c3 head=$m"
done
for m in persist stream; do
  DS_HEAD_MODE=$m DS_CUDNN_BENCHMARK=0 timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 20 > $O/c5_$m.json 2> $O/c5_$m.err; show $O/c5_$m.json "c5 head=$m"
done
DS_EXPERIMENTS=1 timeout 400 python stable-diffusion-webui-depthmap-script_amd/build_native.py --force > $O/build_exp.log 2>&1; tail -1 $O/build_exp.log
for a in 0 1 2 3; do
  echo "DS_HEAD_ABLATE=$a" >> $O/ablate.txt
  DS_HEAD_ABLATE=$a timeout 200 python tools/microbench.py head 2>&1 | grep stream >> $O/ablate.txt
done
cat $O/ablate.txt
