#!/bin/bash
# Round 3, GPU call 16: streaming head-tail kernel, consumer schedule pinned (DS_HEAD_VARIANT=1) against the compiler's own order (0).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call16
rm -rf $O; mkdir -p $O
cd $R
for v in 0 1; do
  DS_HEAD_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "head_tail_stream" > $O/pytest_v$v.log 2>&1; tail -2 $O/pytest_v$v.log
  echo "variant $v" >> $O/microbench_head.txt
  DS_HEAD_VARIANT=$v timeout 300 python tools/microbench.py head 2>&1 | grep "dpt_head" >> $O/microbench_head.txt
done
cat $O/microbench_head.txt
DS_EXPERIMENTS=1 timeout 400 python stable-diffusion-webui-depthmap-script_amd/build_native.py --force > $O/build_exp.log 2>&1; tail -1 $O/build_exp.log
for v in 0 1; do for a in 1 2; do
  echo "variant $v DS_HEAD_ABLATE=$a" >> $O/ablate.txt
  DS_HEAD_VARIANT=$v DS_HEAD_ABLATE=$a timeout 200 python tools/microbench.py head 2>&1 | grep stream >> $O/ablate.txt
done; done
cat $O/ablate.txt
