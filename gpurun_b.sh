python -m pytest tests/test_gpu_models.py -x -q -k "head_tail" 2>&1 | tail -2
DS_HEAD_RPW=1 python tools/microbench.py head
DS_HEAD_RPW=2 python tools/microbench.py head
DS_HEAD_RPW=2 python -m pytest tests/test_gpu_models.py -x -q -k "head_tail" 2>&1 | tail -2
python tools/microbench.py attention rln upsample stereo
