python -m pytest tests/test_gpu_models.py -x -q -k "attention or beit or video" 2>&1 | tail -3
python tools/microbench.py attention
