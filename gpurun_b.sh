python -m pytest tests/test_gpu_models.py -x -q -k "attention or beit or dinov2 or dav2" 2>&1 | tail -2
python tools/microbench.py attention
