python -m pytest tests/test_gpu_models.py -x -q 2>&1 | tail -4
