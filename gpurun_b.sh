timeout 400 python -m pytest tests/test_gpu_models.py -x -q -k "zoedepth" 2>&1 | grep -v "Warning\|warn\|^$" | tail -15
