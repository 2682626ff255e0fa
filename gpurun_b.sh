timeout 100 python -m pytest tests/test_gpu_parity.py -x -q -k "simple_mesh" 2>&1 | tail -6
