timeout 100 python -m pytest tests/test_gpu_parity.py -x -q -k "simple_mesh or heatmap" 2>&1 | tail -8
