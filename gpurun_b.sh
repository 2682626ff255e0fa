timeout 500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "Warning\|warn\|^$\|TransformerEncoder" | tail -6
