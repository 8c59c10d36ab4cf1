timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "^>|^E|test_parity_gpu.py:[0-9]+|passed|failed" | head -20
