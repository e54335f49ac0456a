python -m pytest tests/test_gpu_ops.py tests/test_gpu_path.py -m gpu -q -k "two_streams or alternating" -p no:cacheprovider 2>&1 | tail -4
