python -m pytest tests/test_gpu_multi.py tests/test_gpu_comm.py -x -q -m gpu 2>&1 | tail -6
