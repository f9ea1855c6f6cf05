df -h /dev/shm | tail -1
timeout 900 python -m pytest tests/test_gpu_multi.py -q 2>&1 | tail -30
