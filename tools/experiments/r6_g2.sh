python -m pytest tests/test_gpu_gs.py tests/test_gpu_sort.py tests/test_gpu_fullsize.py tests/test_gpu_api.py tests/test_gpu_baseline_size.py -x -q -m gpu 2>&1 | tail -8
ST3R_DEBUG_FLAGS=4 bash tools/quick_bench.sh r6_flag4; bash tools/quick_bench.sh r6_seg
