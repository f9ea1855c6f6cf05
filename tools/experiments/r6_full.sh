mkdir -p gpurun_out
python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r6_gputest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gputest_full.log
tail -22 gpurun_out/r6_gputest_full.log
