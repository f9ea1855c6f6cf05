mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r6_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_t1.log
tail -25 gpurun_out/r6_t1.log
bash tools/variants.sh run > gpurun_out/r6_variants1.log 2>&1; cat gpurun_out/r6_variants1.log
