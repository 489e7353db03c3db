R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_steps.py tests/test_gpu_walk.py tests/test_gpu_epoch.py -m gpu -q 2>&1 | tail -15 > $R/gpurun_out/r5b_tests.txt
python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "schedule" 2>&1 | grep -E "engine|seed-mean|passed|failed|Error|assert" | cut -c1-1500 > $R/gpurun_out/r5b_e2e.txt
python bench.py --no-cpu-baseline > gpurun_out/r5b_bench.json 2> gpurun_out/r5b_bench.err
cat gpurun_out/r5b_tests.txt; tail -5 gpurun_out/r5b_e2e.txt; tail -3 gpurun_out/r5b_bench.err
