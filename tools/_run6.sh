R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_steps.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $R/gpurun_out/r5f_tests.txt
GG_DET_PROFILE=1 python tools/strict_line.py 2>&1 | grep "\[det\]" | tail -4 > gpurun_out/r5f_detprof.txt
python tools/strict_line.py 2>/dev/null | tail -1 > gpurun_out/r5f_strict.txt
GG_DETERMINISTIC=0 python tools/strict_line.py 2>/dev/null | tail -1 > gpurun_out/r5f_strict_atomic.txt
cat gpurun_out/r5f_tests.txt gpurun_out/r5f_detprof.txt gpurun_out/r5f_strict.txt gpurun_out/r5f_strict_atomic.txt
