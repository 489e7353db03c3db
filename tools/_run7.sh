R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_steps.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $R/gpurun_out/r5g_tests.txt
GG_DET_PROFILE=1 python tools/strict_line.py 2> gpurun_out/r5g_det_all.txt > /dev/null
(grep "model 0" gpurun_out/r5g_det_all.txt | tail -2; grep "model 1" gpurun_out/r5g_det_all.txt | tail -2) > gpurun_out/r5g_detprof.txt; rm gpurun_out/r5g_det_all.txt
python tools/strict_line.py 2>/dev/null | tail -1 > gpurun_out/r5g_strict.txt
python tools/strict_line.py 2>/dev/null | tail -1 >> gpurun_out/r5g_strict.txt
GG_DETERMINISTIC=0 python tools/strict_line.py 2>/dev/null | tail -1 > gpurun_out/r5g_strict_atomic.txt
cat gpurun_out/r5g_tests.txt gpurun_out/r5g_detprof.txt gpurun_out/r5g_strict.txt gpurun_out/r5g_strict_atomic.txt
