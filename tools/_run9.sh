R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_steps.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $R/gpurun_out/r5i_tests.txt
for w in 8 4 1; do GG_DET_WORKGROUPS=$w python tools/strict_line.py 2>/dev/null | tail -1 | cut -c1-140 >> gpurun_out/r5i_strict.txt; done
GG_DETERMINISTIC=0 python tools/strict_line.py 2>/dev/null | tail -1 | cut -c1-140 >> gpurun_out/r5i_strict.txt
cat gpurun_out/r5i_tests.txt gpurun_out/r5i_strict.txt
