R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_steps.py tests/test_gpu_scale.py tests/test_gpu_epoch.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > $R/gpurun_out/r5e_tests.txt
python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "not schedule" 2>&1 | grep -E "passed|failed|Error|error" | tail -5 >> $R/gpurun_out/r5e_tests.txt
GG_DET_PROFILE=1 python tools/strict_line.py 2>&1 | grep "\[det\]" | tail -8 > gpurun_out/r5e_detprof.txt
PB="--no-cpu-baseline --no-strict --fresh-batches 0 --overlap-steps 0 --continuity-roots 0 --steps 20 --warmup 3"
for v in A B A B; do
if [ $v = A ]; then python bench.py $PB 2>/dev/null | cut -c1-200 >> gpurun_out/r5e_ab.txt; else GG_NO_EARLY_SLOTS=1 python bench.py $PB 2>/dev/null | cut -c1-200 >> gpurun_out/r5e_ab.txt; fi
done
cat gpurun_out/r5e_tests.txt gpurun_out/r5e_detprof.txt gpurun_out/r5e_ab.txt
