R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
GG_DETERMINISTIC=$m rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5c_kt$m -o s -- python $R/tools/strict_line.py > $R/gpurun_out/r5c_strict$m.txt 2>&1
rm -f $R/gpurun_out/r5c_kt$m/*/s_kernel_trace.csv $R/gpurun_out/r5c_kt$m/s_kernel_trace.csv
done
cd $R
python -m pytest tests/test_gpu_walk.py -m gpu -q -k "non_finite" 2>&1 | tail -3 > gpurun_out/r5c_tests.txt
cat gpurun_out/r5c_tests.txt; tail -1 gpurun_out/r5c_strict1.txt; tail -1 gpurun_out/r5c_strict0.txt
