R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_steps.py -m gpu -q -x 2>&1 | tail -4 > $R/gpurun_out/r5d_tests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5d_kt -o s -- python $R/tools/strict_line.py > $R/gpurun_out/r5d_strict.txt 2>&1
rm -f $R/gpurun_out/r5d_kt/*/s_kernel_trace.csv $R/gpurun_out/r5d_kt/s_kernel_trace.csv
cd $R
python tools/strict_line.py > gpurun_out/r5d_strict_noprof.txt 2>&1
cat gpurun_out/r5d_tests.txt; tail -1 gpurun_out/r5d_strict_noprof.txt
