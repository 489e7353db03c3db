R=$GRAFT_REPO_ROOT
GG_DET_PROFILE=1 python tools/strict_line.py 2> gpurun_out/r5h_det_all.txt > /dev/null
(grep "model 0" gpurun_out/r5h_det_all.txt | tail -2; grep "model 1" gpurun_out/r5h_det_all.txt | tail -2) > gpurun_out/r5h_detprof.txt; rm gpurun_out/r5h_det_all.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5h_kt -o s -- python $R/tools/strict_line.py > /dev/null 2>&1
rm -f $R/gpurun_out/r5h_kt/*/s_kernel_trace.csv $R/gpurun_out/r5h_kt/s_kernel_trace.csv
cd $R; cat gpurun_out/r5h_detprof.txt; head -3 $(find gpurun_out/r5h_kt -name s_kernel_stats.csv) | cut -c1-200
