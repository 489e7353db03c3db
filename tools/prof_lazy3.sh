cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in 0 16384 32768 65536 81920; do
GG_LZ_BUDGET=100000 GG_WALK_EXPERIMENT=$e LAZY_TIME_MODES=1 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lazy_$e -o lazy -- python tools/lazy_time.py 1000000 16384 2 - > gpurun_out/lazy_prof_$e.log 2>&1
grep resolve gpurun_out/prof_lazy_$e/lazy_kernel_stats.csv | cut -d, -f1-4 | sed "s/^/exp $e /"
done
