R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for e in 0 1 2 4 7; do
GG_WALK_EXPERIMENT=$e rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_$e -o b -- python $R/bench.py --no-cpu-baseline --no-strict --fresh-batches 0 --overlap-steps 0 --steps 6 --warmup 3 > /dev/null 2>&1
echo "exp $e: $(grep level_weights $(find $R/gpurun_out/abl_$e -name 'b_kernel_stats.csv') | cut -d, -f1-4)"
rm -rf $R/gpurun_out/abl_$e
done
