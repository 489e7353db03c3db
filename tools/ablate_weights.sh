# Cost of the task classes of level_weights_kernel: run a class TWICE (GG_WALK_EXPERIMENT=64 big, 32 small; idempotent, walks stay
# valid) and compare the kernel's total time with the plain run.   gpurun -- 'bash tools/ablate_weights.sh'
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for e in 0 32 64; do
GG_WALK_EXPERIMENT=$e rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_$e -o b -- python $R/bench.py --no-cpu-baseline --no-strict --fresh-batches 0 --overlap-steps 0 --steps 8 --warmup 2 > /dev/null 2>&1
echo "exp $e: $(grep -E 'level_weights|level_advance' $(find $R/gpurun_out/abl_$e -name 'b_kernel_stats.csv') | cut -d, -f1-4 | tr '\n' ' ')"
rm -rf $R/gpurun_out/abl_$e
done
