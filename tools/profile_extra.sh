#!/bin/bash
# Second box session of a round's profiles (gpurun --timeout 1500 -- 'bash tools/profile_extra.sh r5'): the all-roots epoch at 1M nodes,
# BASELINE configs[2] and the configs[4] size on one GPU, and the per-seed accuracies of the end-to-end float-parity tests.
T=${1:-r5}
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "schedule" 2>&1 | grep -E "^seed|^\.seed|schedule:|passed|failed" | sed 's/^\.//' | cut -c1-1200 > gpurun_out/${T}_e2e_parity.txt
python tools/epoch_1m.py 1000000 gpurun_out/${T}_epoch_1m.json > gpurun_out/${T}_epoch_1m.log 2>&1
python bench.py --nodes 100000 --roots 8192 --no-cpu-baseline --no-strict --fresh-batches 12 > gpurun_out/${T}_bench_100k.json 2> gpurun_out/${T}_bench_100k.err
python bench.py --nodes 10000000 --emb 256 --roots 512 --steps 5 --warmup 2 --no-cpu-baseline --no-strict --fresh-batches 1 --overlap-steps 0 > gpurun_out/${T}_bench_10m.json 2> gpurun_out/${T}_bench_10m.err
tail -3 gpurun_out/${T}_e2e_parity.txt | cut -c1-300; tail -2 gpurun_out/${T}_epoch_1m.log | cut -c1-300; cut -c1-160 gpurun_out/${T}_bench_100k.json; cut -c1-160 gpurun_out/${T}_bench_10m.json
