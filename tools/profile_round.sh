#!/bin/bash
# One box session that produces every file of profiles/ for a round (run through gpurun from the repo root:
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh'), then on the host:
#   python profiles/summarize.py r<N> gpurun_out/r1e_kt gpurun_out/r1e_fetch gpurun_out/r1e_write
# GPU tests + smoke, rocprofv3 kernel trace of the default bench, separate PMC passes (FETCH_SIZE, WRITE_SIZE,
# TCC hit / miss; never combined with trace domains), un-profiled bench with the CPU baseline.
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $R/gpurun_out/r1e_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/r1e_smoke.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1e_kt -o b -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/r1e_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r1e_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r1e_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r1e_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r1e_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/r1e_l2 -o l -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r1e_l2.log 2>&1
cd $R
python bench.py > gpurun_out/r1e_bench.json 2> gpurun_out/r1e_bench.err
rm -f gpurun_out/r1e_kt/b_kernel_trace.csv.bak
cat gpurun_out/r1e_tests.txt; tail -1 gpurun_out/r1e_smoke.txt; cut -c1-400 gpurun_out/r1e_bench.json; ls -la gpurun_out/r1e_kt gpurun_out/r1e_fetch | head -20
