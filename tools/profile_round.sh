#!/bin/bash
# One box session that produces the files of profiles/ for a round (run through gpurun from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r2 [quick]'), then on the host:
#   python profiles/summarize.py r2 gpurun_out/r2_kt gpurun_out/r2_fetch gpurun_out/r2_write [gpurun_out/r2_l2 [gpurun_out/r2_sq [gpurun_out/r2_tl]]]
# GPU tests + smoke, rocprofv3 kernel trace of the default bench, separate PMC passes (FETCH_SIZE, WRITE_SIZE,
# TCC hit / miss; never combined with trace domains), un-profiled bench with the CPU baseline.  "quick": no tests, no L2 pass.
T=${1:-r3}
Q=${2:-full}
R=$GRAFT_REPO_ROOT
if [ "$Q" != "quick" ]; then
  python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $R/gpurun_out/${T}_tests.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/${T}_smoke.txt 2>&1
fi
cd /tmp && export TMPDIR=/tmp
PB="--no-cpu-baseline --no-strict --fresh-batches 1 --overlap-steps 3 --continuity-roots 0"  # (the 3 extra steps measure the side-stream launches solo)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_kt -o b -- python $R/bench.py $PB > $R/gpurun_out/${T}_kt.log 2>&1
# (the bench line of the FETCH_SIZE pass is kept: its algorithmic bytes per launch are what the PMC traffic of the same run is compared with)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 $PB > $R/gpurun_out/${T}_fetch_bench.json 2> $R/gpurun_out/${T}_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_write -o w -- python $R/bench.py --steps 3 --warmup 1 $PB > $R/gpurun_out/${T}_write.log 2>&1
if [ "$Q" != "quick" ]; then
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/${T}_l2 -o l -- python $R/bench.py --steps 3 --warmup 1 $PB > $R/gpurun_out/${T}_l2.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/${T}_sq -o s -- python $R/bench.py --steps 3 --warmup 1 $PB > $R/gpurun_out/${T}_sq.log 2>&1
  # kernel timeline of a few steps (start / end of every kernel on both streams): summarize.py condenses one step
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${T}_tl -o b -- python $R/bench.py --no-cpu-baseline --no-strict --fresh-batches 0 --overlap-steps 0 --continuity-roots 0 --steps 6 --warmup 3 > $R/gpurun_out/${T}_tl.log 2>&1
fi
if [ "$Q" != "quick" ]; then
  # K7 (all-pairs consumer on the matrix cores, configs[4] size): kernel stats + the matrix pipe's busy cycles (own PMC pass)
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_k7 -o a -- python $R/tools/allpairs_bench.py 10000000 256 4096 > $R/gpurun_out/${T}_k7_bench.json 2> $R/gpurun_out/${T}_k7.log
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/${T}_k7pmc -o s -- python $R/tools/allpairs_bench.py 10000000 256 4096 > /dev/null 2> $R/gpurun_out/${T}_k7pmc.log
  python - <<PY
import csv, collections, glob, json
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/${T}_k7pmc/**/s_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "all_score_reduce" in k: out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: {"launches": len(x), "mean": sum(x) / len(x)} for c, x in v.items()} for k, v in out.items()}
json.dump({"what": "SQ counters of the all-pairs consumer kernels, tools/allpairs_bench.py 10000000 256 4096 (sum over the chip's SEs per launch; SQ_BUSY_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES count cycles, SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY quad-cycles)", "kernels": res}, open("$R/gpurun_out/${T}_k7_pmc.json", "w"), indent=1)
PY
  rm -rf $R/gpurun_out/${T}_k7pmc $R/gpurun_out/${T}_k7/a_kernel_trace.csv
fi
cd $R
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
# the raw traces are large: keep the summaries
rm -f gpurun_out/${T}_kt/b_kernel_trace.csv gpurun_out/${T}_kt/*.bak
cat gpurun_out/${T}_tests.txt 2>/dev/null; tail -1 gpurun_out/${T}_smoke.txt 2>/dev/null; cut -c1-300 gpurun_out/${T}_bench.json; ls gpurun_out/${T}_kt gpurun_out/${T}_fetch | head
