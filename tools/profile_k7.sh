#!/bin/bash
# The K7 block of tools/profile_round.sh alone: tools/profile_k7.sh r6  (all-pairs consumer on the matrix cores, configs[4] size:
# the un-profiled bench line, kernel stats, and the matrix pipe's busy cycles in their own PMC pass)
T=${1:-r6}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 300 python -m pytest tests/test_gpu_steps.py -x -q -m gpu -k "all_score" 2>&1 | tail -2
python $R/tools/allpairs_bench.py 10000000 256 4096 > $R/gpurun_out/${T}_k7_bench.json 2> $R/gpurun_out/${T}_k7.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_k7 -o a -- python $R/tools/allpairs_bench.py 10000000 256 4096 > $R/gpurun_out/${T}_k7_bench_under_rocprof.json 2>> $R/gpurun_out/${T}_k7.log
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
python -c "
import json; d=json.load(open('$R/gpurun_out/${T}_k7_bench.json')); print({k:(round(v['kernel_ms'],2), round(v['frac'],3)) for k,v in d.items() if isinstance(v,dict)})"
grep all_score $R/gpurun_out/${T}_k7/a_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
