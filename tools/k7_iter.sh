#!/bin/bash
# one K7 iteration on the GPU box: the all-pairs tests, the bf16 bench at configs[4] size, SQ counters of the same
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_steps.py -x -q -m gpu -k "all_score" 2>&1 | tail -3
ALLPAIRS_ONLY=bf16 timeout 400 python tools/allpairs_bench.py > gpurun_out/k7_x32.json 2> gpurun_out/k7_x32.err
python -c "
import json; d=json.load(open('gpurun_out/k7_x32.json')); print({k:(round(v['kernel_ms'],2), round(v['frac'],3)) for k,v in d.items() if isinstance(v,dict)})"
if [ -n "$K7_PMC" ]; then
rm -rf gpurun_out/k7pmc
ALLPAIRS_ONLY=bf16 timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/k7pmc -o s -- python tools/allpairs_bench.py 10000000 256 4096 > /dev/null 2> gpurun_out/k7pmc.log
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/k7pmc/**/s_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "all_score_reduce" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: "%.3g" % (sum(x) / len(x)) for c, x in v.items()})
PY
rm -rf gpurun_out/k7pmc
fi
