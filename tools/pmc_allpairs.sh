R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/k7_sq -o s -- python $R/tools/allpairs_bench.py 2000000 128 8192 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/k7_sq2 -o s -- python $R/tools/allpairs_bench.py 2000000 128 8192 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, collections, glob
for d in ("k7_sq","k7_sq2"):
    fs=glob.glob("gpurun_out/%s/**/s_counter_collection.csv"%d, recursive=True)
    if not fs: print(d,"no file"); continue
    out=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0]
        if "all_score_reduce" in k: out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in out.items():
        print(k[-60:], {c: (len(x), round(sum(x)/len(x)/1e6,2)) for c,x in v.items()})
PY
rm -rf gpurun_out/k7_sq gpurun_out/k7_sq2
