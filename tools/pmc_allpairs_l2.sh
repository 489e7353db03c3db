# L2-side view of the wide bf16 all-pairs consumer (do its four wavefronts share the B tiles in the CU's vector cache?):
# requests that reach the L2 and bytes fetched from HBM per launch.  Own PMC passes (no trace domain).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/k7_l2 -o s -- python $R/tools/allpairs_bench.py 10000000 256 4096 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/k7_l2b -o s -- python $R/tools/allpairs_bench.py 10000000 256 4096 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, collections, glob
for d in ("k7_l2", "k7_l2b"):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/%s/**/s_counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "all_score_reduce_bf16" in k: out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in out.items():
        print(k[-50:], {c: (len(x), round(sum(x) / len(x) / 1e6, 2)) for c, x in v.items()}, "(millions per launch)")
PY
rm -rf gpurun_out/k7_l2 gpurun_out/k7_l2b
