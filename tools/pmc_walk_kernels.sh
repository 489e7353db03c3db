R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
PB="--no-cpu-baseline --no-strict --fresh-batches 0 --overlap-steps 0 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pw_sq -o s -- python $R/bench.py $PB > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pw_f -o f -- python $R/bench.py $PB > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pw_w -o w -- python $R/bench.py $PB > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, collections, glob
def agg(path, names):
    out=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k=r["Kernel_Name"].split("(")[0]
        if r["Counter_Name"] in names: out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out
sq=agg(glob.glob("gpurun_out/pw_sq/**/s_counter_collection.csv", recursive=True)[0], ["SQ_WAVE_CYCLES","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_INSTS_VALU","SQ_WAIT_INST_ANY","SQ_WAIT_ANY"])
f=agg(glob.glob("gpurun_out/pw_f/**/f_counter_collection.csv", recursive=True)[0], ["FETCH_SIZE"])
w=agg(glob.glob("gpurun_out/pw_w/**/w_counter_collection.csv", recursive=True)[0], ["WRITE_SIZE"])
for k in sorted(sq):
    if not any(x in k for x in ("level_", "staged_opt", "pair_grad", "path_grad")): continue
    d=sq[k]; wc=sum(d["SQ_WAVE_CYCLES"]) or 1
    print("%-40s n=%4d any %.3f valu %.3f wait_inst %.3f wait_any %.3f | fetch %.1f MB write %.1f MB per launch" % (k[-40:], len(d["SQ_WAVE_CYCLES"]), sum(d["SQ_ACTIVE_INST_ANY"])/wc, sum(d["SQ_ACTIVE_INST_VALU"])/wc, sum(d["SQ_WAIT_INST_ANY"])/wc, sum(d["SQ_WAIT_ANY"])/wc,
          2*1024*sum(f[k]["FETCH_SIZE"])/max(len(f[k]["FETCH_SIZE"]),1)/1e6, 1024*sum(w[k]["WRITE_SIZE"])/max(len(w[k]["WRITE_SIZE"]),1)/1e6))
PY
rm -rf gpurun_out/pw_sq gpurun_out/pw_f gpurun_out/pw_w
