cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in 0 4096 12288; do
GG_WALK_EXPERIMENT=$e LAZY_TIME_MODES=1 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lazy_$e -o lazy -- python tools/lazy_time.py 1000000 16384 2 - > gpurun_out/lazy_prof_$e.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_lazy_$e/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:4]:
    if "resolve" in r["Name"]: print("exp $e", r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6)
PY
done
