cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LAZY_TIME_MODES=1 timeout 1700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lazy -o lazy -- python tools/lazy_time.py 1000000 16384 3 - > gpurun_out/lazy_prof.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_lazy/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:28]:
    print(r["Name"][:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
# timeline of the last batch: gaps
t=glob.glob("gpurun_out/prof_lazy/**/*kernel_trace.csv", recursive=True)[0]
ks=sorted(csv.DictReader(open(t)), key=lambda r:int(r["Start_Timestamp"]))
# find the last lazy bfs kernel launch (3rd big batch)
idx=[i for i,k in enumerate(ks) if "bfs_order2_kernel" in k["Kernel_Name"] and "true>" in k["Kernel_Name"].replace(" ","")]
bf=[i for i,k in enumerate(ks) if "bfs_order2" in k["Kernel_Name"]]
print("bfs launches:", [(ks[i]["Kernel_Name"][-30:], (int(ks[i]["End_Timestamp"])-int(ks[i]["Start_Timestamp"]))/1e6) for i in bf])
# batch 3 = from third lazy-build launch to end
lazyb=[i for i in bf if (int(ks[i]["End_Timestamp"])-int(ks[i]["Start_Timestamp"]))>20e6]
s=lazyb[-1]
t0=int(ks[s]["Start_Timestamp"])
prev_end=t0
busy=0
for k in ks[s:]:
    st,en=int(k["Start_Timestamp"]),int(k["End_Timestamp"])
    gap=(st-prev_end)/1e3
    if gap>300: print("gap %.0f us before %s at %.1f ms"%(gap,k["Kernel_Name"][:50],(st-t0)/1e6))
    prev_end=max(prev_end,en)
print("batch span ms", (prev_end-t0)/1e6)
PY
