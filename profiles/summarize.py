#!/usr/bin/env python3
"""Turn rocprofv3 outputs (gpurun_out/, scratch) into the committed per-round summaries.

    python profiles/summarize.py <round-tag> <kernel_trace_dir> <fetch_dir> <write_dir> [<l2_dir> [<sq_dir> [<timeline_dir>]]]

Commands that produced the inputs (on the MI355X box, `cd /tmp && export TMPDIR=/tmp` first):
    rocprofv3 --kernel-trace --stats --output-format csv -d <kernel_trace_dir> -o b -- python bench.py --no-cpu-baseline
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <fetch_dir> -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <write_dir> -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass).  gfx950 correction
(MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128 B request of wide coalesced reads,
so read bytes = 2 * FETCH_SIZE KiB; WRITE_SIZE is taken as reported.
"""
import collections
import csv
import json
import os
import shutil
import sys


def pmc(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag, kt, fd, wd = sys.argv[1:5]
    here = os.path.dirname(os.path.abspath(__file__))
    shutil.copy(os.path.join(kt, "b_kernel_stats.csv"), os.path.join(here, "%s_kernel_stats.csv" % tag))
    fetch = pmc(os.path.join(fd, "f_counter_collection.csv"), "FETCH_SIZE")
    write = pmc(os.path.join(wd, "w_counter_collection.csv"), "WRITE_SIZE")
    out = {}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, []), write.get(name, [])
        if not f:
            continue
        fm = sum(f) / len(f)
        wm = sum(w) / len(w) if w else 0.0
        out[name] = {"launches_sampled": len(f), "FETCH_SIZE_KiB_mean": fm, "WRITE_SIZE_KiB_mean": wm,
                     "hbm_bytes_per_launch": (2.0 * fm + wm) * 1024.0}
    # the bench line printed by the FETCH_SIZE pass itself (tools/profile_round.sh): algorithmic bytes per launch of the SAME run
    same = os.path.join(os.path.dirname(os.path.normpath(fd)), "%s_fetch_bench.json" % tag)
    if os.path.isfile(same):
        try:
            line = [ln for ln in open(same).read().splitlines() if ln.startswith("{")][-1]
            r = json.loads(line)["roofline"]
            out["_same_run"] = {"kernel": r["kernel"], "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
                                "rows_per_launch": r["rows_per_launch"], "launches": r["launches"]}
        except (IndexError, KeyError, ValueError):
            pass
    with open(os.path.join(here, "%s_pmc_hbm_traffic.json" % tag), "w") as fjs:
        json.dump(out, fjs, indent=1, sort_keys=True)
    for k, v in sorted(((k, v) for k, v in out.items() if not k.startswith("_")), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])[:6]:
        print("%-60s %4d launches  %.1f MB/launch" % (k[:60], v["launches_sampled"], v["hbm_bytes_per_launch"] / 1e6))
    if len(sys.argv) > 5:  # optional: directory of the `--pmc TCC_HIT_sum TCC_MISS_sum` pass
        hit = pmc(os.path.join(sys.argv[5], "l_counter_collection.csv"), "TCC_HIT_sum")
        miss = pmc(os.path.join(sys.argv[5], "l_counter_collection.csv"), "TCC_MISS_sum")
        l2 = {}
        for name in sorted(hit):
            h, m = sum(hit[name]), sum(miss.get(name, []))
            if h + m > 0:
                l2[name] = {"l2_hit_rate": h / (h + m), "TCC_HIT_sum_mean": h / len(hit[name]), "TCC_MISS_sum_mean": m / len(hit[name]),
                            "launches_sampled": len(hit[name])}
        with open(os.path.join(here, "%s_pmc_l2_hit_rate.json" % tag), "w") as fjs:
            json.dump(l2, fjs, indent=1, sort_keys=True)
    if len(sys.argv) > 6:  # optional: directory of the SQ activity pass
        rows = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(os.path.join(sys.argv[6], "s_counter_collection.csv"))):
            rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        sq = {}
        for name, v in rows.items():
            if not v.get("SQ_WAVE_CYCLES"):
                continue
            m = {c: sum(x) / len(x) for c, x in v.items()}
            wc = m["SQ_WAVE_CYCLES"]
            if wc > 0:
                sq[name] = {"launches_sampled": len(v["SQ_WAVE_CYCLES"]), "SQ_WAVE_CYCLES": wc, "active_any": m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                            "active_valu": m.get("SQ_ACTIVE_INST_VALU", 0) / wc, "wait_inst_any": m.get("SQ_WAIT_INST_ANY", 0) / wc,
                            "wait_any": m.get("SQ_WAIT_ANY", 0) / wc, "SQ_INSTS_VALU": m.get("SQ_INSTS_VALU", 0)}
        with open(os.path.join(here, "%s_pmc_sq_activity.json" % tag), "w") as fjs:
            json.dump(sq, fjs, indent=1, sort_keys=True)
    if len(sys.argv) > 7:  # optional: directory of the timeline trace -> one step, kernels of both streams in start order
        ev = []
        for r in csv.DictReader(open(os.path.join(sys.argv[7], "b_kernel_trace.csv"))):
            name = r["Kernel_Name"].split("(")[0].replace("void gg::", "").replace("gg::", "")
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Queue_Id"]))
        ev.sort()
        idx = [i for i, e in enumerate(ev) if "walk_init_status" in e[2]]  # two walk launches per step
        best = None
        for k in range(len(idx) - 2, 4, -2):
            a, b = idx[k - 2], idx[k]
            if best is None or ev[b][0] - ev[a][0] < best[0]:
                best = (ev[b][0] - ev[a][0], a, b)
        if best:
            _, a, b = best
            while a > 0 and "walk_reset" in ev[a - 1][2]:
                a -= 1
            t0 = ev[a][0]
            out = ["One step of the default bench (rocprofv3 --kernel-trace; kernels of both streams in start order; q = HIP queue: the main",
                   "stream and the side stream that carries the G-mode walks).  Wall time of this step: %.0f us." % ((ev[b][0] - t0) / 1e3), "",
                   "%10s %9s  %-3s %s" % ("start us", "dur us", "q", "kernel")]
            out += ["%10.1f %9.1f  %-3s %s" % ((s0 - t0) / 1e3, (e0 - s0) / 1e3, q, n[:70]) for s0, e0, n, q in ev[a:b]]
            with open(os.path.join(here, "%s_step_timeline.txt" % tag), "w") as ft:
                ft.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
