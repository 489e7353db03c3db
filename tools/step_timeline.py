#!/usr/bin/env python3
"""One step of the bench as a kernel timeline (both streams) from a rocprofv3 --kernel-trace CSV:
    python tools/step_timeline.py <b_kernel_trace.csv> <out.txt>
(the same condensation as profiles/summarize.py's optional timeline argument, usable on its own)."""
import csv
import sys


def main():
    src, dst = sys.argv[1:3]
    ev = []
    for r in csv.DictReader(open(src)):
        name = r["Kernel_Name"].split("(")[0].replace("void gg::", "").replace("gg::", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Queue_Id"]))
    ev.sort()
    idx = [i for i, e in enumerate(ev) if "walk_init_status" in e[2]]  # two walk launches per step
    best = None
    for k in range(len(idx) - 2, 4, -2):
        a, b = idx[k - 2], idx[k]
        if best is None or ev[b][0] - ev[a][0] < best[0]:
            best = (ev[b][0] - ev[a][0], a, b)
    _, a, b = best
    while a > 0 and "walk_reset" in ev[a - 1][2]:
        a -= 1
    t0 = ev[a][0]
    out = ["One step of the default bench (rocprofv3 --kernel-trace; kernels of both streams in start order; q = HIP queue: the main",
           "stream and the side stream that carries the G-mode walks).  Wall time of this step: %.0f us." % ((ev[b][0] - t0) / 1e3), "",
           "%10s %9s  %-3s %s" % ("start us", "dur us", "q", "kernel")]
    out += ["%10.1f %9.1f  %-3s %s" % ((s0 - t0) / 1e3, (e0 - s0) / 1e3, q, n[:70]) for s0, e0, n, q in ev[a:b]]
    # per-kernel totals inside the step
    tot = {}
    for s0, e0, n, q in ev[a:b]:
        k = n.split("<")[0]
        tot[k] = tot.get(k, [0, 0.0])
        tot[k][0] += 1
        tot[k][1] += (e0 - s0) / 1e3
    out += ["", "kernel totals inside this step (launches, us):"]
    out += ["  %-34s %4d %9.1f" % (k, v[0], v[1]) for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])]
    open(dst, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
