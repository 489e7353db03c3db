#!/usr/bin/env python3
"""How much of the score kernel's work is shared between ROOTS?  Runs the D-mode and G-mode walks of one bench step,
fetches the paths and counts, per hop level: walks alive, distinct (root, node) distributions, distinct nodes, the sum of
graph degrees over both (= rows a per-(root, node) scorer may read at most / rows a per-node scorer would read).
    python tools/share_stats.py [--roots 8192]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphgan_amd as ga  # noqa: E402
from graphgan_amd import workloads  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--nodes", type=int, default=1_000_000)
p.add_argument("--roots", type=int, default=8192)
p.add_argument("--emb", type=int, default=128)
a = p.parse_args()
rowptr, col, emb, ne = workloads.powerlaw_workload(a.nodes, 10, a.emb)
deg = (rowptr[1:] - rowptr[:-1]).astype(np.int64)
eng = ga.Engine(emb, emb, optimizer=ga.GG_OPT_ADAM_LAZY)
eng.set_graph_csr(rowptr, col)
roots = workloads.bench_roots(rowptr, a.roots, 0, 1, 6)
eng.build_trees(roots, device=True)
slots = np.arange(len(roots), dtype=np.int32)
seen_step = np.zeros(a.nodes, bool)
for mode, for_d in (("D", True), ("G", False)):
    nw = deg[roots].astype(np.int32) if for_d else np.full(len(roots), 20, np.int32)
    c0 = eng.counters()
    w = eng.walk_sample(slots, nw, for_d, 6, 0 if for_d else 1)
    c1 = eng.counters()
    print("%s: walks %d hops %d rows_scored %d nbr_reads %d" % (mode, len(w["samples"]), c1["hops"] - c0["hops"], c1["rows_scored"] - c0["rows_scored"], c1["nbr_reads"] - c0["nbr_reads"]))
    slot_of_walk = np.repeat(np.arange(len(roots), dtype=np.int64), nw)
    paths, plen = w["paths"], w["path_len"]
    seen_launch = np.zeros(a.nodes, bool)
    for h in range(paths.shape[1] - 1):
        alive = plen > h + 1          # the walk samples a hop from position h
        if not alive.any():
            break
        cur = paths[alive, h].astype(np.int64)
        key = slot_of_walk[alive] * a.nodes + cur
        uk = np.unique(key)
        un = np.unique(cur)
        new_l = un[~seen_launch[un]]
        new_s = un[~seen_step[un]]
        print("  hop %2d: walks %7d  (root,node) %7d  nodes %7d | sum deg: per (root,node) %9d  per node %9d  new in launch %9d  new in step %9d" % (
            h, alive.sum(), len(uk), len(un), deg[uk % a.nodes].sum(), deg[un].sum(), deg[new_l].sum(), deg[new_s].sum()))
        seen_launch[un] = True
        seen_step[un] = True
eng.close()
