#!/usr/bin/env python3
"""Root batches of an epoch with whole trees against lazy trees on the bench workload:
    python tools/lazy_time.py [n_node] [roots per batch] [batches] [out.json] [node cap]
Per batch gg_epoch_add (BFS + D-mode walks + rows + G-mode walks + pairs); reports wall time, BFS kernel time, walk kernel time,
the lazy statistics, and checks that both modes accumulate the same rows and pairs, bit for bit."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphgan_amd as ga  # noqa: E402
from graphgan_amd import workloads  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    out = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "-" else None
    cap = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    modes = [int(x) for x in os.environ.get("LAZY_TIME_MODES", "0,1").split(",")]
    rowptr, col, emb, ne = workloads.powerlaw_workload(n, 10, 128)
    roots = workloads.bench_roots(rowptr, R * B)
    roots = roots[np.random.RandomState(1).permutation(len(roots))]  # (bench order is hubs first: mix them over the batches)
    rec = {"workload": "power-law %d nodes / %d edges, d = 128; %d batches of %d roots" % (n, ne, B, R), "modes": {}}
    data = {}
    for mode in modes:
        eng = ga.Engine(emb, emb, optimizer=ga.GG_OPT_ADAM_LAZY)
        eng.set_tree_mode(mode, cap)
        eng.set_graph_csr(rowptr, col)
        eng.epoch_begin()
        per = []
        for b in range(B):
            c0 = eng.counters()
            t0 = time.time()
            eng.epoch_add(roots[b * R:(b + 1) * R], seed=7)
            eng.synchronize()
            dt = time.time() - t0
            c1 = eng.counters()
            st = eng.lazy_stats()
            per.append({"wall_s": dt, "bfs_kernel_ms": c1["bfs_kernel_ms"] - c0["bfs_kernel_ms"], "bfs_trees": c1["bfs_trees"] - c0["bfs_trees"],
                        "walk_kernel_ms": c1["walk_kernel_ms"] - c0["walk_kernel_ms"], "hops": c1["hops"] - c0["hops"], "walk_reruns": c1["walk_reruns"] - c0["walk_reruns"],
                        "lazy": st})
            print(mode, b, json.dumps(per[-1]), flush=True)
        eng.epoch_commit(1)
        eng.epoch_commit(0)
        data[mode] = eng.get_d_data() + eng.get_g_data()
        eng.close()
        rec["modes"]["lazy" if mode else "whole"] = per
    if len(data) == 2:
        same = all(np.array_equal(x, y) for x, y in zip(data[0], data[1]))
        rec["rows_and_pairs_identical"] = bool(same)
        rec["d_rows"], rec["g_pairs"] = int(len(data[0][0])), int(len(data[0][3]))
        assert same
    print(json.dumps(rec))
    if out:
        json.dump(rec, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
