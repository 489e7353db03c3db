#!/usr/bin/env python3
"""Time gg_build_trees_device on the bench workload: python tools/bfs_time.py [n_node] [roots] [emb]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphgan_amd as ga  # noqa: E402
from graphgan_amd import workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
d = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rowptr, col, emb, ne = workloads.powerlaw_workload(n, 10, d)
roots = workloads.bench_roots(rowptr, R)
eng = ga.Engine(emb, emb, optimizer=ga.GG_OPT_SGD)
eng.set_tree_mode(int(os.environ.get("BFS_TIME_MODE", "0")))  # 1: lazy trees
eng.set_graph_csr(rowptr, col)
for rep in range(3):
    c0 = eng.counters()
    t0 = time.time()
    eng.build_trees(roots, device=True)
    dt = time.time() - t0
    c1 = eng.counters()
    ms = c1["bfs_kernel_ms"] - c0["bfs_kernel_ms"]
    print("n=%d edges=%d roots=%d: call %.3f s, kernel %.1f ms = %.1f us/tree, depth %d" % (n, ne, len(roots), dt, ms, 1e3 * ms / len(roots), eng.max_depth), flush=True)
eng.close()
