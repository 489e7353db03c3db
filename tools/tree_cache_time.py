#!/usr/bin/env python3
"""Tree cache (gg_save_trees / gg_load_trees, the replacement of the reference's pickle, graph_gan.py:31-46): time to
rebuild the BFS trees on the GPU vs to read them back from the cache file.
    python tools/tree_cache_time.py [n_node] [roots] [dir]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphgan_amd as ga  # noqa: E402
from graphgan_amd import workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
out = sys.argv[3] if len(sys.argv) > 3 else "/tmp"
rowptr, col, emb, ne = workloads.powerlaw_workload(n, 10, 8)
roots = workloads.bench_roots(rowptr, R)
eng = ga.Engine(emb, emb, optimizer=ga.GG_OPT_SGD)
eng.set_graph_csr(rowptr, col)
path = os.path.join(out, "trees_%d_%d.ggtr" % (n, R))
for rep in range(2):
    t0 = time.time(); eng.build_trees(roots, device=True); t_build = time.time() - t0
    t0 = time.time(); eng.save_trees(path); t_save = time.time() - t0
    t0 = time.time(); eng.load_trees(path); t_load = time.time() - t0
    gb = os.path.getsize(path) / 1e9
    print("n=%d roots=%d: GPU BFS %.3f s | save %.3f s (%.2f GB, %.2f GB/s) | load %.3f s (%.2f GB/s)" % (n, R, t_build, t_save, gb, gb / t_save, t_load, gb / t_load), flush=True)
os.remove(path)
eng.close()
