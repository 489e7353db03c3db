#!/usr/bin/env python3
"""What of a BFS tree do the walks of one step read?  (round 6, VERDICT item 1: measured before the lazy trees were built)
    python tools/tree_reads.py [n_node] [roots] [out.json]
Bench workload (power-law, d = 128), full trees of `roots` bench roots, one D-mode and one G-mode walk launch (deg(root) / 20
walks per root, graph_gan.py:191,210).  Per root: nodes per tree level, the level holding most nodes, the deepest level a walk
stands on, and the distinct (root, node) children lists the walks read per level -- against the nodes the BFS wrote."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphgan_amd as ga  # noqa: E402
from graphgan_amd import workloads  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    out = sys.argv[3] if len(sys.argv) > 3 else None
    rowptr, col, emb, ne = workloads.powerlaw_workload(n, 10, 128)
    roots = workloads.bench_roots(rowptr, workloads.BENCH_ROOTS)
    roots = roots[:: max(1, len(roots) // R)][:R]  # hubs first in bench order: a spread sample
    R = len(roots)
    deg = (rowptr[1:] - rowptr[:-1]).astype(np.int32)
    eng = ga.Engine(emb, emb, optimizer=ga.GG_OPT_SGD)
    eng.set_graph_csr(rowptr, col)
    eng.build_trees(roots, device=True)
    base, order, cstart, edge, _ = eng.get_tree_order()
    L = 16
    level_nodes = np.zeros((R, L), np.int64)
    level_deg = np.zeros((R, L), np.int64)  # adjacency entries of the level's nodes = what expanding it scans
    for r in range(R):
        C = int(base[r + 1] - base[r])
        cs = cstart[base[r] + r: base[r] + r + C + 1]
        od = order[base[r]: base[r] + C]
        lo, hi, lev = 0, 1, 0
        while lo < hi and lev < L:
            level_nodes[r, lev] = hi - lo
            level_deg[r, lev] = int(deg[od[lo:hi]].sum())
            lo, hi = hi, int(cs[hi])
            lev += 1
    slots = np.arange(R, dtype=np.int32)
    reads = np.zeros((2, R, L), np.int64)      # distinct (root, node) lists read at level l (the node a walk stands on when it samples)
    deepest = np.zeros((2, R), np.int64)
    hops = np.zeros(2, np.int64)
    for mode, (for_d, nw) in enumerate(((True, deg[roots]), (False, np.full(R, 20, np.int32)))):
        got = eng.walk_sample(slots, nw, for_d, 11, mode, stride=eng.max_depth + 3)
        paths, plen = got["paths"], got["path_len"]
        wroot = np.repeat(np.arange(R), nw)
        hops[mode] = int(np.maximum(plen - 1, 0).sum())
        for l in range(L):
            m = plen >= l + 2  # the walk samples a hop while standing on path[l]
            if not m.any():
                break
            key = np.unique(wroot[m].astype(np.int64) * n + paths[m, l])
            np.add.at(reads[mode, :, l], (key // n).astype(np.int64), 1)
            np.maximum.at(deepest[mode], wroot[m], l)
    big = level_nodes.argmax(1)
    written = level_nodes.sum(1)
    both = reads.sum(0)
    rec = {
        "workload": "power-law %d nodes / %d edges, %d of the %d bench roots (every %d-th in bench order)" % (n, ne, R, workloads.BENCH_ROOTS, max(1, workloads.BENCH_ROOTS // R)),
        "tree_nodes_written_per_root_mean": float(written.mean()),
        "level_nodes_mean": [float(x) for x in level_nodes.mean(0)],
        "level_adjacency_entries_mean": [float(x) for x in level_deg.mean(0)],
        "largest_level_histogram": {str(k): int(v) for k, v in zip(*np.unique(big, return_counts=True))},
        "largest_level_share_of_tree_mean": float((level_nodes.max(1) / written).mean()),
        "lists_read_per_root_mean": {"d_mode": float(reads[0].sum(1).mean()), "g_mode": float(reads[1].sum(1).mean())},
        "lists_read_per_level_mean": {"d_mode": [float(x) for x in reads[0].mean(0)], "g_mode": [float(x) for x in reads[1].mean(0)]},
        "lists_read_share_of_written": float(both.sum() / (2.0 * written.sum())),
        "deepest_level_stood_on_histogram": {"d_mode": {str(k): int(v) for k, v in zip(*np.unique(deepest[0], return_counts=True))},
                                             "g_mode": {str(k): int(v) for k, v in zip(*np.unique(deepest[1], return_counts=True))}},
        "deepest_minus_largest_level_histogram": {str(k): int(v) for k, v in zip(*np.unique(deepest.max(0) - big, return_counts=True))},
        "hops": {"d_mode": int(hops[0]), "g_mode": int(hops[1])},
        # what an exact BFS through the level BEFORE the largest one scans and appends (lazy trees: the rest is resolved on demand)
        "bfs_through_level_before_largest": {
            "adjacency_entries_scanned_mean": float(np.mean([level_deg[r, :max(big[r] - 1, 0)].sum() for r in range(R)])),
            "nodes_appended_mean": float(np.mean([level_nodes[r, :big[r]].sum() for r in range(R)])),
            "full_bfs_adjacency_entries": int(rowptr[-1]), "full_bfs_nodes": float(written.mean())},
        "lists_read_at_or_below_largest_level_mean": float(np.mean([both[r, big[r] - 1:].sum() for r in range(R)])),
    }
    eng.close()
    print(json.dumps(rec))
    if out:
        json.dump(rec, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
