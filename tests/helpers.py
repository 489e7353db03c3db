"""Shared test helpers (fixtures -> graphs, oracle/engine drivers)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def graph_from_edges(train, test):
    graph, nodes = {}, set()
    for a, b in np.asarray(train).reshape(-1, 2).tolist():
        nodes.update((a, b))
        graph.setdefault(a, [])
        graph.setdefault(b, [])
        graph[a].append(b)
        graph[b].append(a)
    for a, b in np.asarray(test).reshape(-1, 2).tolist():
        nodes.update((a, b))
        graph.setdefault(a, [])
        graph.setdefault(b, [])
    return len(nodes), graph


def load_small(gi):
    g = np.load(os.path.join(GOLD, "ref_small_%d.npz" % gi))
    n, graph = graph_from_edges(g["train"], g["test"])
    return g, n, graph


def load_ca_grqc():
    d = np.load(os.path.join(GOLD, "ca_grqc.npz"))
    n, graph = graph_from_edges(d["train"], d["test"])
    return d, n, graph


def ca_grqc_init_embeddings(d, n, seed=0):
    """read_embeddings semantics (utils.py:57-67): shipped rows, U[0,1) for the missing ids."""
    emb = np.random.RandomState(seed).rand(n, d["emb_rows"].shape[1])
    emb[d["emb_ids"]] = d["emb_rows"].astype(np.float64)
    return emb


def star_graph_edges(n_leaves, extra_chain=3):
    """hub 0 with n_leaves leaves, each leaf i has a private child (so D-mode does not abort),
    plus a short chain hanging off the hub: exercises k > 64 and k > 1024 score passes."""
    edges = []
    nxt = 1 + n_leaves
    for i in range(1, n_leaves + 1):
        edges.append((0, i))
    for i in range(1, n_leaves + 1, 7):
        edges.append((i, nxt))
        nxt += 1
    prev = 0
    for _ in range(extra_chain):
        edges.append((prev, nxt))
        prev = nxt
        nxt += 1
    return np.array(edges, dtype=np.int32), nxt
