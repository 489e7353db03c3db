"""Synthetic benchmark workloads (BASELINE.json configs[2..4]; recipe: SURVEY.md section 8d) shared by
``bench.py`` and the parity tests, so that what is measured is exactly what is checked against the oracle."""
from __future__ import annotations

import numpy as np

from . import engine as _engine


def powerlaw_workload(n_node, m=10, n_emb=128, seed_graph=1, seed_perm=2, seed_emb=5):
    """Barabasi-Albert graph (m edges per new node, ids permuted) in the reference's adjacency order and
    N(0, sigma^2) "pre-trained" embeddings with the dot-product scale of the shipped CA-GrQc file.
    Returns (rowptr int64 [N+1], col int32, emb fp32 [N, d], n_edges)."""
    edges = _engine.synth_powerlaw(n_node, m, seed_graph, seed_perm)
    rowptr, col = _engine.edges_to_csr(n_node, edges)
    sigma = 0.6 * np.sqrt(50.0 / n_emb)
    rs = np.random.default_rng(seed_emb)
    emb = rs.standard_normal((n_node, n_emb), dtype=np.float32) * np.float32(sigma)
    return rowptr, col, emb, len(edges)


def powerlaw_split_workload(n_node, m=10, n_emb=128, test_frac=0.1, seed_graph=1, seed_perm=2, seed_split=3, seed_neg=4, seed_emb=5):
    """SURVEY.md section 8d, config 3 in full: the power-law graph with 10 % of its edges held out as the link-prediction
    test set (seed 3) and one negative per test edge drawn like the reference's ``generate_neg_links``
    (src/utils.py:96-128: a uniform node that is neither the edge's first endpoint nor one of its neighbours in
    train + test; seed 4).  The training adjacency keeps the file order of the remaining edges.
    Returns dict(rowptr, col, emb, n_train_edges, test [T, 2], test_neg [T, 2])."""
    edges = _engine.synth_powerlaw(n_node, m, seed_graph, seed_perm)
    n_edges = len(edges)
    held = np.zeros(n_edges, dtype=bool)
    held[np.random.RandomState(seed_split).permutation(n_edges)[: int(round(test_frac * n_edges))]] = True
    test = np.ascontiguousarray(edges[held])
    train = np.ascontiguousarray(edges[~held])
    rowptr, col = _engine.edges_to_csr(n_node, train)
    # neighbours over train + test, as sorted rows for the membership test of the rejection sampler
    frow, fcol = _engine.edges_to_csr(n_node, edges)
    key = np.sort(np.repeat(np.arange(n_node, dtype=np.int64), np.diff(frow)) * n_node + fcol.astype(np.int64))
    rs = np.random.RandomState(seed_neg)
    a = test[:, 0].astype(np.int64)
    neg = rs.randint(0, n_node, len(test)).astype(np.int64)
    for _ in range(64):  # uniform over the complement of {a} + neighbours(a): redraw the rejected ones
        k = a * n_node + neg
        pos = np.searchsorted(key, k)
        bad = (neg == a) | ((pos < len(key)) & (key[np.minimum(pos, len(key) - 1)] == k))
        if not bad.any():
            break
        neg[bad] = rs.randint(0, n_node, int(bad.sum()))
    test_neg = np.stack([a, neg], axis=1).astype(np.int32)
    sigma = 0.6 * np.sqrt(50.0 / n_emb)
    emb = np.random.default_rng(seed_emb).standard_normal((n_node, n_emb), dtype=np.float32) * np.float32(sigma)
    return dict(rowptr=rowptr, col=col, emb=emb, n_train_edges=len(train), test=test, test_neg=test_neg)


# roots per GPU and step of the default bench: 16 384 trees of the 1M-node graph = 197 GB of the 288 GB HBM (12 B per tree node)
BENCH_ROOTS = 16384
BENCH_ROOTS_ROUND2 = 8192  # rounds 1-2 ran 8 192 roots per step: bench.py reports that batch too (same graph, same kernels)


def bench_roots(rowptr, roots_per_rank, rank=0, world=1, seed=6):
    """The roots a rank walks per bench step: a seeded random sample of the nodes with train edges, this rank's
    contiguous share, longest (hub) roots first (LPT order for the walk scheduler)."""
    deg = rowptr[1:] - rowptr[:-1]
    cand = np.flatnonzero(deg > 0)
    R = min(int(roots_per_rank), len(cand) // max(world, 1))
    all_roots = np.random.RandomState(seed).permutation(cand)[: R * world].astype(np.int32)
    roots = np.ascontiguousarray(all_roots[rank * R:(rank + 1) * R])
    return roots[np.argsort(-deg[roots], kind="stable")]
