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


def bench_roots(rowptr, roots_per_rank, rank=0, world=1, seed=6):
    """The roots a rank walks per bench step: a seeded random sample of the nodes with train edges, this rank's
    contiguous share, longest (hub) roots first (LPT order for the walk scheduler)."""
    deg = rowptr[1:] - rowptr[:-1]
    cand = np.flatnonzero(deg > 0)
    R = min(int(roots_per_rank), len(cand) // max(world, 1))
    all_roots = np.random.RandomState(seed).permutation(cand)[: R * world].astype(np.int32)
    roots = np.ascontiguousarray(all_roots[rank * R:(rank + 1) * R])
    return roots[np.argsort(-deg[roots], kind="stable")]
