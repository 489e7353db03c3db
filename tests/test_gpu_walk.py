"""GPU parity: the HIP walk sampler (K1, through the C ABI) against the spec-arithmetic CPU
oracle (oracle/walk_oracle.c) on the same seeded inputs -- BIT-EXACT on samples, paths, path
lengths, root status and the in-place tree mutations (integer work: no tolerance)."""
import numpy as np
import pytest

from oracle import graphgan_oracle as orc
from tests.helpers import load_ca_grqc, load_small, star_graph_edges, ca_grqc_init_embeddings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ga():
    import graphgan_amd
    return graphgan_amd


def run_both(ga, n, rowptr, col, roots, E, b, rounds, seed, n_sample=20, stride=None, device_bfs=False):
    """D, G, D, G ... on engine and oracle; returns nothing, asserts equality."""
    roots = np.asarray(roots, dtype=np.int32)
    deg = (rowptr[1:] - rowptr[:-1]).astype(np.int32)
    eng = ga.Engine(E, E)
    eng.set_bias(0, b)
    eng.set_graph_csr(rowptr, col)
    eng.build_trees(roots, device=device_bfs)
    off, nbr, base, dmax = orc.c_build_trees(n, rowptr, col, roots)
    toff, tnbr, tbase = eng.get_trees()
    assert np.array_equal(toff, off) and np.array_equal(tnbr, nbr) and np.array_equal(tbase, base)
    assert eng.max_depth == dmax
    stride = stride or dmax + 3
    Ep = orc.pad_rows(E)
    slots = np.arange(len(roots), dtype=np.int32)
    nbr = nbr.copy()
    hops = 0
    prev_ctr = 0
    for r in range(rounds):
        for_d = r % 2 == 0
        nw = deg[roots] if for_d else np.full(len(roots), n_sample, dtype=np.int32)
        want = orc.c_walk_sample(Ep, b, off, nbr, base, roots, slots, nw, for_d, seed, r, stride)
        got = eng.walk_sample(slots, nw, for_d, seed, r, stride=stride)
        assert np.array_equal(got["root_status"], want["root_status"]), "round %d status" % r
        assert np.array_equal(got["path_len"], want["path_len"]), "round %d path_len" % r
        assert np.array_equal(got["samples"], want["samples"]), "round %d samples" % r
        m = np.arange(stride)[None, :] < want["path_len"][:, None]
        assert np.array_equal(got["paths"][m], want["paths"][m]), "round %d paths" % r
        _, tnbr, _ = eng.get_trees()
        assert np.array_equal(tnbr, nbr), "round %d tree mutation state (Q3)" % r
        hops += want["hops"]
        c = eng.counters()
        # G-mode: hop counters agree exactly.  D-mode: the reference stops a root at its first
        # aborting walk; the kernel runs all walks of a root concurrently and discards them.
        delta = c["hops"] - prev_ctr
        assert (delta == want["hops"]) if not for_d else (delta >= want["hops"])
        prev_ctr = c["hops"]
    eng.close()
    return hops


@pytest.fixture(params=["finisher", "hybrid2", "levels", "adaptive", "split", "share_all", "share_off"])
def walk_mode(request, monkeypatch):
    """The decompositions of the sampler: GG_WALK_LEVELS = 0 (one wavefront per walk, the
    finisher kernel alone), 2 (two hops through the level pipeline, the finisher takes over), 64 (level pipeline to
    the end; the default), and 64 with GG_FIN_THRESHOLD (the hand-over level follows the previous launch of the mode:
    the later rounds of a test switch to the finisher's walk list wherever fewer than that many walks were left); "split"
    runs the sync-free launches as two halves of their walks on two streams (opt-in, GG_WALK_SPLIT=1; from 512 walks here).
    "share_all" / "share_off": the level pipeline with the edge-score cache forced on for every node (GG_ES_MODE=2: a stale
    node's whole adjacency is scored once, every (root, node) distribution gathers) / switched off (0: every distribution
    scores its own candidates); the other level modes run the default policy (1), "split" without the cache.
    gg_create reads the variables, so they are set before the engine exists."""
    monkeypatch.setenv("GG_WALK_LEVELS", {"finisher": "0", "hybrid2": "2"}.get(request.param, "64"))
    if request.param in ("share_all", "share_off"):
        monkeypatch.setenv("GG_ES_MODE", "2" if request.param == "share_all" else "0")
    if request.param == "adaptive":
        monkeypatch.setenv("GG_FIN_THRESHOLD", "2000")
    if request.param == "split":  # two-half launches (two streams, per-half buffers and counters) already for small launches
        monkeypatch.setenv("GG_WALK_SPLIT", "1")
        monkeypatch.setenv("GG_WALK_SPLIT_MIN", "512")
    return request.param


@pytest.mark.parametrize("gi", [0, 1, 2, 3])
def test_small_graphs_bit_exact(ga, gi, walk_mode):
    g, n, graph = load_small(gi)
    rowptr, col = ga.graph_to_csr(n, graph)
    hops = run_both(ga, n, rowptr, col, np.arange(n), g["E"], g["b"], rounds=4, seed=1234 + gi)
    assert hops > 100


@pytest.mark.parametrize("d", [4, 50, 128, 200, 256])
def test_embedding_widths(ga, d):
    g, n, graph = load_small(1)
    rs = np.random.RandomState(d)
    E = (rs.randn(n, d) * (1.5 / np.sqrt(d))).astype(np.float32)
    b = (rs.randn(n) * 0.2).astype(np.float32)
    rowptr, col = ga.graph_to_csr(n, graph)
    run_both(ga, n, rowptr, col, np.arange(n), E, b, rounds=2, seed=99)


def test_ca_grqc_all_roots_bit_exact(ga, walk_mode):
    """BASELINE.json configs[1]: CA-GrQc, n_emb = 50, every root, D then G then D then G."""
    d, n, graph = load_ca_grqc()
    E = ca_grqc_init_embeddings(d, n).astype(np.float32)
    b = (np.random.RandomState(1).randn(n) * 0.05).astype(np.float32)
    rowptr, col = ga.graph_to_csr(n, graph)
    hops = run_both(ga, n, rowptr, col, np.arange(n), E, b, rounds=4, seed=2026)
    assert hops > 500000


@pytest.mark.parametrize("leaves", [70, 300, 1500])
def test_hub_lists_longer_than_one_pass(ga, leaves, walk_mode):
    """k > 64 (multi-block scan) and k > 1024.  In the "finisher" and "hybrid2" modes the per-walk kernel scores the
    1 501-candidate hub list itself: more than its 1 024 LDS score slots, so the HBM-scratch branch (sbuf_glb) runs;
    in "levels" mode the same list goes through the workgroup-per-hub weights path."""
    edges, n = star_graph_edges(leaves)
    rowptr, col = ga.edges_to_csr(n, edges)
    rs = np.random.RandomState(leaves)
    E = (rs.randn(n, 32) * 0.4).astype(np.float32)
    b = (rs.randn(n) * 0.5).astype(np.float32)
    roots = np.array([0, 1, 2, n - 1, n - 2], dtype=np.int32)
    run_both(ga, n, rowptr, col, roots, E, b, rounds=4, seed=5, n_sample=200)


def test_slot_order_and_batching_do_not_change_walks(ga):
    """Counter RNG keyed by (root id, walk, hop, stream): any slot order / batch split gives the same walks."""
    g, n, graph = load_small(3)
    rowptr, col = ga.graph_to_csr(n, graph)
    eng = ga.Engine(g["E"], g["E"])
    eng.set_bias(0, g["b"])
    eng.set_graph_csr(rowptr, col)
    roots = np.arange(n, dtype=np.int32)
    eng.build_trees(roots)
    nw = np.full(n, 20, dtype=np.int32)
    full = eng.walk_sample(np.arange(n), nw, False, 7, 3)
    perm = np.random.RandomState(0).permutation(n).astype(np.int32)
    a = eng.walk_sample(perm[: n // 2], nw[: n // 2], False, 7, 3, stride=full["paths"].shape[1])
    bb = eng.walk_sample(perm[n // 2:], nw[n // 2:], False, 7, 3, stride=full["paths"].shape[1])
    stride = full["paths"].shape[1]

    def masked(res):
        m = np.arange(stride)[None, :] < res["path_len"][:, None]
        return np.where(m, res["paths"], -1).reshape(-1, 20, stride), res["path_len"].reshape(-1, 20)

    pa, la = masked(a)
    pb, lb = masked(bb)
    pf, lf = masked(full)
    assert np.array_equal(np.concatenate([pa, pb]), pf[perm])
    assert np.array_equal(np.concatenate([la, lb]), lf[perm])
    eng.close()


def test_errors_are_codes_not_crashes(ga):
    g, n, graph = load_small(0)
    rowptr, col = ga.graph_to_csr(n, graph)
    eng = ga.Engine(g["E"], g["E"])
    with pytest.raises(ga.GraphGANHipError) as ei:  # no trees yet
        eng.walk_sample([0], [1], False, 0, 0, stride=8)
    assert ei.value.code == -1
    eng.set_graph_csr(rowptr, col)
    eng.build_trees(np.arange(n))
    with pytest.raises(ga.GraphGANHipError) as ei:  # slot out of range
        eng.walk_sample([n + 3], [1], False, 0, 0)
    assert ei.value.code == -1
    with pytest.raises(ga.GraphGANHipError) as ei:  # stride too small -> GG_ECAPACITY, not memory corruption
        eng.walk_sample(np.arange(n), np.full(n, 20), False, 0, 0, stride=2)
    assert ei.value.code == -2
    # empty launch is fine
    out = eng.walk_sample(np.zeros(0, np.int32), np.zeros(0, np.int32), False, 0, 0)
    assert len(out["samples"]) == 0
    eng.close()


@pytest.mark.parametrize("levels", ["0", "64"])
def test_non_finite_generator_scores_are_an_error_not_a_fault(ga, levels, monkeypatch):
    """A diverged generator (Inf / NaN rows) gives softmax weights that sum to 0: the sampler must report
    GG_EINVAL instead of indexing the candidate list with -1 (finisher) or sampling garbage (level pipeline)."""
    monkeypatch.setenv("GG_WALK_LEVELS", levels)
    g, n, graph = load_small(3)
    rowptr, col = ga.graph_to_csr(n, graph)
    E = g["E"].copy()
    E[:] = np.inf
    eng = ga.Engine(E, g["E"])
    eng.set_graph_csr(rowptr, col)
    eng.build_trees(np.arange(n, dtype=np.int32))
    with pytest.raises(ga.GraphGANHipError) as ei:
        eng.walk_sample(np.arange(n), np.full(n, 5), False, 1, 1)
    assert ei.value.code == ga.GG_EINVAL and "non-finite" in str(ei.value)
    eng.set_embeddings(0, g["E"])  # the context stays usable
    out = eng.walk_sample(np.arange(n), np.full(n, 5), False, 1, 1)
    assert (out["path_len"][out["root_status"].repeat(5) == 0] >= 3).all()
    eng.close()


@pytest.mark.parametrize("levels", ["0", "64"])
def test_non_finite_rows_behind_unscored_hops_are_an_error_too(ga, levels, monkeypatch):
    """Hops with ONE candidate (an only child; a leaf's back-step) score nothing -- the pick is certain -- so a non-finite row that
    walks reach only through such hops never meets a softmax.  The reference fails there (softmax([inf]) = [nan],
    np.random.choice raises, graph_gan.py:261-262): the engine reports GG_EINVAL through the finiteness flag the optimizer
    kernels and the table uploads maintain (gg_ctx::table_bad).  Graph: a pendant pair 0 - 1 beside a component that is
    scored normally; the walks from root 0 are 0 -> 1 (only child) -> 0 (leaf back-step)."""
    monkeypatch.setenv("GG_WALK_LEVELS", levels)
    edges = np.array([[0, 1], [2, 3], [3, 4], [4, 5], [2, 5], [3, 5]], dtype=np.int32)
    n, d = 6, 8
    rowptr, col = ga.edges_to_csr(n, edges)
    rs = np.random.RandomState(3)
    E = (rs.randn(n, d) * 0.3).astype(np.float32)
    bad = E.copy()
    bad[1, 3] = np.inf
    roots = np.arange(n, dtype=np.int32)
    eng = ga.Engine(bad, E, optimizer=ga.GG_OPT_ADAM_LAZY)
    eng.set_graph_csr(rowptr, col)
    eng.build_trees(roots)
    with pytest.raises(ga.GraphGANHipError) as ei:   # (a) uploaded non-finite row
        eng.walk_sample(roots, np.full(n, 4), False, 1, 1)
    assert ei.value.code == ga.GG_EINVAL and "non-finite" in str(ei.value)
    eng.set_embeddings(0, E)                          # a finite table clears the flag
    out = eng.walk_sample(roots, np.full(n, 4), False, 1, 1)
    assert (out["path_len"][:4] == 3).all() and (out["paths"][:4, :3] == [0, 1, 0]).all() and (out["samples"][:4] == 1).all()
    # (b) an optimizer step that diverges row 1: batch 64 path (fused update) and a large fused batch (staged / sparse optimizer)
    for B in (8, 40000):
        eng.set_embeddings(0, E)
        eng.set_bias(0, np.zeros(n, np.float32))       # (the diverged step below also leaves a NaN bias behind)
        eng.walk_sample(roots, np.full(n, 4), False, 1, 1)
        u, v = np.zeros(B, np.int32), np.ones(B, np.int32)
        eng.g_step(u, v, np.full(B, np.float32(np.inf)))   # dL/ds = -inf: m / sqrt(v) = inf / inf
        assert not np.isfinite(eng.get_embeddings(0)[1]).all()
        with pytest.raises(ga.GraphGANHipError) as ei:
            eng.walk_sample(roots, np.full(n, 4), False, 1, 1)
        assert ei.value.code == ga.GG_EINVAL
    # the discriminator's tables do not concern the walks
    eng.set_embeddings(0, E)
    eng.set_bias(0, np.zeros(n, np.float32))
    eng.set_embeddings(1, bad)
    eng.walk_sample(roots, np.full(n, 4), False, 1, 1)
    eng.close()


def test_speculative_level_buffers_overflow_is_retried(ga, monkeypatch):
    """The level pipeline sizes its score buffers from earlier launches and runs without host
    synchronisation; a launch that needs more raises a device flag and is rerun with exact sizing.
    Small launch first (learns a tiny capacity), then a 40x bigger one: results still bit-exact."""
    monkeypatch.setenv("GG_WALK_LEVELS", "64")
    g, n, graph = load_small(3)
    rowptr, col = ga.graph_to_csr(n, graph)
    roots = np.arange(n, dtype=np.int32)
    eng = ga.Engine(g["E"], g["E"])
    eng.set_bias(0, g["b"])
    eng.set_graph_csr(rowptr, col)
    eng.build_trees(roots)
    off, nbr, base, dmax = orc.c_build_trees(n, rowptr, col, roots)
    Ep = orc.pad_rows(g["E"])
    stride = dmax + 3
    for slots, nw in ((np.array([5], np.int32), np.array([2], np.int32)),
                      (np.arange(n, dtype=np.int32), np.full(n, 40, np.int32)),
                      (np.arange(n, dtype=np.int32), np.full(n, 40, np.int32))):
        want = orc.c_walk_sample(Ep, g["b"], off, nbr, base, roots, slots, nw, False, 3, 9, stride)
        got = eng.walk_sample(slots, nw, False, 3, 9, stride=stride)
        assert np.array_equal(got["path_len"], want["path_len"]) and np.array_equal(got["samples"], want["samples"])
        m = np.arange(stride)[None, :] < want["path_len"][:, None]
        assert np.array_equal(got["paths"][m], want["paths"][m])
    eng.close()


def dense_layers_edges(widths=(1, 120, 150, 90), seed=3):
    """Layered graph whose consecutive layers are COMPLETELY connected, edges in a shuffled file order: a BFS from the
    single top node finds every node of layer l + 1 from every node of layer l -- 120 x 150 = 18 000 edges into 150 new nodes
    inside two consecutive chunks of the GPU BFS's edge stream, i.e. thousands of in-chunk duplicates: the overflow branch of
    its duplicate list (resolved through the per-workgroup key array)."""
    ids, nxt = [], 0
    for w in widths:
        ids.append(np.arange(nxt, nxt + w))
        nxt += w
    edges = np.concatenate([np.stack(np.meshgrid(a, b, indexing="ij"), -1).reshape(-1, 2) for a, b in zip(ids[:-1], ids[1:])])
    edges = edges[np.random.RandomState(seed).permutation(len(edges))]
    return edges.astype(np.int32), nxt


@pytest.fixture(params=["lds_bitmap", "lds_bitmap_whole_levels", "lds_bitmap_sparse_levels_forced", "lds_bitmap_sparse_levels_forced_3_buckets", "global_bitmap",
                        "v1_lds_bitmap", "v1_global_bitmap"])
def bfs_bitmap(request, monkeypatch):
    """Both instances of the BFS kernel: the visited bitmap in LDS (graphs up to ~1.06 M nodes) and in global memory
    (GG_BFS_GLOBAL_BITMAP=1 forces what larger graphs -- BASELINE.json configs[4], 10^7 nodes -- take by themselves); both
    kernels: the scan / claim kernel of round 4 (default) and the chunk kernel of rounds 2-3 (GG_BFS_V1=1); and, for the default
    kernel with the LDS bitmap, its sparse-level path (a level popped through the candidate fathers of the unseen nodes only):
    as the heuristic takes it (large late levels), never (GG_BFS_SPARSE=0), and FORCED on every level of every tree
    (GG_BFS_SPARSE_K=0, GG_BFS_SPARSE_MIN=1), with 16 rank buckets and with 3."""
    env = {"GG_BFS_GLOBAL_BITMAP": "1" if "global" in request.param else None,
           "GG_BFS_V1": "1" if request.param.startswith("v1") else None,
           "GG_BFS_SPARSE": "0" if "whole_levels" in request.param else None,
           "GG_BFS_SPARSE_K": "0" if "forced" in request.param else None,
           "GG_BFS_SPARSE_MIN": "1" if "forced" in request.param else None,
           "GG_BFS_SPARSE_BUCKETS": "3" if "3_buckets" in request.param else None}
    for var, val in env.items():
        if val is not None:
            monkeypatch.setenv(var, val)
        else:
            monkeypatch.delenv(var, raising=False)
    return request.param


@pytest.mark.parametrize("case", ["small0", "small3", "star", "big_star", "ca_grqc", "powerlaw", "dense_layers", "grid", "two_components"])
def test_gpu_bfs_builds_the_reference_trees(ga, case, bfs_bitmap):
    """gg_build_trees_device == the host builder == reference construct_trees (pop order, child order,
    self-loops, isolated nodes, several components), offsets / lists / depth / longest list."""
    if case == "dense_layers":
        edges, n = dense_layers_edges()
        rowptr, col = ga.edges_to_csr(n, edges)
        roots = np.array([0, 1, 130, n - 1, 200], dtype=np.int32)
    elif case.startswith("small"):
        g, n, graph = load_small(int(case[-1]))
        rowptr, col = ga.graph_to_csr(n, graph)
        roots = np.arange(n, dtype=np.int32)
    elif case in ("star", "big_star"):
        # big_star: a hub whose 20 003 adjacency entries exceed one scan window (16 384 positions) -- scanned in segments -- and
        # are ALL new when the hub is the root: more candidates than claim threads, the window is rescanned shorter
        edges, n = star_graph_edges(300 if case == "star" else 20000)
        rowptr, col = ga.edges_to_csr(n, edges)
        roots = np.array([0, 1, 5, n - 1, 17], dtype=np.int32)
    elif case == "ca_grqc":
        d, n, graph = load_ca_grqc()
        rowptr, col = ga.graph_to_csr(n, graph)
        roots = np.arange(n, dtype=np.int32)
    elif case == "grid":
        # 120 x 100 lattice, edges shuffled: 218 levels of ~100 nodes -- long queues of tiny levels (forced sparse levels: 218 father
        # searches per tree, most unseen nodes several levels away from the level being popped: they stay on the list through every pass)
        W_, H_ = 120, 100
        idx = np.arange(W_ * H_).reshape(H_, W_)
        edges = np.concatenate([np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()], 1), np.stack([idx[:-1].ravel(), idx[1:].ravel()], 1)])
        edges = edges[np.random.RandomState(4).permutation(len(edges))].astype(np.int32)
        n = W_ * H_
        rowptr, col = ga.edges_to_csr(n, edges)
        roots = np.array([0, W_ - 1, n // 2 + 17, n - 1, 4242], dtype=np.int32)
    elif case == "two_components":
        # two power-law components of 30 000 and 20 000 nodes with interleaved ids + 500 isolated nodes: the unseen-node list of a sparse
        # level holds the OTHER component's nodes as well (they never hit; the search ends when only they are left), and when they
        # outnumber the scratch the level falls back to being popped whole
        na, nb_, iso = 30_000, 20_000, 500
        ea, eb = ga.synth_powerlaw(na, 10, 1, 2), ga.synth_powerlaw(nb_, 8, 3, 4)
        n = na + nb_ + iso
        perm = np.random.RandomState(9).permutation(n).astype(np.int32)
        edges = np.concatenate([perm[ea], perm[na + eb]]).astype(np.int32)
        rowptr, col = ga.edges_to_csr(n, edges)
        roots = np.concatenate([perm[:40], perm[na:na + 40], perm[na + nb_:na + nb_ + 3]]).astype(np.int32)
    else:
        n = 50_000
        rowptr, col = ga.edges_to_csr(n, ga.synth_powerlaw(n, 10, 1, 2))
        roots = np.random.RandomState(1).choice(n, 300, replace=False).astype(np.int32)
    E = np.zeros((n, 8), np.float32)
    eng = ga.Engine(E, E)
    eng.set_graph_csr(rowptr, col)
    eng.build_trees(roots, device=True)
    off, nbr, base = eng.get_trees()
    woff, wnbr, wbase, wdepth = ga.host_build_trees(n, rowptr, col, roots)
    assert np.array_equal(base, wbase) and np.array_equal(off, woff) and np.array_equal(nbr, wnbr)
    assert eng.max_depth == wdepth
    # BFS-order form + the edge each node was appended at (what the edge-score cache is indexed by): written by the BFS
    # kernel itself here, derived by tree_edges_kernel for the host-built trees -- both must name, for every non-root
    # rank, the FIRST occurrence of the node in its father's adjacency (graph_gan.py:101-107)
    tb, order, cstart, edge, valid = eng.get_tree_order()
    assert valid
    eng.build_trees(roots)  # host builder + derived edges
    tb2, order2, cstart2, edge2, valid2 = eng.get_tree_order()
    assert valid2 and np.array_equal(tb, tb2) and np.array_equal(order, order2) and np.array_equal(cstart, cstart2)
    assert np.array_equal(edge, edge2)
    row_of_edge = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    for r in range(min(len(roots), 40)):
        o, e = order[tb[r]:tb[r + 1]], edge[tb[r]:tb[r + 1]]
        cs = cstart[tb[r] + r: tb[r + 1] + r + 1]
        C = len(o)
        assert o[0] == roots[r] and e[0] == -1 and cs[0] == 1 and cs[C] == C
        if C == 1:
            continue
        father_rank = np.repeat(np.arange(C), np.diff(cs))          # father of ranks 1 .. C-1
        assert len(father_rank) == C - 1
        assert np.array_equal(col[e[1:]], o[1:])                      # the edge leads to the node ...
        assert np.array_equal(row_of_edge[e[1:]], o[father_rank])     # ... from its father
        for i in range(1, min(C, 200)):                               # first occurrence in the father's adjacency
            f = o[father_rank[i - 1]]
            adj = col[rowptr[f]:rowptr[f + 1]]
            assert rowptr[f] + int(np.flatnonzero(adj == o[i])[0]) == e[i]
    eng.close()


@pytest.mark.parametrize("case", ["small1", "powerlaw"])
def test_walks_on_device_built_trees_bit_exact(ga, case, bfs_bitmap):
    """The whole path -- trees from gg_build_trees_device (either bitmap instance), then D / G / D / G walks -- against the
    oracle's trees and walks (graph_gan.py:84-108,225-270): what a graph above the LDS bitmap's size runs."""
    if case == "small1":
        g, n, graph = load_small(1)
        rowptr, col = ga.graph_to_csr(n, graph)
        roots, E, b = np.arange(n, dtype=np.int32), g["E"], g["b"]
    else:
        n = 20_000
        rowptr, col = ga.edges_to_csr(n, ga.synth_powerlaw(n, 10, 1, 2))
        deg = rowptr[1:] - rowptr[:-1]
        rs = np.random.RandomState(2)
        roots = np.unique(np.concatenate([np.argsort(-deg)[:4], rs.choice(n, 60, replace=False)])).astype(np.int32)
        E = (rs.randn(n, 32) * 0.3).astype(np.float32)
        b = (rs.randn(n) * 0.1).astype(np.float32)
    hops = run_both(ga, n, rowptr, col, roots, E, b, rounds=4, seed=77, device_bfs=True)
    assert hops > 500


def test_split_launch_with_skewed_halves(ga, monkeypatch):
    """Two-half launches (GG_WALK_SPLIT=1) whose first half needs nearly ALL the chunks of a level: hub roots first, then
    many low-degree roots, D-mode walks = degree.  Each half may use up to the learned capacity `cap` of score chunks and a
    chunk descriptor is two int4: the halves' descriptor regions must be 2 * cap int4 apart (round-3 advisor finding: they
    were cap apart, so a half that used more than cap / 2 chunks overwrote the other half's descriptors)."""
    monkeypatch.setenv("GG_WALK_LEVELS", "64")
    monkeypatch.setenv("GG_WALK_SPLIT", "1")
    monkeypatch.setenv("GG_WALK_SPLIT_MIN", "512")
    n = 20_000
    rowptr, col = ga.edges_to_csr(n, ga.synth_powerlaw(n, 10, 1, 2))
    deg = rowptr[1:] - rowptr[:-1]
    by_deg = np.argsort(-deg, kind="stable")
    hubs, small = by_deg[:24], by_deg[-700:]
    assert deg[hubs].sum() > 0.8 * deg[small].sum()    # the first half of the D-mode walks is hub walks only
    roots = np.concatenate([hubs, small]).astype(np.int32)
    rs = np.random.RandomState(4)
    E = (rs.randn(n, 32) * 0.3).astype(np.float32)
    b = (rs.randn(n) * 0.1).astype(np.float32)
    hops = run_both(ga, n, rowptr, col, roots, E, b, rounds=6, seed=31)  # round 0 sized (learns cap), 1.. sync-free = split
    assert hops > 20000


def test_deep_chain_crosses_the_level_cap(ga):
    """A path graph whose embeddings pull every walk to the far end: trees up to 149 levels deep, paths of up to
    150 hops -- more than the 64 hops the level pipeline handles, so the per-walk finisher takes over the
    survivors mid-walk.  Same bit-exact comparison as everywhere else."""
    n = 150
    graph = {v: [u for u in (v - 1, v + 1) if 0 <= u < n] for v in range(n)}
    rowptr, col = ga.graph_to_csr(n, graph)
    E = np.zeros((n, 8), dtype=np.float32)
    E[:, 0] = np.sqrt(0.05) * np.arange(n)          # score(v, u) = 0.05 v u: the higher-numbered neighbour wins
    E[:, 1] = 0.01 * np.cos(np.arange(n))
    b = (0.1 * np.sin(np.arange(n))).astype(np.float32)
    hops = run_both(ga, n, rowptr, col, np.arange(n), E, b, rounds=4, seed=4242)
    assert hops > 64 * 150  # many walks ran past the level cap


def test_trees_in_the_reference_shape_round_trip_through_set_trees(ga):
    """gg_get_trees hands the resident trees over as the reference's lists (father entries removed by D-mode shown as
    -1); gg_set_trees takes such lists back.  A second engine fed that way samples the same walks, mutation state included."""
    g, n, graph = load_small(2)
    rowptr, col = ga.graph_to_csr(n, graph)
    deg = (rowptr[1:] - rowptr[:-1]).astype(np.int32)
    roots = np.arange(n, dtype=np.int32)
    a = ga.Engine(g["E"], g["E"])
    a.set_bias(0, g["b"])
    a.set_graph_csr(rowptr, col)
    a.build_trees(roots, device=True)
    a.walk_sample(roots, deg, True, 9, 0)              # D-mode: removes father entries (Q3)
    off, nbr, base = a.get_trees()
    assert (nbr < 0).any()
    b = ga.Engine(g["E"], g["E"])
    b.set_bias(0, g["b"])
    b.set_trees(roots, off, nbr, base)                # no graph needed for that
    assert b.max_depth == a.max_depth
    off2, nbr2, base2 = b.get_trees()
    assert np.array_equal(off, off2) and np.array_equal(nbr, nbr2) and np.array_equal(base, base2)
    for stream, for_d, nw in ((1, False, np.full(n, 20, np.int32)),):
        wa = a.walk_sample(roots, nw, for_d, 9, stream)
        wb = b.walk_sample(roots, nw, for_d, 9, stream)
        for k in ("samples", "path_len", "root_status"):
            assert np.array_equal(wa[k], wb[k]), k
        m = np.arange(wa["paths"].shape[1])[None, :] < wa["path_len"][:, None]  # entries behind a path's end are undefined
        assert np.array_equal(wa["paths"][m], wb["paths"][m])
    bad = nbr.copy()
    bad[base[3] + off[3, roots[3]] + 1:base[3] + off[3, roots[3]] + 2] = n + 5  # a child id out of range
    with pytest.raises(ga.GraphGANHipError):
        b.set_trees(roots, off, bad, base)
    a.close()
    b.close()
