"""GPU parity of the LAZY trees (round 6; include/graphgan_hip.h "LAZY trees"): gg_build_trees_device builds a root's tree
exactly only through a level, the walks resolve deeper children lists on demand -- and every walk, path, status and every
resolved list must equal what the whole trees (construct_trees, graph_gan.py:84-108) give: bit-exact against the C oracle's
walks on the oracle's own whole trees, and list for list against the whole trees' BFS order."""
import numpy as np
import pytest

from oracle import graphgan_oracle as orc
from tests.helpers import load_ca_grqc, load_small, star_graph_edges, ca_grqc_init_embeddings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ga():
    import graphgan_amd
    return graphgan_amd


@pytest.fixture(params=["levels", "finisher", "hybrid2", "levels_global_bits", "levels_coop", "levels_coop_global_bits"])
def walk_mode(request, monkeypatch):
    """The decompositions of the sampler a lazy tree must serve: the level pipeline with the resolve kernel between the two
    halves of the advance kernel (visited words of the slot in LDS; "levels_global_bits": read from the index in global memory,
    the path of graphs above ~1.2 M nodes), the per-walk finisher alone (it resolves inline), and two levels + the finisher.
    "levels_coop*": every adjacency of more than 64 entries is resolved by the whole workgroup (default: more than 256 -- on
    these small graphs that path would hardly run)."""
    monkeypatch.setenv("GG_WALK_LEVELS", {"finisher": "0", "hybrid2": "2"}.get(request.param, "64"))
    if request.param.endswith("global_bits"):
        monkeypatch.setenv("GG_LZ_NO_LDS", "1")
    if "coop" in request.param:
        monkeypatch.setenv("GG_LZ_COOP_MIN", "64")
        monkeypatch.setenv("GG_LZ_STATS", "1")
    return request.param


def whole_order(ga, rowptr, col, roots, E):
    """BFS-order form of the whole trees (oracle-checked in test_gpu_walk.py): per slot (order, cstart)."""
    eng = ga.Engine(E, E)
    eng.set_tree_mode(0)
    eng.set_graph_csr(rowptr, col)
    eng.build_trees(roots, device=True)
    base, order, cstart, edge, _ = eng.get_tree_order()
    eng.close()
    out = []
    for r in range(len(roots)):
        C = int(base[r + 1] - base[r])
        out.append((order[base[r]: base[r] + C].copy(), cstart[base[r] + r: base[r] + r + C + 1].copy(), edge[base[r]: base[r] + C].copy()))
    return out


def check_lazy_arrays(eng, whole, n):
    """Every exact rank and every RESOLVED list of the resident lazy trees against the whole trees."""
    z = eng.get_lazy_trees()
    st = eng.lazy_stats()
    resolved = 0
    for r, (w_order, w_cs, w_edge) in enumerate(whole):
        lzs, P, L, seg = (int(x) for x in z["info"][r])
        b = int(z["base"][r])
        C = len(w_order)
        rank_of = np.full(n, -1, np.int64)
        rank_of[w_order] = np.arange(C)
        if lzs == P:  # a whole tree (small component, or rebuilt in the arena)
            assert P == C
            assert np.array_equal(z["order"][b: b + C], w_order)
            assert np.array_equal(z["cstart"][b: b + C + 1], w_cs)
            assert np.array_equal(z["edge"][b: b + C], w_edge)
            continue
        assert 1 <= lzs < P <= C
        assert np.array_equal(z["order"][b: b + P], w_order[:P]), "slot %d: exact ranks" % r
        assert np.array_equal(z["edge"][b: b + P], w_edge[:P])
        assert np.array_equal(z["cstart"][b: b + lzs + 1], w_cs[: lzs + 1]), "slot %d: built lists" % r
        assert int(w_cs[lzs]) == P  # the children of the last built rank end where the exact ranks end
        # resolved lists: a pair carries the build's stamp (low 12 bits, the same for the whole build)
        pairs = z["pair"][b + lzs: b + seg]
        stamps = pairs & np.uint64(0xFFF)
        cnt = (pairs >> np.uint64(12)) & np.uint64(0xFFFFF)
        live = np.flatnonzero((stamps != 0) & (cnt != 0xFFFFF))
        assert len(np.unique(stamps[live])) <= 1
        for i in live:
            rank = lzs + int(i)
            start, c = int(pairs[i] >> np.uint64(32)), int(cnt[i])
            node = int(z["order"][b + rank])
            wr = int(rank_of[node])
            want = w_order[w_cs[wr]: w_cs[wr + 1]]
            assert c == len(want), "slot %d node %d: %d children resolved, the BFS appends %d" % (r, node, c, len(want))
            assert np.array_equal(z["order"][b + start: b + start + c], want), "slot %d node %d" % (r, node)
            assert np.array_equal(z["edge"][b + start: b + start + c], w_edge[w_cs[wr]: w_cs[wr + 1]])
            resolved += 1
    return resolved, st


def run_lazy(ga, n, rowptr, col, roots, E, b, rounds, seed, node_cap, n_sample=20, expect_lazy=True, monkeypatch=None):
    roots = np.asarray(roots, dtype=np.int32)
    if monkeypatch is not None:  # room for every slot's whole tree: the batch stays lazy whatever the walks ask for
        monkeypatch.setenv("GG_LZ_ARENA", str(len(roots)))
    deg = (rowptr[1:] - rowptr[:-1]).astype(np.int32)
    whole = whole_order(ga, rowptr, col, roots, E)
    eng = ga.Engine(E, E)
    eng.set_bias(0, b)
    eng.set_tree_mode(1, node_cap)
    eng.set_graph_csr(rowptr, col)
    eng.build_trees(roots, device=True)
    st0 = eng.lazy_stats()
    assert st0["lazy"]
    if expect_lazy:
        assert st0["lazy_slots"] > 0
    off, nbr, base, dmax = orc.c_build_trees(n, rowptr, col, roots)
    Ep = orc.pad_rows(E)
    slots = np.arange(len(roots), dtype=np.int32)
    nbr = nbr.copy()
    hops = 0
    for r in range(rounds):
        for_d = r % 2 == 0
        nw = deg[roots] if for_d else np.full(len(roots), n_sample, dtype=np.int32)
        stride = dmax + 3
        want = orc.c_walk_sample(Ep, b, off, nbr, base, roots, slots, nw, for_d, seed, r, stride)
        got = eng.walk_sample(slots, nw, for_d, seed, r, stride=stride)
        assert np.array_equal(got["root_status"], want["root_status"]), "round %d status" % r
        assert np.array_equal(got["path_len"], want["path_len"]), "round %d path_len" % r
        assert np.array_equal(got["samples"], want["samples"]), "round %d samples" % r
        m = np.arange(stride)[None, :] < want["path_len"][:, None]
        assert np.array_equal(got["paths"][m], want["paths"][m]), "round %d paths" % r
        hops += want["hops"]
    resolved, st = check_lazy_arrays(eng, whole, n)
    eng.close()
    return hops, resolved, st


@pytest.mark.parametrize("gi,cap", [(0, 6), (0, 12), (1, 8), (1, 24), (2, 4), (3, 8), (3, 30)])
def test_small_graphs_lazy_bit_exact(ga, gi, cap, walk_mode, monkeypatch):
    g, n, graph = load_small(gi)
    rowptr, col = ga.graph_to_csr(n, graph)
    hops, resolved, st = run_lazy(ga, n, rowptr, col, np.arange(n), g["E"], g["b"], rounds=4, seed=4321 + gi, node_cap=cap, expect_lazy=False, monkeypatch=monkeypatch)
    assert hops > 100


@pytest.mark.parametrize("cap", [40, 300, 2000])
def test_ca_grqc_lazy_bit_exact(ga, cap, walk_mode, monkeypatch):
    """CA-GrQc (5 242 nodes, self-loops: the first-occurrence tests are live), every root; small node limits put the exact
    part at levels 1-4 of trees up to 17 levels deep: resolutions at all three depths, slots rebuilt whole, reruns."""
    d, n, graph = load_ca_grqc()
    E = ca_grqc_init_embeddings(d, n).astype(np.float32)
    b = (np.random.RandomState(1).randn(n) * 0.05).astype(np.float32)
    rowptr, col = ga.graph_to_csr(n, graph)
    hops, resolved, st = run_lazy(ga, n, rowptr, col, np.arange(n), E, b, rounds=4, seed=2026, node_cap=cap, monkeypatch=monkeypatch)
    assert hops > 500000
    assert resolved > 1000
    assert st["fallback_roots"] > 0  # deep components: their walks leave the two resolvable levels


def test_powerlaw_lazy_bit_exact(ga, walk_mode, monkeypatch):
    """A power-law graph (20 000 nodes, m = 5): the shape of the bench workload -- a level that holds most of the nodes behind
    three small ones.  Node limit 6 000: exact through level 2-3, walks resolve levels 3-5."""
    n = 20000
    edges = ga.synth_powerlaw(n, 5, 1, 2)
    rowptr, col = ga.edges_to_csr(n, edges)
    rs = np.random.RandomState(3)
    E = (rs.randn(n, 32) * 0.5).astype(np.float32)
    b = (rs.randn(n) * 0.1).astype(np.float32)
    roots = rs.permutation(n)[:300].astype(np.int32)
    hops, resolved, st = run_lazy(ga, n, rowptr, col, roots, E, b, rounds=4, seed=77, node_cap=6000, monkeypatch=monkeypatch)
    assert resolved > 3000
    assert st["lazy_slots"] + st["fallback_roots"] == len(roots) and st["lazy_slots"] > len(roots) // 2
    if "coop" in walk_mode:
        assert st["coop_lists"] > 100  # hubs' lists went through the workgroup path (and every one of them equals the whole tree's)


def test_lazy_arena_overflow_rebuilds_the_batch_whole(ga, monkeypatch):
    """More slots ask for their whole tree than the arena holds: the batch is rebuilt as whole trees, same walks."""
    monkeypatch.setenv("GG_LZ_ARENA", "1")
    d, n, graph = load_ca_grqc()
    E = ca_grqc_init_embeddings(d, n).astype(np.float32)
    b = np.zeros(n, np.float32)
    rowptr, col = ga.graph_to_csr(n, graph)
    roots = np.arange(0, n, 7, dtype=np.int32)
    deg = (rowptr[1:] - rowptr[:-1]).astype(np.int32)
    eng = ga.Engine(E, E)
    eng.set_tree_mode(1, 40)
    eng.set_graph_csr(rowptr, col)
    eng.build_trees(roots, device=True)
    assert eng.lazy_stats()["lazy"]
    off, nbr, base, dmax = orc.c_build_trees(n, rowptr, col, roots)
    slots = np.arange(len(roots), dtype=np.int32)
    want = orc.c_walk_sample(orc.pad_rows(E), b, off, nbr.copy(), base, roots, slots, deg[roots], True, 5, 0, dmax + 3)
    got = eng.walk_sample(slots, deg[roots], True, 5, 0, stride=dmax + 3)
    for k in ("samples", "path_len", "root_status"):
        assert np.array_equal(got[k], want[k]), k
    assert not eng.lazy_stats()["lazy"]  # the batch is whole now
    eng.close()


def test_lazy_pool_overflow_falls_back(ga, monkeypatch):
    """A pool too small for the lists the walks resolve: the slots get their whole trees, same walks."""
    monkeypatch.setenv("GG_LZ_POOL", "8")
    n = 5000
    edges = ga.synth_powerlaw(n, 5, 1, 2)
    rowptr, col = ga.edges_to_csr(n, edges)
    rs = np.random.RandomState(4)
    E = (rs.randn(n, 16) * 0.5).astype(np.float32)
    b = (rs.randn(n) * 0.1).astype(np.float32)
    roots = np.arange(0, n, 50, dtype=np.int32)
    hops, resolved, st = run_lazy(ga, n, rowptr, col, roots, E, b, rounds=2, seed=9, node_cap=600, monkeypatch=monkeypatch)
    assert st["fallback_roots"] > 0


def test_lazy_prepare_calls_equal_whole_trees(ga):
    """gg_prepare_d / gg_prepare_g (rows, pairs, rewards) on lazy trees = on whole trees, including the launches repeated for
    slots that were rebuilt whole and the paths they lengthen."""
    d, n, graph = load_ca_grqc()
    E = ca_grqc_init_embeddings(d, n).astype(np.float32)
    rowptr, col = ga.graph_to_csr(n, graph)
    roots = np.arange(n, dtype=np.int32)
    slots = np.arange(n, dtype=np.int32)
    out = []
    for mode, cap in ((0, 0), (1, 300)):
        eng = ga.Engine(E, E)
        eng.set_tree_mode(mode, cap)
        eng.set_graph_csr(rowptr, col)
        eng.build_trees(roots, device=True)
        res = []
        for it in range(2):
            res.append(eng.prepare_d(slots, 11 + it, 0))
            res.append(eng.prepare_g(slots, 20, 11 + it, 1))
        if mode:
            assert eng.lazy_stats()["fallback_roots"] > 0
        out.append(res)
        eng.close()
    for a, b in zip(*out):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_lazy_epoch_equals_whole_epoch(ga):
    """gg_epoch_add over root batches (epoch.hip) with lazy trees: the accumulated rows and pairs of the whole-tree epoch."""
    n = 20000
    edges = ga.synth_powerlaw(n, 5, 1, 2)
    rowptr, col = ga.edges_to_csr(n, edges)
    rs = np.random.RandomState(8)
    E = (rs.randn(n, 32) * 0.5).astype(np.float32)
    roots = rs.permutation(n)[:600].astype(np.int32)
    out = []
    for mode in (0, 1):
        eng = ga.Engine(E, E)
        eng.set_tree_mode(mode, 2000)
        eng.set_graph_csr(rowptr, col)
        for ep in range(2):  # the second epoch reads the Q3 bits the first one stored
            eng.epoch_begin()
            for i in range(0, len(roots), 200):
                eng.epoch_add(roots[i: i + 200], seed=3 + ep)
            eng.epoch_commit(1)
            eng.epoch_commit(0)
            out.append((mode, ep, eng.get_d_data(), eng.get_g_data()))
        eng.close()
    for ep in range(2):
        a = [o for o in out if o[0] == 0 and o[1] == ep][0]
        b = [o for o in out if o[0] == 1 and o[1] == ep][0]
        for x, y in zip(a[2] + a[3], b[2] + b[3]):
            assert np.array_equal(x, y)
