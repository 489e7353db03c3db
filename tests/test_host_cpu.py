"""CPU-side checks of the product: the C-ABI library loads and exports every symbol that
include/graphgan_hip.h declares, host-only entry points (tree builder, synthetic graphs)
match the reference-derived fixtures, and the engine FAILS LOUDLY without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.helpers import GOLD, load_ca_grqc, load_small

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ga():
    so = os.path.join(ROOT, "graphgan_amd", "libgraphgan_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    import graphgan_amd
    return graphgan_amd


def test_library_exports_every_declared_symbol(ga):
    hdr = open(os.path.join(ROOT, "include", "graphgan_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    from graphgan_amd import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert raw.gg_abi_version() == _lib.ABI_VERSION == _lib.header_abi_version()


def test_graft_entry_build_runs_end_to_end(ga):
    """The driver's "does it build" check: make (a no-op when everything is up to date), import, and the three ABI
    numbers -- header, binding, library -- agree.  Round 3's entry point asserted a stale literal and raised."""
    import __graft_entry__
    __graft_entry__.build()
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert not re.search(r"gg_abi_version\(\)\s*==\s*\d", src), "compare against the header, not a literal"


def test_no_gpu_means_loud_failure(ga):
    try:
        import subprocess
        has_gpu = subprocess.run(["bash", "-c", "test -e /dev/kfd"]).returncode == 0
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(ga.GraphGANHipError) as ei:
        ga.Engine(np.zeros((4, 8), np.float32), np.zeros((4, 8), np.float32))
    assert ei.value.code == -3 and "no CPU fallback" in str(ei.value)


@pytest.mark.parametrize("gi", [0, 1, 2, 3])
def test_host_tree_builder_small(ga, gi):
    g, n, graph = load_small(gi)
    rowptr, col = ga.graph_to_csr(n, graph)
    for threads in (1, 3):
        off, nbr, base, dmax = ga.host_build_trees(n, rowptr, col, np.arange(n), n_threads=threads)
        assert np.array_equal(off, g["tree_off"]) and np.array_equal(nbr, g["tree_nbr"]) and np.array_equal(base, g["tree_base"])
        assert dmax >= 1


def test_host_tree_builder_ca_grqc(ga):
    d, n, graph = load_ca_grqc()
    g = np.load(os.path.join(GOLD, "ref_ca_grqc.npz"))
    rowptr, col = ga.graph_to_csr(n, graph)
    off, nbr, base, dmax = ga.host_build_trees(n, rowptr, col, g["roots"], n_threads=4)
    assert np.array_equal(off, g["tree_off"]) and np.array_equal(nbr, g["tree_nbr"]) and np.array_equal(base, g["tree_base"])
    # edges_to_csr (vectorised) == dict-based adjacency
    r2, c2 = ga.edges_to_csr(n, d["train"])
    assert np.array_equal(r2, rowptr) and np.array_equal(c2, col)


def test_host_tree_builder_errors(ga):
    from graphgan_amd._lib import lib
    rowptr = np.array([0, 1, 2], dtype=np.int64)
    col = np.array([1, 0], dtype=np.int32)
    roots = np.array([5], dtype=np.int32)
    base = np.zeros(2, dtype=np.int64)
    rc = lib.gg_host_build_trees(2, rowptr.ctypes.data, col.ctypes.data, roots.ctypes.data, 1, None, None, base.ctypes.data, 0, 1, None)
    assert rc == -1 and b"out of range" in lib.gg_last_error(None)


def test_synth_powerlaw(ga):
    e = ga.synth_powerlaw(20000, 10, 1, 2)
    assert e.shape == (19990 * 10, 2) and e.min() == 0 and e.max() == 19999
    assert np.array_equal(e, ga.synth_powerlaw(20000, 10, 1, 2))
    deg = np.bincount(e.ravel(), minlength=20000)
    assert deg.min() >= 10 and deg.max() > 300  # heavy tail
    assert not np.any(e[:, 0] == e[:, 1])
    u = np.unique(np.sort(e, 1), axis=0)
    assert len(u) == len(e)  # simple graph


def test_native_embedding_writer_is_byte_identical(ga, tmp_path):
    """gg_host_write_embeddings == the reference's write_embeddings_to_file text (graph_gan.py:293-306:
    hstack promotes fp32 -> fp64, str(x) per value), including the notation switches of repr()."""
    rs = np.random.RandomState(0)
    special = np.array([0.0, -0.0, 1.0, -1.0, 123.0, 1e-4, 9.999e-5, 1e-5, 1.5e-5, 1e15, 1e16, 1.2345e20, 3.4e38, 1e-38, 1.4e-45,
                        0.1, 0.5, 2.5, 100000.0, 16777216.0, 0.22650299966335297, -0.016766, 65504.0, 1e-7, 123456.789], dtype=np.float32)
    emb = np.concatenate([special, (rs.randn(2975) * 10.0 ** rs.randint(-8, 8, 2975)).astype(np.float32)]).reshape(-1, 50)
    emb = np.concatenate([emb, rs.randn(5000, 50).astype(np.float32)])
    n, d = emb.shape
    want = [str(n) + "\t" + str(d) + "\n"]
    mat = np.hstack([np.array(range(n)).reshape(-1, 1), emb]).tolist()
    want += [str(int(r[0])) + "\t" + "\t".join(str(x) for x in r[1:]) + "\n" for r in mat]
    for threads in (1, 5):
        path = tmp_path / ("emb%d.txt" % threads)
        ga.host_write_embeddings(path, emb, n_threads=threads)
        assert open(path).read() == "".join(want)


def test_native_edge_ingest_matches_read_edges(ga, tmp_path):
    """gg_host_read_edges == utils.read_edges semantics (utils.py:12-54): list order, both directions,
    self-loops listed twice, test-only nodes with empty lists, n_node = number of distinct ids."""
    from graphgan_amd import utils
    from tests.helpers import load_ca_grqc
    d, n, graph = load_ca_grqc()
    tr, te = tmp_path / "train.txt", tmp_path / "test.txt"
    with open(tr, "w") as f:
        f.writelines("%d\t%d\n" % (a, b) for a, b in d["train"].tolist())
    with open(te, "w") as f:
        f.writelines("%d %d\r\n" % (a, b) for a, b in d["test"].tolist())  # other separators / line ends
    n2, rowptr, col = ga.read_edges_csr(tr, te)
    n3, graph3 = utils.read_edges(str(tr), str(te))
    want_rowptr, want_col = ga.graph_to_csr(n3, graph3)
    assert n2 == n3 == n == 5242
    assert np.array_equal(rowptr, want_rowptr) and np.array_equal(col, want_col)
    g = ga.CSRGraph(rowptr, col)
    assert len(g) == n and g[4095] == graph[4095] and g[int(np.flatnonzero(np.diff(rowptr) == 0)[0])] == []
    # errors are codes with messages
    bad = tmp_path / "bad.txt"
    bad.write_text("0 1\n2\n")
    with pytest.raises(ga.GraphGANHipError) as ei:
        ga.read_edges_csr(bad)
    assert ei.value.code == -1 and "one id" in str(ei.value)
    gap = tmp_path / "gap.txt"
    gap.write_text("0 1\n1 5\n")
    with pytest.raises(ga.GraphGANHipError) as ei:
        ga.read_edges_csr(gap)
    assert "not 0..N-1" in str(ei.value)
    with pytest.raises(ga.GraphGANHipError) as ei:
        ga.read_edges_csr(tmp_path / "missing.txt")
    assert ei.value.code == -6


def test_powerlaw_split_workload_follows_the_survey_recipe():
    """10 % of the edges held out (seed 3), one negative per test edge that is neither the edge's first
    endpoint nor one of its neighbours in train + test (src/utils.py:96-128 semantics, seed 4), train adjacency in file order of
    the remaining edges; the generator is seeded: two calls give the same split."""
    import numpy as np
    from graphgan_amd import workloads, engine
    n = 3000
    w = workloads.powerlaw_split_workload(n, 10, 16)
    w2 = workloads.powerlaw_split_workload(n, 10, 16)
    for k in ("rowptr", "col", "test", "test_neg", "emb"):
        assert np.array_equal(w[k], w2[k]), k
    edges = engine.synth_powerlaw(n, 10, 1, 2)
    assert len(w["test"]) == round(0.1 * len(edges)) and w["n_train_edges"] == len(edges) - len(w["test"])
    assert w["rowptr"][-1] == 2 * w["n_train_edges"] and w["emb"].shape == (n, 16) and w["emb"].dtype == np.float32
    es = set(map(tuple, edges.tolist())) | set((b, a) for a, b in edges.tolist())
    ts = set(map(tuple, w["test"].tolist()))
    assert ts <= set(map(tuple, edges.tolist()))
    # the train CSR holds exactly the other edges, both directions
    src = np.repeat(np.arange(n), np.diff(w["rowptr"]))
    train_dir = set(zip(src.tolist(), w["col"].tolist()))
    assert all((a, b) not in train_dir and (b, a) not in train_dir for a, b in w["test"].tolist())
    assert len(train_dir) == 2 * w["n_train_edges"]
    # negatives: same first endpoint as their test edge, never a neighbour (train or test) nor the node itself
    assert np.array_equal(w["test_neg"][:, 0], w["test"][:, 0])
    assert all(a != b and (a, b) not in es for a, b in w["test_neg"].tolist())
