"""Drives the host-side C++ of libgraphgan_hip (edge ingest, BFS tree builder, .emb writer, graph synthesis) with
well-formed and malformed inputs.  Run by tests/test_host_sanitizers.py against an AddressSanitizer + UBSan build:
    LD_PRELOAD=<libasan.so> python tests/support/fuzz_host.py <libhost_asan.so>"""
import ctypes, os, sys, tempfile
import numpy as np
lib = ctypes.CDLL(sys.argv[1])
P, i32, i64, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
class GGGraph(ctypes.Structure):
    _fields_ = [("n_node", i32), ("nnz", i64), ("n_train_edges", i64), ("n_test_edges", i64), ("rowptr", ctypes.POINTER(i64)), ("col", ctypes.POINTER(i32))]
lib.gg_host_build_trees.restype = i64
lib.gg_host_build_trees.argtypes = [i32, P, P, P, i32, P, P, P, i64, i32, P]
lib.gg_host_write_embeddings.argtypes = [P, i64, i32, ctypes.c_char_p, i32]
lib.gg_synth_powerlaw.restype = i64
lib.gg_synth_powerlaw.argtypes = [i32, i32, u64, u64, P, i64]
lib.gg_host_read_edges.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(GGGraph)]
lib.gg_host_free_graph.argtypes = [ctypes.POINTER(GGGraph)]
def p(a): return a.ctypes.data_as(P)
rs = np.random.RandomState(0)
tmp = tempfile.mkdtemp()
# --- ingest: well-formed and malformed files
cases = ["0 1\n1 2\n2 0\n", "0\t1\n\n 3   4 \n5 5\n", "0 1", "", "\n\n", "a b\n", "1\n", "0 1 2\n", "-1 2\n", "0 99999999999999\n", "4294967296 1\n",
         "1 2\r\n3 4\r\n", "0 1\n" * 1000, " ".join(str(rs.randint(0, 50)) for _ in range(2001)) + "\n", "7 8\n" + "\x00\x01\n", "1e3 4\n", "+3 +4\n", "3 4 # c\n"]
for ci, txt in enumerate(cases):
    for test_txt in (None, "", "1 2\n", "x\n", "999 1000\n"):
        f = os.path.join(tmp, "tr%d.txt" % ci); open(f, "w").write(txt)
        ft = None
        if test_txt is not None:
            ft = os.path.join(tmp, "te.txt"); open(ft, "w").write(test_txt)
        g = GGGraph()
        rc = lib.gg_host_read_edges(f.encode(), ft.encode() if ft else None, ctypes.byref(g))
        if rc == 0:
            n = g.n_node
            rp = np.ctypeslib.as_array(g.rowptr, shape=(n + 1,)).copy() if n >= 0 and g.rowptr else None
            if rp is not None and g.nnz > 0:
                col = np.ctypeslib.as_array(g.col, shape=(g.nnz,)).copy()
                assert rp[0] == 0 and rp[-1] == g.nnz and (np.diff(rp) >= 0).all() and col.min() >= 0 and col.max() < n, (ci, txt[:30])
            lib.gg_host_free_graph(ctypes.byref(g))
rc = lib.gg_host_read_edges(b"/nonexistent/file", None, ctypes.byref(GGGraph())); assert rc < 0
# --- tree builder on random graphs incl. isolated nodes, self loops, duplicates
for trial in range(60):
    n = int(rs.randint(1, 60)); m = int(rs.randint(0, 150))
    e = rs.randint(0, n, size=(m, 2))
    adj = [[] for _ in range(n)]
    for a, b in e.tolist():
        adj[a].append(b); adj[b].append(a)
    rowptr = np.zeros(n + 1, np.int64); rowptr[1:] = np.cumsum([len(x) for x in adj])
    col = np.array([v for x in adj for v in x], np.int32) if rowptr[-1] else np.zeros(0, np.int32)
    roots = rs.randint(0, n, size=int(rs.randint(1, 9))).astype(np.int32)
    base = np.zeros(len(roots) + 1, np.int64); depth = np.zeros(1, np.int32)
    tot = lib.gg_host_build_trees(n, p(rowptr), p(col), p(roots), len(roots), None, None, p(base), 0, 3, None)
    assert tot >= 0
    off = np.zeros(len(roots) * (n + 1), np.int32); nbr = np.zeros(max(tot, 1), np.int32)
    tot2 = lib.gg_host_build_trees(n, p(rowptr), p(col), p(roots), len(roots), p(off), p(nbr), p(base), tot, int(rs.randint(1, 5)), p(depth))
    assert tot2 == tot, (tot, tot2)
    # too small capacity / bad roots must be error codes
    if tot > 0:
        assert lib.gg_host_build_trees(n, p(rowptr), p(col), p(roots), len(roots), p(off), p(nbr), p(base), tot - 1, 2, None) < 0
    bad = roots.copy(); bad[0] = n
    assert lib.gg_host_build_trees(n, p(rowptr), p(col), p(bad), len(bad), None, None, p(base), 0, 2, None) < 0
# --- writer: specials
for d in (1, 3, 50, 128):
    n = 37
    emb = rs.randn(n, d).astype(np.float32)
    emb.flat[::7] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 3.4e38][: len(emb.flat[::7])] if emb.size >= 49 else emb.flat[::7]
    out = os.path.join(tmp, "e.emb")
    assert lib.gg_host_write_embeddings(p(emb), n, d, out.encode(), 3) == 0
    lines = open(out).read().split("\n")
    assert lines[0] == "%d\t%d" % (n, d) and len(lines) == n + 2
    for i in range(n):
        got = lines[1 + i].split("\t"); assert got[0] == str(i)
        want = [str(float(x)) for x in emb[i].astype(np.float64)]
        assert got[1:] == want, (i, got[1:4], want[:4])
assert lib.gg_host_write_embeddings(p(emb), n, d, b"/nonexistent/dir/x.emb", 2) < 0
# --- synth
for n, m in ((12, 3), (1000, 10), (11, 10)):
    cap = lib.gg_synth_powerlaw(n, m, 1, 2, None, 0)
    assert cap > 0
    ed = np.zeros((cap, 2), np.int32)
    assert lib.gg_synth_powerlaw(n, m, 1, 2, p(ed), cap) == cap
    assert ed.min() >= 0 and ed.max() < n and (ed[:, 0] != ed[:, 1]).all()
    assert lib.gg_synth_powerlaw(n, m, 1, 2, p(ed), cap - 1) < 0
assert lib.gg_synth_powerlaw(5, 10, 1, 2, None, 0) < 0
print("host sanitizer run: ok")
