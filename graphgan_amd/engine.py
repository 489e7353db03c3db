"""numpy-facing wrapper of the C ABI (``include/graphgan_hip.h``).

``Engine`` is what the ``graph_gan.py`` mirror drives in place of the reference's
``tf.Session``: each method corresponds to one ``sess.run`` call site or host-side
sampler of ``src/GraphGAN/graph_gan.py`` (cited per method).  All numerics run in
``libgraphgan_hip.so``; nothing here computes scores, samples or updates.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import GGConfig, GGCounters, check, lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def graph_to_csr(n_node, graph):
    """Adjacency dict of ``utils.read_edges`` -> (rowptr int64 [N+1], col int32), list order kept."""
    deg = np.fromiter((len(graph.get(v, ())) for v in range(n_node)), dtype=np.int64, count=n_node)
    rowptr = np.zeros(n_node + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    col = np.empty(int(rowptr[-1]), dtype=np.int32)
    for v in range(n_node):
        lst = graph.get(v)
        if lst:
            col[rowptr[v]:rowptr[v + 1]] = lst
    return rowptr, col


def edges_to_csr(n_node, edges):
    """Vectorised ``read_edges`` adjacency for large synthetic graphs: for edge k = (a, b) the
    reference appends b to graph[a] and then a to graph[b] (utils.py:36-37), in file order."""
    edges = np.asarray(edges).reshape(-1, 2)
    if len(edges) and (int(edges.min()) < 0 or int(edges.max()) >= n_node):  # (the C pass below writes unchecked)
        raise ValueError("edges_to_csr: node id outside [0, %d)" % n_node)
    if 2 * len(edges) < 2 ** 31 - 1:
        try:  # one stable counting pass in C (scipy's COO -> CSR kernel, called directly: no duplicate merging, no index sorting)
            from scipy.sparse import _sparsetools
            src = np.empty(2 * len(edges), dtype=np.int32)
            dst = np.empty(2 * len(edges), dtype=np.int32)
            src[0::2], dst[0::2] = edges[:, 0], edges[:, 1]
            src[1::2], dst[1::2] = edges[:, 1], edges[:, 0]
            indptr = np.zeros(n_node + 1, dtype=np.int32)
            col = np.empty(len(src), dtype=np.int32)
            nothing = np.zeros(len(src), dtype=np.int8)
            _sparsetools.coo_tocsr(n_node, n_node, len(src), src, dst, nothing, indptr, col, np.empty_like(nothing))
            return indptr.astype(np.int64), col
        except Exception:  # a private scipy routine: any change of its signature / dtype dispatch falls back to numpy
            pass
    edges = edges.astype(np.int64)
    src = np.empty(2 * len(edges), dtype=np.int64)
    dst = np.empty(2 * len(edges), dtype=np.int32)
    src[0::2], dst[0::2] = edges[:, 0], edges[:, 1]
    src[1::2], dst[1::2] = edges[:, 1], edges[:, 0]
    order = np.argsort(src, kind="stable")
    rowptr = np.zeros(n_node + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n_node), out=rowptr[1:])
    return rowptr, np.ascontiguousarray(dst[order])


def host_build_trees(n_node, rowptr, col, roots, n_threads=0):
    """``construct_trees`` (graph_gan.py:84-108) on host threads; no GPU needed.
    Returns (off int32 [R, N+1], nbr int32, nbr_base int64 [R+1], max_depth)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = _i32(col)
    roots = _i32(roots)
    R = len(roots)
    base = np.zeros(R + 1, dtype=np.int64)
    total = check(lib.gg_host_build_trees(n_node, _ptr(rowptr), _ptr(col), _ptr(roots), R, None, None, _ptr(base), 0,
                                          n_threads, None))
    off = np.zeros((R, n_node + 1), dtype=np.int32)
    nbr = np.zeros(max(int(total), 1), dtype=np.int32)
    dmax = np.zeros(1, dtype=np.int32)
    check(lib.gg_host_build_trees(n_node, _ptr(rowptr), _ptr(col), _ptr(roots), R, _ptr(off), _ptr(nbr), _ptr(base),
                                  int(total), n_threads, _ptr(dmax)))
    return off, nbr[: int(total)], base, int(dmax[0])


def read_edges_csr(train_filename, test_filename=""):
    """``utils.read_edges`` (utils.py:12-54) natively -> (n_node, rowptr int64 [N+1], col int32), list order kept."""
    g = _lib.GGGraph()
    check(lib.gg_host_read_edges(str(train_filename).encode(), str(test_filename or "").encode(), ctypes.byref(g)))
    try:
        rowptr = np.ctypeslib.as_array(g.rowptr, shape=(g.n_node + 1,)).copy()
        col = np.ctypeslib.as_array(g.col, shape=(max(g.nnz, 1),))[: g.nnz].copy()
        return int(g.n_node), rowptr, col
    finally:
        lib.gg_host_free_graph(ctypes.byref(g))


class CSRGraph:
    """Read-only stand-in for the reference's adjacency dict (``graph[i]`` -> list of neighbours)."""

    def __init__(self, rowptr, col):
        self.rowptr, self.col = rowptr, col

    def __len__(self):
        return len(self.rowptr) - 1

    def __getitem__(self, v):
        return self.col[self.rowptr[v]:self.rowptr[v + 1]].tolist()

    def get(self, v, default=None):
        return self[v] if 0 <= v < len(self) else default

    def keys(self):
        return range(len(self))

    def __iter__(self):
        return iter(range(len(self)))

    def __contains__(self, v):
        return 0 <= v < len(self)


def host_write_embeddings(path, emb, n_threads=0):
    """The reference's ``.emb`` text (graph_gan.py:293-306), byte-identical, from host threads."""
    emb = np.ascontiguousarray(emb, dtype=np.float32)
    check(lib.gg_host_write_embeddings(_ptr(emb), emb.shape[0], emb.shape[1], str(path).encode(), n_threads))


def synth_powerlaw(n_node, m, seed_graph=1, seed_perm=2):
    """Barabasi-Albert edge list [E, 2] int32 (SURVEY.md section 8d recipe)."""
    n = check(lib.gg_synth_powerlaw(n_node, m, seed_graph, seed_perm, None, 0))
    edges = np.empty((int(n), 2), dtype=np.int32)
    check(lib.gg_synth_powerlaw(n_node, m, seed_graph, seed_perm, _ptr(edges), int(n)))
    return edges


class Engine:
    """One HIP context on one MI355X: both embedding models, graph, trees, sample buffers."""

    def __init__(self, emb_gen, emb_dis, lr_gen=1e-3, lr_dis=1e-3, lambda_gen=1e-5, lambda_dis=1e-5, window_size=2,
                 optimizer=_lib.GG_OPT_ADAM_DENSE, device=0, adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8):
        eg = np.ascontiguousarray(emb_gen, dtype=np.float32)  # Q6: fp64 init rounded to fp32 once
        ed = np.ascontiguousarray(emb_dis, dtype=np.float32)
        assert eg.shape == ed.shape and eg.ndim == 2
        self.n_node, self.n_emb = int(eg.shape[0]), int(eg.shape[1])
        cfg = GGConfig(lr_gen, lr_dis, lambda_gen, lambda_dis, adam_beta1, adam_beta2, adam_eps, window_size, optimizer, device)
        self._ctx = ctypes.c_void_p()
        check(lib.gg_create(self.n_node, self.n_emb, _ptr(eg), _ptr(ed), ctypes.byref(cfg), ctypes.byref(self._ctx)))
        self.tree_roots = np.zeros(0, dtype=np.int32)
        self.max_depth = 0
        self._rowptr = None

    # ------------------------------------------------------------------ life cycle
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            lib.gg_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        return check(rc, self._ctx)

    # ------------------------------------------------------------------ graph / trees
    def set_graph_csr(self, rowptr, col):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = _i32(col)
        self._ck(lib.gg_set_graph_csr(self._ctx, _ptr(rowptr), _ptr(col)))
        self._rowptr = rowptr

    def build_trees(self, roots, n_threads=0, device=False):
        """graph_gan.py:31-46,84-108: BFS trees of ``roots`` -> device tree CSR (slot i = roots[i]).
        device=False: threaded host BFS + upload; device=True: BFS on the GPU (same trees)."""
        roots = _i32(roots)
        if device:
            self._ck(lib.gg_build_trees_device(self._ctx, _ptr(roots), len(roots)))
        else:
            self._ck(lib.gg_build_trees(self._ctx, _ptr(roots), len(roots), n_threads))
        self._after_trees(roots)

    def tree_bytes_estimate(self, n_roots):
        """Upper bound of the HBM bytes ``n_roots`` resident trees need (every root reaching every node)."""
        return float(n_roots) * 12.0 * (self.n_node + 1)  # pop order + first-child ranks + edge indices

    def set_tree_mode(self, mode, node_cap=0):
        """gg_set_tree_mode: 0 = whole trees, 1 = lazy trees (exact through a level, children lists below it resolved by the
        walks that need them -- same walks, bit for bit), -1 = lazy from 2^18 nodes on (the default)."""
        self._ck(lib.gg_set_tree_mode(self._ctx, int(mode), int(node_cap)))

    def lazy_stats(self):
        out = np.zeros(24, dtype=np.int64)
        self._ck(lib.gg_lazy_stats(self._ctx, _ptr(out)))
        return dict(lazy=bool(out[0]), min_level=int(out[1]), fallback_roots=int(out[2]), fallback_rounds=int(out[3]), exact_nodes=int(out[4]),
                    pool_entries=int(out[5]), lazy_slots=int(out[6]), max_level=int(out[7]), resolved=[int(x) for x in out[8:11]],
                    candidates=int(out[11]), scan_rounds=int(out[12]), max_rounds=int(out[13]), max_degree_resolved=int(out[14]), coop_lists=int(out[15]),
                    slots_by_level=[int(x) for x in out[16:24]])

    def get_lazy_trees(self):
        """Raw arrays of the resident LAZY trees (tests): dict(info [R, 4], base [R], order, cstart, edge, pair)."""
        R = len(self.tree_roots)
        n = ctypes.c_int64()
        self._ck(lib.gg_get_lazy_trees(self._ctx, ctypes.byref(n), None, None, None, None, None, None))
        n = n.value
        info, base = np.zeros((R, 4), np.int32), np.zeros(R, np.int64)
        order, cstart, edge, pair = np.zeros(n, np.int32), np.zeros(n + R, np.int32), np.zeros(n, np.int32), np.zeros(n, np.uint64)
        self._ck(lib.gg_get_lazy_trees(self._ctx, None, _ptr(info), _ptr(base), _ptr(order), _ptr(cstart), _ptr(edge), _ptr(pair)))
        return dict(info=info, base=base, order=order, cstart=cstart, edge=edge, pair=pair)

    def set_trees(self, roots, off, nbr, nbr_base, max_depth=0):
        roots, off, nbr = _i32(roots), _i32(off), _i32(nbr)
        nbr_base = np.ascontiguousarray(nbr_base, dtype=np.int64)
        self._ck(lib.gg_set_trees(self._ctx, _ptr(roots), len(roots), _ptr(off), _ptr(nbr), _ptr(nbr_base), max_depth))
        self._after_trees(roots)

    def _after_trees(self, roots):
        self.tree_roots = roots.copy()
        nr, ne, md = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int32()
        self._ck(lib.gg_tree_info(self._ctx, ctypes.byref(nr), ctypes.byref(ne), ctypes.byref(md)))
        self.tree_entries, self.max_depth = ne.value, md.value

    def save_trees(self, path):
        """The tree cache (reference: pickle.dump(self.trees), graph_gan.py:43-46)."""
        self._ck(lib.gg_save_trees(self._ctx, str(path).encode()))

    def load_trees(self, path):
        """pickle.load of the cache (graph_gan.py:35-38): the resident trees as saved by ``save_trees``; refuses a
        cache built from another graph."""
        self._ck(lib.gg_load_trees(self._ctx, str(path).encode()))
        nr = ctypes.c_int32()
        self._ck(lib.gg_tree_info(self._ctx, ctypes.byref(nr), None, None))
        roots = np.zeros(nr.value, dtype=np.int32)
        self._ck(lib.gg_tree_roots(self._ctx, _ptr(roots)))
        self._after_trees(roots)

    def get_trees(self):
        R = len(self.tree_roots)
        off = np.zeros((R, self.n_node + 1), dtype=np.int32)
        nbr = np.zeros(max(self.tree_entries, 1), dtype=np.int32)
        base = np.zeros(R + 1, dtype=np.int64)
        self._ck(lib.gg_get_trees(self._ctx, _ptr(off), _ptr(nbr), _ptr(base)))
        return off, nbr[: self.tree_entries], base

    def get_tree_order(self):
        """The resident trees in BFS-order form: (base [R+1], order, cstart, edge, edges_valid) -- see gg_get_tree_order."""
        R = len(self.tree_roots)
        base = np.zeros(R + 1, dtype=np.int64)
        self._ck(lib.gg_get_tree_order(self._ctx, _ptr(base), None, None, None, None))
        nodes = int(base[R])
        order, cstart, edge = np.zeros(nodes, np.int32), np.zeros(nodes + R, np.int32), np.zeros(nodes, np.int32)
        valid = ctypes.c_int32()
        self._ck(lib.gg_get_tree_order(self._ctx, _ptr(base), _ptr(order), _ptr(cstart), _ptr(edge), ctypes.byref(valid)))
        return base, order, cstart, edge, bool(valid.value)

    # ------------------------------------------------------------------ K1
    def walk_sample(self, slots, n_walks, for_d, seed, stream, stride=None, fetch=True):
        """GraphGAN.sample (graph_gan.py:225-270) for many roots at once."""
        slots, n_walks = _i32(slots), _i32(n_walks)
        stride = int(stride or (self.max_depth + 3))
        total = int(n_walks.sum())
        if not fetch:
            self._ck(lib.gg_walk_sample(self._ctx, _ptr(slots), _ptr(n_walks), len(slots), int(bool(for_d)), seed, stream,
                                        None, None, None, stride, None))
            return None
        samples = np.full(total, -1, dtype=np.int32)
        paths = np.full((total, stride), -1, dtype=np.int32)
        plen = np.zeros(total, dtype=np.int32)
        status = np.zeros(len(slots), dtype=np.int32)
        self._ck(lib.gg_walk_sample(self._ctx, _ptr(slots), _ptr(n_walks), len(slots), int(bool(for_d)), seed, stream,
                                    _ptr(samples), _ptr(paths), _ptr(plen), stride, _ptr(status)))
        self._after_trees(self.tree_roots)  # (lazy trees: slots rebuilt whole may have raised the depth)
        return dict(samples=samples, paths=paths, path_len=plen, root_status=status)

    def get_walks(self):
        """Walk outputs left resident by the last walk_sample / prepare_d / prepare_g call (the (samples, paths)
        ``sample`` returned inside ``prepare_data_for_*``, graph_gan.py:191,210)."""
        tot, stride, ns = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
        self._ck(lib.gg_walk_info(self._ctx, ctypes.byref(tot), ctypes.byref(stride), ctypes.byref(ns)))
        total, stride, ns = tot.value, stride.value, ns.value
        samples = np.full(total, -1, dtype=np.int32)
        paths = np.full((total, max(stride, 1)), -1, dtype=np.int32)
        plen = np.zeros(total, dtype=np.int32)
        status = np.zeros(ns, dtype=np.int32)
        self._ck(lib.gg_get_walks(self._ctx, _ptr(samples), _ptr(paths), _ptr(plen), _ptr(status)))
        return dict(samples=samples, paths=paths, path_len=plen, root_status=status)

    # ------------------------------------------------------------------ prepared data
    def prepare_d(self, slots, seed, stream, fetch=True):
        """prepare_data_for_d (graph_gan.py:182-202) -> (center, neighbor, label, root_status)."""
        slots = _i32(slots)
        n = ctypes.c_int64()
        status = np.zeros(len(slots), dtype=np.int32) if fetch else None  # no status read-back for resident-only use
        self._ck(lib.gg_prepare_d(self._ctx, _ptr(slots), len(slots), seed, stream, ctypes.byref(n), _ptr(status) if fetch else None))
        self.d_rows = n.value
        if not fetch:
            return n.value
        c, nb, lab = np.zeros(n.value, np.int32), np.zeros(n.value, np.int32), np.zeros(n.value, np.float32)
        self._ck(lib.gg_get_d_data(self._ctx, _ptr(c), _ptr(nb), _ptr(lab)))
        return c, nb, lab, status

    def prepare_g(self, slots, n_sample, seed, stream, fetch=True):
        """prepare_data_for_g (graph_gan.py:204-223) -> (node_1, node_2, reward, root_status)."""
        slots = _i32(slots)
        n = ctypes.c_int64()
        status = np.zeros(len(slots), dtype=np.int32) if fetch else None
        self._ck(lib.gg_prepare_g(self._ctx, _ptr(slots), len(slots), n_sample, seed, stream, ctypes.byref(n), _ptr(status) if fetch else None))
        self.g_pairs = n.value
        if not fetch:
            return n.value
        a, b, r = np.zeros(n.value, np.int32), np.zeros(n.value, np.int32), np.zeros(n.value, np.float32)
        self._ck(lib.gg_get_g_data(self._ctx, _ptr(a), _ptr(b), _ptr(r)))
        return a, b, r, status

    def prepare_g_begin(self, slots, n_sample, seed, stream):
        """Optional head start: enqueue the walks of the NEXT prepare_g (same arguments) on the side stream now -- before
        the d_pass that precedes it -- without waiting for anything (gg_prepare_g_begin)."""
        slots = _i32(slots)
        self._ck(lib.gg_prepare_g_begin(self._ctx, _ptr(slots), len(slots), n_sample, seed, stream))

    # ------------------------------------------------------------------ an epoch over root batches (trees not all resident)
    def epoch_begin(self, reset_d=True, reset_g=True):
        """Empty the accumulated discriminator rows / generator pairs of gg_epoch_add."""
        self._ck(lib.gg_epoch_begin(self._ctx, int(bool(reset_d)), int(bool(reset_g))))

    def epoch_add(self, roots, do_d=True, do_g=True, n_sample=20, seed=0, stream_d=0, stream_g=1):
        """prepare_data_for_d / prepare_data_for_g (graph_gan.py:182-223) for one BATCH of roots (node ids): trees built on
        the GPU, the roots' Q3 bits restored from / saved to the persistent store, rows and pairs appended to the epoch's
        arrays.  Returns (rows accumulated so far, pairs accumulated so far)."""
        roots = _i32(roots)
        rows, pairs = ctypes.c_int64(), ctypes.c_int64()
        self._ck(lib.gg_epoch_add(self._ctx, _ptr(roots), len(roots), int(bool(do_d)), int(bool(do_g)), int(n_sample), seed, stream_d,
                                  stream_g, ctypes.byref(rows), ctypes.byref(pairs)))
        if len(roots) and (do_d or do_g):
            self._after_trees(roots)
        return rows.value, pairs.value

    def epoch_commit(self, which):
        """The accumulated rows (which = 1) / pairs with their rewards (which = 0) become the data of d_pass / g_pass; returns their number."""
        n = ctypes.c_int64()
        self._ck(lib.gg_epoch_commit(self._ctx, int(which), ctypes.byref(n)))
        if which == 1:
            self.d_rows = n.value
        else:
            self.g_pairs = n.value
        return n.value

    def get_d_data(self):
        """The resident discriminator rows (center, neighbor, label) -- of the last prepare_d or epoch_commit(1)."""
        n = self.d_rows
        c, nb, lab = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.float32)
        self._ck(lib.gg_get_d_data(self._ctx, _ptr(c), _ptr(nb), _ptr(lab)))
        return c, nb, lab

    def get_g_data(self):
        """The resident generator pairs (node_1, node_2, reward) -- of the last prepare_g or epoch_commit(0)."""
        n = self.g_pairs
        a, b, r = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.float32)
        self._ck(lib.gg_get_g_data(self._ctx, _ptr(a), _ptr(b), _ptr(r)))
        return a, b, r

    def q3_clear(self):
        """Forget every in-place tree mutation (graph_gan.py:258-259) kept for the root-batched epochs."""
        self._ck(lib.gg_q3_clear(self._ctx))

    def q3_get(self):
        """(word_off int64 [N+1], words uint32): bit j of root v's words = the father entry of its (j+1)-th tree child was removed."""
        off = np.zeros(self.n_node + 1, dtype=np.int64)
        self._ck(lib.gg_q3_get(self._ctx, _ptr(off), None))
        words = np.zeros(max(int(off[-1]), 1), dtype=np.uint32)
        self._ck(lib.gg_q3_get(self._ctx, None, _ptr(words)))
        return off, words[: int(off[-1])]

    def d_pass(self, starts, batch_size):
        """One inner D epoch over the prepared rows (graph_gan.py:149-157)."""
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        self._ck(lib.gg_d_pass(self._ctx, _ptr(starts), len(starts), batch_size))

    def g_pass(self, starts, batch_size):
        """One inner G epoch over the prepared pairs (graph_gan.py:168-176)."""
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        self._ck(lib.gg_g_pass(self._ctx, _ptr(starts), len(starts), batch_size))

    # ------------------------------------------------------------------ sess.run call sites with host buffers
    def pair_reward(self, u, v):
        """sess.run(discriminator.reward) (graph_gan.py:220-222)."""
        u, v = _i32(u), _i32(v)
        out = np.zeros(len(u), dtype=np.float32)
        self._ck(lib.gg_pair_reward(self._ctx, _ptr(u), _ptr(v), len(u), _ptr(out)))
        return out

    def d_step(self, u, v, label):
        """sess.run(discriminator.d_updates) (graph_gan.py:154-157)."""
        u, v, label = _i32(u), _i32(v), np.ascontiguousarray(label, dtype=np.float32)
        self._ck(lib.gg_d_step(self._ctx, _ptr(u), _ptr(v), _ptr(label), len(u)))

    def g_step(self, u, v, reward):
        """sess.run(generator.g_updates) (graph_gan.py:173-176)."""
        u, v, reward = _i32(u), _i32(v), np.ascontiguousarray(reward, dtype=np.float32)
        self._ck(lib.gg_g_step(self._ctx, _ptr(u), _ptr(v), _ptr(reward), len(u)))

    def all_score(self, rows=None):
        """sess.run(generator.all_score) (graph_gan.py:238) for ``rows`` (None: all) -> fp32 [len(rows), n_node]."""
        if rows is None:
            out = np.zeros((self.n_node, self.n_node), dtype=np.float32)
            self._ck(lib.gg_all_score(self._ctx, None, 0, _ptr(out)))
            return out
        rows = _i32(rows)
        out = np.zeros((len(rows), self.n_node), dtype=np.float32)
        self._ck(lib.gg_all_score(self._ctx, _ptr(rows), len(rows), _ptr(out)))
        return out

    def all_score_reduce(self, rows=None, precision="fp32", logsumexp=True):
        """Rows of ``generator.all_score`` (generator.py:21) streamed through a fused consumer: per row the maximum, its
        column and log-sum-exp over ALL nodes; nothing of size rows x N is materialised.  precision "fp32" (exact) or
        "bf16" (bf16 inputs on the matrix cores, fp32 accumulate).  Returns dict(max, argmax, logsumexp, kernel_ms)."""
        n_rows = self.n_node if rows is None else len(rows)
        rows_a = None if rows is None else _i32(rows)
        mx = np.zeros(n_rows, dtype=np.float32)
        am = np.zeros(n_rows, dtype=np.int32)
        lse = np.zeros(n_rows, dtype=np.float32) if logsumexp else None
        ms = ctypes.c_double()
        self._ck(lib.gg_all_score_reduce(self._ctx, _ptr(rows_a), n_rows, {"fp32": 0, "bf16": 1}[precision], int(bool(logsumexp)),
                                         _ptr(mx), _ptr(am), _ptr(lse), ctypes.byref(ms)))
        return dict(max=mx, argmax=am, logsumexp=lse, kernel_ms=ms.value)

    def get_embeddings(self, which):
        """sess.run(embedding_matrix) (graph_gan.py:298); which: 0 = gen, 1 = dis."""
        out = np.zeros((self.n_node, self.n_emb), dtype=np.float32)
        self._ck(lib.gg_get_embeddings(self._ctx, which, _ptr(out)))
        return out

    def write_embeddings(self, which, path, n_threads=0):
        """write_embeddings_to_file (graph_gan.py:293-306) for one model, natively."""
        self._ck(lib.gg_write_embeddings(self._ctx, which, str(path).encode(), n_threads))

    def write_embeddings_bin(self, which, path):
        """Binary side-car of the ``.emb`` text (same fp32 numbers; header "GGEB", version, n_emb, n_node)."""
        self._ck(lib.gg_write_embeddings_bin(self._ctx, which, str(path).encode()))

    def edge_scores(self, which, u, v):
        """Evaluator scores np.dot(emd[u], emd[v]) (link_prediction.py:26-27) from the resident table, float64."""
        u, v = _i32(u), _i32(v)
        out = np.zeros(len(u), dtype=np.float64)
        self._ck(lib.gg_edge_scores(self._ctx, which, _ptr(u), _ptr(v), len(u), _ptr(out)))
        return out

    def get_bias(self, which):
        out = np.zeros(self.n_node, dtype=np.float32)
        self._ck(lib.gg_get_bias(self._ctx, which, _ptr(out)))
        return out

    def set_embeddings(self, which, emb):
        emb = np.ascontiguousarray(emb, dtype=np.float32)
        assert emb.shape == (self.n_node, self.n_emb)
        self._ck(lib.gg_set_embeddings(self._ctx, which, _ptr(emb)))

    def set_bias(self, which, bias):
        bias = np.ascontiguousarray(bias, dtype=np.float32)
        assert bias.shape == (self.n_node,)
        self._ck(lib.gg_set_bias(self._ctx, which, _ptr(bias)))

    def save_state(self, path):
        self._ck(lib.gg_save_state(self._ctx, path.encode()))

    def load_state(self, path):
        self._ck(lib.gg_load_state(self._ctx, path.encode()))

    def counters(self):
        c = GGCounters()
        self._ck(lib.gg_get_counters(self._ctx, ctypes.byref(c)))
        return {k: getattr(c, k) for k, _ in GGCounters._fields_ if k != "reserved"}

    def set_profiling(self, every_n):
        """HIP events around every ``every_n``-th walk launch (1 = every launch and every pass, the default;
        0 = none).  With ``every_n != 1`` ``d_pass`` / ``g_pass`` return once their kernels are enqueued."""
        self._ck(lib.gg_set_profiling(self._ctx, int(every_n)))

    def set_profiling_solo(self, solo):
        """solo=True (default): profiled side-stream walks are measured alone; False: overlapped with the D update."""
        self._ck(lib.gg_set_profiling_solo(self._ctx, int(bool(solo))))

    def synchronize(self):
        self._ck(lib.gg_synchronize(self._ctx))

    # ------------------------------------------------------------------ multi-GPU
    @staticmethod
    def comm_unique_id():
        buf = (ctypes.c_char * 128)()
        check(lib.gg_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (ctypes.c_char * 128).from_buffer_copy(unique_id)
        self._ck(lib.gg_comm_init(self._ctx, buf, rank, world))

    def comm_stats(self):
        """Gradient-exchange statistics of this rank: dict(sparse_steps, dense_steps, bytes_sent, world) + what kind the sparse
        steps were (pack_steps: all-gather of fixed-capacity row packs; owner_steps: owner-partitioned send / recv + gather) and
        whether the owner-partitioned exchange moves bf16 rows."""
        out = np.zeros(8, dtype=np.int64)
        self._ck(lib.gg_comm_stats_ex(self._ctx, _ptr(out)))
        return dict(sparse_steps=int(out[0] + out[1]), dense_steps=int(out[2]), bytes_sent=int(out[3]), world=int(out[4]),
                    pack_steps=int(out[0]), owner_steps=int(out[1]), bf16_rows=bool(out[5]))

    def comm_barrier(self):
        self._ck(lib.gg_comm_barrier(self._ctx))
