"""CPU oracle for the GraphGAN hot path -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference (hwwang55/GraphGAN, /root/reference, never read
at run time).  Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package
``graphgan_amd`` never does.

What follows which reference lines:

=========================  =====================================================
``read_edges``             src/utils.py:12-54
``read_embeddings``        src/utils.py:57-67
``construct_trees``        src/GraphGAN/graph_gan.py:84-108
``GraphGANOracle.sample``  src/GraphGAN/graph_gan.py:225-270 + utils.py:131-133
``prepare_data_for_d``     src/GraphGAN/graph_gan.py:182-202
``prepare_data_for_g``     src/GraphGAN/graph_gan.py:204-223
``pairs_from_path``        src/GraphGAN/graph_gan.py:272-291
``reward``                 src/GraphGAN/discriminator.py:21-24,33-34
``d_step``                 src/GraphGAN/discriminator.py:21-32
``g_step``                 src/GraphGAN/generator.py:22-31
``TF1Adam``                tf.train.AdamOptimizer of tensorflow==1.8.0 (README.md:26;
                           un-vendored third party; algorithm restated from its
                           published ``_apply_sparse_shared``: duplicate indices
                           summed first, ``m``/``v`` decayed over ALL rows, then
                           ``var -= lr_t * m / (sqrt(v) + eps)`` over ALL rows,
                           ``lr_t = lr*sqrt(1-b2^t)/(1-b1^t)``, eps not corrected)
``train``                  src/GraphGAN/graph_gan.py:122-180
``write_embeddings``       src/GraphGAN/graph_gan.py:293-306
``eval_link_prediction``   src/evaluation/link_prediction.py:19-38
=========================  =====================================================

Pinning status: the reference ships no tests.  The integer part
(trees, walks, window pairs) is pinned by executing the reference's own
``construct_trees`` / ``sample`` / ``get_node_pairs_from_path`` under a stub
``tensorflow`` module (tests/golden/make_golden.py -> tests/golden/*.npz) and
comparing this restatement in ``rng='reference'`` mode draw for draw.  The TF1
float graphs (reward, d/g updates, Adam) cannot run here (no TensorFlow for
Python 3.10, no network): **parity unpinned** for those; they are cross-checked
against torch-CPU autograd and a hand-computed Adam example instead.

Two RNG modes:
  * ``rng='reference'``: consumes the global legacy ``np.random`` stream in the
    reference's order (one ``rand()`` per root per prepare, one ``choice`` per hop,
    one ``shuffle`` per pass) -- used to pin this file against the reference code.
  * ``rng='counter'``: Philox4x32-10 keyed by (seed; hop, walk, root, stream) -- the
    order-independent stream the HIP kernels use.
Two arithmetic modes for the walk:
  * ``arith='numpy'``: what the reference computes (BLAS fp32 scores, fp32 softmax,
    fp64 cumsum + searchsorted-right of ``np.random.choice``).
  * ``arith='spec'``: the exactly-specified arithmetic of DESIGN.md section 3 via the C
    oracle (oracle/walk_oracle.c); bit-exact target for the HIP walk kernel.
"""
from __future__ import annotations

import collections
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# ----------------------------------------------------------------------------- I/O


def read_edges_from_file(filename):
    """utils.py:50-54 -- whitespace-split ints, one edge per line."""
    with open(filename, "r") as f:
        return [[int(t) for t in line.split()] for line in f.readlines()]


def read_edges(train_filename, test_filename):
    """utils.py:12-47 -- graph dict keeps FILE ORDER of neighbours; both directions for
    train edges (a self-loop ``a a`` appends ``a`` twice, Q4); test nodes get empty lists."""
    graph = {}
    nodes = set()
    train = read_edges_from_file(train_filename)
    test = read_edges_from_file(test_filename) if test_filename != "" else []
    for a, b in train:
        nodes.add(a)
        nodes.add(b)
        graph.setdefault(a, [])
        graph.setdefault(b, [])
        graph[a].append(b)
        graph[b].append(a)
    for a, b in test:
        nodes.add(a)
        nodes.add(b)
        graph.setdefault(a, [])
        graph.setdefault(b, [])
    return len(nodes), graph


def read_embeddings(filename, n_node, n_embed):
    """utils.py:57-67 -- float64 [n_node, n_embed]; rows absent from the file keep
    ``np.random.rand`` values (global RNG, Q5)."""
    with open(filename, "r") as f:
        lines = f.readlines()[1:]
    emb = np.random.rand(n_node, n_embed)
    for line in lines:
        t = line.split()
        emb[int(t[0]), :] = [float(x) for x in t[1:]]
    return emb


def graph_to_csr(n_node, graph):
    """adjacency dict -> (rowptr int64 [N+1], col int32) preserving list order."""
    rowptr = np.zeros(n_node + 1, dtype=np.int64)
    for v in range(n_node):
        rowptr[v + 1] = rowptr[v] + len(graph.get(v, ()))
    col = np.zeros(int(rowptr[-1]), dtype=np.int32)
    for v in range(n_node):
        lst = graph.get(v, ())
        col[rowptr[v]:rowptr[v + 1]] = lst
    return rowptr, col


def write_embeddings(filename, emb_fp32, n_emb):
    """graph_gan.py:293-306 -- header ``N\\td``; rows ``id\\tv0\\t...``; values are the
    fp32 numbers widened to float64 and printed with ``str`` (hstack promotes)."""
    n = emb_fp32.shape[0]
    index = np.array(range(n)).reshape(-1, 1)
    mat = np.hstack([index, emb_fp32])
    rows = mat.tolist()
    lines = [str(n) + "\t" + str(n_emb) + "\n"]
    lines += [str(int(r[0])) + "\t" + "\t".join(str(x) for x in r[1:]) + "\n" for r in rows]
    with open(filename, "w+") as f:
        f.writelines(lines)


def eval_link_prediction(emb_f64, test_edges, test_neg_edges):
    """link_prediction.py:19-38 -- dot per edge, label = score >= median, accuracy with
    the first half (positives) labelled 1."""
    edges = list(test_edges) + list(test_neg_edges)
    score = np.array([np.dot(emb_f64[a], emb_f64[b]) for a, b in edges])
    median = np.median(score)
    pred = (score >= median).astype(np.float64)
    true = np.zeros(len(edges))
    true[: len(edges) // 2] = 1
    return float(np.mean(pred == true))


# ----------------------------------------------------------------------------- trees


def construct_trees(graph, nodes):
    """graph_gan.py:84-108 -- dict root -> dict node -> [father, child...]; FIFO BFS,
    adjacency order; ``trees[root][root] = [root, child...]``."""
    trees = {}
    for root in nodes:
        t = {root: [root]}
        used = set()
        queue = collections.deque([root])
        while queue:
            cur = queue.popleft()
            used.add(cur)
            for sub in graph[cur]:
                if sub not in used:
                    t[cur].append(sub)
                    t[sub] = [cur]
                    queue.append(sub)
                    used.add(sub)
        trees[root] = t
    return trees


def trees_to_csr(trees, roots, n_node):
    """dict trees -> the tree-CSR layout of DESIGN.md section 2 (off int32 [R, N+1] relative to
    nbr_base int64 [R+1]; lists in node-id order)."""
    off = np.zeros((len(roots), n_node + 1), dtype=np.int32)
    base = np.zeros(len(roots) + 1, dtype=np.int64)
    chunks = []
    for i, r in enumerate(roots):
        t = trees[r]
        run = 0
        lst = []
        for v in range(n_node):
            off[i, v] = run
            if v in t:
                lst.extend(t[v])
                run += len(t[v])
        off[i, n_node] = run
        base[i + 1] = base[i] + run
        chunks.append(np.asarray(lst, dtype=np.int32))
    nbr = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.int32)
    return off, nbr, base


# ----------------------------------------------------------------------------- C oracle

class _Counters(ctypes.Structure):
    _fields_ = [("hops", ctypes.c_int64), ("nbr_reads", ctypes.c_int64), ("walks", ctypes.c_int64)]


_C = None


def c_oracle():
    """Load oracle/_build/libwalk_oracle.so (built by ``make -C oracle`` /
    ``__graft_entry__.build()``)."""
    global _C
    if _C is not None:
        return _C
    path = os.path.join(_HERE, "_build", "libwalk_oracle.so")
    if not os.path.exists(path):
        raise RuntimeError("C oracle not built: run `make -C oracle` or __graft_entry__.build()")
    lib = ctypes.CDLL(path)
    P = ctypes.c_void_p
    lib.orc_philox4x32_10.argtypes = [P, P, P]
    lib.orc_philox4x32_10.restype = None
    lib.orc_uniform53.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    lib.orc_uniform53.restype = ctypes.c_uint64
    lib.orc_expf.argtypes = [ctypes.c_float]
    lib.orc_expf.restype = ctypes.c_float
    lib.orc_dot16.argtypes = [P, P, ctypes.c_int]
    lib.orc_dot16.restype = ctypes.c_float
    lib.orc_weight.argtypes = [ctypes.c_float]
    lib.orc_weight.restype = ctypes.c_uint64
    lib.orc_build_trees.argtypes = [ctypes.c_int, P, P, P, ctypes.c_int, P, P, P, ctypes.c_int64, P]
    lib.orc_build_trees.restype = ctypes.c_int64
    lib.orc_walk_sample.argtypes = [ctypes.c_int, ctypes.c_int, P, P, P, P, P, P, P, P, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_uint64, ctypes.c_uint32, P, P, P, ctypes.c_int, P, P]
    lib.orc_walk_sample.restype = ctypes.c_int
    lib.orc_pairs_from_path.argtypes = [P, ctypes.c_int, ctypes.c_int, P, P]
    lib.orc_pairs_from_path.restype = ctypes.c_int
    lib.orc_all_score_rows.argtypes = [P, P, ctypes.c_int, ctypes.c_int, P, ctypes.c_int, P]
    lib.orc_all_score_rows.restype = None
    _C = lib
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def pad_rows(emb, mult=4):
    """fp32 [N, d] -> fp32 [N, ld] with ld = d rounded up to a multiple of 4 (zero padded)."""
    emb = np.asarray(emb, dtype=np.float32)
    n, d = emb.shape
    ld = (d + mult - 1) // mult * mult
    out = np.zeros((n, ld), dtype=np.float32)
    out[:, :d] = emb
    return out


def c_build_trees(n_node, rowptr, col, roots):
    """C restatement of construct_trees -> (off [R,N+1], nbr, nbr_base [R+1], max_depth)."""
    lib = c_oracle()
    roots = np.ascontiguousarray(roots, dtype=np.int32)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    R = len(roots)
    off = np.zeros((R, n_node + 1), dtype=np.int32)
    base = np.zeros(R + 1, dtype=np.int64)
    dmax = np.zeros(1, dtype=np.int32)
    total = lib.orc_build_trees(n_node, _p(rowptr), _p(col), _p(roots), R, _p(off), None, _p(base), 0, _p(dmax))
    nbr = np.zeros(max(int(total), 1), dtype=np.int32)
    got = lib.orc_build_trees(n_node, _p(rowptr), _p(col), _p(roots), R, _p(off), _p(nbr), _p(base), int(total), _p(dmax))
    assert got == total
    return off, nbr[: int(total)], base, int(dmax[0])


def c_all_score_rows(emb_pad, bias, rows):
    """generator.py:21 for selected rows, k-ordered fp32 fmaf chain + bias (the MFMA kernel's arithmetic)."""
    lib = c_oracle()
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    n = emb_pad.shape[0]
    out = np.zeros((len(rows), n), dtype=np.float32)
    lib.orc_all_score_rows(_p(emb_pad), _p(bias), n, emb_pad.shape[1], _p(rows), len(rows), _p(out))
    return out


def c_walk_sample(emb_pad, bias, off, nbr, base, tree_root, slots, n_walks, for_d, seed, stream, stride):
    """Spec-arithmetic sequential walk sampler (mutates ``nbr`` in place in D-mode, Q3).
    Returns dict(samples, paths, path_len, root_status, hops, nbr_reads, walks)."""
    lib = c_oracle()
    n_node = off.shape[1] - 1
    slots = np.ascontiguousarray(slots, dtype=np.int32)
    n_walks = np.ascontiguousarray(n_walks, dtype=np.int32)
    tree_root = np.ascontiguousarray(tree_root, dtype=np.int32)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    assert emb_pad.dtype == np.float32 and emb_pad.flags.c_contiguous and emb_pad.shape[1] % 4 == 0
    assert nbr.dtype == np.int32 and nbr.flags.c_contiguous and off.flags.c_contiguous
    tot = int(n_walks.sum())
    samples = np.full(tot, -1, dtype=np.int32)
    paths = np.full((tot, stride), -1, dtype=np.int32)
    plen = np.zeros(tot, dtype=np.int32)
    status = np.zeros(len(slots), dtype=np.int32)
    ctr = _Counters()
    rc = lib.orc_walk_sample(n_node, emb_pad.shape[1], _p(emb_pad), _p(bias), _p(off), _p(nbr), _p(base),
                             _p(tree_root), _p(slots), _p(n_walks), len(slots), int(bool(for_d)),
                             ctypes.c_uint64(seed), ctypes.c_uint32(stream),
                             _p(samples), _p(paths), _p(plen), stride, _p(status), ctypes.byref(ctr))
    if rc != 0:
        raise RuntimeError("orc_walk_sample failed rc=%d (path stride too small?)" % rc)
    return dict(samples=samples, paths=paths, path_len=plen, root_status=status,
                hops=ctr.hops, nbr_reads=ctr.nbr_reads, walks=ctr.walks)


# ----------------------------------------------------------------------------- python spec pieces

def philox4x32_10(ctr, key):
    """Pure-python Philox4x32-10 (spec S4), independent of the C oracle."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c0, c1, c2, c3 = [int(x) & 0xFFFFFFFF for x in ctr]
    k0, k1 = [int(x) & 0xFFFFFFFF for x in key]
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, \
                         ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def uniform53(seed, stream, root, walk, hop):
    """53-bit numerator m of the hop's uniform u = m / 2**53 (numpy legacy random_sample
    layout: (a >> 5) * 2**26 + (b >> 6))."""
    o = philox4x32_10((hop, walk, root, stream), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    return ((o[0] >> 5) << 26) | (o[1] >> 6)


def softmax(x):
    """utils.py:131-133, in the dtype of x."""
    e = np.exp(x - np.max(x))
    return e / e.sum()


def choice_index(p, u):
    """np.random.choice(p=p) for a given uniform u (legacy RandomState.choice):
    cdf = cumsum(float64(p)); cdf /= cdf[-1]; searchsorted(cdf, u, side='right')."""
    cdf = np.cumsum(np.asarray(p, dtype=np.float64))
    cdf /= cdf[-1]
    return int(np.searchsorted(cdf, u, side="right"))


def pairs_from_path(path, window_size=2):
    """graph_gan.py:272-291."""
    path = path[:-1]
    pairs = []
    for i in range(len(path)):
        for j in range(max(i - window_size, 0), min(i + window_size + 1, len(path))):
            if i != j:
                pairs.append([path[i], path[j]])
    return pairs


# ----------------------------------------------------------------------------- TF1 Adam

class TF1Adam:
    """tf.train.AdamOptimizer(lr) with IndexedSlices gradients, tensorflow==1.8.0 semantics
    (see module docstring).  ``apply(var, idx, grad_rows)`` updates var in place."""

    def __init__(self, shapes, lr, beta1=0.9, beta2=0.999, eps=1e-8, lazy=False):
        self.lr, self.b1, self.b2, self.eps = np.float32(lr), np.float32(beta1), np.float32(beta2), np.float32(eps)
        self.m = [np.zeros(s, dtype=np.float32) for s in shapes]
        self.v = [np.zeros(s, dtype=np.float32) for s in shapes]
        self.t = 0
        self.lazy = lazy
        # TF keeps beta1_power / beta2_power as fp32 variables multiplied once per step
        self.b1p = np.float32(beta1)
        self.b2p = np.float32(beta2)

    def lr_t(self):
        return np.float32(self.lr * np.sqrt(np.float32(1) - self.b2p) / (np.float32(1) - self.b1p))

    def step(self, variables, sparse_grads):
        """sparse_grads: list of (indices int array, values [n, ...]) per variable;
        duplicates are summed first (_deduplicate_indexed_slices)."""
        self.t += 1
        lr_t = self.lr_t()
        one = np.float32(1)
        for var, m, v, (idx, val) in zip(variables, self.m, self.v, sparse_grads):
            uniq, inv = np.unique(idx, return_inverse=True)
            summed = np.zeros((len(uniq),) + val.shape[1:], dtype=np.float32)
            np.add.at(summed, inv, val.astype(np.float32))
            if self.lazy:
                m[uniq] = m[uniq] * self.b1 + (one - self.b1) * summed
                v[uniq] = v[uniq] * self.b2 + (one - self.b2) * summed * summed
                var[uniq] -= lr_t * m[uniq] / (np.sqrt(v[uniq]) + self.eps)
            else:
                m *= self.b1
                m[uniq] += (one - self.b1) * summed
                v *= self.b2
                v[uniq] += (one - self.b2) * summed * summed
                var -= lr_t * m / (np.sqrt(v) + self.eps)
        self.b1p = np.float32(self.b1p * self.b1)
        self.b2p = np.float32(self.b2p * self.b2)


# ----------------------------------------------------------------------------- models

def _sigmoid(x):
    """tf.nn.sigmoid on fp32 tensors: evaluated in fp32 (saturates to exactly 1.0 for x > ~16.6, like
    TF's fp32 kernel), not in fp64."""
    if os.environ.get("GG_ORACLE_SIGMOID64") == "1":
        # numeric-variant switch for sensitivity studies (tests/run_oracle_epochs.py): the sigmoid evaluated in fp64 and
        # rounded once, instead of TF-like fp32 -- a difference of at most 1 ulp per value
        with np.errstate(over="ignore"):
            return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)
    x = x.astype(np.float32)
    with np.errstate(over="ignore"):
        return (np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32)


class PairModel:
    """Embedding table + bias vector with the pair score ``e_u . e_v + b[v]``
    (generator.py:11-25 / discriminator.py:11-24)."""

    def __init__(self, emb_init, lr, lazy=False):
        self.E = np.array(emb_init, dtype=np.float32)  # Q6: fp64 init rounded to fp32 once
        self.b = np.zeros(self.E.shape[0], dtype=np.float32)
        self.opt = TF1Adam([self.E.shape, self.b.shape], lr, lazy=lazy)

    def score(self, u, v):
        return np.sum(self.E[u] * self.E[v], axis=1, dtype=np.float32) + self.b[v]

    def _bias_slices(self, u, v, gb):
        """Gradient slices of the bias vector.  Dense TF1-Adam (the reference): indices v only (the gather of the
        graph), every element decays anyway.  Lazy mode -- the engine's scale mode, not a reference mode -- is
        NODE-granular: a node a step touches advances its embedding row AND its bias (gradient 0 where the node
        was only a centre), which is what the engine's per-row optimizer kernels do."""
        if self.opt.lazy:
            return np.concatenate([np.asarray(u), np.asarray(v)]), np.concatenate([np.zeros(len(u), np.float32), gb])
        return np.asarray(v), gb


class Discriminator(PairModel):
    def reward(self, u, v):
        """discriminator.py:33-34 -- log(1 + exp(clip(score, -10, 10)))."""
        s = np.clip(self.score(u, v), -10.0, 10.0).astype(np.float32)
        return np.log(np.float32(1) + np.exp(s)).astype(np.float32)

    def loss_and_grads(self, u, v, label, lam):
        """discriminator.py:26-30: sum sigmoid-CE + lam*(l2(E_v)+l2(E_u)+l2(b_v)), l2 = sum(x^2)/2.
        Returns (loss, dE_u rows, dE_v rows, db_v)."""
        s = self.score(u, v)
        y = label.astype(np.float32)
        loss = np.sum(np.maximum(s, 0) - s * y + np.log1p(np.exp(-np.abs(s))))
        loss += lam * 0.5 * (np.sum(self.E[v] ** 2) + np.sum(self.E[u] ** 2) + np.sum(self.b[v] ** 2))
        ds = (_sigmoid(s) - y).astype(np.float32)
        lam = np.float32(lam)
        gu = ds[:, None] * self.E[v] + lam * self.E[u]
        gv = ds[:, None] * self.E[u] + lam * self.E[v]
        gb = ds + lam * self.b[v]
        return float(loss), gu.astype(np.float32), gv.astype(np.float32), gb.astype(np.float32)

    def d_step(self, u, v, label, lam):
        _, gu, gv, gb = self.loss_and_grads(u, v, label, lam)
        idx = np.concatenate([u, v])
        self.opt.step([self.E, self.b], [(idx, np.concatenate([gu, gv])), self._bias_slices(u, v, gb)])


class Generator(PairModel):
    def all_score(self):
        """generator.py:21 -- E.E^T + b (bias broadcast over columns)."""
        return self.E @ self.E.T + self.b

    def loss_and_grads(self, u, v, reward, lam):
        """generator.py:26-29: -mean(log(clip(sigmoid(s),1e-5,1)) * r) + lam*(l2(E_v)+l2(E_u))."""
        s = self.score(u, v)
        sg = _sigmoid(s)
        p = np.clip(sg, np.float32(1e-5), np.float32(1))
        r = reward.astype(np.float32)
        B = np.float32(len(u))
        loss = -np.mean(np.log(p) * r) + lam * 0.5 * (np.sum(self.E[v] ** 2) + np.sum(self.E[u] ** 2))
        inside = (sg >= np.float32(1e-5)) & (sg <= np.float32(1))
        ds = np.where(inside, -(r / B) * (np.float32(1) - sg), np.float32(0)).astype(np.float32)
        lam = np.float32(lam)
        gu = ds[:, None] * self.E[v] + lam * self.E[u]
        gv = ds[:, None] * self.E[u] + lam * self.E[v]
        gb = ds
        return float(loss), gu.astype(np.float32), gv.astype(np.float32), gb.astype(np.float32)

    def g_step(self, u, v, reward, lam):
        _, gu, gv, gb = self.loss_and_grads(u, v, reward, lam)
        idx = np.concatenate([u, v])
        self.opt.step([self.E, self.b], [(idx, np.concatenate([gu, gv])), self._bias_slices(u, v, gb)])


# ----------------------------------------------------------------------------- the trainer

class Config:
    """config.py:1-25 defaults."""
    batch_size_gen = 64
    batch_size_dis = 64
    lambda_gen = 1e-5
    lambda_dis = 1e-5
    n_sample_gen = 20
    lr_gen = 1e-3
    lr_dis = 1e-3
    n_epochs = 20
    n_epochs_gen = 30
    n_epochs_dis = 30
    gen_interval = 30
    dis_interval = 30
    update_ratio = 1
    n_emb = 50
    window_size = 2


STREAM_D = 0  # stream id = 2*epoch + {0: D-prepare, 1: G-prepare}
STREAM_G = 1


class GraphGANOracle:
    """The reference trainer restated on numpy (graph_gan.py:17-61,122-291)."""

    def __init__(self, n_node, graph, emb_init_g, emb_init_d, cfg=None, rng="counter", arith="numpy",
                 seed=0, hoist_all_score=True, trees=None, lazy_adam=False, all_score_fn=None):
        self.cfg = cfg or Config()
        self.n_node, self.graph = n_node, graph
        self.root_nodes = list(range(n_node))
        self.rng, self.arith, self.seed = rng, arith, seed
        self.hoist = hoist_all_score
        self.all_score_fn = all_score_fn  # optional override of generator.py:21 (golden pin uses an fp64 product)
        self.generator = Generator(emb_init_g, self.cfg.lr_gen, lazy=lazy_adam)
        self.discriminator = Discriminator(emb_init_d, self.cfg.lr_dis, lazy=lazy_adam)
        self.csr = None
        if arith == "spec":
            assert rng == "counter", "spec arithmetic is defined on the counter RNG"
            self.trees = None  # tree CSR is built lazily for self.root_nodes (C oracle)
        else:
            self.trees = trees if trees is not None else construct_trees(graph, self.root_nodes)
        self.stream = 0
        self.host_rng = np.random.RandomState(seed)  # batch-order shuffles in counter mode
        self.counters = dict(hops=0, nbr_reads=0)
        self._all_score = None

    # -- graph_gan.py:225-270
    def sample(self, root, tree, sample_num, for_d):
        g = self.generator
        if self.hoist:
            all_score = None
        else:
            # graph_gan.py:238, once per call as the reference does
            all_score = self.all_score_fn(g.E, g.b) if self.all_score_fn else g.all_score()
        samples, paths = [], []
        n = 0
        while len(samples) < sample_num:
            cur, prev = root, -1
            paths.append([cur])
            is_root = True
            hop = 0
            while True:
                nbrs = tree[cur][1:] if is_root else tree[cur]
                is_root = False
                if len(nbrs) == 0:
                    return None, None
                if for_d:
                    if nbrs == [root]:
                        return None, None
                    if root in nbrs:
                        nbrs.remove(root)  # in place for non-root cur (Q3)
                if all_score is None:
                    sc = (g.E[nbrs] @ g.E[cur] + g.b[nbrs]).astype(np.float32)
                else:
                    sc = all_score[cur, nbrs]
                p = softmax(sc)
                if self.rng == "reference":
                    nxt = np.random.choice(nbrs, size=1, p=p)[0]
                else:
                    u = uniform53(self.seed, self.stream, root, n, hop) / 9007199254740992.0
                    nxt = nbrs[choice_index(p, u)]
                self.counters["hops"] += 1
                self.counters["nbr_reads"] += len(nbrs)
                paths[n].append(nxt)
                hop += 1
                if nxt == prev:
                    samples.append(cur)
                    break
                prev, cur = cur, nxt
            n += 1
        return samples, paths

    def _take_root(self):
        if self.rng == "reference":
            return np.random.rand() < self.cfg.update_ratio
        return True if self.cfg.update_ratio >= 1 else self.host_rng.rand() < self.cfg.update_ratio

    # -- graph_gan.py:182-202
    def _spec_setup(self):
        if self.csr is None:
            self._rowptr, self._col = graph_to_csr(self.n_node, self.graph)
            roots = np.asarray(self.root_nodes, dtype=np.int32)
            off, nbr, base, dmax = c_build_trees(self.n_node, self._rowptr, self._col, roots)
            self.csr = dict(off=off, nbr=nbr.copy(), base=base, roots=roots, stride=dmax + 3)
        return self.csr

    def _spec_walks(self, n_walks, for_d):
        t = self._spec_setup()
        slots = np.arange(len(t["roots"]), dtype=np.int32)
        res = c_walk_sample(pad_rows(self.generator.E), self.generator.b, t["off"], t["nbr"], t["base"], t["roots"], slots,
                            n_walks, for_d, self.seed, self.stream, t["stride"])
        self.counters["hops"] += res["hops"]
        self.counters["nbr_reads"] += res["nbr_reads"]
        return res

    def prepare_data_for_d(self):
        if self.arith == "spec":
            assert self.cfg.update_ratio >= 1
            t = self._spec_setup()
            deg = (self._rowptr[1:] - self._rowptr[:-1]).astype(np.int32)[t["roots"]]
            res = self._spec_walks(deg, True)
            centers, neighbors, labels, w = [], [], [], 0
            for i, r in enumerate(t["roots"]):
                k = int(deg[i])
                if k and res["root_status"][i] == 0:
                    centers += [int(r)] * (2 * k)
                    neighbors += self.graph[int(r)] + res["samples"][w:w + k].tolist()
                    labels += [1] * k + [0] * k
                w += k
            return centers, neighbors, labels
        centers, neighbors, labels = [], [], []
        for i in self.root_nodes:
            if self._take_root():
                pos = self.graph[i]
                neg, _ = self.sample(i, self.trees[i], len(pos), for_d=True)
                if len(pos) != 0 and neg is not None:
                    centers.extend([i] * len(pos))
                    neighbors.extend(pos)
                    labels.extend([1] * len(pos))
                    centers.extend([i] * len(pos))
                    neighbors.extend(neg)
                    labels.extend([0] * len(neg))
        return centers, neighbors, labels

    # -- graph_gan.py:204-223
    def prepare_data_for_g(self):
        if self.arith == "spec":
            assert self.cfg.update_ratio >= 1
            t = self._spec_setup()
            res = self._spec_walks(np.full(len(t["roots"]), self.cfg.n_sample_gen, dtype=np.int32), False)
            node_1, node_2 = [], []
            for w in range(len(res["path_len"])):
                L = int(res["path_len"][w])
                if L:
                    for a, b in pairs_from_path(res["paths"][w, :L].tolist(), self.cfg.window_size):
                        node_1.append(a)
                        node_2.append(b)
            reward = self.discriminator.reward(np.array(node_1, dtype=np.int64), np.array(node_2, dtype=np.int64))
            return node_1, node_2, reward
        paths = []
        for i in self.root_nodes:
            if self._take_root():
                _, pi = self.sample(i, self.trees[i], self.cfg.n_sample_gen, for_d=False)
                if pi is not None:
                    paths.extend(pi)
        node_1, node_2 = [], []
        for path in paths:
            for a, b in pairs_from_path(path, self.cfg.window_size):
                node_1.append(a)
                node_2.append(b)
        reward = self.discriminator.reward(np.array(node_1, dtype=np.int64), np.array(node_2, dtype=np.int64))
        return node_1, node_2, reward

    def _shuffle(self, start_list):
        if self.rng == "reference":
            np.random.shuffle(start_list)
        else:
            self.host_rng.shuffle(start_list)

    # -- graph_gan.py:133-176 (one outer epoch)
    def train_epoch(self, epoch):
        cfg = self.cfg
        centers = neighbors = labels = None
        for d_epoch in range(cfg.n_epochs_dis):
            if d_epoch % cfg.dis_interval == 0:
                self.stream = 2 * (epoch * max(cfg.n_epochs_dis, 1) + d_epoch) + STREAM_D
                centers, neighbors, labels = self.prepare_data_for_d()
                centers, neighbors = np.array(centers, dtype=np.int64), np.array(neighbors, dtype=np.int64)
                labels = np.array(labels, dtype=np.float32)
            start_list = list(range(0, len(centers), cfg.batch_size_dis))
            self._shuffle(start_list)
            for s in start_list:
                e = s + cfg.batch_size_dis
                self.discriminator.d_step(centers[s:e], neighbors[s:e], labels[s:e], cfg.lambda_dis)
        node_1 = node_2 = reward = None
        for g_epoch in range(cfg.n_epochs_gen):
            if g_epoch % cfg.gen_interval == 0:
                self.stream = 2 * (epoch * max(cfg.n_epochs_gen, 1) + g_epoch) + STREAM_G
                node_1, node_2, reward = self.prepare_data_for_g()
                node_1, node_2 = np.array(node_1, dtype=np.int64), np.array(node_2, dtype=np.int64)
            start_list = list(range(0, len(node_1), cfg.batch_size_gen))
            self._shuffle(start_list)
            for s in start_list:
                e = s + cfg.batch_size_gen
                self.generator.g_step(node_1[s:e], node_2[s:e], reward[s:e], cfg.lambda_gen)
