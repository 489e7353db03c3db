#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by EXECUTING THE REFERENCE'S OWN CODE.

Runs only in the build container (needs /root/reference); the fixtures it writes are
committed so that tests and the GPU box never read /root/reference.

How the reference is executed without TensorFlow: a stub ``tensorflow``
module is put in ``sys.modules`` (the model modules only *define* classes at import),
``graph_gan.GraphGAN`` is instantiated with ``object.__new__`` and given ``.graph``,
``.n_node``, a fake ``.sess.run`` (returns ``E.E^T + b`` for ``generator.all_score`` and
zeros for ``discriminator.reward``) -- then the reference's unmodified
``construct_trees`` (graph_gan.py:84-108), ``sample`` (:225-270),
``prepare_data_for_d`` (:182-202), ``prepare_data_for_g`` (:204-223) and
``get_node_pairs_from_path`` (:272-291) run as shipped, under ``np.random.seed``.
``utils.read_edges`` / ``read_embeddings`` (utils.py:12-67) and
``LinkPredictEval`` (link_prediction.py:10-38) are imported and run unmodified.

Outputs:
  ca_grqc.npz          the shipped CA-GrQc fixture (train/test/test_neg edges, pre-trained rows as fp32)
  ref_small_*.npz      reference outputs on small synthetic graphs (trees, D+G prepare sequences)
  ref_ca_grqc.npz      reference outputs on CA-GrQc for a subset of roots
  ref_misc.json        docstring vector, epoch-0 accuracy of the shipped embeddings
"""
import copy
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    tf = types.ModuleType("tensorflow")
    sys.modules["tensorflow"] = tf
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "src", "GraphGAN"))
    import graph_gan  # noqa
    from src import utils  # noqa
    from src.evaluation import link_prediction  # noqa
    return graph_gan, utils, link_prediction


def all_score_f64(E, b):
    """Fake ``sess.run(generator.all_score)``: fp64 product rounded once to fp32 (BLAS-order
    independent), bias added in fp32 -- the oracle's ``all_score_fn`` in the pin test."""
    return (E.astype(np.float64) @ E.T.astype(np.float64)).astype(np.float32) + b.astype(np.float32)


class FakeSess:
    def __init__(self, E, b, gen, dis):
        self.E, self.b, self.gen, self.dis = E, b, gen, dis

    def run(self, fetch, feed_dict=None):
        if fetch is self.gen.all_score:
            return all_score_f64(self.E, self.b)
        if fetch is self.dis.reward:
            n = len(feed_dict[self.dis.node_id])
            return np.zeros(n, dtype=np.float32)
        raise RuntimeError("unexpected fetch")


def make_ref_instance(graph_gan, n_node, graph, E, b):
    g = object.__new__(graph_gan.GraphGAN)
    g.n_node, g.graph = n_node, graph
    g.root_nodes = list(range(n_node))
    gen = types.SimpleNamespace(all_score=object())
    dis = types.SimpleNamespace(reward=object(), node_id=object(), node_neighbor_id=object())
    g.generator, g.discriminator = gen, dis
    g.sess = FakeSess(E, b, gen, dis)
    return g


def trees_to_arrays(trees, roots, n_node):
    off = np.zeros((len(roots), n_node + 1), dtype=np.int32)
    base = np.zeros(len(roots) + 1, dtype=np.int64)
    chunks = []
    for i, r in enumerate(roots):
        t, run, lst = trees[r], 0, []
        for v in range(n_node):
            off[i, v] = run
            if v in t:
                lst.extend(t[v])
                run += len(t[v])
        off[i, n_node] = run
        base[i + 1] = base[i] + run
        chunks.append(np.asarray(lst, dtype=np.int32))
    return off, np.concatenate(chunks), base


def small_graph(rs, n, n_edges, n_self, n_isolated):
    """edge list with self-loops, duplicate-free otherwise, a few isolated (test-only) nodes."""
    live = n - n_isolated
    edges = set()
    while len(edges) < n_edges:
        a, b = rs.randint(0, live, 2)
        if a != b and (a, b) not in edges and (b, a) not in edges:
            edges.add((int(a), int(b)))
    seen = set(x for e in edges for x in e)
    for v in range(live):  # every live id must appear (README.md:32-39: ids are 0..N-1)
        if v not in seen:
            edges.add((v, int((v + 1 + rs.randint(0, live - 1)) % live)))
    edges = sorted(edges)
    rs.shuffle(edges)
    for _ in range(n_self):
        a = int(rs.randint(0, live))
        edges.insert(int(rs.randint(0, len(edges))), (a, a))
    # make sure every live node id appears; isolated ones appear only in "test"
    test = [(int(rs.randint(0, live)), live + i) for i in range(n_isolated)]
    return edges, test


def flatten_paths(paths):
    lens = np.array([len(p) for p in paths], dtype=np.int32)
    flat = np.array([x for p in paths for x in p], dtype=np.int32)
    return flat, lens


def run_prepare_sequence(graph_gan, ref, trees, seed, n_rounds):
    """D-prepare then G-prepare, n_rounds times, on one (mutating) tree dict."""
    import config  # the reference's config module
    ref.trees = trees
    np.random.seed(seed)
    out = {}
    for r in range(n_rounds):
        c, nb, lab = ref.prepare_data_for_d()
        out["d%d_center" % r] = np.array(c, dtype=np.int32)
        out["d%d_neighbor" % r] = np.array(nb, dtype=np.int32)
        out["d%d_label" % r] = np.array(lab, dtype=np.int32)
        n1, n2, _ = ref.prepare_data_for_g()
        out["g%d_node1" % r] = np.array(n1, dtype=np.int32)
        out["g%d_node2" % r] = np.array(n2, dtype=np.int32)
    out["n_sample_gen"] = np.int32(config.n_sample_gen)
    out["window_size"] = np.int32(config.window_size)
    return out


def main():
    graph_gan, utils, lp = import_reference()
    os.chdir(os.path.join(REF, "src", "GraphGAN"))
    import config

    # ---------------------------------------------------------------- CA-GrQc fixture
    n_node, graph = utils.read_edges(config.train_filename, config.test_filename)
    train = np.array(utils.read_edges_from_file(config.train_filename), dtype=np.int32)
    test = np.array(utils.read_edges_from_file(config.test_filename), dtype=np.int32)
    test_neg = np.array(utils.read_edges_from_file(config.test_neg_filename), dtype=np.int32)
    with open(config.pretrain_emb_filename_g) as f:
        lines = f.readlines()[1:]
    ids = np.array([int(l.split()[0]) for l in lines], dtype=np.int32)
    rows64 = np.array([[float(x) for x in l.split()[1:]] for l in lines], dtype=np.float64)
    rows32 = rows64.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "ca_grqc.npz"), n_node=np.int32(n_node), train=train, test=test,
                        test_neg=test_neg, emb_ids=ids, emb_rows=rows32)

    misc = {}
    # docstring known answer (graph_gan.py:276-277), executed
    misc["pairs_docstring_path"] = [1, 0, 2, 4, 2]
    misc["pairs_docstring_out"] = graph_gan.GraphGAN.get_node_pairs_from_path([1, 0, 2, 4, 2])
    rs = np.random.RandomState(7)
    extra = []
    for L in (2, 3, 4, 5, 6, 9, 13):
        p = [int(x) for x in rs.randint(0, 50, L)]
        extra.append({"path": p, "pairs": graph_gan.GraphGAN.get_node_pairs_from_path(p)})
    misc["pairs_extra"] = extra

    # epoch-0 accuracy of the shipped embeddings under the shipped evaluator, 3 seeds
    accs = []
    for s in (0, 1, 2):
        np.random.seed(s)
        e = lp.LinkPredictEval(config.pretrain_emb_filename_g, config.test_filename, config.test_neg_filename,
                               n_node, config.n_emb)
        accs.append(e.eval_link_prediction())
    misc["epoch0_accuracy"] = accs

    # ---------------------------------------------------------------- small graphs
    specs = [(40, 70, 3, 2, 11), (64, 200, 0, 0, 12), (25, 24, 2, 3, 13), (90, 130, 5, 4, 14)]
    for gi, (n, ne, nself, niso, seed) in enumerate(specs):
        rs = np.random.RandomState(seed)
        edges, tedges = small_graph(rs, n, ne, nself, niso)
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            trf, tef = os.path.join(td, "tr.txt"), os.path.join(td, "te.txt")
            with open(trf, "w") as f:
                f.writelines("%d\t%d\n" % e for e in edges)
            with open(tef, "w") as f:
                f.writelines("%d\t%d\n" % e for e in tedges)
            nn, gr = utils.read_edges(trf, tef)
        # ids must be 0..nn-1 for the reference (README.md:32-39); regenerate if a live id never appeared
        assert set(gr.keys()) == set(range(nn)), "graph %d has gaps; change seed" % gi
        d = 6
        E = (rs.randn(nn, d) * 0.9).astype(np.float32)
        b = (rs.randn(nn) * 0.3).astype(np.float32)
        ref = make_ref_instance(graph_gan, nn, gr, E, b)
        trees = ref.construct_trees(list(range(nn)))
        off, nbr, base = trees_to_arrays(trees, list(range(nn)), nn)
        out = run_prepare_sequence(graph_gan, ref, copy.deepcopy(trees), seed=100 + gi, n_rounds=2)
        np.savez_compressed(os.path.join(HERE, "ref_small_%d.npz" % gi), n_node=np.int32(nn),
                            train=np.array(edges, dtype=np.int32), test=np.array(tedges, dtype=np.int32).reshape(-1, 2),
                            E=E, b=b, tree_off=off, tree_nbr=nbr, tree_base=base, seed=np.int32(100 + gi), **out)

    # ---------------------------------------------------------------- CA-GrQc subset
    rs = np.random.RandomState(5)
    np.random.seed(5)
    E64 = utils.read_embeddings(config.pretrain_emb_filename_g, n_node, config.n_emb)
    E = E64.astype(np.float32)
    b = (rs.randn(n_node) * 0.05).astype(np.float32)
    deg = np.array([len(graph[v]) for v in range(n_node)])
    roots = sorted(set([int(np.argmax(deg))] + [int(x) for x in rs.choice(n_node, 47, replace=False)]
                       + [int(np.where(deg == 0)[0][0]), int(np.where(deg == 1)[0][0])]))
    ref = make_ref_instance(graph_gan, n_node, graph, E, b)
    ref.root_nodes = roots
    trees = ref.construct_trees(roots)
    off, nbr, base = trees_to_arrays(trees, roots, n_node)
    out = run_prepare_sequence(graph_gan, ref, copy.deepcopy(trees), seed=77, n_rounds=2)
    np.savez_compressed(os.path.join(HERE, "ref_ca_grqc.npz"), roots=np.array(roots, dtype=np.int32), E=E, b=b,
                        tree_off=off, tree_nbr=nbr, tree_base=base, seed=np.int32(77), **out)

    with open(os.path.join(HERE, "ref_misc.json"), "w") as f:
        json.dump(misc, f, indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
