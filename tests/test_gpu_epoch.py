"""GPU parity of the ROOT-BATCHED epoch (gg_epoch_begin / gg_epoch_add / gg_epoch_commit, ABI 5): when the trees of all roots
cannot be resident, prepare_data_for_d / prepare_data_for_g (reference src/GraphGAN/graph_gan.py:182-223) run batch by batch
with the in-place tree mutations (Q3, :258-259) kept in a store that outlives the trees.  Rows, pairs and mutation state must
equal (a) the all-resident engine calls over the same roots and (b) the oracle, over two outer epochs -- the second epoch's
walks depend on the bits the first one left."""
import numpy as np
import pytest

from oracle import graphgan_oracle as orc
from tests.helpers import load_ca_grqc, ca_grqc_init_embeddings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ga():
    import graphgan_amd
    return graphgan_amd


def _oracle_d_rows(n, rowptr, col, deg, want):
    c, nb, lab, w = [], [], [], 0
    for i in range(n):
        k = int(deg[i])
        if want["root_status"][i] == 0 and k > 0:
            c += [i] * (2 * k)
            nb += list(col[rowptr[i]:rowptr[i + 1]]) + list(want["samples"][w:w + k])
            lab += [1] * k + [0] * k
        w += k
    return np.array(c, np.int32), np.array(nb, np.int32), np.array(lab, np.float32)


def _oracle_g_pairs(want, n_walks):
    n1, n2 = [], []
    for wlk in range(n_walks):
        L = want["path_len"][wlk]
        if L > 0:
            for a, b in orc.pairs_from_path(list(want["paths"][wlk, :L]), 2):
                n1.append(a)
                n2.append(b)
    return np.array(n1, np.int32), np.array(n2, np.int32)


def _oracle_q3_bits(n, off, nbr, base, word_off):
    """Mutation state of the oracle's trees in the store's layout: bit j of root r = the father entry of its (j + 1)-th tree child is gone."""
    words = np.zeros(int(word_off[-1]), np.uint32)
    for r in range(n):
        lst = nbr[base[r] + off[r, r]: base[r] + off[r, r + 1]]          # [r, child_1, ..., child_k]
        for j, c in enumerate(lst[1:]):
            if nbr[base[r] + off[r, c]] < 0:
                words[word_off[r] + j // 32] |= np.uint32(1 << (j % 32))
    return words


def test_root_batched_epochs_equal_the_resident_calls_and_the_oracle(ga):
    d, n, graph = load_ca_grqc()
    rowptr, col = ga.graph_to_csr(n, graph)
    deg = (rowptr[1:] - rowptr[:-1]).astype(np.int32)
    E = ca_grqc_init_embeddings(d, n).astype(np.float32)
    rs = np.random.RandomState(8)
    Ed = (E + 0.05 * rs.randn(*E.shape)).astype(np.float32)
    b = (rs.randn(n) * 0.05).astype(np.float32)
    roots = np.arange(n, dtype=np.int32)
    slots = roots.copy()
    engs = []
    for _ in range(2):
        e = ga.Engine(E, Ed, optimizer=ga.GG_OPT_SGD)  # (SGD: table differences stay proportional to the fp32 summation-order noise)
        e.set_bias(0, b)
        e.set_graph_csr(rowptr, col)
        engs.append(e)
    A, B = engs
    A.build_trees(roots, device=True)              # every tree resident
    off, nbr, base, dmax = orc.c_build_trees(n, rowptr, col, roots)
    nbr = nbr.copy()
    stride = dmax + 3
    batch = 1100                                   # 5 batches of the 5 242 roots
    seed = 21
    word_off, _ = B.q3_get()
    assert np.array_equal(np.diff(word_off), (deg + 31) // 32)
    for epoch in range(2):
        sd, sg = 2 * epoch, 2 * epoch + 1
        gen_E, gen_b = A.get_embeddings(0), A.get_bias(0)
        Ep = orc.pad_rows(gen_E)
        # ---- oracle: D walks (mutating), then G walks on the mutated trees
        want_d = orc.c_walk_sample(Ep, gen_b, off, nbr, base, roots, slots, deg, True, seed, sd, stride)
        oc, onb, olab = _oracle_d_rows(n, rowptr, col, deg, want_d)
        nw = np.full(n, 20, np.int32)
        want_g = orc.c_walk_sample(Ep, gen_b, off, nbr, base, roots, slots, nw, False, seed, sg, stride)
        o1, o2 = _oracle_g_pairs(want_g, 20 * n)
        # ---- A: all trees resident (the reference's shape)
        cA, nA, lA, _ = A.prepare_d(slots, seed, sd)
        A.d_pass([0], 1 << 30)
        g1A, g2A, rA, _ = A.prepare_g(slots, 20, seed, sg)
        # ---- B: root batches; the G-mode walks share the trees of the D prepare
        B.epoch_begin()
        for k in range(0, n, batch):
            rows, pairs = B.epoch_add(roots[k:k + batch], True, True, 20, seed, sd, sg)
        assert B.epoch_commit(1) == rows == len(cA)
        cB, nB, lB = B.get_d_data()
        B.d_pass([0], 1 << 30)
        assert B.epoch_commit(0) == pairs == len(g1A)
        g1B, g2B, rB = B.get_g_data()
        for name, x, y, z in (("center", cA, cB, oc), ("neighbor", nA, nB, onb), ("label", lA, lB, olab), ("node_1", g1A, g1B, o1), ("node_2", g2A, g2B, o2)):
            assert np.array_equal(x, y), "epoch %d: %s, batched vs resident" % (epoch, name)
            assert np.array_equal(x, z), "epoch %d: %s vs the oracle" % (epoch, name)
        assert np.max(np.abs(rA - rB)) <= 1e-5      # same pairs; the D tables differ by the order of the fp32 atomics only
        # mutation state: oracle lists == A's resident trees == B's persistent store
        _, tnbr, _ = A.get_trees()
        assert np.array_equal(tnbr, nbr)
        _, words = B.q3_get()
        assert np.array_equal(words, _oracle_q3_bits(n, off, nbr, base, word_off))
        assert words.any()
        # generator pass on both; then B takes A's tables so that the next epoch's walks start from identical floats
        A.g_pass([0], 1 << 30)
        B.g_pass([0], 1 << 30)
        for which in (0, 1):
            assert np.max(np.abs(A.get_embeddings(which) - B.get_embeddings(which))) <= 2e-5
            B.set_embeddings(which, A.get_embeddings(which))
            B.set_bias(which, A.get_bias(which))
    # the second epoch DID depend on the stored bits: without them its G-mode walks come out differently
    B.q3_clear()
    B.epoch_begin()
    for k in range(0, n, batch):
        B.epoch_add(roots[k:k + batch], False, True, 20, seed, 0, 3)
    B.epoch_commit(0)
    h1, _, _ = B.get_g_data()
    B.set_embeddings(0, gen_E)
    # a root twice in one batch would make two slots share one row of the persistent Q3 store: refused
    with pytest.raises(ga.GraphGANHipError) as ei:
        B.epoch_add(np.array([3, 8, 3], np.int32), True, True, 20, seed, 0, 1)
    assert ei.value.code == ga.GG_EINVAL and "twice" in str(ei.value)
    A.close()
    B.close()
    assert len(h1) != len(o1) or not np.array_equal(h1, o1)


def test_trainer_runs_an_epoch_over_root_batches(ga, tmp_path):
    """GraphGAN.train() with the tree budget forced below N trees and update_ratio = 1 (the case that used to end in
    GG_ENOMEM at scale): one outer epoch of the fused schedule runs through gg_epoch_*, and its tables equal those of the same
    schedule with every tree resident."""
    from tests.test_gpu_e2e import make_cfg, write_reference_layout
    from graphgan_amd import graph_gan
    outs = []
    for budget in (160.0, None):
        base = str(tmp_path / ("b%s" % budget))
        import os
        os.makedirs(base)
        d, n, graph = write_reference_layout(base)
        kw = dict(n_epochs=1, n_epochs_dis=2, n_epochs_gen=2, dis_interval=2, gen_interval=2, batch_size_dis=1 << 30, batch_size_gen=1 << 30,
                  engine_optimizer="sgd", engine_seed=5)
        if budget is None:
            kw.update(engine_tree_budget_gb=900.5 * 12.0 * (n + 1) / 2.0 ** 30)   # room for 900 of the 5 242 trees: 6 batches
        else:
            kw.update(engine_tree_budget_gb=budget)
        g = graph_gan.GraphGAN(make_cfg(base, **kw))
        assert g._all_resident == (budget is not None)
        g.train()
        outs.append((g.engine.get_embeddings(0), g.engine.get_embeddings(1), g.engine.counters()))
        g.engine.close()
    (g0, d0, c0), (g1, d1, c1) = outs
    assert c1["bfs_trees"] >= n and c0["hops"] == c1["hops"] and c0["d_pairs"] == c1["d_pairs"] and c0["g_pairs"] == c1["g_pairs"]
    assert np.max(np.abs(d0 - d1)) <= 2e-5 and np.max(np.abs(g0 - g1)) <= 2e-5
