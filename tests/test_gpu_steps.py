"""GPU parity of the float kernels (through the C ABI) against the numpy oracle:
K2 pair_reward (abs 1e-5), K3/K4 d_step / g_step with TF1-Adam (rows to 1e-6 relative,
atol 1e-7: fp32 atomics sum duplicates in a different order than np.add.at), lazy-Adam and
SGD variants, prepared-data pipelines (integer arrays bit-exact), state save / restore."""
import os

import numpy as np
import pytest

from oracle import graphgan_oracle as orc
from tests.helpers import load_ca_grqc, load_small, ca_grqc_init_embeddings

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-6, 2e-7


@pytest.fixture(scope="module")
def ga():
    import graphgan_amd
    return graphgan_amd


def make_models(n, d, seed):
    rs = np.random.RandomState(seed)
    Eg = (rs.randn(n, d) * 0.5).astype(np.float32)
    Ed = (rs.randn(n, d) * 0.5).astype(np.float32)
    bg = (rs.randn(n) * 0.1).astype(np.float32)
    bd = (rs.randn(n) * 0.1).astype(np.float32)
    return Eg, Ed, bg, bd


def engine_with(ga, Eg, Ed, bg, bd, **kw):
    eng = ga.Engine(Eg, Ed, **kw)
    eng.set_bias(0, bg)
    eng.set_bias(1, bd)
    return eng


@pytest.mark.parametrize("d", [8, 50, 128])
def test_pair_reward(ga, d):
    n = 500
    Eg, Ed, bg, bd = make_models(n, d, d)
    Ed *= np.float32(np.sqrt(5.0 / np.sqrt(d)) / 0.5)  # dot std ~5: some scores beyond the +-10 clip
    eng = engine_with(ga, Eg, Ed, bg, bd)
    rs = np.random.RandomState(0)
    u, v = rs.randint(0, n, 10000), rs.randint(0, n, 10000)
    dis = orc.Discriminator(Ed, 1e-3)
    dis.b[:] = bd
    want = dis.reward(u, v)
    got = eng.pair_reward(u, v)
    # 1e-5 absolute (summation order of the fp32 dot differs from numpy's)
    assert np.max(np.abs(got - want)) <= 1e-5
    assert (np.abs(want - np.log1p(np.exp(10.0))) < 1e-5).any()  # the clip was exercised
    assert len(eng.pair_reward(np.zeros(0, np.int32), np.zeros(0, np.int32))) == 0
    with pytest.raises(ga.GraphGANHipError):
        eng.pair_reward([0], [n])
    eng.close()


@pytest.mark.parametrize("d,steps", [(50, 12), (128, 6)])
def test_dense_adam_steps_match_tf1_semantics(ga, d, steps):
    """B = 64 batches with duplicate rows, both models, several consecutive steps
    (m, v, beta powers carried on both sides)."""
    n = 300
    Eg, Ed, bg, bd = make_models(n, d, 7)
    eng = engine_with(ga, Eg, Ed, bg, bd)
    gen, dis = orc.Generator(Eg, 1e-3), orc.Discriminator(Ed, 1e-3)
    gen.b[:] = bg
    dis.b[:] = bd
    rs = np.random.RandomState(3)
    for t in range(steps):
        B = 64 if t != steps - 1 else 37  # ragged last chunk (graph_gan.py:153)
        u = rs.randint(0, n, B)
        v = rs.randint(0, n, B)
        u[: B // 2] = u[0]  # contiguous chunks of one root's rows: heavy duplication (a13)
        lab = (rs.rand(B) < 0.5).astype(np.float32)
        rew = (rs.rand(B) * 2).astype(np.float32)
        dis.d_step(u, v, lab, 1e-5)
        eng.d_step(u, v, lab)
        gen.g_step(u, v, rew, 1e-5)
        eng.g_step(u, v, rew)
        # one step: 1e-6 relative.  Later steps: Adam's m/sqrt(v) normalisation turns the
        # relative rounding noise of near-cancelling duplicate sums into absolute noise of order lr * 1e-3.
        rtol, atol = (RTOL, ATOL) if t == 0 else (1e-5, 3e-6)
        for which, m in ((0, gen), (1, dis)):
            dE = np.abs(eng.get_embeddings(which) - m.E)
            assert np.all(dE <= atol + rtol * np.abs(m.E)), (t, which, float(dE.max()))
            db = np.abs(eng.get_bias(which) - m.b)
            assert np.all(db <= atol + rtol * np.abs(m.b)), (t, which, float(db.max()))
    # rows never touched did not move (their m, v are exactly zero)
    c = eng.counters()
    assert c["d_steps"] == steps and c["g_steps"] == steps
    eng.close()


@pytest.mark.parametrize("mode", ["lazy", "sgd"])
def test_scale_mode_optimizers(ga, mode):
    n, d = 400, 64
    Eg, Ed, bg, bd = make_models(n, d, 11)
    opt = ga.GG_OPT_ADAM_LAZY if mode == "lazy" else ga.GG_OPT_SGD
    eng = engine_with(ga, Eg, Ed, bg, bd, optimizer=opt)
    dis = orc.Discriminator(Ed, 1e-3, lazy=True)
    dis.b[:] = bd
    rs = np.random.RandomState(5)
    for t in range(5):
        B = 3000
        u, v = rs.randint(0, n // 2, B), rs.randint(0, n // 2, B)  # rows >= n/2 stay untouched
        lab = (rs.rand(B) < 0.5).astype(np.float32)
        if mode == "lazy":
            dis.d_step(u, v, lab, 1e-5)
        else:
            _, gu, gv, gb = dis.loss_and_grads(u, v, lab, 1e-5)
            GE, Gb = np.zeros_like(dis.E, dtype=np.float64), np.zeros(n)
            np.add.at(GE, u, gu)
            np.add.at(GE, v, gv)
            np.add.at(Gb, v, gb)
            dis.E -= (1e-3 * GE).astype(np.float32)
            dis.b -= (1e-3 * Gb).astype(np.float32)
        eng.d_step(u, v, lab)
        assert np.allclose(eng.get_embeddings(1), dis.E, rtol=2e-5, atol=1e-6), t
        assert np.allclose(eng.get_bias(1), dis.b, rtol=2e-5, atol=1e-6), t
    assert np.array_equal(eng.get_embeddings(1)[n // 2:], Ed[n // 2:])
    eng.close()


def _setup_graph_engine(ga, gi=3, **kw):
    g, n, graph = load_small(gi)
    rs = np.random.RandomState(0)
    Ed = (rs.randn(n, g["E"].shape[1]) * 0.8).astype(np.float32)
    eng = ga.Engine(g["E"], Ed, **kw)
    eng.set_bias(0, g["b"])
    rowptr, col = ga.graph_to_csr(n, graph)
    eng.set_graph_csr(rowptr, col)
    eng.build_trees(np.arange(n))
    return g, n, graph, rowptr, col, Ed, eng


def test_prepare_d_and_g_match_oracle(ga):
    """prepare_data_for_d / prepare_data_for_g (graph_gan.py:182-223) device pipelines:
    integer arrays bit-exact against the spec oracle walks + the reference's row layout."""
    g, n, graph, rowptr, col, Ed, eng = _setup_graph_engine(ga)
    roots = np.arange(n, dtype=np.int32)
    off, nbr, base, dmax = orc.c_build_trees(n, rowptr, col, roots)
    Ep = orc.pad_rows(g["E"])
    deg = (rowptr[1:] - rowptr[:-1]).astype(np.int32)
    slots = np.arange(n, dtype=np.int32)
    dis = orc.Discriminator(Ed, 1e-3)
    for rnd in range(2):
        # D
        want = orc.c_walk_sample(Ep, g["b"], off, nbr, base, roots, slots, deg, True, 9, 2 * rnd, dmax + 3)
        c, nb, lab = [], [], []
        w = 0
        for i in range(n):
            k = int(deg[i])
            if want["root_status"][i] == 0 and k > 0:
                c += [i] * k + [i] * k
                nb += list(col[rowptr[i]:rowptr[i + 1]]) + list(want["samples"][w:w + k])
                lab += [1] * k + [0] * k
            w += k
        gc, gn, gl, st = eng.prepare_d(slots, 9, 2 * rnd)
        assert np.array_equal(st, want["root_status"])
        assert np.array_equal(gc, np.array(c, np.int32)) and np.array_equal(gn, np.array(nb, np.int32))
        assert np.array_equal(gl, np.array(lab, np.float32))
        assert len(gc) > 0 and (st == 1).any()
        # G
        nw = np.full(n, 20, np.int32)
        want = orc.c_walk_sample(Ep, g["b"], off, nbr, base, roots, slots, nw, False, 9, 2 * rnd + 1, dmax + 3)
        n1, n2 = [], []
        for wlk in range(len(nw) * 20):
            L = want["path_len"][wlk]
            if L > 0:
                for a, b in orc.pairs_from_path(list(want["paths"][wlk, :L]), 2):
                    n1.append(a)
                    n2.append(b)
        g1, g2, rew, st = eng.prepare_g(slots, 20, 9, 2 * rnd + 1)
        assert np.array_equal(st, want["root_status"])
        assert np.array_equal(g1, np.array(n1, np.int32)) and np.array_equal(g2, np.array(n2, np.int32))
        assert np.max(np.abs(rew - dis.reward(g1.astype(np.int64), g2.astype(np.int64)))) <= 1e-5
        # the whole-walk reward kernel (every row read once per walk) == the per-pair kernel, bit for bit
        os.environ["GG_NO_PATH_REWARD"] = "1"
        try:
            h1, h2, rew_pairs, _ = eng.prepare_g(slots, 20, 9, 2 * rnd + 1)
        finally:
            del os.environ["GG_NO_PATH_REWARD"]
        assert np.array_equal(h1, g1) and np.array_equal(h2, g2)
        assert np.array_equal(rew_pairs.view(np.uint32), rew.view(np.uint32))
        assert np.array_equal(eng.pair_reward(g1, g2).view(np.uint32), rew.view(np.uint32))
    eng.close()


def test_passes_equal_sequences_of_steps(ga):
    """gg_d_pass / gg_g_pass over resident rows == the reference's minibatch loops
    (graph_gan.py:149-157, 168-176) issued one sess.run-equivalent at a time."""
    g, n, graph, rowptr, col, Ed, eng = _setup_graph_engine(ga)
    _, _, _, _, _, _, eng2 = _setup_graph_engine(ga)
    slots = np.arange(n, dtype=np.int32)
    c, nb, lab, _ = eng.prepare_d(slots, 4, 0)
    eng2.prepare_d(slots, 4, 0, fetch=False)
    starts = np.arange(0, len(c), 64)
    np.random.RandomState(0).shuffle(starts)
    eng.d_pass(starts, 64)
    for s in starts:
        eng2.d_step(c[s:s + 64], nb[s:s + 64], lab[s:s + 64])
    assert np.allclose(eng.get_embeddings(1), eng2.get_embeddings(1), rtol=1e-6, atol=1e-7)
    n1, n2, rew, _ = eng.prepare_g(slots, 20, 4, 1)
    r2 = eng2.pair_reward(n1, n2)
    assert np.allclose(rew, r2, rtol=1e-6, atol=1e-7)
    starts = np.arange(0, len(n1), 64)[:200]
    eng.g_pass(starts, 64)
    for s in starts:
        eng2.g_step(n1[s:s + 64], n2[s:s + 64], rew[s:s + 64])
    assert np.allclose(eng.get_embeddings(0), eng2.get_embeddings(0), rtol=1e-6, atol=1e-7)
    with pytest.raises(ga.GraphGANHipError):
        eng.g_pass(np.array([len(n1) + 5]), 64)
    eng.close()
    eng2.close()


def test_state_save_restore(ga, tmp_path):
    n, d = 100, 50
    Eg, Ed, bg, bd = make_models(n, d, 1)
    eng = engine_with(ga, Eg, Ed, bg, bd)
    rs = np.random.RandomState(0)
    u, v = rs.randint(0, n, 64), rs.randint(0, n, 64)
    lab = (rs.rand(64) < 0.5).astype(np.float32)
    eng.d_step(u, v, lab)
    eng.g_step(u, v, lab)
    path = str(tmp_path / "model.ggst")
    eng.save_state(path)
    eng.d_step(u, v, lab)
    eng.g_step(v, u, lab)
    want = [eng.get_embeddings(0), eng.get_embeddings(1), eng.get_bias(0), eng.get_bias(1)]
    eng2 = engine_with(ga, Eg * 0, Ed * 0, bg * 0, bd * 0)
    eng2.load_state(path)
    eng2.d_step(u, v, lab)
    eng2.g_step(v, u, lab)
    got = [eng2.get_embeddings(0), eng2.get_embeddings(1), eng2.get_bias(0), eng2.get_bias(1)]
    for a, b in zip(want, got):
        assert np.allclose(a, b, rtol=1e-6, atol=1e-7)
    with pytest.raises(ga.GraphGANHipError):
        eng2.load_state(str(tmp_path / "missing"))
    # the save is atomic (temporary + rename: no half-written checkpoint at the final path) ...
    assert not os.path.exists(path + ".tmp")
    # ... and a truncated file is rejected BEFORE any device table changes
    blob = open(path, "rb").read()
    bad = str(tmp_path / "truncated.ggst")
    open(bad, "wb").write(blob[: len(blob) // 2])
    with pytest.raises(ga.GraphGANHipError) as ei:
        eng2.load_state(bad)
    assert ei.value.code == ga.GG_EIO and "nothing was loaded" in str(ei.value)
    for a, b in zip(got, [eng2.get_embeddings(0), eng2.get_embeddings(1), eng2.get_bias(0), eng2.get_bias(1)]):
        assert np.array_equal(a, b)
    eng.close()
    eng2.close()


@pytest.mark.parametrize("mode", ["sgd", "lazy"])
def test_generator_mean_uses_the_pairs_of_all_ranks(ga, mode, monkeypatch):
    """generator.py:28 takes the MEAN over the batch: with replicas the batch of a step is the union of the ranks'
    slices, so 1/B must count the pairs of ALL ranks (one 8-byte all-reduce per step).  Two simulated ranks holding
    the same 500 pairs (GG_COMM_FAKE_WORLD=2) must move the tables exactly like one rank stepping on the 1 000-pair
    union; with per-rank means the data term would be twice too large against the L2 term."""
    n, d = 300, 64
    Eg, Ed, bg, bd = make_models(n, d, 33)
    opt = ga.GG_OPT_SGD if mode == "sgd" else ga.GG_OPT_ADAM_LAZY
    rs = np.random.RandomState(2)
    u, v = rs.randint(0, n, 500), rs.randint(0, n, 500)
    r = (rs.rand(500) * 3).astype(np.float32)
    monkeypatch.setenv("GG_COMM_FAKE_WORLD", "2")
    two = engine_with(ga, Eg, Ed, bg, bd, optimizer=opt, lr_gen=0.05, lambda_gen=0.05)
    monkeypatch.delenv("GG_COMM_FAKE_WORLD")
    one = engine_with(ga, Eg, Ed, bg, bd, optimizer=opt, lr_gen=0.05, lambda_gen=0.05)
    two.g_step(u, v, r)
    one.g_step(np.concatenate([u, u]), np.concatenate([v, v]), np.concatenate([r, r]))
    assert np.abs(one.get_embeddings(0) - Eg).max() > 1e-3
    assert np.allclose(two.get_embeddings(0), one.get_embeddings(0), rtol=2e-5, atol=2e-6)
    assert np.allclose(two.get_bias(0), one.get_bias(0), rtol=2e-5, atol=2e-6)
    # and it is NOT what per-rank means would give
    gen = orc.Generator(Eg, 0.05, lazy=True)
    gen.b[:] = bg
    _, gu, gv, gb = gen.loss_and_grads(u, v, r, 0.05)
    assert np.abs(gu).max() > 0
    one.close()
    two.close()


@pytest.mark.parametrize("mode", ["dense", "lazy", "lazy-dense-fallback"])
def test_rccl_plumbing_with_one_rank_communicator(ga, mode, monkeypatch):
    """gpurun boxes have ONE GPU: a 1-rank RCCL communicator (GG_COMM_FORCE=1) still exercises
    dlopen(librccl), the enum values, the all-reduce on the engine's stream and, in lazy mode, the
    rebuild of the touched-row list from the reduced gradient.  Results must equal the no-comm run."""
    monkeypatch.setenv("GG_COMM_FORCE", "1")
    if mode.endswith("dense-fallback"):  # all-reduce of accumulators + int32 flag union instead of packs
        monkeypatch.setenv("GG_COMM_DENSE_RATIO", "0")
        mode = "lazy"
    n, d = 300, 50
    Eg, Ed, bg, bd = make_models(n, d, 2)
    opt = ga.GG_OPT_ADAM_DENSE if mode == "dense" else ga.GG_OPT_ADAM_LAZY
    a = engine_with(ga, Eg, Ed, bg, bd, optimizer=opt)
    b = engine_with(ga, Eg, Ed, bg, bd, optimizer=opt)
    b.comm_init(ga.Engine.comm_unique_id(), 0, 1)
    rs = np.random.RandomState(0)
    for t in range(3):
        u, v = rs.randint(0, n, 500), rs.randint(0, n, 500)
        lab = (rs.rand(500) < 0.5).astype(np.float32)
        a.d_step(u, v, lab)
        b.d_step(u, v, lab)
        a.g_step(v, u, lab)
        b.g_step(v, u, lab)
    b.comm_barrier()
    for which in (0, 1):
        assert np.allclose(a.get_embeddings(which), b.get_embeddings(which), rtol=1e-5, atol=1e-6)
        assert np.allclose(a.get_bias(which), b.get_bias(which), rtol=1e-5, atol=1e-6)
    a.close()
    b.close()


@pytest.mark.parametrize("opt", ["lazy", "dense"])
def test_whole_walk_g_pass_equals_generic_pair_kernel(ga, opt, monkeypatch):
    """A fused G pass over all prepared pairs takes the path-structured kernel (one row read and
    one gradient flush per path node); it must equal the generic per-pair kernel and the oracle's
    g_step on the same pair list."""
    optimizer = ga.GG_OPT_ADAM_LAZY if opt == "lazy" else ga.GG_OPT_ADAM_DENSE
    g, n, graph, rowptr, col, Ed, eng = _setup_graph_engine(ga, optimizer=optimizer)
    _, _, _, _, _, _, eng2 = _setup_graph_engine(ga, optimizer=optimizer)
    slots = np.arange(n, dtype=np.int32)
    n1, n2, rew, _ = eng.prepare_g(slots, 20, 4, 1)
    eng2.prepare_g(slots, 20, 4, 1, fetch=False)
    eng.g_pass([0], len(n1))                      # path-structured kernel
    monkeypatch.setenv("GG_NO_PATH_GRAD", "1")
    eng2.g_pass([0], len(n1))                     # generic pair kernel
    monkeypatch.delenv("GG_NO_PATH_GRAD")
    gen = orc.Generator(g["E"], 1e-3, lazy=(opt == "lazy"))
    gen.b[:] = g["b"]
    gen.g_step(n1.astype(np.int64), n2.astype(np.int64), rew, 1e-5)
    for e in (eng, eng2):
        assert np.allclose(e.get_embeddings(0), gen.E, rtol=2e-5, atol=2e-6)
        assert np.allclose(e.get_bias(0), gen.b, rtol=2e-5, atol=2e-6)
    c = eng.counters()
    assert c["g_pairs"] == len(n1) and c["g_steps"] == 1
    eng.close()
    eng2.close()


@pytest.mark.parametrize("opt", ["lazy", "sgd"])
@pytest.mark.parametrize("threshold", ["64", "3"])
def test_staged_generator_gradient_equals_the_atomic_path(ga, opt, threshold, monkeypatch):
    """The fused G pass stages its gradient rows (plain stores into per-row segments, summed by the optimizer kernel)
    instead of adding them with fp32 atomics; rows with more gradients than the threshold keep the atomic path.  Both
    ways, and the mix (threshold 3: most rows of this graph are "hubs"), give the oracle's update; two passes in a row
    check that every row's count is reset by whichever kernel updated it."""
    optimizer = ga.GG_OPT_ADAM_LAZY if opt == "lazy" else ga.GG_OPT_SGD
    monkeypatch.setenv("GG_STAGE_T", threshold)
    g, n, graph, rowptr, col, Ed, eng = _setup_graph_engine(ga, optimizer=optimizer)
    monkeypatch.setenv("GG_STAGE_T", "0")  # everything atomic
    _, _, _, _, _, _, eng2 = _setup_graph_engine(ga, optimizer=optimizer)
    slots = np.arange(n, dtype=np.int32)
    gen = orc.Generator(g["E"], 1e-3, lazy=True) if opt == "lazy" else None
    if gen is not None:
        gen.b[:] = g["b"]
    for rnd in range(2):
        n1, n2, rew, _ = eng.prepare_g(slots, 20, 4, 2 * rnd + 1)
        eng2.prepare_g(slots, 20, 4, 2 * rnd + 1, fetch=False)
        eng.g_pass([0], len(n1))
        eng2.g_pass([0], len(n1))
        assert np.allclose(eng.get_embeddings(0), eng2.get_embeddings(0), rtol=2e-5, atol=2e-6)
        assert np.allclose(eng.get_bias(0), eng2.get_bias(0), rtol=2e-5, atol=2e-6)
        if gen is not None and rnd == 0:
            gen.g_step(n1.astype(np.int64), n2.astype(np.int64), rew, 1e-5)
            assert np.allclose(eng.get_embeddings(0), gen.E, rtol=2e-5, atol=2e-6)
            assert np.allclose(eng.get_bias(0), gen.b, rtol=2e-5, atol=2e-6)
    # The staged sum is deterministic when no row is a hub: the slot of a gradient row inside its segment is the arrival
    # order of an atomic counter (run dependent), but the reducing kernel adds a segment in the order of the rows' SOURCES
    # (path position).  Four engines repeat the staged G pass bit for bit (threshold 100000 on this 90-node graph: segments
    # of hundreds of rows, the minimum-selection path; test_staged_sums_are_bit_reproducible covers the in-register ranking)
    if threshold == "64":
        monkeypatch.setenv("GG_STAGE_T", "100000")
        tabs = []
        for _ in range(4):
            _, _, _, _, _, _, e = _setup_graph_engine(ga, optimizer=optimizer)
            e.prepare_g(slots, 20, 4, 1, fetch=False)
            e.g_pass([0], 1 << 30)
            tabs.append((e.get_embeddings(0).view(np.uint32), e.get_bias(0).view(np.uint32)))
            e.close()
        for k in range(1, 4):
            assert np.array_equal(tabs[0][0], tabs[k][0]) and np.array_equal(tabs[0][1], tabs[k][1])
    eng.close()
    eng2.close()


@pytest.mark.parametrize("opt", ["lazy", "sgd"])
def test_staged_sums_are_bit_reproducible(ga, opt):
    """Fused D pass (>= 16 384 pairs) on a table large enough that no row collects more than the default 64 gradients:
    every row is staged and summed in source order -- segments of 1 .. 64 rows, i.e. one to four keys per lane in the
    ranking -- so repeated runs give bit-identical tables, and the result is the oracle's."""
    n, d = 12000, 40
    rs = np.random.RandomState(4)
    Eg = (rs.randn(n, d) * 0.3).astype(np.float32)
    Ed = (rs.randn(n, d) * 0.3).astype(np.float32)
    hot = rs.choice(n, 200, replace=False)                        # 200 rows with 30-50 gradients each, the rest 1-6
    v = np.concatenate([rs.randint(0, n, 14000), np.repeat(hot, 30)]).astype(np.int32)
    rs.shuffle(v)
    u = np.repeat(rs.randint(0, n, len(v) // 8 + 1), 8)[: len(v)].astype(np.int32)
    lab = (rs.rand(len(v)) < 0.5).astype(np.float32)
    c1, c2 = np.bincount(np.concatenate([u[::8], v]), minlength=n), np.bincount(np.concatenate([v, u]), minlength=n)
    assert 32 < c1.max() <= 64 and 32 < c2.max() <= 64 and len(v) >= 16384
    tabs = []
    for _ in range(4):
        e = ga.Engine(Eg, Ed, optimizer=ga.GG_OPT_ADAM_LAZY if opt == "lazy" else ga.GG_OPT_SGD)
        e.d_step(u, v, lab)
        e.d_step(v, u, lab)   # second pass: counts were reset, no runs on the u side
        tabs.append((e.get_embeddings(1), e.get_bias(1)))
        e.close()
    for k in range(1, 4):
        assert np.array_equal(tabs[0][0].view(np.uint32), tabs[k][0].view(np.uint32))
        assert np.array_equal(tabs[0][1].view(np.uint32), tabs[k][1].view(np.uint32))
    if opt == "lazy":
        dis = orc.Discriminator(Ed, 1e-3, lazy=True)
        dis.d_step(u.astype(np.int64), v.astype(np.int64), lab, 1e-5)
        dis.d_step(v.astype(np.int64), u.astype(np.int64), lab, 1e-5)
        assert np.allclose(tabs[0][0], dis.E, rtol=2e-5, atol=2e-6)
        assert np.allclose(tabs[0][1], dis.b, rtol=2e-5, atol=2e-6)


def test_fused_d_pass_wide_rows_keep_every_column(ga):
    """ld = 512 (the widest gradient kernel): a fused batch >= 16 384 pairs must update ALL columns -- the reducing kernel of
    the staged path holds 256 floats per row, so wider tables take the atomic kernels (columns >= 256 were
    silently dropped).  SGD, so that the comparison is linear in the gradient (a first Adam step is lr * sign(g): rows
    whose ~100 gradients cancel would flip on summation order)."""
    n, d = 400, 512
    rs = np.random.RandomState(2)
    Eg = (rs.randn(n, d) * 0.05).astype(np.float32)
    Ed = (rs.randn(n, d) * 0.05).astype(np.float32)
    eng = ga.Engine(Eg, Ed, optimizer=ga.GG_OPT_SGD)
    u = np.repeat(rs.randint(0, n, 2500), 8).astype(np.int64)
    v = rs.randint(0, n, len(u)).astype(np.int64)
    lab = (rs.rand(len(u)) < 0.5).astype(np.float32)
    dis = orc.Discriminator(Ed, 1e-3)
    _, gu, gv, gb = dis.loss_and_grads(u, v, lab, 1e-5)
    GE, Gb = np.zeros_like(dis.E, dtype=np.float64), np.zeros(n)
    np.add.at(GE, u, gu)
    np.add.at(GE, v, gv)
    np.add.at(Gb, v, gb)
    want_E, want_b = Ed - (1e-3 * GE).astype(np.float32), -(1e-3 * Gb).astype(np.float32)
    eng.d_step(u, v, lab)
    got = eng.get_embeddings(1)
    assert np.abs(got[:, 256:] - Ed[:, 256:]).max() > 1e-4
    assert np.allclose(got, want_E, rtol=2e-5, atol=1e-6)
    assert np.allclose(eng.get_bias(1), want_b, rtol=2e-5, atol=1e-6)
    eng.close()


@pytest.mark.parametrize("dense_ratio", ["100", "0"])
def test_staged_fused_passes_with_simulated_replicas(ga, dense_ratio, monkeypatch):
    """With replicas (GG_COMM_FAKE_WORLD=2: the exchange code runs, every simulated rank brings this rank's gradient) the
    staged rows are summed into the gradient accumulators and travel through the row packs (ratio 100) or the dense
    exchange (ratio 0) like an atomically accumulated gradient: same tables as the atomic path, for the whole-walk G pass
    and a fused D pass."""
    monkeypatch.setenv("GG_COMM_FAKE_WORLD", "2")
    monkeypatch.setenv("GG_COMM_DENSE_RATIO", dense_ratio)
    monkeypatch.setenv("GG_STAGE_T", "64")
    g, n, graph, rowptr, col, Ed, a = _setup_graph_engine(ga, optimizer=ga.GG_OPT_ADAM_LAZY)
    monkeypatch.setenv("GG_STAGE_T", "0")
    _, _, _, _, _, _, b = _setup_graph_engine(ga, optimizer=ga.GG_OPT_ADAM_LAZY)
    for k in ("GG_COMM_FAKE_WORLD", "GG_COMM_DENSE_RATIO", "GG_STAGE_T"):
        monkeypatch.delenv(k)
    slots = np.arange(n, dtype=np.int32)
    for e in (a, b):
        pairs = e.prepare_g(slots, 20, 4, 1, fetch=False)
        e.g_pass([0], pairs)
    assert np.abs(a.get_embeddings(0) - g["E"]).max() > 1e-4
    assert np.allclose(a.get_embeddings(0), b.get_embeddings(0), rtol=2e-5, atol=2e-6)
    assert np.allclose(a.get_bias(0), b.get_bias(0), rtol=2e-5, atol=2e-6)
    rs = np.random.RandomState(5)
    u, v = rs.randint(0, n, 20000), rs.randint(0, n, 20000)
    lab = (rs.rand(20000) < 0.5).astype(np.float32)
    for e in (a, b):
        e.d_step(u, v, lab)
    assert np.allclose(a.get_embeddings(1), b.get_embeddings(1), rtol=2e-5, atol=2e-6)
    assert np.allclose(a.get_bias(1), b.get_bias(1), rtol=2e-5, atol=2e-6)
    a.close()
    b.close()


@pytest.mark.parametrize("d", [50, 200, 256])
def test_fused_g_pass_embedding_widths(ga, d, monkeypatch):
    """Whole-walk reward + staged path gradient + reducing optimizer for rows that are not a multiple of 16 floats (d = 50:
    the reference's own width, 200) and for the widest template instance (d = 256), against the per-pair kernels."""
    g, n, graph = load_small(3)
    rs = np.random.RandomState(d)
    Eg = (rs.randn(n, d) * (1.2 / np.sqrt(d))).astype(np.float32)
    Ed = (rs.randn(n, d) * (1.2 / np.sqrt(d))).astype(np.float32)
    rowptr, col = ga.graph_to_csr(n, graph)
    engs = []
    for _ in range(2):
        e = ga.Engine(Eg, Ed, optimizer=ga.GG_OPT_ADAM_LAZY)
        e.set_graph_csr(rowptr, col)
        e.build_trees(np.arange(n))
        engs.append(e)
    slots = np.arange(n, dtype=np.int32)
    n1, n2, rew, _ = engs[0].prepare_g(slots, 20, 7, 1)
    monkeypatch.setenv("GG_NO_PATH_REWARD", "1")
    m1, m2, rew2, _ = engs[1].prepare_g(slots, 20, 7, 1)
    monkeypatch.delenv("GG_NO_PATH_REWARD")
    assert np.array_equal(n1, m1) and np.array_equal(n2, m2)
    assert np.array_equal(rew.view(np.uint32), rew2.view(np.uint32))
    engs[0].g_pass([0], len(n1))
    monkeypatch.setenv("GG_NO_PATH_GRAD", "1")
    engs[1].g_pass([0], len(n1))
    monkeypatch.delenv("GG_NO_PATH_GRAD")
    assert np.allclose(engs[0].get_embeddings(0), engs[1].get_embeddings(0), rtol=2e-5, atol=2e-6)
    assert np.allclose(engs[0].get_bias(0), engs[1].get_bias(0), rtol=2e-5, atol=2e-6)
    gen = orc.Generator(Eg, 1e-3, lazy=True)
    gen.g_step(n1.astype(np.int64), n2.astype(np.int64), rew, 1e-5)
    assert np.allclose(engs[0].get_embeddings(0), gen.E, rtol=2e-5, atol=2e-6)
    for e in engs:
        e.close()


def test_empty_draw_resets_the_resident_rows(ga):
    """update_ratio < 1 can select no root at all: gg_prepare_* with an empty slot list yields 0 rows / pairs and replaces
    what the previous call left resident (the trainer always makes the call -- with replicas it ends with a collective)."""
    g, n, graph, rowptr, col, Ed, eng = _setup_graph_engine(ga)
    slots = np.arange(n, dtype=np.int32)
    none = np.zeros(0, dtype=np.int32)
    assert eng.prepare_d(slots, 4, 4, fetch=False) > 0 and eng.prepare_g(slots, 20, 4, 5, fetch=False) > 0
    assert eng.prepare_d(none, 4, 6, fetch=False) == 0 and eng.prepare_g(none, 20, 4, 7, fetch=False) == 0
    before = [eng.get_embeddings(w).copy() for w in (0, 1)]
    eng.d_pass([], 64)
    eng.g_pass([], 64)
    eng.d_pass([0], 64)   # nothing resident: no stale row is trained on
    eng.g_pass([0], 1 << 30)
    for w in (0, 1):
        assert np.array_equal(before[w], eng.get_embeddings(w))
    eng.close()


def test_pairs_are_expanded_on_first_use_and_guarded(ga):
    """gg_prepare_g keeps the walks; the (node_1, node_2) arrays are written when something reads them.  A minibatch pass
    straight after an un-fetched prepare sees the same pairs as a fetched one; reading them after the walks were
    overwritten by another launch is refused instead of returning stale ids."""
    g, n, graph, rowptr, col, Ed, eng = _setup_graph_engine(ga)
    _, _, _, _, _, _, eng2 = _setup_graph_engine(ga)
    slots = np.arange(n, dtype=np.int32)
    n1, n2, rew, _ = eng.prepare_g(slots, 20, 4, 1)           # fetched: pairs expanded by the fetch
    pairs = eng2.prepare_g(slots, 20, 4, 1, fetch=False)      # not fetched
    assert pairs == len(n1)
    starts = np.arange(0, len(n1), 4096, dtype=np.int64)
    eng.g_pass(starts, 4096)
    eng2.g_pass(starts, 4096)                                  # minibatches: expands the pairs itself
    assert np.allclose(eng.get_embeddings(0), eng2.get_embeddings(0), rtol=2e-5, atol=2e-6)
    eng2.prepare_g(slots, 20, 4, 3, fetch=False)
    eng2.prepare_d(slots, 4, 4, fetch=False)                   # overwrites the walk buffers
    with pytest.raises(ga.GraphGANHipError):
        eng2.g_pass(starts, 4096)
    eng.close()
    eng2.close()


@pytest.mark.parametrize("n,d", [(300, 50), (1000, 128), (77, 6), (4100, 256)])
def test_all_score_rows_on_mfma(ga, n, d):
    """K7: rows of generator.all_score (generator.py:21) from the fp32 MFMA kernel: BIT-EXACT against
    the k-ordered fmaf chain of the oracle (the instruction's documented arithmetic), and within
    1e-5 of a float64 product; bias is added per column."""
    Eg, Ed, bg, bd = make_models(n, d, n)
    eng = engine_with(ga, Eg, Ed, bg, bd)
    rs = np.random.RandomState(1)
    rows = rs.choice(n, min(n, 70), replace=False).astype(np.int32)
    got = eng.all_score(rows)
    want = orc.c_all_score_rows(orc.pad_rows(Eg), bg, rows)
    assert np.array_equal(got, want)
    ref = Eg[rows].astype(np.float64) @ Eg.T.astype(np.float64) + bg.astype(np.float64)
    assert np.max(np.abs(got - ref)) < 1e-5 * max(1.0, np.abs(ref).max())
    if n <= 300:
        full = eng.all_score()
        assert full.shape == (n, n) and np.array_equal(full[rows], got)
        assert not np.allclose(full, full.T)  # column bias makes S asymmetric
    with pytest.raises(ga.GraphGANHipError):
        eng.all_score([n])
    eng.close()


@pytest.mark.parametrize("mode", ["sgd", "lazy", "sgd-dense-fallback", "lazy-staged", "sgd-staged-dense-fallback"])
def test_sparse_exchange_with_simulated_ranks(ga, mode, monkeypatch):
    """The replica gradient exchange of the lazy / sgd modes packs touched rows, all-gathers the
    packs and adds them in rank order.  gpurun has one GPU, so GG_COMM_FAKE_WORLD=3 feeds the
    3-rank code path with three copies of the local pack: the applied gradient must be exactly
    3x the local one (offsets, counts, per-rank launches, flag union all exercised)."""
    monkeypatch.setenv("GG_COMM_FAKE_WORLD", "3")
    # "-staged": batches large enough for the staged gradient (>= 16 384 pairs; every row counts as small): with replicas the
    # reducing kernel writes the row sums into the accumulators and the exchange takes them from there
    B = 2000
    if "staged" in mode:
        B = 20000
        monkeypatch.setenv("GG_STAGE_T", "1000000")
        mode = mode.replace("-staged", "")
    if mode.endswith("dense-fallback"):  # force the "replicas may touch most of the table" branch
        monkeypatch.setenv("GG_COMM_DENSE_RATIO", "0")
        mode = "sgd"
    else:  # fixed-capacity row packs (capacity min(N, 2 * batch) = 400 rows, 200 of them used: the rest is -1 padding)
        monkeypatch.setenv("GG_COMM_DENSE_RATIO", "100")
    n, d = 400, 64
    Eg, Ed, bg, bd = make_models(n, d, 21)
    opt = ga.GG_OPT_SGD if mode == "sgd" else ga.GG_OPT_ADAM_LAZY
    eng = engine_with(ga, Eg, Ed, bg, bd, optimizer=opt)
    monkeypatch.delenv("GG_COMM_FAKE_WORLD")
    monkeypatch.delenv("GG_COMM_DENSE_RATIO", raising=False)
    monkeypatch.delenv("GG_STAGE_T", raising=False)
    dis = orc.Discriminator(Ed, 1e-3, lazy=True)
    dis.b[:] = bd
    rs = np.random.RandomState(8)
    for t in range(3):
        u, v = rs.randint(0, n // 2, B), rs.randint(0, n // 2, B)
        lab = (rs.rand(B) < 0.5).astype(np.float32)
        _, gu, gv, gb = dis.loss_and_grads(u, v, lab, 1e-5)
        GE, Gb = np.zeros((n, d), np.float64), np.zeros(n)
        np.add.at(GE, u, gu)
        np.add.at(GE, v, gv)
        np.add.at(Gb, v, gb)
        GE, Gb = (3 * GE).astype(np.float32), (3 * Gb).astype(np.float32)
        rows = np.flatnonzero((np.abs(GE).sum(1) > 0) | (Gb != 0))
        if mode == "sgd":
            dis.E -= np.float32(1e-3) * GE
            dis.b -= np.float32(1e-3) * Gb
        else:
            dis.opt.step([dis.E, dis.b], [(rows, GE[rows]), (rows, Gb[rows])])
        eng.d_step(u, v, lab)
        assert np.allclose(eng.get_embeddings(1), dis.E, rtol=3e-5, atol=2e-6), t
        assert np.allclose(eng.get_bias(1), dis.b, rtol=3e-5, atol=2e-6), t
    assert np.array_equal(eng.get_embeddings(1)[n // 2:], Ed[n // 2:])
    eng.close()


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("mode", ["sgd", "lazy", "sgd-bf16", "lazy-bf16"])
def test_owner_partitioned_exchange_with_simulated_ranks(ga, mode, world, monkeypatch):
    """The owner-partitioned sparse exchange (round 4): touched rows travel to their owner (row mod P), the owner adds the
    sources in rank order, the reduced rows are gathered back and OVERWRITE the receivers' accumulator rows.  One GPU here:
    GG_COMM_FAKE_WORLD = P makes this rank play every owner in turn with P - 1 copies of its own rows as the other sources,
    so the applied gradient must be exactly P x the local one -- per-owner counting, segment offsets, slot assignment, the
    rank-order adds, the gather pack and the overwrite kernel all run.  The branch is forced (dense ratio 0: the row packs are
    ruled out) and must really be taken (statistics).
    -bf16 (GG_COMM_BF16=1, round 5): the rows travel as bf16 -- half the bytes on xGMI -- and are summed in fp32 at the owner:
    own fp32 contribution + (P - 1) x bf16(contribution) in rank order, the sum rounded to bf16 once more for the way back (the
    owner keeps the rounded sum too: all replicas hold the same bits).  The oracle restates exactly that; the engine's LOCAL
    gradient is an atomic sum whose last fp32 bit may differ from numpy's, which moves a bf16 rounding in a few elements: those may
    be off by one bf16 step (2^-8 relative), everything else agrees to the fp32 tolerances."""
    bf16 = mode.endswith("-bf16")
    mode = mode.split("-")[0]
    monkeypatch.setenv("GG_COMM_FAKE_WORLD", str(world))
    monkeypatch.setenv("GG_COMM_DENSE_RATIO", "0")
    monkeypatch.setenv("GG_COMM_OWNER_MIN", "0")
    if bf16:
        monkeypatch.setenv("GG_COMM_BF16", "1")
    n, d = 1200, 64
    Eg, Ed, bg, bd = make_models(n, d, 33)
    opt = ga.GG_OPT_SGD if mode == "sgd" else ga.GG_OPT_ADAM_LAZY
    eng = engine_with(ga, Eg, Ed, bg, bd, optimizer=opt)
    for k in ("GG_COMM_FAKE_WORLD", "GG_COMM_DENSE_RATIO", "GG_COMM_OWNER_MIN"):
        monkeypatch.delenv(k)
    if bf16:
        monkeypatch.delenv("GG_COMM_BF16")
    dis = orc.Discriminator(Ed, 1e-3, lazy=True)
    dis.b[:] = bd
    rs = np.random.RandomState(world)
    B = 700

    def exchanged(local):  # what every replica holds after the exchange, from this rank's fp32 gradient
        if not bf16:
            return (world * local).astype(np.float32)
        local = local.astype(np.float32)
        piece = _bf16_round(local)
        tot = local.copy()
        for _ in range(world - 1):
            tot = (tot + piece).astype(np.float32)
        return _bf16_round(tot).reshape(local.shape)

    moved = 0.0
    for t in range(3):
        u, v = rs.randint(0, n // 4, B), rs.randint(0, n // 4, B)   # ~ 270 of the 1 200 rows touched: P x that stays below the table
        lab = (rs.rand(B) < 0.5).astype(np.float32)
        _, gu, gv, gb = dis.loss_and_grads(u, v, lab, 1e-5)
        GE, Gb = np.zeros((n, d), np.float64), np.zeros(n)
        np.add.at(GE, u, gu)
        np.add.at(GE, v, gv)
        np.add.at(Gb, v, gb)
        plain = (world * GE).astype(np.float32)
        GE, Gb = exchanged(GE), exchanged(Gb)
        moved = max(moved, float(np.abs(GE - plain).max()))
        rows = np.flatnonzero((np.abs(GE).sum(1) > 0) | (Gb != 0))
        if mode == "sgd":
            dis.E -= np.float32(1e-3) * GE
            dis.b -= np.float32(1e-3) * Gb
        else:
            dis.opt.step([dis.E, dis.b], [(rows, GE[rows]), (rows, Gb[rows])])
        eng.d_step(u, v, lab)
        for got, want, gmax in ((eng.get_embeddings(1), dis.E, np.abs(GE).max()), (eng.get_bias(1), dis.b, np.abs(Gb).max())):
            if not bf16:
                assert np.allclose(got, want, rtol=3e-5, atol=2e-6), t
            else:
                off = np.abs(got - want) > 3e-5 * np.abs(want) + 2e-6
                assert off.mean() < 0.02, (t, off.mean())
                # one bf16 step of the largest gradient through the step size (lazy Adam's first steps move by ~lr whatever the size)
                assert np.abs(got - want).max() <= (1e-3 * gmax * 2.0 ** -7 if mode == "sgd" else 2.5e-3), t
        if bf16:   # later steps start from the engine's own tables: rounding flips do not accumulate into the comparison
            dis.E[:] = eng.get_embeddings(1)
            dis.b[:] = eng.get_bias(1)
    assert (moved > 1e-3) == bf16          # the rounding is really there (and only there)
    assert np.array_equal(eng.get_embeddings(1)[n // 4:], Ed[n // 4:])
    st = eng.comm_stats()
    assert st["sparse_steps"] == 3 and st["dense_steps"] == 0   # the owner path ran every time (it counts as a sparse step)
    eng.close()


def test_small_batches_are_atomic_free_and_reproducible(ga, monkeypatch):
    """The reference's batches (<= 256 pairs: batch 64, graph_gan.py:149-157,168-176) run the atomic-free single-workgroup
    gradient kernel BY DEFAULT: two engines give BIT-IDENTICAL tables after many steps with heavy row duplication, and the
    result stays within tolerance of the atomic kernel (GG_DETERMINISTIC=0) and of the oracle."""
    n, d = 300, 50
    Eg, Ed, bg, bd = make_models(n, d, 9)
    a = engine_with(ga, Eg, Ed, bg, bd)
    b = engine_with(ga, Eg, Ed, bg, bd)
    monkeypatch.setenv("GG_DETERMINISTIC", "0")
    c = engine_with(ga, Eg, Ed, bg, bd)  # atomic mode
    monkeypatch.delenv("GG_DETERMINISTIC")
    gen, dis = orc.Generator(Eg, 1e-3), orc.Discriminator(Ed, 1e-3)
    gen.b[:] = bg
    dis.b[:] = bd
    rs = np.random.RandomState(4)
    for t in range(25):
        B = (64, 37, 256, 130, 1)[t % 5]
        u, v = rs.randint(0, n, B), rs.randint(0, n, B)
        u[: B // 2] = u[0]
        v[B // 2:] = rs.randint(0, 5, B - B // 2)
        lab = (rs.rand(B) < 0.5).astype(np.float32)
        rew = (rs.rand(B) * 2).astype(np.float32)
        for e in (a, b, c):
            e.d_step(u, v, lab)
            e.g_step(v, u, rew)
        if t == 0:
            dis.d_step(u, v, lab, 1e-5)
            gen.g_step(v, u, rew, 1e-5)
            assert np.allclose(a.get_embeddings(1), dis.E, rtol=1e-6, atol=2e-7)
            assert np.allclose(a.get_embeddings(0), gen.E, rtol=1e-6, atol=2e-7)
    for which in (0, 1):
        assert np.array_equal(a.get_embeddings(which), b.get_embeddings(which))
        assert np.array_equal(a.get_bias(which), b.get_bias(which))
        diff = np.abs(a.get_embeddings(which) - c.get_embeddings(which))
        assert np.quantile(diff, 0.99) < 1e-4
    for e in (a, b, c):
        e.close()


@pytest.mark.parametrize("d", [50, 128, 256, 300])
def test_atomic_free_kernel_row_widths(ga, d):
    """The atomic-free kernel keeps the batch's rows in LDS when 2 n rows fit (batch 64 up to d = 288), and re-reads the
    partner rows from the table when they do not (n = 256 at d >= 128; every n at d = 300): both against the oracle's step,
    duplicates on both sides of the pairs included."""
    n = 500
    Eg, Ed, bg, bd = make_models(n, d, 21)
    eng = engine_with(ga, Eg, Ed, bg, bd)
    gen, dis = orc.Generator(Eg, 1e-3), orc.Discriminator(Ed, 1e-3)
    gen.b[:] = bg
    dis.b[:] = bd
    rs = np.random.RandomState(6)
    for B in (64, 256, 3):
        u, v = rs.randint(0, n, B), rs.randint(0, 40, B)
        u[::3] = u[0]
        lab = (rs.rand(B) < 0.5).astype(np.float32)
        rew = (rs.rand(B) * 2).astype(np.float32)
        dis.d_step(u, v, lab, 1e-5)
        gen.g_step(v, u, rew, 1e-5)
        eng.d_step(u, v, lab)
        eng.g_step(v, u, rew)
        assert np.allclose(eng.get_embeddings(1), dis.E, rtol=3e-5, atol=2e-6), B
        assert np.allclose(eng.get_bias(1), dis.b, rtol=3e-5, atol=2e-6), B
        assert np.allclose(eng.get_embeddings(0), gen.E, rtol=3e-5, atol=2e-6), B
        assert np.allclose(eng.get_bias(0), gen.b, rtol=3e-5, atol=2e-6), B
    eng.close()


@pytest.mark.parametrize("opt", ["dense", "lazy"])
def test_atomic_free_kernel_batch_shapes(ga, opt):
    """Edge shapes of the reference's batches through the atomic-free kernel, each against the oracle's step: every size around the
    16-slot words and the 64-pair boundary of the register-resident id list (1 .. 256 pairs), one centre for the whole batch (a
    discriminator slice: graph_gan.py:193-201), one neighbour for the whole batch, SELF pairs (u == v: the reference's self-loop
    positives, utils.py:36-37 -- both slots of the pair name one row), and a batch in which every pair is the same pair."""
    n, d = 400, 50
    Eg, Ed, bg, bd = make_models(n, d, 17)
    lazy = opt == "lazy"
    eng = engine_with(ga, Eg, Ed, bg, bd, optimizer=ga.GG_OPT_ADAM_LAZY if lazy else ga.GG_OPT_ADAM_DENSE)
    gen, dis = orc.Generator(Eg, 1e-3, lazy=lazy), orc.Discriminator(Ed, 1e-3, lazy=lazy)
    gen.b[:] = bg
    dis.b[:] = bd
    rs = np.random.RandomState(12)
    shapes = []
    for B in (1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256):
        shapes.append((rs.randint(0, n, B), rs.randint(0, n, B)))
    shapes.append((np.full(64, 7), rs.randint(0, n, 64)))            # one centre
    shapes.append((rs.randint(0, n, 64), np.full(64, 9)))            # one neighbour
    u = rs.randint(0, n, 64)
    shapes.append((u, u.copy()))                                     # self pairs only
    u = rs.randint(0, 20, 200)
    v = rs.randint(0, 20, 200)
    v[::5] = u[::5]                                                  # a fifth of them self pairs, heavy duplication
    shapes.append((u, v))
    shapes.append((np.full(37, 3), np.full(37, 5)))                  # the same pair 37 times
    shapes.append((np.full(64, 11), np.full(64, 11)))                # the same SELF pair 64 times
    for t, (u, v) in enumerate(shapes):
        B = len(u)
        lab = (rs.rand(B) < 0.5).astype(np.float32)
        rew = (rs.rand(B) * 2).astype(np.float32)
        dis.d_step(u, v, lab, 1e-5)
        gen.g_step(v, u, rew, 1e-5)
        eng.d_step(u, v, lab)
        eng.g_step(v, u, rew)
        for which, model in ((1, dis), (0, gen)):
            assert np.allclose(eng.get_embeddings(which), model.E, rtol=3e-5, atol=3e-6), (t, B, which)
            assert np.allclose(eng.get_bias(which), model.b, rtol=3e-5, atol=3e-6), (t, B, which)
    eng.close()


@pytest.mark.parametrize("mode", ["lazy", "sgd"])
def test_small_batch_steps_apply_the_optimizer_in_the_gradient_kernel(ga, mode, monkeypatch):
    """The strict schedule at scale (batch 64 with lazy Adam / SGD on one replica): ONE launch per step -- the owner of a row in the
    atomic-free gradient kernel applies the optimizer to it.  Against the oracle's lazy Adam / plain SGD, and BIT-IDENTICAL to the
    unfused sequence gradient kernel -> flag compaction -> sparse_opt_kernel (GG_NO_FUSED_SMALL_STEP=1)."""
    n, d = 2000, 128
    Eg, Ed, bg, bd = make_models(n, d, 13)
    opt = ga.GG_OPT_ADAM_LAZY if mode == "lazy" else ga.GG_OPT_SGD
    fused = engine_with(ga, Eg, Ed, bg, bd, optimizer=opt)
    plain = engine_with(ga, Eg, Ed, bg, bd, optimizer=opt)
    dis = orc.Discriminator(Ed, 1e-3, lazy=True)
    gen = orc.Generator(Eg, 1e-3, lazy=True)
    dis.b[:] = bd
    gen.b[:] = bg
    rs = np.random.RandomState(8)
    for t in range(12):
        B = (64, 64, 17, 1)[t % 4]
        u, v = rs.randint(0, n // 2, B), rs.randint(0, n // 2, B)  # rows >= n/2 stay untouched
        u[: B // 3] = u[0]
        v[B // 2:] = v[-1]
        lab = (rs.rand(B) < 0.5).astype(np.float32)
        rew = (rs.rand(B) * 2).astype(np.float32)
        fused.d_step(u, v, lab)
        fused.g_step(v, u, rew)
        monkeypatch.setenv("GG_NO_FUSED_SMALL_STEP", "1")
        plain.d_step(u, v, lab)
        plain.g_step(v, u, rew)
        monkeypatch.delenv("GG_NO_FUSED_SMALL_STEP")
        if mode == "lazy":
            dis.d_step(u, v, lab, 1e-5)
            gen.g_step(v, u, rew, 1e-5)
        else:
            for model, (a, b, x), is_d in ((dis, (u, v, lab), True), (gen, (v, u, rew), False)):
                _, gu, gv, gb = model.loss_and_grads(a, b, x, 1e-5)
                GE, Gb = np.zeros_like(model.E, dtype=np.float64), np.zeros(n)
                np.add.at(GE, a, gu)
                np.add.at(GE, b, gv)
                np.add.at(Gb, b, gb)
                model.E -= (1e-3 * GE).astype(np.float32)
                model.b -= (1e-3 * Gb).astype(np.float32)
        for which, model in ((1, dis), (0, gen)):
            assert np.allclose(fused.get_embeddings(which), model.E, rtol=2e-5, atol=1e-6), (t, which)
            assert np.allclose(fused.get_bias(which), model.b, rtol=2e-5, atol=1e-6), (t, which)
    for which in (0, 1):
        assert np.array_equal(fused.get_embeddings(which), plain.get_embeddings(which))
        assert np.array_equal(fused.get_bias(which), plain.get_bias(which))
        assert np.array_equal(fused.get_embeddings(which)[n // 2:], (Eg, Ed)[which][n // 2:])
    assert fused.counters()["d_steps"] == 12 and fused.counters()["g_steps"] == 12
    fused.close()
    plain.close()


def test_strict_inner_pass_tables_match_the_oracle(ga):
    """Table-level float parity over a WHOLE inner pass of the reference's schedule on CA-GrQc (graph_gan.py:144-176, batch 64,
    dense TF1 Adam): prepare_data_for_d -> every D minibatch of the pass in shuffled order (131 optimizer steps) -> prepare_data_for_g
    -> the first 400 G minibatches, engine against the numpy restatement on the same integer data.  Gate: max |difference|
    <= 1e-5 over all four tables (the updates are ~1e-3 per step; the two sides differ in fp32 summation order and in the last
    bit of exp / sqrt), and a second engine gives the SAME BITS (no atomics on this path)."""
    d0, n, graph = load_ca_grqc()
    init = ca_grqc_init_embeddings(d0, n, seed=0).astype(np.float32)
    rowptr, col = ga.graph_to_csr(n, graph)
    slots = np.arange(n, dtype=np.int32)
    engines = []
    for _ in range(2):
        eng = ga.Engine(init, init)
        eng.set_graph_csr(rowptr, col)
        eng.build_trees(np.arange(n))
        engines.append(eng)
    gen, dis = orc.Generator(init, 1e-3), orc.Discriminator(init, 1e-3)
    c, nb, lab, _ = engines[0].prepare_d(slots, 31, 0)
    engines[1].prepare_d(slots, 31, 0, fetch=False)
    starts = np.arange(0, len(c), 64)
    np.random.RandomState(1).shuffle(starts)
    assert len(starts) >= 130
    for e in engines:
        e.d_pass(starts, 64)
    c64, nb64 = c.astype(np.int64), nb.astype(np.int64)
    for s in starts:
        dis.d_step(c64[s:s + 64], nb64[s:s + 64], lab[s:s + 64], 1e-5)
    n1, n2, rew, _ = engines[0].prepare_g(slots, 20, 31, 1)
    engines[1].prepare_g(slots, 20, 31, 1, fetch=False)
    assert np.max(np.abs(rew - dis.reward(n1.astype(np.int64), n2.astype(np.int64)))) <= 1e-5  # rewards of the UPDATED discriminator
    starts = np.arange(0, len(n1), 64)
    np.random.RandomState(2).shuffle(starts)
    starts = starts[:400]
    for e in engines:
        e.g_pass(starts, 64)
    a64, b64 = n1.astype(np.int64), n2.astype(np.int64)
    for s in starts:
        gen.g_step(a64[s:s + 64], b64[s:s + 64], rew[s:s + 64], 1e-5)
    worst = 0.0
    for which, model in ((0, gen), (1, dis)):
        E, b = engines[0].get_embeddings(which), engines[0].get_bias(which)
        worst = max(worst, float(np.abs(E - model.E).max()), float(np.abs(b - model.b).max()))
        assert np.abs(model.E - init).max() > 1e-2     # the pass moved the table
        assert np.array_equal(E, engines[1].get_embeddings(which)) and np.array_equal(b, engines[1].get_bias(which))
    print("strict inner pass: max |engine - oracle| over the four tables = %.3g" % worst)
    assert worst <= 1e-5
    for e in engines:
        e.close()


@pytest.mark.parametrize("every", [0, 3])
def test_sampled_profiling_and_early_returning_passes(ga, every, monkeypatch):
    """gg_set_profiling(k != 1): events only on every k-th walk launch, gg_*_pass return before their kernels have
    finished (stream-ordered).  Results and work counters are those of the default (every launch timed,
    synchronous passes) mode; the timing counters cover exactly the profiled launches."""
    # (batch-64 steps are atomic-free by default: the two runs are comparable bit for bit)
    # every stale node is scored whole exactly once: rows_scored is then a pure function of the walks (under the default
    # policy it also depends on which of two racing roots asks for a node first)
    monkeypatch.setenv("GG_ES_MODE", "2")
    g, n, graph, rowptr, col, Ed, eng = _setup_graph_engine(ga)
    _, _, _, _, _, _, ref = _setup_graph_engine(ga)
    eng.set_profiling(every)
    slots = np.arange(n, dtype=np.int32)
    for e in (eng, ref):
        for it in range(3):
            rows = e.prepare_d(slots, 4, 2 * it, fetch=False)
            starts = np.arange(0, rows, 64)
            e.d_pass(starts, 64)
            pairs = e.prepare_g(slots, 20, 4, 2 * it + 1, fetch=False)
            e.g_pass(np.arange(0, pairs, 64)[:150], 64)
    eng.synchronize()
    ca, cb = eng.counters(), ref.counters()
    for k in ("walks", "hops", "nbr_reads", "rows_scored", "d_pairs", "g_pairs", "d_steps", "g_steps", "reward_pairs"):
        assert ca[k] == cb[k], k
    assert cb["walk_launches"] == 6 and ca["walk_launches"] == (2 if every == 3 else 0)
    assert (ca["score_launches"] > 0) == (every == 3) and (ca["score_rows"] > 0) == (every == 3)
    # every launch profiled: the rows of the timed score kernels are all rows but the few the per-walk finisher scores itself
    # (round 4: walks still going behind the learned number of levels are finished by it instead of rerunning the launch)
    assert 0.9 * cb["rows_scored"] <= cb["score_rows"] <= cb["rows_scored"]
    for which in (0, 1):
        assert np.array_equal(eng.get_embeddings(which), ref.get_embeddings(which))
    with pytest.raises(ga.GraphGANHipError):
        eng.set_profiling(-1)
    eng.close()
    ref.close()


def _bf16_round(x):
    """fp32 -> bf16 (round to nearest even) -> fp32, like v_cvt_pk_bf16_f32"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("n,d", [(700, 50), (3000, 128), (1029, 256), (300, 8), (600, 300)])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_all_score_streamed_consumer(ga, n, d, precision):
    """generator.all_score (generator.py:21) streamed through the fused consumer (max, argmax, log-sum-exp per row; nothing
    of size rows x N materialised) against numpy on the materialised rows.  fp32: the matrix-core scores are exact fp32
    (max equal to the oracle's k-ordered chain, argmax identical, lse to 1e-5); bf16: against numpy on the bf16-rounded
    table (fp32 accumulation order differs: 2e-3 abs on scores of magnitude ~10)."""
    Eg, Ed, bg, bd = make_models(n, d, n + d)
    Eg = Eg * np.float32(1.5)
    eng = engine_with(ga, Eg, Ed, bg, bd)
    rows = np.unique(np.random.RandomState(3).randint(0, n, 150)).astype(np.int32)
    res = eng.all_score_reduce(rows, precision=precision)
    ref = _bf16_round(Eg) if precision == "bf16" else Eg
    S = (ref[rows].astype(np.float64) @ ref.T.astype(np.float64)) + bg.astype(np.float64)[None, :]
    tol = 2e-3 if precision == "bf16" else 2e-5
    assert np.max(np.abs(res["max"] - S.max(1))) <= tol * max(1.0, np.abs(S).max())
    lse = S.max(1) + np.log(np.exp(S - S.max(1, keepdims=True)).sum(1))
    assert np.max(np.abs(res["logsumexp"] - lse)) <= tol * max(1.0, np.abs(S).max())
    # argmax: identical unless two scores are closer than the tolerance
    am = S.argmax(1)
    close = np.abs(S[np.arange(len(rows)), res["argmax"]] - S.max(1)) <= tol * max(1.0, np.abs(S).max())
    assert close.all() and (res["argmax"] == am).mean() > (0.97 if precision == "bf16" else 0.999)
    if precision == "fp32":
        want = orc.c_all_score_rows(orc.pad_rows(Eg), bg, rows)  # the oracle's exact fp32 rows
        assert np.array_equal(res["max"], want.max(1)) and np.array_equal(res["argmax"], want.argmax(1))
    all_rows = eng.all_score_reduce(None, precision=precision, logsumexp=False)
    assert np.array_equal(all_rows["argmax"][rows], res["argmax"]) and all_rows["logsumexp"] is None
    assert res["kernel_ms"] > 0
    if precision == "bf16" and n >= 512:
        # many rows: the workgroup's wavefronts take four row blocks and share the table sweep; same scores, same answers
        wide = eng.all_score_reduce(None, precision=precision)
        os.environ["GG_ALLPAIRS_NARROW"] = "1"
        try:
            narrow = eng.all_score_reduce(None, precision=precision)
        finally:
            del os.environ["GG_ALLPAIRS_NARROW"]
        assert np.array_equal(wide["max"], narrow["max"]) and np.array_equal(wide["argmax"], narrow["argmax"])
        assert np.allclose(wide["logsumexp"], narrow["logsumexp"], rtol=1e-5, atol=1e-5)
        assert np.array_equal(wide["argmax"][rows], res["argmax"])
        # ... and for a LIST of >= 512 requested rows (the bench's call: the row ids go through the kernel's index array;
        # a last row tile that is not full)
        some = np.random.RandomState(8).permutation(n)[:523].astype(np.int32)
        listed = eng.all_score_reduce(some, precision=precision)
        assert np.array_equal(listed["max"], narrow["max"][some]) and np.array_equal(listed["argmax"], narrow["argmax"][some])
        assert np.allclose(listed["logsumexp"], narrow["logsumexp"][some], rtol=1e-5, atol=1e-5)
    eng.close()


def test_all_score_wide_consumer_falls_back_when_its_sum_leaves_the_fp32_range(ga):
    """The wide bf16 consumer accumulates sum_j exp(S[i, j]) without a running reference (cheaper per score); scores beyond
    +-85 overflow / underflow that fp32 sum: the kernel raises a flag and the call is repeated with the running-max kernel --
    same max / argmax / log-sum-exp as the narrow path asked for directly."""
    n, d = 1200, 64
    rs = np.random.RandomState(12)
    Eg = (rs.randn(n, d) * 2.0).astype(np.float32)         # scores ~ N(0, 32): the row maxima are far beyond 85
    bg = (rs.randn(n) * 0.1).astype(np.float32)
    eng = engine_with(ga, Eg, Eg, bg, bg)
    wide = eng.all_score_reduce(None, precision="bf16")
    os.environ["GG_ALLPAIRS_NARROW"] = "1"
    try:
        narrow = eng.all_score_reduce(None, precision="bf16")
    finally:
        del os.environ["GG_ALLPAIRS_NARROW"]
    assert wide["max"].max() > 120.0
    assert np.array_equal(wide["max"], narrow["max"]) and np.array_equal(wide["argmax"], narrow["argmax"])
    assert np.isfinite(wide["logsumexp"]).all() and np.allclose(wide["logsumexp"], narrow["logsumexp"], rtol=1e-6, atol=1e-5)
    # very negative scores: the reference-free sum underflows to 0 -> same fall-back
    eng.set_bias(0, np.full(n, -400.0, np.float32))
    low = eng.all_score_reduce(None, precision="bf16")
    ref = _bf16_round(Eg)
    S = ref.astype(np.float64) @ ref.T.astype(np.float64) - 400.0
    lse = S.max(1) + np.log(np.exp(S - S.max(1, keepdims=True)).sum(1))
    assert np.isfinite(low["logsumexp"]).all() and np.max(np.abs(low["logsumexp"] - lse)) <= 2e-3 * np.abs(S).max()
    eng.close()


def test_evaluator_scores_on_the_device_and_binary_sidecar(ga, tmp_path):
    """src/evaluation/link_prediction.py:19-38 with the per-edge dots computed by gg_edge_scores: the accuracy of the shipped
    pre-trained embeddings is the reference's 0.7598343685300207 to the digit, scores equal np.dot to 1e-12, and the
    binary side-car holds exactly the numbers of the text file."""
    from graphgan_amd.evaluation import link_prediction as lp
    from graphgan_amd import utils
    d, n, graph = load_ca_grqc()
    emb = ca_grqc_init_embeddings(d, n).astype(np.float32)
    eng = ga.Engine(emb, emb * np.float32(0.5))
    te, neg = str(tmp_path / "test.txt"), str(tmp_path / "neg.txt")
    for path, key in ((te, "test"), (neg, "test_neg")):
        with open(path, "w") as f:
            f.writelines("%d\t%d\n" % (a, b) for a, b in d[key].tolist())
    acc_dev = lp.LinkPredictEval("unused", te, neg, n, 50, engine=eng, which=0).eval_link_prediction()
    acc_ref = lp.LinkPredictEval("unused", te, neg, n, 50, emd=emb.astype(np.float64)).eval_link_prediction()
    assert acc_dev == acc_ref == 0.7598343685300207
    edges = np.array(d["test"].tolist() + d["test_neg"].tolist())
    got = eng.edge_scores(1, edges[:, 0], edges[:, 1])
    e64 = (emb * np.float32(0.5)).astype(np.float64)
    want = np.array([np.dot(e64[a], e64[b]) for a, b in edges])
    assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, np.abs(want).max())
    with pytest.raises(ga.GraphGANHipError):
        eng.edge_scores(0, [0], [n])
    txt, bn = str(tmp_path / "gen.emb"), str(tmp_path / "gen.emb.bin")
    eng.write_embeddings(0, txt)
    eng.write_embeddings_bin(0, bn)
    assert os.path.getsize(bn) == 20 + 4 * n * 50 and os.path.getsize(txt) > 3 * os.path.getsize(bn)
    back = utils.read_embeddings_bin(bn)
    assert back.dtype == np.float32 and np.array_equal(back, emb)
    np.random.seed(0)
    assert np.array_equal(utils.read_embeddings(txt, n, 50), emb.astype(np.float64))
    eng.close()
