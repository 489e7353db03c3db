"""End-to-end on CA-GrQc (BASELINE.json configs[1]): the ``graph_gan.py`` mirror on the HIP
engine against the oracle trainer with the same seed, schedule and optimizer mode.
Gates: epoch-0 line == 0.7598343685300207; integer sample data identical;
final gen/dis accuracy within +-0.5 % absolute; output file formats as the reference writes them."""
import os
import subprocess
import sys
import types

import numpy as np
import pytest

from oracle import graphgan_oracle as orc
from tests.helpers import load_ca_grqc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_reference_layout(base):
    """materialise the CA-GrQc fixture in the reference's directory layout under ``base``"""
    d, n, graph = load_ca_grqc()
    os.makedirs(os.path.join(base, "data", "link_prediction"))
    os.makedirs(os.path.join(base, "pre_train", "link_prediction"))
    for name, key in (("train", "train"), ("test", "test"), ("test_neg", "test_neg")):
        with open(os.path.join(base, "data", "link_prediction", "CA-GrQc_%s.txt" % name), "w") as f:
            f.writelines("%d\t%d\n" % (a, b) for a, b in d[key].tolist())
    with open(os.path.join(base, "pre_train", "link_prediction", "CA-GrQc_pre_train.emb"), "w") as f:
        f.write("%d %d\n" % (len(d["emb_ids"]), d["emb_rows"].shape[1]))
        for i, row in zip(d["emb_ids"].tolist(), d["emb_rows"].astype(np.float64).tolist()):
            f.write(str(i) + " " + " ".join(repr(x) for x in row) + "\n")
    return d, n, graph


def make_cfg(base, **over):
    from graphgan_amd import config as base_cfg
    cfg = types.SimpleNamespace(**{k: getattr(base_cfg, k) for k in dir(base_cfg) if not k.startswith("_")})
    app, ds = cfg.app, cfg.dataset
    cfg.train_filename = "%s/data/%s/%s_train.txt" % (base, app, ds)
    cfg.test_filename = "%s/data/%s/%s_test.txt" % (base, app, ds)
    cfg.test_neg_filename = "%s/data/%s/%s_test_neg.txt" % (base, app, ds)
    cfg.pretrain_emb_filename_d = cfg.pretrain_emb_filename_g = "%s/pre_train/%s/%s_pre_train.emb" % (base, app, ds)
    cfg.emb_filenames = ["%s/results/%s/%s_gen_.emb" % (base, app, ds), "%s/results/%s/%s_dis_.emb" % (base, app, ds)]
    cfg.result_filename = "%s/results/%s/%s.txt" % (base, app, ds)
    cfg.model_log = "%s/log/" % base
    cfg.cache_filename = "%s/cache/%s.pkl" % (base, ds)  # used only when the directory exists (README.md:43: `mkdir cache`)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def test_reduced_schedule_matches_oracle(tmp_path):
    base = str(tmp_path)
    d, n, graph = write_reference_layout(base)
    cfg = make_cfg(base, n_epochs=1, n_epochs_dis=2, n_epochs_gen=2, dis_interval=2, gen_interval=2, engine_seed=17)
    from graphgan_amd.graph_gan import GraphGAN
    np.random.seed(123)
    g = GraphGAN(cfg)
    assert g.n_node == 5242
    subset = list(range(0, n, 7))
    g.root_nodes = subset
    g.trees = g.construct_trees(subset)
    g.train()

    # ---- the oracle, same init, same schedule, same seed, dense TF1-Adam
    ocfg = orc.Config()
    ocfg.n_epochs_dis = ocfg.n_epochs_gen = ocfg.dis_interval = ocfg.gen_interval = 2
    o = orc.GraphGANOracle(n, graph, g.node_embed_init_g, g.node_embed_init_d, cfg=ocfg, rng="counter", arith="spec", seed=17)
    o.root_nodes = subset
    o.train_epoch(0)

    eg, ed = g.generator.embedding_matrix, g.discriminator.embedding_matrix
    # trajectories agree closely (different fp32 summation orders inside 1.5k Adam steps)
    for got, want in ((eg, o.generator.E), (ed, o.discriminator.E)):
        diff = np.abs(got - want)
        print("abs diff: mean %.3g p99 %.3g p99.99 %.3g max %.3g" % (diff.mean(), np.quantile(diff, 0.99), np.quantile(diff, 0.9999), diff.max()))
        assert diff.mean() < 5e-5 and np.quantile(diff, 0.99) < 5e-4
    assert np.abs(eg - g.node_embed_init_g.astype(np.float32)).max() > 1e-2  # training moved the tables

    lines = open(cfg.result_filename).read().split()
    assert lines[0] == "gen:0.7598343685300207" and lines[1] == "dis:0.7598343685300207"
    assert len(lines) == 4 and lines[2].startswith("gen:") and lines[3].startswith("dis:")
    acc_g, acc_d = float(lines[2][4:]), float(lines[3][4:])
    oacc_g = orc.eval_link_prediction(o.generator.E.astype(np.float64), d["test"].tolist(), d["test_neg"].tolist())
    oacc_d = orc.eval_link_prediction(o.discriminator.E.astype(np.float64), d["test"].tolist(), d["test_neg"].tolist())
    assert abs(acc_g - oacc_g) <= 0.005 and abs(acc_d - oacc_d) <= 0.005

    # ---- file format (graph_gan.py:293-306): header, tab separated, str(float64(fp32))
    with open(cfg.emb_filenames[0]) as f:
        assert f.readline() == "5242\t50\n"
        row0 = f.readline().rstrip("\n").split("\t")
    assert row0[0] == "0" and len(row0) == 51
    assert [float(x) for x in row0[1:]] == eg[0].astype(np.float64).tolist()
    assert row0[1] == str(float(eg[0, 0]))


def test_fast_mode_schedule_matches_oracle(tmp_path):
    """The scale mode through the user-facing trainer: one fused batch per inner epoch (batch sizes above the prepared rows),
    lazy Adam, asynchronous passes -- i.e. whole-walk rewards, staged path gradient + reducing optimizer for G, the generic
    pair kernel for D, distribution cache between the prepares -- against the oracle trainer in the same mode."""
    base = str(tmp_path)
    d, n, graph = write_reference_layout(base)
    big = 1 << 30
    cfg = make_cfg(base, n_epochs=1, n_epochs_dis=3, n_epochs_gen=3, dis_interval=3, gen_interval=3, engine_seed=23,
                   batch_size_gen=big, batch_size_dis=big, engine_optimizer="adam_lazy", engine_profile_every=3)
    from graphgan_amd.graph_gan import GraphGAN
    g = GraphGAN(cfg)
    g.train()
    ocfg = orc.Config()
    ocfg.n_epochs_dis = ocfg.n_epochs_gen = ocfg.dis_interval = ocfg.gen_interval = 3
    ocfg.batch_size_gen = ocfg.batch_size_dis = big
    o = orc.GraphGANOracle(n, graph, g.node_embed_init_g, g.node_embed_init_d, cfg=ocfg, rng="counter", arith="spec", seed=23, lazy_adam=True)
    o.train_epoch(0)
    eg, ed = g.generator.embedding_matrix, g.discriminator.embedding_matrix
    for got, want in ((eg, o.generator.E), (ed, o.discriminator.E)):
        diff = np.abs(got - want)
        print("abs diff: mean %.3g p99 %.3g max %.3g" % (diff.mean(), np.quantile(diff, 0.99), diff.max()))
        assert diff.mean() < 2e-5 and np.quantile(diff, 0.99) < 2e-4
    assert np.abs(eg - g.node_embed_init_g.astype(np.float32)).max() > 1e-3  # training moved the tables
    lines = open(cfg.result_filename).read().split()
    acc_g, acc_d = float(lines[2][4:]), float(lines[3][4:])
    oacc_g = orc.eval_link_prediction(o.generator.E.astype(np.float64), d["test"].tolist(), d["test_neg"].tolist())
    oacc_d = orc.eval_link_prediction(o.discriminator.E.astype(np.float64), d["test"].tolist(), d["test_neg"].tolist())
    assert abs(acc_g - oacc_g) <= 0.005 and abs(acc_d - oacc_d) <= 0.005


def test_user_facing_api_shapes(tmp_path):
    """prepare_data_for_d / prepare_data_for_g / sample / get_node_pairs_from_path keep the reference's
    return shapes (graph_gan.py:182-291)."""
    base = str(tmp_path)
    d, n, graph = write_reference_layout(base)
    cfg = make_cfg(base)
    from graphgan_amd.graph_gan import GraphGAN
    np.random.seed(1)
    g = GraphGAN(cfg)
    g.root_nodes = list(range(0, n, 11))
    g.trees = g.construct_trees(g.root_nodes)
    c, nb, lab = g.prepare_data_for_d()
    assert len(c) == len(nb) == len(lab) > 0 and set(lab) == {0, 1}
    r0 = c[0]
    k = len(graph[r0])
    assert nb[:k] == graph[r0] and lab[:k] == [1] * k and lab[k:2 * k] == [0] * k  # positives first per root
    n1, n2, rew = g.prepare_data_for_g()
    assert len(n1) == len(n2) == len(rew) > 0 and rew.dtype == np.float32 and rew.min() > 0
    root = g.root_nodes[3]
    samples, paths = g.sample(root, None, 20, for_d=False)
    if samples is not None:
        assert len(samples) == len(paths) == 20 and all(p[0] == root and p[-1] == p[-3] for p in paths if len(p) > 2)
    isolated = [v for v in range(n) if len(graph[v]) == 0][0]
    g.trees = g.construct_trees([isolated])
    assert g.sample(isolated, None, 20, for_d=False) == (None, None)
    assert g.get_node_pairs_from_path([1, 0, 2, 4, 2]) == [[1, 0], [1, 2], [0, 1], [0, 2], [0, 4], [2, 1], [2, 0], [2, 4], [4, 0], [4, 2]]


def test_script_entry_point_with_user_config(tmp_path):
    """``cd <dir with config.py> && python graph_gan.py`` -- the reference's way of running (README.md:43-45)."""
    base = str(tmp_path)
    write_reference_layout(base)
    work = os.path.join(base, "src", "GraphGAN")
    os.makedirs(work)
    src = open(os.path.join(ROOT, "graphgan_amd", "config.py")).read()
    src = src.replace("n_epochs = 20", "n_epochs = 1").replace("n_epochs_gen = 30", "n_epochs_gen = 1").replace("n_epochs_dis = 30", "n_epochs_dis = 1")
    with open(os.path.join(work, "config.py"), "w") as f:
        f.write(src)
    env = dict(os.environ)
    env.pop("GRAPHGAN_ROOT", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "graphgan_amd", "graph_gan.py")], cwd=work, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "training completes" in p.stdout
    res = open(os.path.join(base, "results", "link_prediction", "CA-GrQc.txt")).read().split()
    assert res[0] == "gen:0.7598343685300207" and len(res) == 4
    for name in ("CA-GrQc_gen_.emb", "CA-GrQc_dis_.emb"):
        assert os.path.getsize(os.path.join(base, "results", "link_prediction", name)) > 1e6


def test_update_ratio_below_one_selects_resident_slots(tmp_path):
    """config.update_ratio < 1 (graph_gan.py:189,209): every prepare draws its own subset of roots.  All trees stay
    resident (the reference's self.trees): the draw only selects slots, so the D-mode mutations (Q3) persist; with a tree
    budget below N trees the mirror runs every prepare over root batches of ITS draw (gg_epoch_*: the mutations then live in
    the engine's persistent store)."""
    base = str(tmp_path)
    d, n, graph = write_reference_layout(base)
    for budget, resident in ((160.0, True), (0.0, False)):
        cfg = make_cfg(base, n_epochs=1, n_epochs_dis=2, n_epochs_gen=2, dis_interval=1, gen_interval=1, update_ratio=0.05,
                       engine_optimizer="adam_lazy", engine_profile_every=0, engine_tree_budget_gb=budget, engine_batch_roots=100)  # no events: passes return early, G walks beside D updates
        if os.path.exists(cfg.result_filename):
            os.remove(cfg.result_filename)
        from graphgan_amd.graph_gan import GraphGAN
        g = GraphGAN(cfg)
        assert (g.trees is not None) == resident
        g.train()
        lines = open(cfg.result_filename).read().split()
        assert len(lines) == 4 and lines[0] == "gen:0.7598343685300207"
        c = g.engine.counters()
        assert c["d_steps"] > 0 and c["g_steps"] > 0 and 0 < c["walks"] < 4 * 0.2 * n * 25
        nroots = len(g.engine.tree_roots)
        if resident:
            assert nroots == n and c["bfs_trees"] == n
        else:  # four prepares (2 D + 2 G), each over its own ~5 % draw in batches of 100 roots; the last batch is resident
            assert 0 < nroots <= 100 and 4 * 0.02 * n < c["bfs_trees"] < 4 * 0.09 * n
            assert g.engine.q3_get()[1].any()   # D-mode mutations were kept across the batches
        g.engine.close()


# ---- float parity over whole schedules (north star: +-0.5 % absolute) -------------------------------
# The strict schedule (batch 64, dense TF1 Adam) runs without a single float atomic (pair_grad_det_kernel, steps.hip), so one
# build gives the SAME BITS for a seed every time: these gates compare fixed numbers, not samples of a scheduling-dependent
# process.  What stays chaotic is the training process itself: the oracle's own <= 1-ulp variant (sigmoid evaluated in fp64,
# tests/golden/oracle_epochs_multiseed.json "epochs_sigmoid64") moves a seed's accuracy by up to 1.0 % once the generator has
# fallen to chance (default schedule, after the second outer epoch) and on the generator's falling edge (short schedule, after
# the third), because one flipped walk changes every batch behind it.  There, +-0.5 % is asserted for the MEAN over the seeds --
# the quantity a last-bit difference cannot move -- and per seed everywhere else.  No gate in this file is wider than 0.005.
GATE = 0.005
SHORT_SEEDS = list(range(8))
FULL_SEEDS = list(range(8))


def _train_schedule(base, seed, **over):
    import json
    from tests.helpers import ca_grqc_init_embeddings
    d, n, graph = write_reference_layout(base)
    cfg = make_cfg(base, engine_seed=seed, **over)
    from graphgan_amd.graph_gan import GraphGAN
    g = GraphGAN(cfg)
    init = ca_grqc_init_embeddings(d, n, seed=0).astype(np.float32)
    g.engine.set_embeddings(0, init)
    g.engine.set_embeddings(1, init)
    g.train()
    lines = open(cfg.result_filename).read().split()
    acc = [[float(lines[2 * i][4:]), float(lines[2 * i + 1][4:])] for i in range(len(lines) // 2)]
    c = g.engine.counters()
    tables = [g.engine.get_embeddings(0), g.engine.get_embeddings(1)]
    g.engine.close()
    return acc, c, tables


@pytest.fixture(scope="module")
def short_runs(tmp_path_factory):
    """the reference's schedule with 2 + 2 inner passes per outer epoch (batch 64, dense TF1 Adam, all 5 242 roots, ~15 600
    optimizer steps per epoch), three outer epochs, every seed of tests/golden/oracle_epochs_short.json"""
    out = {}
    for seed in SHORT_SEEDS:
        out[seed] = _train_schedule(str(tmp_path_factory.mktemp("short%d" % seed)), seed, n_epochs=3, n_epochs_dis=2, n_epochs_gen=2,
                                    dis_interval=2, gen_interval=2)
    print("short schedule, engine [gen, dis] per seed:", {k: v[0] for k, v in out.items()})
    return out


@pytest.fixture(scope="module")
def full_runs(tmp_path_factory):
    """TWO outer epochs of the reference's DEFAULT schedule (30 + 30 inner passes, batch 64, dense TF1 Adam, ~330 k optimizer
    steps each), seeds of tests/golden/oracle_epochs_multiseed.json (~7 CPU-minutes per epoch on the oracle side)"""
    out = {}
    for seed in FULL_SEEDS:
        out[seed] = _train_schedule(str(tmp_path_factory.mktemp("full%d" % seed)), seed, n_epochs=2)
    print("default schedule, engine [gen, dis] per seed:", {k: v[0] for k, v in out.items()})
    return out


def _golden(name):
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", name)))["epochs"]


@pytest.mark.parametrize("seed", SHORT_SEEDS)
def test_short_schedule_epochs_match_the_oracle_per_seed(short_runs, seed):
    """The INFORMATIVE float-parity workload: the short schedule moves the generator UP, 0.760 -> 0.873 -> 0.876, and keeps the
    discriminator near 0.78 -- far from the start and far from chance, with a seed-to-seed spread of 0.1-0.2 % on the oracle side.
    Gate: +-0.5 % PER SEED for both models after outer epochs 0 and 1, and for the discriminator after epoch 2 (where the
    generator turns and falls: see the seed-mean test)."""
    want = _golden("oracle_epochs_short.json")[str(seed)]
    acc = short_runs[seed][0]
    print("seed %d engine %s oracle %s" % (seed, acc, want[:4]))
    assert len(acc) == 4 and acc[0] == want[0]
    for ep in (1, 2):
        assert abs(acc[ep][0] - want[ep][0]) <= GATE and abs(acc[ep][1] - want[ep][1]) <= GATE, (ep, acc[ep], want[ep])
        assert acc[ep][0] > 0.85 and 0.75 < acc[ep][1] < 0.82       # informative: nowhere near chance
    assert abs(acc[3][1] - want[3][1]) <= GATE, (acc[3], want[3])
    c = short_runs[seed][1]
    assert c["d_steps"] > 3 * 2 * 125 and c["g_steps"] > 3 * 2 * 7000


def test_short_schedule_seed_means_match_the_oracle(short_runs):
    """After the third outer epoch the generator is on its falling edge (0.876 -> 0.63-0.65 within the epoch; the oracle's seeds
    span 2 % there): the +-0.5 % bar is asserted for the mean over the 8 seeds, for every epoch and both models."""
    want = _golden("oracle_epochs_short.json")
    E = np.array([short_runs[s][0] for s in SHORT_SEEDS])
    O = np.array([want[str(s)][:4] for s in SHORT_SEEDS])
    diff = (E - O).mean(0)
    print("short schedule: seed-mean engine - oracle per epoch [gen, dis] =", diff.tolist(), "max per-seed |diff| =", np.abs(E - O).max(0).tolist())
    assert np.abs(diff).max() <= GATE
    assert 0.55 < E[:, 3, 0].mean() < 0.75   # it HAS fallen, and not to chance -- on both sides
    assert 0.55 < O[:, 3, 0].mean() < 0.75


@pytest.mark.parametrize("seed", FULL_SEEDS)
def test_full_schedule_first_epoch_matches_the_oracle_per_seed(full_runs, seed):
    """Default schedule, after the first outer epoch (~330 k optimizer steps): +-0.5 % per seed for both models."""
    want = _golden("oracle_epochs_multiseed.json")[str(seed)]
    acc, c, _ = full_runs[seed]
    print("seed %d engine %s oracle %s" % (seed, acc, want[:3]))
    assert len(acc) == 3
    assert acc[0] == want[0]  # before training: the shipped embeddings under the shipped evaluator
    assert abs(acc[1][0] - want[1][0]) <= GATE and abs(acc[1][1] - want[1][1]) <= GATE, (acc[1], want[1])
    assert c["d_steps"] > 6000 and c["g_steps"] > 400000


def test_full_schedule_seed_means_match_the_oracle(full_runs):
    """Default schedule, after the second outer epoch: the generator has fallen to chance on BOTH sides (0.50-0.53) and the
    trajectories are different samples of one chaotic process -- the oracle's own fp64-sigmoid variant differs from the oracle by
    up to 1.0 % (generator) / 0.9 % (discriminator) per seed there.  Gate: +-0.5 % on the mean over the seeds, every epoch, both
    models; and every engine value inside the band the oracle's 15 seeds span, widened by the same 0.5 %."""
    gold = _golden("oracle_epochs_multiseed.json")
    E = np.array([full_runs[s][0] for s in FULL_SEEDS])
    O = np.array([gold[str(s)][:3] for s in FULL_SEEDS])
    diff = (E - O).mean(0)
    print("default schedule: seed-mean engine - oracle per epoch [gen, dis] =", diff.tolist(), "max per-seed |diff| =", np.abs(E - O).max(0).tolist())
    assert np.abs(diff).max() <= GATE
    allo = np.array([v[:3] for v in gold.values()])
    assert (E >= allo.min(0) - GATE).all() and (E <= allo.max(0) + GATE).all()


def test_schedules_are_bit_reproducible(short_runs, tmp_path):
    """A second training run of one seed -- new process state, new engine, same build -- ends on the SAME accuracies and the same
    bits in both tables: nothing on the strict path depends on the order in which the hardware retires work."""
    acc, _, tables = _train_schedule(str(tmp_path), 3, n_epochs=3, n_epochs_dis=2, n_epochs_gen=2, dis_interval=2, gen_interval=2)
    assert acc == short_runs[3][0]
    assert np.array_equal(tables[0], short_runs[3][2][0]) and np.array_equal(tables[1], short_runs[3][2][1])


def test_tree_cache_file_replaces_the_pickle(tmp_path):
    """graph_gan.py:31-46: build the trees once, cache them, read them back in the next process.  The second GraphGAN
    must come up from the cache (no BFS), hold the same trees, and sample the same walks; a cache of another graph or a
    truncated file is refused and rebuilt."""
    import graphgan_amd as ga
    base = str(tmp_path)
    d, n, graph = write_reference_layout(base)
    os.makedirs(os.path.join(base, "cache"))
    cfg = make_cfg(base, engine_seed=5)
    from graphgan_amd.graph_gan import GraphGAN
    open(cfg.cache_filename, "wb").write(b"\x80\x04a reference pickle")  # the reference's own cache: never touched
    cache = cfg.cache_filename + ".ggtr"                                  # the native cache lives next to it
    g1 = GraphGAN(cfg)
    assert open(cfg.cache_filename, "rb").read() == b"\x80\x04a reference pickle"
    assert os.path.getsize(cache) > 8 * 16e6 and not os.path.exists(cache + ".tmp")  # 16.3 M (root, node) pairs
    assert g1.engine.counters()["bfs_trees"] == n
    t1 = g1.engine.get_trees()
    w1 = g1.engine.walk_sample(np.arange(n), np.full(n, 5), False, 3, 1)
    g1.engine.close()
    g2 = GraphGAN(cfg)
    assert g2.engine.counters()["bfs_trees"] == 0 and g2.trees is not None and g2._slot_of_root[17] == 17
    t2 = g2.engine.get_trees()
    for a, b in zip(t1, t2):
        assert np.array_equal(a, b)
    assert g2.engine.max_depth == g1.engine.max_depth
    w2 = g2.engine.walk_sample(np.arange(n), np.full(n, 5), False, 3, 1)
    for k in ("samples", "path_len", "root_status"):
        assert np.array_equal(w1[k], w2[k])
    m = np.arange(w1["paths"].shape[1])[None, :] < w1["path_len"][:, None]
    assert np.array_equal(w1["paths"][m], w2["paths"][m])
    # another graph: refused with GG_EINVAL, nothing loaded
    rowptr, col = ga.edges_to_csr(n, d["train"][:-7])
    other = ga.Engine(np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32))
    other.set_graph_csr(rowptr, col)
    with pytest.raises(ga.GraphGANHipError) as ei:
        other.load_trees(cache)
    assert ei.value.code == ga.GG_EINVAL
    other.close()
    # truncated: refused with GG_EIO; the trainer rebuilds and rewrites it
    blob = open(cache, "rb").read()
    open(cache, "wb").write(blob[: len(blob) // 3])
    with pytest.raises(ga.GraphGANHipError) as ei:
        g2.engine.load_trees(cache)
    assert ei.value.code == ga.GG_EIO
    g2.engine.close()
    g3 = GraphGAN(cfg)
    assert g3.engine.counters()["bfs_trees"] == n and os.path.getsize(cache) == len(blob)
    # right size, corrupt contents (a node id out of range; a child range that points backwards): refused with GG_EIO --
    # the walk kernels index the tree arrays without bounds tests -- and nothing stays resident
    hdr = 48 + 4 * n + 8 * (n + 1)
    nodes = (len(blob) - hdr - 4 * n) // 8
    for pos, val in ((hdr + 4 * 12345, 0x7FFFFFF0), (hdr + 4 * nodes + 4 * 777, 0)):
        bad = bytearray(blob)
        bad[pos:pos + 4] = int(val).to_bytes(4, "little")
        open(cache, "wb").write(bytes(bad))
        with pytest.raises(ga.GraphGANHipError) as ei:
            g3.engine.load_trees(cache)
        assert ei.value.code == ga.GG_EIO and "corrupt" in str(ei.value)
        with pytest.raises(ga.GraphGANHipError):
            g3.engine.walk_sample([0], [1], False, 1, 1)   # no trees loaded
    open(cache, "wb").write(blob)
    g3.engine.load_trees(cache)
    g3.engine.close()


def test_bench_gpus_2_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher: bench.py spawns the two ranks itself (gloo control plane, rank 0 prints
    the one JSON line).  This box has one GPU, so the ranks share device 0 and skip RCCL (GG_BENCH_SHARE_GPU=1: plumbing
    only); each rank walks its own roots of the same graph, so the whole-job hop count is about twice the one-rank run's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = ["--nodes", "20000", "--roots", "256", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-strict",
             "--fresh-batches", "0", "--overlap-steps", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    lines = {}
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n)] + flags, capture_output=True, text=True,
                           env=dict(env, GG_BENCH_SHARE_GPU="1"), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(out) == 1
        lines[n] = json.loads(out[0])
    assert lines[1]["n_gpus"] == 1 and lines[2]["n_gpus"] == 2 and lines[2]["scaling"] == "weak"
    h1 = lines[1]["value"] * lines[1]["ms_per_step"]   # ~ hops per step (all ranks)
    h2 = lines[2]["value"] * lines[2]["ms_per_step"]
    assert 1.6 < h2 / h1 < 2.4
    assert 1.6 < lines[2]["g_pairs_per_step_all_ranks"] / lines[1]["g_pairs_per_step_all_ranks"] < 2.4
