"""Host-side mirror of the reference's user contract (config names, readers, evaluator, window
pairs) -- no GPU needed: these modules do not touch the engine."""
import json
import os

import numpy as np
import pytest

from oracle import graphgan_oracle as orc
from tests.helpers import GOLD, load_ca_grqc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# names defined by the reference's flag module (src/GraphGAN/config.py:1-41)
REFERENCE_CONFIG_NAMES = """modes batch_size_gen batch_size_dis lambda_gen lambda_dis n_sample_gen lr_gen lr_dis n_epochs
n_epochs_gen n_epochs_dis gen_interval dis_interval update_ratio load_model save_steps n_emb multi_processing window_size app
dataset train_filename test_filename test_neg_filename pretrain_emb_filename_d pretrain_emb_filename_g emb_filenames
result_filename cache_filename model_log""".split()


@pytest.fixture(scope="module")
def pkg():
    so = os.path.join(ROOT, "graphgan_amd", "libgraphgan_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    import graphgan_amd
    return graphgan_amd


def test_config_keeps_every_reference_name_and_default(pkg):
    from graphgan_amd import config
    for name in REFERENCE_CONFIG_NAMES:
        assert hasattr(config, name), name
    assert (config.batch_size_gen, config.batch_size_dis, config.n_sample_gen, config.n_epochs) == (64, 64, 20, 20)
    assert (config.lambda_gen, config.lambda_dis, config.lr_gen, config.lr_dis) == (1e-5, 1e-5, 1e-3, 1e-3)
    assert (config.n_epochs_gen, config.n_epochs_dis, config.gen_interval, config.dis_interval) == (30, 30, 30, 30)
    assert (config.update_ratio, config.load_model, config.save_steps, config.n_emb, config.window_size) == (1, False, 10, 50, 2)
    assert config.modes == ["gen", "dis"] and config.app == "link_prediction" and config.dataset == "CA-GrQc"
    assert config.emb_filenames[0].endswith("CA-GrQc_gen_.emb") and config.emb_filenames[1].endswith("CA-GrQc_dis_.emb")
    assert config.result_filename.endswith("results/link_prediction/CA-GrQc.txt")


def test_readers_and_evaluator_on_the_shipped_fixture(pkg, tmp_path):
    from graphgan_amd import utils
    from graphgan_amd.evaluation import link_prediction as lp
    d, n, graph = load_ca_grqc()
    tr, te, neg, emb = [str(tmp_path / x) for x in ("train.txt", "test.txt", "neg.txt", "pre.emb")]
    for path, key in ((tr, "train"), (te, "test"), (neg, "test_neg")):
        with open(path, "w") as f:
            f.writelines("%d\t%d\n" % (a, b) for a, b in d[key].tolist())
    with open(emb, "w") as f:
        f.write("%d 50\n" % len(d["emb_ids"]))
        for i, row in zip(d["emb_ids"].tolist(), d["emb_rows"].astype(np.float64).tolist()):
            f.write(str(i) + " " + " ".join(repr(x) for x in row) + "\n")
    n2, g2 = utils.read_edges(tr, te)
    n3, g3 = orc.read_edges(tr, te)
    assert n2 == n3 == 5242 and g2 == g3
    assert utils.read_edges_from_file(te) == d["test"].tolist()
    np.random.seed(4)
    a = utils.read_embeddings(emb, n, 50)
    np.random.seed(4)
    b = orc.read_embeddings(emb, n, 50)
    assert a.dtype == np.float64 and np.array_equal(a, b)
    misc = json.load(open(os.path.join(GOLD, "ref_misc.json")))
    for seed in (0, 1, 2):  # the seeds the reference's own evaluator was run with (make_golden.py)
        np.random.seed(seed)
        acc = lp.LinkPredictEval(emb, te, neg, n, 50).eval_link_prediction()
        assert acc == misc["epoch0_accuracy"][0] == 0.7598343685300207
    # fed from memory == re-parsing the text
    acc2 = lp.LinkPredictEval(emb, te, neg, n, 50, emd=a).eval_link_prediction()
    np.random.seed(4)
    assert acc2 == lp.LinkPredictEval(emb, te, neg, n, 50).eval_link_prediction()


def test_window_pairs_docstring_vector(pkg):
    from graphgan_amd.graph_gan import GraphGAN
    misc = json.load(open(os.path.join(GOLD, "ref_misc.json")))
    assert GraphGAN.get_node_pairs_from_path(misc["pairs_docstring_path"]) == misc["pairs_docstring_out"]
    for c in misc["pairs_extra"]:
        assert GraphGAN.get_node_pairs_from_path(c["path"]) == c["pairs"]
    assert GraphGAN.stream_id(0, 0, 30, True) == 0 and GraphGAN.stream_id(0, 0, 30, False) == 1
    assert GraphGAN.stream_id(2, 0, 30, True) == 120 and len({GraphGAN.stream_id(e, i, 30, f) for e in range(3) for i in range(30) for f in (0, 1)}) == 180


class _FakeEngine(object):
    """Records what graph_gan.GraphGAN asks of the engine (the five sess.run call sites + sampler), no GPU."""

    def __init__(self, emb_g, emb_d, **kw):
        self.kw = kw
        self.E = [np.array(emb_g, np.float32), np.array(emb_d, np.float32)]
        self.n_node, self.n_emb = self.E[0].shape
        self.calls = []
        self.tree_roots = None

    def tree_bytes_estimate(self, n_roots): return float(n_roots) * 8.0 * (self.n_node + 1)
    def set_profiling(self, k): self.calls.append(("set_profiling", k))
    def set_graph_csr(self, rowptr, col): self.calls.append(("set_graph_csr", len(rowptr) - 1, len(col)))
    def build_trees(self, roots, **kw): self.tree_roots = list(roots); self.calls.append(("build_trees", len(roots), kw.get("device")))
    def save_trees(self, path): open(path, "w").write("trees"); self.calls.append(("save_trees", path))
    def load_trees(self, path): self.tree_roots = list(range(self.n_node)); self.calls.append(("load_trees", path))
    def load_state(self, path): self.calls.append(("load_state", path))
    def save_state(self, path): open(path, "w").write("x"); self.calls.append(("save_state", path))
    def prepare_d(self, slots, seed, stream, fetch=True):
        self.calls.append(("prepare_d", len(slots), seed, stream, fetch))
        self.last_slots = np.array(slots)
        return 200 if len(slots) else 0   # gg_prepare_d with no slots: zero rows (and still its collective)

    def prepare_g(self, slots, n_sample, seed, stream, fetch=True):
        self.calls.append(("prepare_g", len(slots), n_sample, seed, stream, fetch))
        self.last_slots = np.array(slots)
        return 333 if len(slots) else 0
    def prepare_g_begin(self, slots, n_sample, seed, stream): self.calls.append(("prepare_g_begin", len(slots), n_sample, seed, stream))
    def counters(self): return {k: 0 for k in ("walks", "hops", "rows_scored", "d_pairs", "g_pairs", "d_steps", "g_steps", "bfs_trees", "bfs_kernel_ms", "walk_reruns")}
    def epoch_begin(self, reset_d=True, reset_g=True): self.calls.append(("epoch_begin", bool(reset_d), bool(reset_g)))
    def epoch_add(self, roots, do_d, do_g, n_sample, seed, stream_d, stream_g):
        self.calls.append(("epoch_add", [int(r) for r in roots], bool(do_d), bool(do_g), n_sample, seed, stream_d, stream_g))
    def epoch_commit(self, which): self.calls.append(("epoch_commit", which)); return 200 if which == 1 else 333
    def d_pass(self, starts, batch): self.calls.append(("d_pass", list(starts), batch))
    def g_pass(self, starts, batch): self.calls.append(("g_pass", list(starts), batch))
    def get_embeddings(self, which): return self.E[which]
    def edge_scores(self, which, u, v):
        e = self.E[which].astype(np.float64)
        return np.array([np.dot(e[a], e[b]) for a, b in zip(u, v)])
    def get_bias(self, which): return np.zeros(self.n_node, np.float32)
    def write_embeddings(self, which, path):
        from graphgan_amd import engine
        engine.host_write_embeddings(path, self.E[which])
        self.calls.append(("write_embeddings", which))


def test_training_schedule_drives_the_engine_like_the_reference(pkg, tmp_path, monkeypatch):
    """graph_gan.py:133-178 on a fake engine: prepare cadence (dis_interval / gen_interval), one shuffled list of
    contiguous batch starts per inner epoch (a13), Philox stream per (epoch, inner epoch, phase), checkpoint
    cadence (save_steps, load_model), embeddings + results written before training and after every epoch."""
    from tests.test_gpu_e2e import make_cfg, write_reference_layout
    from graphgan_amd import engine as eng_mod, graph_gan
    base = str(tmp_path)
    d, n, graph = write_reference_layout(base)
    monkeypatch.setattr(eng_mod, "Engine", _FakeEngine)
    cfg = make_cfg(base, n_epochs=3, n_epochs_dis=4, n_epochs_gen=5, dis_interval=2, gen_interval=3, save_steps=2,
                   load_model=True, engine_seed=11, engine_profile_every=0)
    os.makedirs(cfg.model_log)
    open(os.path.join(cfg.model_log, "model.checkpoint.ggst"), "w").write("x")
    g = graph_gan.GraphGAN(cfg)
    g.train()
    calls = g.engine.calls
    names = [c[0] for c in calls]
    assert names[:3] == ["set_profiling", "set_graph_csr", "build_trees"] and calls[2][1] == n and calls[2][2] is True
    assert names[3] == "load_state"                                 # load_model and the checkpoint exists (:124-127)
    assert names[4:6] == ["write_embeddings", "write_embeddings"]  # before training (:129)
    # (prepare_g_begin: the walks of the epoch's first prepare_g are started before the last D pass is enqueued)
    per_epoch = ["prepare_d", "d_pass", "d_pass", "prepare_d", "d_pass", "prepare_g_begin", "d_pass",
                 "prepare_g", "g_pass", "g_pass", "g_pass", "prepare_g", "g_pass", "g_pass", "write_embeddings", "write_embeddings"]
    want = per_epoch + per_epoch + ["save_state"] + per_epoch  # save at epoch 2 (epoch > 0 and epoch % save_steps == 0, :136-138)
    assert names[6:] == want
    # streams: 2 * (epoch * n_inner + inner) + {0: D, 1: G}; every root slot, fetch=False (resident data)
    pd = [c for c in calls if c[0] == "prepare_d"]
    pg = [c for c in calls if c[0] == "prepare_g"]
    assert [c[3] for c in pd] == [2 * (e * 4 + i) for e in range(3) for i in (0, 2)]
    assert [c[4] for c in pg] == [2 * (e * 5 + i) + 1 for e in range(3) for i in (0, 3)]
    assert all(c[1] == n and c[2] == 11 and c[4] is False for c in pd) and all(c[2] == 20 for c in pg)
    # the head start names exactly the arguments of the prepare_g that adopts it (slots, n_sample, seed, stream)
    pb = [c for c in calls if c[0] == "prepare_g_begin"]
    first_pg = [c for c in pg if (c[4] - 1) // 2 % 5 == 0]
    assert [c[1:] for c in pb] == [c[1:5] for c in first_pg] and len(pb) == 3
    # batches: a permutation of the contiguous starts 0, 64, ... of the prepared size
    for c in calls:
        if c[0] == "d_pass":
            assert sorted(c[1]) == list(range(0, 200, 64)) and c[2] == 64
        if c[0] == "g_pass":
            assert sorted(c[1]) == list(range(0, 333, 64)) and c[2] == 64
    orders = [tuple(c[1]) for c in calls if c[0] == "g_pass"]
    assert len(set(orders)) > 1                                     # reshuffled per inner epoch (:151,170)
    res = open(cfg.result_filename).read().split()
    assert len(res) == 8 and res[0] == "gen:0.7598343685300207" and all(r[:4] in ("gen:", "dis:") for r in res)
    assert open(cfg.emb_filenames[0]).readline() == "5242\t50\n"


def test_root_batched_epoch_when_the_trees_do_not_fit(pkg, tmp_path, monkeypatch):
    """All N trees over the budget (N^2 storage): the mirror runs the reference's schedule (graph_gan.py:133-176) over ROOT
    BATCHES -- gg_epoch_begin / gg_epoch_add per batch / gg_epoch_commit -- instead of failing: every root of root_nodes in
    order, the G-mode walks of the epoch's first generator prepare in the same batches as the LAST discriminator prepare (one
    BFS per root and outer epoch), later generator prepares with batches of their own; passes and streams as before; one
    JSON line per outer epoch beside the results file."""
    import json
    from tests.test_gpu_e2e import make_cfg, write_reference_layout
    from graphgan_amd import engine as eng_mod, graph_gan
    base = str(tmp_path)
    d, n, graph = write_reference_layout(base)
    monkeypatch.setattr(eng_mod, "Engine", _FakeEngine)
    per_tree = 8.0 * (n + 1)                        # the fake engine's estimate
    cfg = make_cfg(base, n_epochs=2, n_epochs_dis=4, n_epochs_gen=5, dis_interval=2, gen_interval=3, engine_seed=11,
                   engine_tree_budget_gb=1200.5 * per_tree / 2.0 ** 30)  # room for 1 200 of the 5 242 trees
    g = graph_gan.GraphGAN(cfg)
    assert not g._all_resident and g._batch_roots == 1200 and g.trees is None
    g.train()
    calls = g.engine.calls
    names = [c[0] for c in calls]
    assert "build_trees" not in names and "prepare_d" not in names and "prepare_g" not in names and "prepare_g_begin" not in names
    adds = ["epoch_add"] * 5                        # ceil(5242 / 1200) batches
    per_epoch = (["epoch_begin"] + adds + ["epoch_commit", "d_pass", "d_pass"]          # D prepare at inner epoch 0
                 + ["epoch_begin"] + adds + ["epoch_commit", "d_pass", "d_pass"]        # ... at inner epoch 2: with the G walks
                 + ["epoch_commit", "g_pass", "g_pass", "g_pass"]                       # G epoch 0: the pairs are there already
                 + ["epoch_begin"] + adds + ["epoch_commit", "g_pass", "g_pass"]        # G prepare at inner epoch 3: own batches
                 + ["write_embeddings", "write_embeddings"])
    i0 = names.index("epoch_begin")
    assert names[i0:] == per_epoch + per_epoch
    ep = [c for c in calls if c[0] in ("epoch_begin", "epoch_add", "epoch_commit")]
    begins = [c[1:] for c in ep if c[0] == "epoch_begin"]
    assert begins == [(True, False), (True, True), (False, True)] * 2
    commits = [c[1] for c in ep if c[0] == "epoch_commit"]
    assert commits == [1, 1, 0, 0] * 2
    a = [c for c in ep if c[0] == "epoch_add"]
    for k in range(0, len(a), 5):                   # every prepare covers root_nodes once, in order
        assert sum((c[1] for c in a[k:k + 5]), []) == list(range(n))
    flags = [(c[2], c[3]) for c in a[::5]]
    assert flags == [(True, False), (True, True), (False, True)] * 2
    sd = lambda e, i: 2 * (e * 4 + i)               # noqa: E731  streams as in the resident schedule
    sg = lambda e, i: 2 * (e * 5 + i) + 1           # noqa: E731
    assert [(c[6], c[7]) for c in a[::5]] == [(sd(0, 0), sg(0, 0)), (sd(0, 2), sg(0, 0)), (0, sg(0, 3)),
                                               (sd(1, 0), sg(1, 0)), (sd(1, 2), sg(1, 0)), (0, sg(1, 3))]
    assert all(c[4] == 20 and c[5] == 11 for c in a)
    perf = [json.loads(l) for l in open(cfg.result_filename + ".perf.jsonl")]
    assert [p["epoch"] for p in perf] == [0, 1] and perf[0]["trees"] == "root batches of 1200" and len(perf[0]["results"]) == 2


def test_update_ratio_selects_roots_per_prepare(pkg, tmp_path, monkeypatch):
    """``np.random.rand() < update_ratio`` per root (graph_gan.py:189,209).  The reference keeps ALL trees in
    self.trees and only skips roots per prepare, so the D-mode mutations (Q3) persist: the mirror builds every
    tree once and passes the drawn subset of slots to each prepare call (one draw per root, in root order)."""
    from tests.test_gpu_e2e import make_cfg, write_reference_layout
    from graphgan_amd import engine as eng_mod, graph_gan
    base = str(tmp_path)
    d, n, graph = write_reference_layout(base)
    monkeypatch.setattr(eng_mod, "Engine", _FakeEngine)
    cfg = make_cfg(base, n_epochs=1, n_epochs_dis=2, n_epochs_gen=2, dis_interval=1, gen_interval=2, update_ratio=0.1, engine_seed=3)
    g = graph_gan.GraphGAN(cfg)
    assert g.trees is not None and [c[1] for c in g.engine.calls if c[0] == "build_trees"] == [n]
    seen = []
    orig_d, orig_g = g.engine.prepare_d, g.engine.prepare_g
    g.engine.prepare_d = lambda slots, *a, **k: (seen.append(np.array(slots)), orig_d(slots, *a, **k))[1]
    g.engine.prepare_g = lambda slots, *a, **k: (seen.append(np.array(slots)), orig_g(slots, *a, **k))[1]
    g.train()
    calls = g.engine.calls
    assert len([c for c in calls if c[0] == "build_trees"]) == 1   # never rebuilt
    assert len(seen) == 3                                          # D at inner epochs 0, 1; G at inner epoch 0
    rng = np.random.RandomState(3)                                 # host RNG = RandomState(engine_seed); shuffles interleave
    for sl in seen:
        assert 0.06 * n < len(sl) < 0.14 * n and np.all(np.diff(sl) > 0) and sl.max() < n
    assert len({len(sl) for sl in seen}) > 1                       # a fresh draw per prepare
    first = np.flatnonzero(rng.rand(n) < 0.1)
    assert np.array_equal(seen[0], first)                          # one draw per root, in root order, before any shuffle


def test_update_ratio_with_trees_over_budget_batches_each_draw(pkg, tmp_path, monkeypatch):
    """When all N trees cannot stay resident (engine_tree_budget_gb) and update_ratio < 1, every prepare runs root batches
    over ITS OWN draw (graph_gan.py:189,209 draw per prepare: the G-mode walks cannot share the D prepare's trees); an empty
    draw adds nothing but still commits (0 rows; the commit holds the replicas' collective); sample() of a root builds that
    root's tree on demand."""
    from tests.test_gpu_e2e import make_cfg, write_reference_layout
    from graphgan_amd import engine as eng_mod, graph_gan
    base = str(tmp_path)
    d, n, graph = write_reference_layout(base)
    monkeypatch.setattr(eng_mod, "Engine", _FakeEngine)
    cfg = make_cfg(base, n_epochs=1, n_epochs_dis=2, n_epochs_gen=2, dis_interval=1, gen_interval=2, update_ratio=0.1, engine_seed=3,
                   engine_tree_budget_gb=0.0, engine_batch_roots=100)
    g = graph_gan.GraphGAN(cfg)
    assert g.trees is None and not any(c[0] == "build_trees" for c in g.engine.calls) and g._batch_roots == 100
    g.train()
    calls = g.engine.calls
    assert not any(c[0] in ("build_trees", "prepare_d", "prepare_g") for c in calls)
    begins = [i for i, c in enumerate(calls) if c[0] == "epoch_begin"]
    assert [calls[i][1:] for i in begins] == [(True, False), (True, False), (False, True)]   # D, D, G: three draws
    drawn = []
    for i in begins:
        j = i + 1
        roots = []
        while calls[j][0] == "epoch_add":
            assert len(calls[j][1]) <= 100 and calls[j][2:4] == calls[i][1:]
            roots += calls[j][1]
            j += 1
        assert calls[j][0] == "epoch_commit" and calls[j][1] == (1 if calls[i][1] else 0)
        assert 0.06 * n < len(roots) < 0.14 * n and roots == sorted(set(roots))   # ~10 % of the roots, in root order
        drawn.append(roots)
    assert len({tuple(r) for r in drawn}) == 3                                      # a fresh draw per prepare
    assert drawn[0] == np.flatnonzero(np.random.RandomState(3).rand(n) < 0.1).tolist()
    # empty draw: no batch, zero rows -- but begin + commit are issued (the commit is a collective every replica joins)
    g.config.update_ratio = 0.0
    k = len(g.engine.calls)
    g.engine.get_d_data = lambda: (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
    g.engine.epoch_commit = lambda which: (g.engine.calls.append(("epoch_commit", which)), 0)[1]
    assert g.prepare_data_for_d() == ([], [], [])
    assert [c[0] for c in g.engine.calls[k:]] == ["epoch_begin", "epoch_commit"]
    # sample() of a root: its tree is built on demand
    g.engine.walk_sample = lambda slots, nw, for_d, seed, stream: dict(root_status=np.array([1]), paths=None, path_len=None, samples=None)
    assert g.sample(77, None, 5, False) == (None, None)
    assert g.engine.tree_roots == [77]


def test_tree_cache_is_read_or_written_like_the_pickle(pkg, tmp_path, monkeypatch):
    """graph_gan.py:31-46 on the fake engine: no cache directory -> build only; directory but no file -> build + save;
    file present -> load, no build."""
    from tests.test_gpu_e2e import make_cfg, write_reference_layout
    from graphgan_amd import engine as eng_mod, graph_gan
    base = str(tmp_path)
    d, n, graph = write_reference_layout(base)
    monkeypatch.setattr(eng_mod, "Engine", _FakeEngine)
    cfg = make_cfg(base)
    g = graph_gan.GraphGAN(cfg)
    names = [c[0] for c in g.engine.calls]
    assert "build_trees" in names and "save_trees" not in names and "load_trees" not in names
    os.makedirs(os.path.dirname(cfg.cache_filename))
    g = graph_gan.GraphGAN(cfg)
    names = [c[0] for c in g.engine.calls]
    assert names.index("build_trees") < names.index("save_trees") and os.path.isfile(cfg.cache_filename + ".ggtr")
    g = graph_gan.GraphGAN(cfg)
    names = [c[0] for c in g.engine.calls]
    assert "load_trees" in names and "build_trees" not in names and g.trees[5] == 5 and g._all_resident
