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
