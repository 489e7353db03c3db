"""Link-prediction evaluator with the reference's semantics
(``src/evaluation/link_prediction.py:10-38``): re-read the ``.emb`` text, score every test and
negative edge by a dot product, predict "edge" when the score is at or above the median, report
the accuracy against (first half = positives, second half = negatives)."""
import numpy as np

from .. import utils


class LinkPredictEval(object):
    def __init__(self, embed_filename, test_filename, test_neg_filename, n_node, n_embed, emd=None, engine=None, which=0):
        self.embed_filename = embed_filename
        self.test_filename = test_filename
        self.test_neg_filename = test_neg_filename
        self.n_node = n_node
        self.n_embed = n_embed
        # ``emd`` (float64 [n_node, n_embed]) skips re-parsing the text that was just written: the file holds
        # repr(float64(fp32)) of exactly these numbers, so parsing it back gives the same matrix
        # ``engine`` (+ ``which``: 0 = generator, 1 = discriminator): the per-edge dots are computed on the device from
        # the resident table (gg_edge_scores, float64 like np.dot on the reference's float64 arrays); no N x d matrix
        # is fetched or parsed at all
        self.engine, self.which = engine, which
        if engine is not None:
            self.emd = None
        else:
            self.emd = emd if emd is not None else utils.read_embeddings(embed_filename, n_node=n_node, n_embed=n_embed)

    def eval_link_prediction(self):
        edges = np.array(utils.read_edges_from_file(self.test_filename) +
                         utils.read_edges_from_file(self.test_neg_filename), dtype=np.int64)
        # per-edge np.dot like the reference (link_prediction.py:26-27): identical fp64 rounding
        if self.engine is not None:
            score = self.engine.edge_scores(self.which, edges[:, 0], edges[:, 1])
        else:
            score = np.array([np.dot(self.emd[a], self.emd[b]) for a, b in edges])
        predicted = score >= np.median(score)
        truth = np.arange(len(edges)) < len(edges) // 2
        return float(np.mean(predicted == truth))
