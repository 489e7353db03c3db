"""GraphGAN trainer: the reference's entry point (``src/GraphGAN/graph_gan.py``) on the MI355X engine.

Same class, method names, return shapes, schedule, file formats and ``config`` names as the
reference; every ``tf.Session.run`` call site and the interpreted walk sampler are replaced by
calls into ``libgraphgan_hip.so`` (``graphgan_amd.engine.Engine``):

    reference                                             here
    ----------------------------------------------------  -------------------------------------------
    utils.read_edges (utils.py:12-54)                     read_edges_csr (native ingest, same list order)
    construct_trees (:84-108) + pickle cache (:31-46)     Engine.build_trees (BFS on the GPU, or threaded host BFS)
    sample (:225-270) incl. sess.run(all_score) (:238)    Engine.walk_sample / prepare_d / prepare_g (K1)
    get_node_pairs_from_path (:272-291)                   device kernel inside prepare_g (K6)
    sess.run(discriminator.reward) (:220-222)             device kernel inside prepare_g (K2)
    sess.run(d_updates) loop (:149-157)                   Engine.d_pass (K3 + K5)
    sess.run(g_updates) loop (:168-176)                   Engine.g_pass (K4 + K5)
    sess.run(embedding_matrix) (:298) + text formatting   Engine.get_embeddings / write_embeddings (native, same bytes)
    tf.train.Saver (:55,124-127,137-138)                  Engine.save_state / load_state

Run exactly like the reference: ``cd src/GraphGAN && python <this file>`` -- a ``config.py`` in
the working directory (the reference's own) is honoured; otherwise ``graphgan_amd/config.py``.
Randomness: the reference draws from the unseeded global numpy RNG (Q5); here the walk sampler
uses Philox keyed by ``config.engine_seed`` and the host decisions (root selection, batch order)
use ``numpy.random.RandomState(engine_seed)``, so runs are reproducible.
"""
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if __package__ in (None, ""):  # executed as a script: make the package importable
    sys.path.insert(0, os.path.dirname(_HERE))

try:  # the user's flag module in the working directory wins, as with the reference (graph_gan.py:8)
    if os.path.isfile(os.path.join(os.getcwd(), "config.py")) and os.getcwd() != _HERE:
        sys.path.insert(0, os.getcwd())
        import config  # noqa: E402
    else:
        raise ImportError
except ImportError:
    from graphgan_amd import config  # noqa: E402

from graphgan_amd import _lib, engine as _engine, parallel, utils  # noqa: E402
from graphgan_amd.evaluation import link_prediction as lp  # noqa: E402

_OPTIMIZERS = {"adam_dense": _lib.GG_OPT_ADAM_DENSE, "adam_lazy": _lib.GG_OPT_ADAM_LAZY, "sgd": _lib.GG_OPT_SGD}


def _cfg(cfg, name, default):
    return getattr(cfg, name, default)


class GraphGAN(object):
    def __init__(self, cfg=None):
        self.config = cfg if cfg is not None else config
        cfg = self.config
        print("reading graphs...")
        # native ingest (same adjacency as utils.read_edges, utils.py:12-47); self.graph[i] still lists i's neighbours
        self.n_node, self._rowptr, self._col = _engine.read_edges_csr(cfg.train_filename, cfg.test_filename)
        self.graph = _engine.CSRGraph(self._rowptr, self._col)
        # One process per GPU (torch.distributed.run exports RANK / WORLD_SIZE / LOCAL_RANK; a plain `python graph_gan.py`
        # is world 1).  The reference is a single tf.Session (:57-61): every rank here holds full replicas of both models,
        # walks ITS share of the roots (:188, :208 -- the walk RNG is keyed by root id, so the share does not change a
        # walk) and the replicas sum their gradients inside every optimizer step (:154, :173; RCCL inside the engine).
        self.ctl = parallel.Control()
        self.rank, self.world = self.ctl.rank, self.ctl.world
        self.all_root_nodes = [i for i in range(self.n_node)]
        deg = (self._rowptr[1:] - self._rowptr[:-1]).astype(np.int64)
        # degree-balanced shares (D-mode walks per root = its degree); world 1: all roots, in id order like the reference
        self.root_nodes = [int(r) for r in parallel.shard_roots(np.arange(self.n_node), self.rank, self.world,
                                                                weights=deg if self.world > 1 else None)]
        self._prepare_no = 0

        self.seed = int(_cfg(cfg, "engine_seed", 0))
        # rows missing from the pre-trained file are drawn from the global numpy RNG (utils.py:63); the
        # reference never seeds it (Q5) -- seeding it here makes the whole run reproducible
        np.random.seed(self.seed)
        print("reading initial embeddings...")
        self.node_embed_init_d = utils.read_embeddings(filename=cfg.pretrain_emb_filename_d, n_node=self.n_node,
                                                       n_embed=cfg.n_emb)
        self.node_embed_init_g = utils.read_embeddings(filename=cfg.pretrain_emb_filename_g, n_node=self.n_node,
                                                       n_embed=cfg.n_emb)

        self.host_rng = np.random.RandomState(self.seed if self.world == 1 else [self.seed, self.rank])

        print("building GAN model...")
        self.engine = None
        self.generator = None
        self.discriminator = None
        self.build_generator()
        self.build_discriminator()

        # BFS trees live in HBM; like the reference's self.trees (:31-46) every root stays resident for the whole
        # run -- also with update_ratio < 1, where each prepare call only SELECTS a subset of the resident slots, so
        # the in-place D-mode mutations (Q3, :258-259) persist across epochs exactly as in the reference.  When all N
        # trees cannot fit the budget (config.engine_tree_budget_gb; N^2 storage: 12 TB at 10^6 nodes) the epoch runs
        # over ROOT BATCHES instead (_prepare_root_batches / gg_epoch_*): the same schedule -- prepare over all (selected)
        # roots, then the passes over all rows -- with one batch of trees resident at a time and the mutation bits kept
        # in a store that outlives the trees.
        print("constructing BFS-trees...")
        self.trees = None
        self._slot_of_root = None
        self._all_resident = False
        self._g_pairs_ready = False
        budget = float(_cfg(cfg, "engine_tree_budget_gb", 160.0)) * 2.0 ** 30
        # roots per batch of a root-batched epoch (all N trees do not fit: N^2 storage): what the budget holds
        self._batch_roots = int(_cfg(cfg, "engine_batch_roots", 0)) or max(1, min(16384, int(budget // self.engine.tree_bytes_estimate(1))))
        if self.engine.tree_bytes_estimate(len(self.root_nodes)) <= budget:
            # construct or read BFS-trees (reference :31-46; the cache is a flat GGTR file instead of a pickle)
            cache = self._tree_cache_path(_cfg(cfg, "cache_filename", None))
            if cache and os.path.isfile(cache) and self._load_tree_cache(cache):
                print("reading BFS-trees from cache...")
            else:
                self.trees = self.construct_trees(self.root_nodes)
                # the reference needs `mkdir cache` too (README.md:43); a file of another format is never overwritten
                if cache and os.path.isdir(os.path.dirname(cache) or ".") and self._is_ours_or_absent(cache):
                    self.engine.save_trees(cache)
            self._all_resident = True

        self.latest_checkpoint = os.path.join(cfg.model_log, "model.checkpoint.ggst")

    # ------------------------------------------------------------------ model construction
    def _ensure_engine(self):
        if self.engine is None:
            cfg = self.config
            self.engine = _engine.Engine(
                self.node_embed_init_g, self.node_embed_init_d, lr_gen=cfg.lr_gen, lr_dis=cfg.lr_dis,
                lambda_gen=cfg.lambda_gen, lambda_dis=cfg.lambda_dis, window_size=cfg.window_size,
                optimizer=_OPTIMIZERS[_cfg(cfg, "engine_optimizer", "adam_dense")],
                device=int(_cfg(cfg, "engine_device", 0)) if self.world == 1 else self.ctl.local_rank)
            if int(_cfg(cfg, "engine_profile_every", 1)) != 1:
                self.engine.set_profiling(int(_cfg(cfg, "engine_profile_every", 1)))
            self.engine.set_graph_csr(self._rowptr, self._col)
            self.ctl.connect_engine(self.engine)  # world > 1: RCCL communicator (unique id over the gloo control plane)
        return self.engine

    def build_generator(self):
        """initializing the generator (table [n_node, n_emb] from the pre-trained rows, zero bias)"""
        self._ensure_engine()
        self.generator = _ModelView(self.engine, 0)

    def build_discriminator(self):
        """initializing the discriminator"""
        self._ensure_engine()
        self.discriminator = _ModelView(self.engine, 1)

    def construct_trees(self, nodes):
        """BFS trees of ``nodes`` (reference :84-108) -> resident tree CSR; returns the slot map
        root -> slot (the reference returns the dict of dicts itself)."""
        nodes = np.asarray(list(nodes), dtype=np.int32)
        self.engine.build_trees(nodes, n_threads=int(_cfg(self.config, "engine_tree_threads", 0)),
                                device=bool(_cfg(self.config, "engine_tree_device", True)))
        self._slot_of_root = {int(r): i for i, r in enumerate(nodes)}
        return self._slot_of_root

    def _tree_cache_path(self, name):
        """File of the native tree cache for ``config.cache_filename``.  The reference pickles to that very name
        (``cache/<dataset>.pkl``, :36-46); a pickle found there is neither readable by gg_load_trees nor ours to replace, so
        the flat GGTR cache lives NEXT to it: ``<name>.ggtr`` (per rank with replicas: every rank caches its own roots)."""
        if not name:
            return None
        path = name if name.endswith(".ggtr") else name + ".ggtr"
        if self.world > 1:
            path = "%s.rank%dof%d" % (path, self.rank, self.world)
        return path

    @staticmethod
    def _is_ours_or_absent(path):
        if not os.path.exists(path):
            return True
        try:
            with open(path, "rb") as f:
                return f.read(4) == b"GGTR"
        except OSError:
            return False

    def _load_tree_cache(self, path):
        """Resident trees from the cache file if it holds exactly the trees of ``root_nodes`` for this graph."""
        try:
            self.engine.load_trees(path)
        except _lib.GraphGANHipError as e:  # foreign / stale / truncated file: rebuild (and overwrite) instead of failing
            print("tree cache %s not used: %s" % (path, e))
            return False
        roots = np.asarray(self.engine.tree_roots)
        if len(roots) != len(self.root_nodes) or not np.array_equal(roots, np.asarray(self.root_nodes, dtype=np.int32)):
            print("tree cache %s holds other roots: rebuilding" % path)
            return False
        self._slot_of_root = {int(r): i for i, r in enumerate(roots)}
        self.trees = self._slot_of_root
        return True

    def construct_trees_with_mp(self, nodes):
        """kept for API compatibility (reference :63-82): the C++ builder is always multi-threaded"""
        self.trees = self.construct_trees(nodes)

    # ------------------------------------------------------------------ training
    @staticmethod
    def stream_id(epoch, inner_epoch, n_inner, for_d):
        """Philox stream of a prepare call: distinct per (outer epoch, inner epoch, phase)."""
        return 2 * (epoch * max(n_inner, 1) + inner_epoch) + (0 if for_d else 1)

    def train(self):
        cfg = self.config
        if os.path.isfile(self.latest_checkpoint) and cfg.load_model:
            print("loading the checkpoint: %s" % self.latest_checkpoint)
            self.engine.load_state(self.latest_checkpoint)

        if self.rank == 0:  # the replicas are identical: one of them writes
            self.write_embeddings_to_file()
            self.evaluation(self)

        print("start training...")
        for epoch in range(cfg.n_epochs):
            print("epoch %d" % epoch)

            if epoch > 0 and epoch % cfg.save_steps == 0 and self.rank == 0:
                os.makedirs(cfg.model_log, exist_ok=True)
                self.engine.save_state(self.latest_checkpoint)

            # D-steps
            t_epoch = time.time()
            train_size = 0
            self._g_pairs_ready = False
            for d_epoch in range(cfg.n_epochs_dis):
                if d_epoch % cfg.dis_interval == 0:
                    self._stream = self.stream_id(epoch, d_epoch, cfg.n_epochs_dis, True)
                    if self._all_resident:
                        train_size = self._prepare_d_resident()
                    else:
                        # the G-mode walks of the epoch's first generator prepare read the generator, the trees and the Q3 bits
                        # only (reference :204-216) -- nothing the remaining D passes change: they share the trees of the LAST
                        # discriminator prepare of the epoch (one BFS per root and outer epoch instead of two).  With
                        # update_ratio < 1 the two prepares draw their own roots (:189, :209): no sharing.
                        last = d_epoch + cfg.dis_interval >= cfg.n_epochs_dis
                        g_too = last and cfg.n_epochs_gen > 0 and cfg.update_ratio >= 1
                        train_size = self._prepare_root_batches(True, g_too, self._stream, self.stream_id(epoch, 0, cfg.n_epochs_gen, False))
                        self._g_pairs_ready = g_too
                if self._all_resident and d_epoch == cfg.n_epochs_dis - 1 and cfg.n_epochs_gen > 0 and cfg.update_ratio >= 1:
                    # the G-mode walks of the first G epoch read the generator only (reference :204-216): started on the
                    # engine's side stream before the last D pass is enqueued, adopted by prepare_g below (same arguments; with
                    # update_ratio < 1 the root draw of that call is not known yet, and nothing is begun)
                    self.engine.prepare_g_begin(self._select_slots(), cfg.n_sample_gen, self.seed, self.stream_id(epoch, 0, cfg.n_epochs_gen, False))
                self.engine.d_pass(self._batch_starts(train_size, cfg.batch_size_dis), cfg.batch_size_dis)

            # G-steps
            train_size = 0
            for g_epoch in range(cfg.n_epochs_gen):
                if g_epoch % cfg.gen_interval == 0:
                    self._stream = self.stream_id(epoch, g_epoch, cfg.n_epochs_gen, False)
                    if self._all_resident:
                        train_size = self._prepare_g_resident()
                    elif g_epoch == 0 and self._g_pairs_ready:
                        train_size = self.engine.epoch_commit(0)  # pairs sampled beside the last D prepare; rewards evaluated now
                        self._g_pairs_ready = False
                    else:
                        train_size = self._prepare_root_batches(False, True, 0, self._stream)
                self.engine.g_pass(self._batch_starts(train_size, cfg.batch_size_gen), cfg.batch_size_gen)

            if self.rank == 0:
                self.write_embeddings_to_file()
                results = self.evaluation(self)
                self._write_perf_log(epoch, time.time() - t_epoch, results)
        print("training completes")

    def _batch_starts(self, train_size, batch_size):
        """One inner epoch's minibatches (reference :149-152, :168-171): the shuffled list of contiguous batch starts over
        THIS rank's prepared rows.  With replicas, step k of the pass is the union of every rank's k-th batch (a global
        batch of up to world * batch_size rows whose gradients the engine sums); ranks with fewer batches pad their list
        with -1 = "nothing from me in this step", so that all ranks issue the same number of steps (each step is a
        collective)."""
        start_list = list(range(0, train_size, batch_size))
        self.host_rng.shuffle(start_list)
        if self.world > 1:
            n_steps = int(self.ctl.max(len(start_list)))
            start_list += [-1] * (n_steps - len(start_list))
        return start_list

    # ------------------------------------------------------------------ sample preparation
    def _draw_roots(self):
        """``np.random.rand() < update_ratio`` per root (reference :189, :209): one draw per root, in root order
        (no draws with update_ratio >= 1: every root is taken, and the oracle trainer does the same).  Returns indices
        into ``root_nodes``."""
        cfg = self.config
        if cfg.update_ratio >= 1:
            return np.arange(len(self.root_nodes), dtype=np.int32)
        self._prepare_no += 1
        if self.world == 1:
            take = np.flatnonzero(self.host_rng.rand(len(self.root_nodes)) < cfg.update_ratio)
        else:  # one draw per ROOT of the whole graph, keyed by the prepare call: the selection does not depend on the sharding
            draws = np.random.RandomState([self.seed, self._prepare_no]).rand(self.n_node)
            take = np.flatnonzero(draws[np.asarray(self.root_nodes, dtype=np.int64)] < cfg.update_ratio)
        return take.astype(np.int32)

    def _select_slots(self):
        """The resident slots of this prepare call's roots (slot i holds root_nodes[i])."""
        assert self._all_resident
        return self._draw_roots()

    def _prepare_root_batches(self, do_d, do_g, stream_d, stream_g):
        """prepare_data_for_d and / or prepare_data_for_g (reference :182-223) over this call's roots when their trees
        cannot all be resident: batch by batch (gg_epoch_add: BFS trees on the GPU, Q3 bits restored / saved, walks, rows
        and pairs appended in root order), then the discriminator rows -- or, for a generator-only call, the pairs -- become
        the prepared data of the passes.  A rank that drew no root still commits (the call holds the replicas' collective)."""
        cfg = self.config
        roots = np.asarray(self.root_nodes, dtype=np.int32)[self._draw_roots()]
        self.engine.epoch_begin(reset_d=do_d, reset_g=do_g)
        for k in range(0, len(roots), self._batch_roots):
            self.engine.epoch_add(roots[k:k + self._batch_roots], do_d, do_g, cfg.n_sample_gen, self.seed, stream_d, stream_g)
        self._slot_of_root = None  # (the resident slots are the last batch's)
        return self.engine.epoch_commit(1 if do_d else 0)

    # An empty draw (update_ratio < 1) still goes through the engine: gg_prepare_* ends with a collective over the replicas
    # (the max of the ranks' row counts) that every rank must join, and it resets the resident row count that the passes and
    # their row-pack capacity read -- skipping the call left stale rows behind and, with world > 1, the other ranks waiting.
    def _prepare_d_resident(self):
        slots = self._select_slots()
        return self.engine.prepare_d(slots, self.seed, self._stream, fetch=False)

    def _prepare_g_resident(self):
        slots = self._select_slots()
        return self.engine.prepare_g(slots, self.config.n_sample_gen, self.seed, self._stream, fetch=False)

    def prepare_data_for_d(self):
        """generate positive and negative samples for the discriminator (reference :182-202);
        returns (center_nodes, neighbor_nodes, labels) and leaves them resident for d_pass"""
        if not hasattr(self, "_stream"):
            self._stream = 0
        if not self._all_resident:
            self._prepare_root_batches(True, False, self._stream, 0)
            center, neighbor, label = self.engine.get_d_data()
            return center.tolist(), neighbor.tolist(), label.astype(np.int64).tolist()
        slots = self._select_slots()
        if len(slots) == 0:
            return [], [], []
        center, neighbor, label, _ = self.engine.prepare_d(slots, self.seed, self._stream)
        return center.tolist(), neighbor.tolist(), label.astype(np.int64).tolist()

    def prepare_data_for_g(self):
        """sample nodes for the generator (reference :204-223); returns (node_1, node_2, reward)"""
        if not hasattr(self, "_stream"):
            self._stream = 1
        if not self._all_resident:
            self._prepare_root_batches(False, True, 0, self._stream)
            n1, n2, reward = self.engine.get_g_data()
            return n1.tolist(), n2.tolist(), reward
        slots = self._select_slots()
        if len(slots) == 0:
            return [], [], np.zeros(0, dtype=np.float32)
        n1, n2, reward, _ = self.engine.prepare_g(slots, self.config.n_sample_gen, self.seed, self._stream)
        return n1.tolist(), n2.tolist(), reward

    def sample(self, root, tree, sample_num, for_d):
        """sample nodes from the BFS-tree of ``root`` (reference :225-270).  ``tree`` is accepted
        for signature compatibility; the resident tree of ``root`` is used.
        Returns (samples, paths), or (None, None) when the reference would."""
        if self._slot_of_root is None or int(root) not in self._slot_of_root:
            if self._all_resident:
                raise KeyError("root %d has no resident tree (not in root_nodes)" % int(root))
            self.trees = self.construct_trees([int(root)])  # on-demand mode: build this root's tree
        slot = self._slot_of_root[int(root)]
        stream = getattr(self, "_stream", 0)
        res = self.engine.walk_sample([slot], [sample_num], for_d, self.seed, stream)
        if res["root_status"][0] == _lib.GG_ROOT_ABORTED:
            return None, None
        paths = [res["paths"][j, : res["path_len"][j]].tolist() for j in range(sample_num)]
        return res["samples"].tolist(), paths

    @staticmethod
    def get_node_pairs_from_path(path):
        """window pairs of a path whose last element (the back-step) is dropped (reference :272-291);
        host twin of the device kernel, kept for API compatibility"""
        window = config.window_size
        nodes = path[:-1]
        pairs = []
        for i, center in enumerate(nodes):
            lo, hi = max(i - window, 0), min(i + window + 1, len(nodes))
            pairs.extend([center, nodes[j]] for j in range(lo, hi) if j != i)
        return pairs

    # ------------------------------------------------------------------ outputs
    def write_embeddings_to_file(self):
        """write embeddings of the generator and the discriminator to files (reference :293-306):
        header ``N<TAB>d``, then ``id<TAB>v0<TAB>...``; values are the fp32 numbers widened to
        fp64 and printed with ``str`` (what ``np.hstack([index, matrix]).tolist()`` gives)"""
        cfg = self.config
        for i in range(2):
            os.makedirs(os.path.dirname(cfg.emb_filenames[i]) or ".", exist_ok=True)
            if _cfg(cfg, "engine_emb_text", True):
                self.engine.write_embeddings(i, cfg.emb_filenames[i])  # native formatter, byte-identical text
            if _cfg(cfg, "engine_emb_sidecar", False):
                self.engine.write_embeddings_bin(i, cfg.emb_filenames[i] + ".bin")  # same numbers, 4 B each

    def _write_perf_log(self, epoch, wall_s, results):
        """One JSON line per outer epoch beside the results file (``<result_filename>.perf.jsonl``): wall time, the engine's
        counters (walks, sampled edges, rows scored, pairs through the passes, optimizer steps, tree builds) and the
        accuracies just appended to the results file (reference :316-319 writes only those)."""
        import json
        cfg = self.config
        c = self.engine.counters()
        prev = getattr(self, "_perf_prev", {})
        delta = {k: c[k] - prev.get(k, 0) for k in ("walks", "hops", "rows_scored", "d_pairs", "g_pairs", "d_steps", "g_steps", "bfs_trees", "bfs_kernel_ms", "walk_reruns")}
        self._perf_prev = c
        rec = {"epoch": int(epoch), "wall_s": wall_s, "sampled_edges_per_sec": delta["hops"] / wall_s if wall_s > 0 else None,
               "trees": "resident" if self._all_resident else "root batches of %d" % self._batch_roots, "world": self.world,
               "results": [r.strip() for r in (results or [])], **delta}
        os.makedirs(os.path.dirname(cfg.result_filename) or ".", exist_ok=True)
        with open(cfg.result_filename + ".perf.jsonl", "a") as f:
            f.write(json.dumps(rec) + "\n")

    @staticmethod
    def evaluation(self):
        cfg = self.config
        results = []
        if cfg.app == "link_prediction":
            for i in range(2):
                # the per-edge dots run on the device (gg_edge_scores) instead of re-reading the text just written
                lpe = lp.LinkPredictEval(cfg.emb_filenames[i], cfg.test_filename, cfg.test_neg_filename, self.n_node, cfg.n_emb,
                                         engine=self.engine, which=i)
                result = lpe.eval_link_prediction()
                results.append(cfg.modes[i] + ":" + str(result) + "\n")
        os.makedirs(os.path.dirname(cfg.result_filename) or ".", exist_ok=True)
        with open(cfg.result_filename, mode="a+") as f:
            f.writelines(results)
        return results


class _ModelView(object):
    """Stand-in for the reference's Generator / Discriminator objects: exposes the variables the
    driver fetches (``embedding_matrix``, ``bias_vector``) as numpy arrays read from the engine."""

    def __init__(self, eng, which):
        self._eng, self._which = eng, which

    @property
    def embedding_matrix(self):
        return self._eng.get_embeddings(self._which)

    @property
    def bias_vector(self):
        return self._eng.get_bias(self._which)


if __name__ == "__main__":
    graph_gan = GraphGAN()
    graph_gan.train()
