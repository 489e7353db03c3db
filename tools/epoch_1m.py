#!/usr/bin/env python3
"""One outer epoch of GraphGAN.train() over ALL roots of the 1M-node / 10M-edge power-law workload (BASELINE.json configs[3],
one GPU's share) -- update_ratio = 1, lazy Adam, fused minibatches: 12 TB of trees cannot be resident, so the trainer runs the
epoch over root batches (gg_epoch_*: BFS per batch on the GPU, Q3 bits in the persistent store, D and G walks on the same trees).
    python tools/epoch_1m.py [n_node] [out.json] [tree budget in GB]
Writes the reference's directory layout under a temporary directory, runs the trainer, prints / stores one JSON record."""
import json
import os
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphgan_amd as ga  # noqa: E402
from graphgan_amd import config as base_cfg, workloads  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    out = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 160.0
    d = 128
    import pandas as pd
    t0 = time.time()
    w = workloads.powerlaw_split_workload(n, 10, d, test_frac=0.001)
    base = tempfile.mkdtemp(prefix="gg_epoch_")
    os.makedirs(base + "/data/link_prediction")
    os.makedirs(base + "/pre_train/link_prediction")
    # the training edges in file order: edge k = (col of the k-th "forward" adjacency entry ...) -- rebuild them from the split
    edges = ga.synth_powerlaw(n, 10, 1, 2)
    held = np.zeros(len(edges), dtype=bool)
    held[np.random.RandomState(3).permutation(len(edges))[: int(round(0.001 * len(edges)))]] = True
    pd.DataFrame(edges[~held]).to_csv(base + "/data/link_prediction/pl_train.txt", sep="\t", header=False, index=False)
    pd.DataFrame(w["test"]).to_csv(base + "/data/link_prediction/pl_test.txt", sep="\t", header=False, index=False)
    pd.DataFrame(w["test_neg"]).to_csv(base + "/data/link_prediction/pl_test_neg.txt", sep="\t", header=False, index=False)
    with open(base + "/pre_train/link_prediction/pl_pre_train.emb", "w") as f:
        f.write("0 %d\n" % d)  # no pre-trained rows: the trainer draws them, the workload's rows are uploaded below
    cfg = types.SimpleNamespace(**{k: getattr(base_cfg, k) for k in dir(base_cfg) if not k.startswith("_")})
    cfg.train_filename = base + "/data/link_prediction/pl_train.txt"
    cfg.test_filename = base + "/data/link_prediction/pl_test.txt"
    cfg.test_neg_filename = base + "/data/link_prediction/pl_test_neg.txt"
    cfg.pretrain_emb_filename_d = cfg.pretrain_emb_filename_g = base + "/pre_train/link_prediction/pl_pre_train.emb"
    cfg.emb_filenames = [base + "/results/link_prediction/pl_gen_.emb", base + "/results/link_prediction/pl_dis_.emb"]
    cfg.result_filename = base + "/results/link_prediction/pl.txt"
    cfg.model_log = base + "/log/"
    cfg.cache_filename = base + "/cache/pl.pkl"
    cfg.n_emb = d
    cfg.n_epochs, cfg.n_epochs_dis, cfg.n_epochs_gen, cfg.dis_interval, cfg.gen_interval = 1, 1, 1, 1, 1
    cfg.batch_size_dis = cfg.batch_size_gen = 1 << 22
    cfg.update_ratio = 1
    cfg.engine_optimizer = "adam_lazy"
    cfg.engine_profile_every = 0
    cfg.engine_emb_text, cfg.engine_emb_sidecar = False, True
    cfg.engine_tree_budget_gb = budget
    prep_s = time.time() - t0
    from graphgan_amd.graph_gan import GraphGAN
    t1 = time.time()
    g = GraphGAN(cfg)
    assert not g._all_resident
    g.engine.set_embeddings(0, w["emb"])
    g.engine.set_embeddings(1, w["emb"])
    init_s = time.time() - t1
    c0 = g.engine.counters()
    t2 = time.time()
    g.train()
    g.engine.synchronize()
    train_s = time.time() - t2
    c = g.engine.counters()
    perf = [json.loads(l) for l in open(cfg.result_filename + ".perf.jsonl")]
    rec = {"workload": "synthetic power-law (Barabasi-Albert m=10): %d nodes / %d train edges, n_emb=%d" % (n, int((~held).sum()), d),
           "trainer": "GraphGAN.train(): 1 outer epoch, update_ratio 1, adam_lazy, fused minibatches of %d rows, root batches of %d" % (cfg.batch_size_gen, g._batch_roots),
           "roots": g.n_node, "train_s": train_s, "epoch_wall_s": perf[0]["wall_s"], "init_s": init_s, "files_s": prep_s,
           "bfs_trees": c["bfs_trees"] - c0["bfs_trees"], "bfs_kernel_s": 1e-3 * (c["bfs_kernel_ms"] - c0["bfs_kernel_ms"]),
           "bfs_us_per_tree": 1e3 * (c["bfs_kernel_ms"] - c0["bfs_kernel_ms"]) / max(c["bfs_trees"] - c0["bfs_trees"], 1),
           "hops": c["hops"] - c0["hops"], "d_pairs": c["d_pairs"], "g_pairs": c["g_pairs"], "d_steps": c["d_steps"], "g_steps": c["g_steps"],
           "sampled_edges_per_sec_whole_epoch": (c["hops"] - c0["hops"]) / perf[0]["wall_s"], "walk_reruns": c["walk_reruns"],
           "results": perf[0]["results"]}
    g.engine.close()
    print(json.dumps(rec))
    if out:
        json.dump(rec, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
