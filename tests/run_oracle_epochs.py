#!/usr/bin/env python3
"""Run the CPU oracle trainer (reference schedule, dense TF1-Adam, spec-arithmetic walks) on the
CA-GrQc fixture for a few outer epochs and record the gen/dis link-prediction accuracy after each
-- the curve the HIP engine's run with the same seed is compared against.
    python tests/run_oracle_epochs.py <n_epochs> <out.json> [seed] [n_inner]
n_inner (default 30 = config.py:10-13) sets n_epochs_gen = n_epochs_dis = gen_interval = dis_interval: the SHORT schedule
n_inner = 2 is the informative float-parity workload of round 4 -- the generator's accuracy climbs from 0.76 to ~0.87 and stays
far from chance, so a per-seed +-0.5 % gate means something (tests/golden/oracle_epochs_short.json)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import graphgan_oracle as orc  # noqa: E402
from tests.helpers import ca_grqc_init_embeddings, load_ca_grqc  # noqa: E402


def main():
    n_epochs, out = int(sys.argv[1]), sys.argv[2]
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # walk / shuffle seed; the initial embeddings stay those of seed 0
    n_inner = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    d, n, graph = load_ca_grqc()
    init = ca_grqc_init_embeddings(d, n, seed=0)
    cfg = orc.Config()
    cfg.n_epochs_gen = cfg.n_epochs_dis = cfg.gen_interval = cfg.dis_interval = n_inner
    o = orc.GraphGANOracle(n, graph, init, init, cfg=cfg, rng="counter", arith="spec", seed=seed)
    test, neg = d["test"].tolist(), d["test_neg"].tolist()
    res = {"epochs": [], "seconds": [], "n_inner": n_inner, "seed": seed}
    res["epochs"].append([orc.eval_link_prediction(o.generator.E.astype(np.float64), test, neg),
                          orc.eval_link_prediction(o.discriminator.E.astype(np.float64), test, neg)])
    for ep in range(n_epochs):
        t = time.time()
        o.train_epoch(ep)
        res["seconds"].append(time.time() - t)
        res["epochs"].append([orc.eval_link_prediction(o.generator.E.astype(np.float64), test, neg),
                              orc.eval_link_prediction(o.discriminator.E.astype(np.float64), test, neg)])
        json.dump(res, open(out, "w"))
        print(ep, res["epochs"][-1], res["seconds"][-1], flush=True)


if __name__ == "__main__":
    main()
