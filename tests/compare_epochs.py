#!/usr/bin/env python3
"""Accuracy parity table of DESIGN.md section 8 from the committed runs:
    python tests/compare_epochs.py [tests/golden/oracle_epochs_multiseed.json] [profiles/r1_engine_epochs_multiseed.json]
Both files hold, per walk/shuffle seed, the gen/dis link-prediction accuracy before training and after each outer
epoch of the full reference schedule on CA-GrQc (oracle: tests/run_oracle_epochs.py, engine: tests/run_engine_epochs.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    fo = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "oracle_epochs_multiseed.json")
    fe = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r1_engine_epochs_multiseed.json")
    o, e = json.load(open(fo))["epochs"], json.load(open(fe))["epochs"]
    seeds = sorted(set(o) & set(e), key=int)
    n_ep = max(min(len(o[s]), len(e[s])) for s in seeds)
    print("%-12s %-6s %-19s %-19s %-30s %s" % ("after epoch", "seeds", "oracle mean g/d", "engine mean g/d", "mean paired diff g/d [%] (+-sem)", "max per-seed |diff| g/d [%]"))
    for k in range(n_ep):
        have = [s for s in seeds if min(len(o[s]), len(e[s])) > k]
        O = np.array([o[s][k] for s in have])
        E = np.array([e[s][k] for s in have])
        d = 100 * (E - O)
        sem = d.std(0, ddof=1) / np.sqrt(len(have)) if len(have) > 1 else np.zeros(2)
        mx = np.abs(d).max(0)
        print("%-12s %-6d %.5f / %.5f   %.5f / %.5f   %+.2f (%.2f) / %+.2f (%.2f)        %.2f / %.2f" % (
            "(before)" if k == 0 else str(k - 1), len(have), O[:, 0].mean(), O[:, 1].mean(), E[:, 0].mean(), E[:, 1].mean(),
            d[:, 0].mean(), sem[0], d[:, 1].mean(), sem[1], mx[0], mx[1]))
    variant_table(fo)


def variant_table(fo):
    d = json.load(open(fo))
    if "epochs_sigmoid64" not in d:
        return
    o, v = d["epochs"], d["epochs_sigmoid64"]
    seeds = sorted(set(o) & set(v), key=int)
    n_ep = min(min(len(o[s]), len(v[s])) for s in seeds)
    print("\noracle with the fp64-evaluated sigmoid (<= 1 ulp per value) minus the committed oracle, seeds %s:" % ", ".join(seeds))
    for k in range(n_ep):
        dd = 100 * (np.array([v[s][k] for s in seeds]) - np.array([o[s][k] for s in seeds]))
        sem = dd.std(0, ddof=1) / np.sqrt(len(seeds))
        print("%-12s mean paired diff g/d [%%]: %+.2f (%.2f) / %+.2f (%.2f)" % ("(before)" if k == 0 else str(k - 1), dd[:, 0].mean(), sem[0], dd[:, 1].mean(), sem[1]))


if __name__ == "__main__":
    main()
