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
    n_ep = min(min(len(o[s]) for s in seeds), min(len(e[s]) for s in seeds))
    O = np.array([o[s][:n_ep] for s in seeds])
    E = np.array([e[s][:n_ep] for s in seeds])
    print("seeds: %s" % ", ".join(seeds))
    print("%-16s %-19s %-19s %-21s %s" % ("after epoch", "oracle mean g/d", "engine mean g/d", "diff of means g/d [%]", "max per-seed |diff| g/d [%]"))
    for k in range(n_ep):
        mo, me = O[:, k].mean(0), E[:, k].mean(0)
        dm = 100 * (me - mo)
        mx = 100 * np.abs(E[:, k] - O[:, k]).max(0)
        print("%-16s %.5f / %.5f   %.5f / %.5f   %+.2f / %+.2f          %.2f / %.2f" % (
            "(before)" if k == 0 else str(k - 1), mo[0], mo[1], me[0], me[1], dm[0], dm[1], mx[0], mx[1]))


if __name__ == "__main__":
    main()
