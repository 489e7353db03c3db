#!/usr/bin/env python3
"""Merge per-seed outputs of tests/run_oracle_epochs.py (or run_engine_epochs.py) into one file for tests/compare_epochs.py:
    python tools/merge_epoch_runs.py <out.json> <what> <pattern with {seed}> <seed> [<seed> ...]"""
import json
import sys

out, what, pattern = sys.argv[1:4]
res = {"what": what, "epochs": {}, "seconds": {}}
for s in sys.argv[4:]:
    d = json.load(open(pattern.format(seed=s)))
    res["epochs"][s] = d["epochs"]
    if "seconds" in d:
        res["seconds"][s] = d["seconds"]
    elif "seconds_per_epoch" in d:
        res["seconds"][s] = d["seconds_per_epoch"]
json.dump(res, open(out, "w"), indent=1)
print(out, {s: len(v) for s, v in res["epochs"].items()})
