#!/usr/bin/env python3
"""The strict_mode_1m line of bench.py on its own (batch 64, lazy Adam / SGD on the 1M-node bench graph): python tools/strict_line_1m.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import graphgan_amd as ga  # noqa: E402
from graphgan_amd import _lib, workloads  # noqa: E402

sys.argv = ["bench.py"]
args = bench.parse()
n, rowptr, col, emb, _ = bench.make_workload(args, ga)
roots = workloads.bench_roots(rowptr, 2048, 0, 1, args.seed)
print(json.dumps(bench.strict_mode_at_scale(ga, _lib, n, rowptr, col, emb, roots, args.n_sample_gen, args.seed)))
