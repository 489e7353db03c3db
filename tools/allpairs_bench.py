#!/usr/bin/env python3
"""K7 at BASELINE.json configs[4] size: rows of generator.all_score (generator.py:21) streamed through the fused consumer
(gg_all_score_reduce: max / argmax / log-sum-exp per row, nothing of size rows x N materialised) on the matrix cores.
    python tools/allpairs_bench.py [n_node] [n_emb] [n_rows]
One JSON line: TFLOP/s against the dense MFMA peak of the dtype (MI355X_MICROARCH.md: fp32-input MFMA 157.3 TF, bf16 ~2 500 TF)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphgan_amd as ga  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
r = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
rs = np.random.default_rng(5)
emb = rs.standard_normal((n, d), dtype=np.float32) * np.float32(0.6 * np.sqrt(50.0 / d))
eng = ga.Engine(emb, emb[:1].repeat(n, 0) if False else emb, optimizer=ga.GG_OPT_SGD)  # SGD: no Adam slots (the tables alone are 2 x 10 GB)
rows = np.sort(rs.choice(n, r, replace=False)).astype(np.int32)
out = {"workload": "all-pairs rows: %d rows x %d nodes, n_emb=%d, fused consumer (max, argmax, logsumexp)" % (r, n, d), "flop": 2.0 * r * n * d}
only = os.environ.get("ALLPAIRS_ONLY")  # (kernel work: "bf16" skips the fp32 passes)
ref_max = None
for prec, peak in (("fp32", 157.3), ("bf16", 2500.0)):
    if only and prec != only:
        continue
    for lse in (True, False):  # consumer with the log-sum-exp (one exponential per score) / max + argmax only
        best = None
        for rep in range(3):
            t0 = time.time()
            res = eng.all_score_reduce(rows, precision=prec, logsumexp=lse)
            wall = time.time() - t0
            best = res["kernel_ms"] if best is None else min(best, res["kernel_ms"])
        tf = out["flop"] / (best * 1e-3) / 1e12
        out[prec + ("" if lse else "_max_argmax_only")] = {
            "kernel_ms": best, "call_s_last": wall, "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "bound": "mfma",
            "table_bytes_streamed_per_row_tile": (2 if prec == "bf16" else 4) * n * d,
            "instruction": ("v_mfma_f32_32x32x16_bf16 (rows >= 512: all_score_reduce_bf16_x32_kernel, requested rows on the lanes; fewer rows: all_score_reduce_bf16_kernel)" if prec == "bf16" else "v_mfma_f32_32x32x2_f32")}
        if prec == "fp32" and lse:
            ref_max = res["max"].copy()
        elif prec == "bf16" and lse and ref_max is not None:
            out["bf16_vs_fp32_max_abs_diff_of_row_max"] = float(np.max(np.abs(res["max"] - ref_max)))
eng.close()
print(json.dumps(out))
