#!/usr/bin/env python3
"""bench.py -- the GraphGAN hot path on MI355X, one JSON line.

Metric (BASELINE.json): sampled-edges/sec of the graph-softmax walk sampler, with the D-step /
G-step pairs/sec beside it.  One *step* = one epoch body of the reference's ``train()``
(src/GraphGAN/graph_gan.py:144-176 with one inner pass each) over this rank's R root slots:

    prepare_data_for_d (D-mode walks, deg(root) each)  ->  one D optimizer pass over those rows
    prepare_data_for_g (G-mode walks, n_sample_gen=20, window pairs, rewards) -> one G pass

``value`` = sampled edges (hops of both walk launches, all ranks) / wall time of the K timed
steps, inputs resident in HBM.  Workload at N=1: synthetic power-law graph, 1M nodes / 10M
edges, n_emb = 128 (the configuration the north-star target is quoted on), R roots per step.
Weak scaling: every rank walks its own R roots; replicas exchange gradients through RCCL.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--workload", default="powerlaw", choices=["powerlaw", "ca_grqc"])
    p.add_argument("--nodes", type=int, default=1_000_000)
    p.add_argument("--m", type=int, default=10)
    p.add_argument("--emb", type=int, default=128)
    p.add_argument("--roots", type=int, default=8192, help="root slots per rank per step (8192 trees of the 1M-node graph = 96 GB of the 288 GB HBM)")
    p.add_argument("--n-sample-gen", type=int, default=20)
    p.add_argument("--optimizer", default="adam_lazy", choices=["adam_dense", "adam_lazy", "sgd"])
    p.add_argument("--threads", type=int, default=0, help="host BFS threads (0 = min(64, cores))")
    p.add_argument("--host-bfs", action="store_true", help="build the BFS trees on host threads instead of on the GPU")
    p.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--seed", type=int, default=6)
    p.add_argument("--profile-every", type=int, default=3,
                   help="HIP events around every k-th walk launch of the timed region (an event pair costs ~6 us of stream bubble on "
                        "each side of every level_score_kernel launch; an odd k alternates between the D-mode and G-mode launches)")
    return p.parse_args()


def make_workload(args, ga):
    if args.workload == "ca_grqc":
        g = np.load(os.path.join(ROOT, "tests", "golden", "ca_grqc.npz"))
        n = int(g["n_node"])
        rowptr, col = ga.edges_to_csr(n, g["train"])
        emb = np.random.RandomState(5).rand(n, g["emb_rows"].shape[1])
        emb[g["emb_ids"]] = g["emb_rows"]
        name = "CA-GrQc (5242 nodes, 13046 train edges), n_emb=50, shipped pre-trained embeddings"
        return n, rowptr, col, emb.astype(np.float32), name
    from graphgan_amd import workloads
    n, d = args.nodes, args.emb
    rowptr, col, emb, n_edges = workloads.powerlaw_workload(n, args.m, d)  # SURVEY.md section 8d recipe (also what tests/test_gpu_scale.py checks)
    name = "synthetic power-law (Barabasi-Albert m=%d): %d nodes / %d edges, n_emb=%d" % (args.m, n, n_edges, d)
    return n, rowptr, col, emb, name


def cpu_baseline(args, n, rowptr, col, emb, bias, roots, seconds):
    """The C oracle ('port' of graph_gan.py:225-270 with scores computed on demand, i.e. the
    'hoisted' fair baseline of BASELINE.md section 3) on ONE host core, G-mode walks over a bounded
    sample of this workload's roots; returns (edges/s, sample description)."""
    from oracle import graphgan_oracle as orc
    Ep = orc.pad_rows(emb)
    # a representative sample: evenly spaced through the degree-sorted root list (not just the hub roots)
    rts = np.ascontiguousarray(roots[:: max(1, len(roots) // 48)][:48])
    # trees are a cached precompute in the reference too (graph_gan.py:31-46): not timed
    off, nbr, base, dmax = orc.c_build_trees(n, rowptr, col, rts)
    slots = np.arange(len(rts), dtype=np.int32)
    nw = np.full(len(rts), args.n_sample_gen, dtype=np.int32)
    hops, t, stream = 0, 0.0, 1
    while t < seconds:
        t0 = time.perf_counter()
        res = orc.c_walk_sample(Ep, bias, off, nbr, base, rts, slots, nw, False, args.seed, stream, dmax + 3)
        t += time.perf_counter() - t0
        hops += int(res["hops"])
        stream += 2
    return hops / t, "G-mode walks (%d per root) from %d roots of the same graph, repeated over %d RNG streams: %d hops in %.1f s, scores computed on demand (hoisted flavour)" % (
        args.n_sample_gen, len(rts), stream // 2, hops, t)


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/summarize.py; FETCH_SIZE / WRITE_SIZE collected in separate passes of this same
    command, gfx950 correction applied).  PMC counters cannot be read from inside the bench."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not files:
        return None, None
    data = json.load(open(files[-1]))
    for name, v in data.items():
        if kernel_substr in name:
            return v["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    return None, None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share_gpu = os.environ.get("GG_BENCH_SHARE_GPU") == "1"  # plumbing test on a 1-GPU box: all ranks on device 0, no RCCL
    if share_gpu:
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import graphgan_amd as ga  # loads libgraphgan_hip.so (and with it the HIP runtime) before anything else
    from graphgan_amd import _lib

    from graphgan_amd import parallel
    ctl = parallel.Control(rank, world)  # control plane only (gloo on CPU): barrier, max, RCCL id broadcast

    t_setup = time.time()
    n, rowptr, col, emb, wl_name = make_workload(args, ga)
    opt = {"adam_dense": _lib.GG_OPT_ADAM_DENSE, "adam_lazy": _lib.GG_OPT_ADAM_LAZY, "sgd": _lib.GG_OPT_SGD}[args.optimizer]
    eng = ga.Engine(emb, emb, optimizer=opt, device=local_rank)
    eng.set_graph_csr(rowptr, col)
    from graphgan_amd import workloads
    roots = workloads.bench_roots(rowptr, args.roots, rank, world, args.seed)  # longest (hub) roots first
    R = len(roots)
    threads = args.threads or min(64, os.cpu_count() or 1)
    t_trees = time.time()
    if args.host_bfs:
        eng.build_trees(roots, n_threads=max(1, threads // max(1, min(world, 8))))
    else:
        eng.build_trees(roots, device=True)
    trees_s = time.time() - t_trees
    slots = np.arange(len(roots), dtype=np.int32)
    if not share_gpu:
        ctl.connect_engine(eng)
    setup_s = time.time() - t_setup
    eng.set_profiling(args.profile_every)

    def step(i):
        # every rank issues both passes (they contain the replicas' gradient exchange), even with no rows
        rows = eng.prepare_d(slots, args.seed, 2 * i, fetch=False)
        eng.d_pass(np.zeros(1, np.int64), max(int(rows), 1))
        pairs = eng.prepare_g(slots, args.n_sample_gen, args.seed, 2 * i + 1, fetch=False)
        eng.g_pass(np.zeros(1, np.int64), max(int(pairs), 1))

    def barrier():
        eng.comm_barrier()
        ctl.barrier()

    for i in range(args.warmup):
        step(i)
    barrier()
    eng.set_profiling(args.profile_every)  # restart the cadence: the first launch of the timed region is a profiled one
    c0 = eng.counters()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    c1 = eng.counters()

    hops = c1["hops"] - c0["hops"]
    reads = c1["nbr_reads"] - c0["nbr_reads"]
    rows_scored = c1["rows_scored"] - c0["rows_scored"]
    walk_ms = c1["walk_kernel_ms"] - c0["walk_kernel_ms"]      # HIP events of the profiled walk launches ...
    launches = c1["walk_launches"] - c0["walk_launches"]       # ... and how many of the 2 * steps launches that was
    calls = 2 * args.steps
    dpairs = c1["d_pairs"] - c0["d_pairs"]
    gpairs = c1["g_pairs"] - c0["g_pairs"]
    sums = ctl.sum([hops, dpairs, gpairs])
    tot = np.array([sums[0], sums[1], sums[2], ctl.max(dt)])
    if rank != 0:
        eng.close()
        return

    d = eng.n_emb
    # Dominant kernel: level_score_kernel (streams the neighbour rows).  Algorithmic bytes per 16-lane work
    # item ("chunk" of <= 16 candidates of one (root, node) distribution): 16 B descriptor + 4d current row;
    # per candidate row: 4 B id + 4d row + 4 B bias + 4 B score written = 4(d+3)   
    sc_ms = c1["score_kernel_ms"] - c0["score_kernel_ms"]
    sc_launches = c1["score_launches"] - c0["score_launches"]
    sc_chunks = c1["score_chunks"] - c0["score_chunks"]
    sc_rows = c1["score_rows"] - c0["score_rows"]  # rows / chunks / ms / launches: the same (profiled) launches
    sc_bytes = 4.0 * (d + 3) * sc_rows + (4.0 * d + 16.0) * sc_chunks
    achieved = sc_bytes / (sc_ms * 1e-3) / 1e9 if sc_ms > 0 else 0.0
    # The reference evaluates every hop's distribution from scratch (4k(d+2) + 4d + 12 per hop);
    # the engine evaluates each distinct (root, node) distribution of a launch once.
    ref_bytes = 4.0 * (d + 2) * reads + (4.0 * d + 12.0) * hops
    traffic, traffic_src = pmc_traffic("level_score_kernel") if args.workload == "powerlaw" and args.nodes == 1_000_000 else (None, None)
    out = {
        "metric": "sampled_edges_per_sec",
        "value": tot[0] / tot[3],
        "unit": "edges/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * tot[3] / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl_name, "roots_per_gpu_per_step": int(R), "n_sample_gen": args.n_sample_gen,
                   "optimizer": args.optimizer, "step": "prepare_d + d_pass + prepare_g + g_pass (graph_gan.py:144-176, one inner pass each)",
                   "parallelism": "roots sharded x%d, replicated tables, RCCL sparse gradient all-gather per pass" % world if world > 1 else "single GPU"},
        "d_step_pairs_per_sec": tot[1] / tot[3],
        "g_step_pairs_per_sec": tot[2] / tot[3],
        "walk_kernel_edges_per_sec": (hops / calls) / (walk_ms / launches * 1e-3) if walk_ms > 0 and launches else None,
        "hops_per_step_rank0": hops / args.steps,
        "mean_k": reads / max(hops, 1),
        "rows_scored_per_step_rank0": rows_scored / args.steps,
        "nbr_reads_per_step_rank0": reads / args.steps,
        "setup_s": setup_s,
        "tree_build_s": trees_s,
        "tree_build": "host threads" if args.host_bfs else "gpu bfs",
        "roofline": {"kernel": "level_score_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": sc_bytes / max(sc_launches, 1), "avg_launch_ms": sc_ms / max(sc_launches, 1),
                     "launches": int(sc_launches), "rows_per_launch": sc_rows / max(sc_launches, 1),
                     "timed": "HIP events on the engine's stream around every level_score_kernel launch of every %s walk call of the timed region"
                              % ("" if args.profile_every == 1 else "%d-th" % args.profile_every),
                     # tools/gather_bw*.hip on the same chip: random 512 B row gathers with 16-lane groups
                     # tools/gather_bw2.hip / gather_bw3.hip on the same chip (profiles/r1_gather_bw*.txt), TB/s of rows resp. in this
                     # kernel's accounting: the separate bias gather is what costs a fifth of the plain gather's rate
                     "gather_microbench_GBs": {"plain_row_gather": 7400.0, "with_dot_and_score_store": 7200.0, "with_bias_gather": 5650.0,
                                               "kernel_shape_16_row_items": 6300.0, "kernel_shape_64_row_items": 6200.0}},
        "walk_phase": {"ms_per_walk_sample_call": walk_ms / max(launches, 1), "calls": calls, "calls_timed": int(launches),
                       "reference_equivalent_bytes_per_call": ref_bytes / calls,
                       "reference_equivalent_GBs": (ref_bytes / calls) / (walk_ms / launches * 1e-3) / 1e9 if walk_ms > 0 and launches else None,
                       "distributions_shared": reads / max(rows_scored, 1)},
    }
    if world == 1 and not args.no_cpu_baseline:
        bias = eng.get_bias(0)
        embg = eng.get_embeddings(0)
        v, sample = cpu_baseline(args, n, rowptr, col, embg, bias, roots, args.cpu_baseline_seconds)
        out["cpu_baseline"] = {"value": v, "unit": "edges/s", "cores": 1, "kind": "port", "sample": sample}
        out["walk_kernel_vs_cpu"] = out["walk_kernel_edges_per_sec"] / v if v > 0 and out["walk_kernel_edges_per_sec"] else None
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
