"""world_size-2 gloo tests (CPU) of the multi-GPU construction: root sharding is a partition,
sharded walks equal single-process walks (counter RNG keyed by root id), and summing per-rank
gradients before one optimizer step equals the single-process step on the union batch -- the two
facts the RCCL path relies on.  The compute here is the CPU oracle (tests may use it); the RCCL
all-reduce itself can only run on GPUs."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), GG_CTL_BACKEND="gloo")
    import torch.distributed as dist
    from graphgan_amd import parallel
    from oracle import graphgan_oracle as orc
    from tests.helpers import load_small
    ctl = parallel.Control()
    assert (ctl.rank, ctl.world) == (rank, world)

    g, n, graph = load_small(3)
    rowptr, col = orc.graph_to_csr(n, graph)
    deg = (rowptr[1:] - rowptr[:-1]).astype(np.int32)
    roots = np.arange(n, dtype=np.int32)
    mine = parallel.shard_roots(roots, rank, world, weights=deg)

    # (a) partition: gather ownership
    own = np.zeros(n)
    own[mine] = 1
    assert np.array_equal(ctl.sum(own), np.ones(n))

    # (b) sharded G-mode walks == the single-process walks of the same roots
    Ep = orc.pad_rows(g["E"])
    off, nbr, base, dmax = orc.c_build_trees(n, rowptr, col, mine)
    res = orc.c_walk_sample(Ep, g["b"], off, nbr, base, mine, np.arange(len(mine), dtype=np.int32),
                            np.full(len(mine), 20, np.int32), False, 5, 1, dmax + 3)
    stride = 64
    mine_paths = np.full((n, 20, stride), 0, dtype=np.int64)
    for i, r in enumerate(mine):
        for j in range(20):
            L = res["path_len"][i * 20 + j]
            mine_paths[r, j, :L] = res["paths"][i * 20 + j, :L] + 1
    all_paths = ctl.sum(mine_paths.ravel()).reshape(n, 20, stride)

    # (c) gradient exchange: local dense gradients, summed, one TF1-Adam step
    dis = orc.Discriminator(g["E"], 1e-3)
    rs = np.random.RandomState(0)
    u, v = rs.randint(0, n, 256), rs.randint(0, n, 256)
    lab = (rs.rand(256) < 0.5).astype(np.float32)
    sl = slice(rank, None, world)
    _, gu, gv, gb = dis.loss_and_grads(u[sl], v[sl], lab[sl], 1e-5)
    GE, Gb = np.zeros((n, g["E"].shape[1])), np.zeros(n)
    np.add.at(GE, u[sl], gu)
    np.add.at(GE, v[sl], gv)
    np.add.at(Gb, v[sl], gb)
    GE, Gb = ctl.sum(GE.ravel()).reshape(GE.shape), ctl.sum(Gb)
    rows = np.flatnonzero(np.abs(GE).sum(1) + np.abs(Gb) > 0)
    dis.opt.step([dis.E, dis.b], [(rows, GE[rows].astype(np.float32)), (rows, Gb[rows].astype(np.float32))])
    t_max = ctl.max(float(rank + 1))
    ctl.barrier()
    if rank == 0:
        np.savez(out, paths=all_paths, E=dis.E, b=dis.b, t_max=t_max)
    dist.destroy_process_group()


def test_two_rank_sharding_and_gradient_sum(tmp_path):
    import torch.multiprocessing as mp
    from oracle import graphgan_oracle as orc
    from tests.helpers import load_small
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    assert got["t_max"] == 2.0

    g, n, graph = load_small(3)
    rowptr, col = orc.graph_to_csr(n, graph)
    roots = np.arange(n, dtype=np.int32)
    off, nbr, base, dmax = orc.c_build_trees(n, rowptr, col, roots)
    res = orc.c_walk_sample(orc.pad_rows(g["E"]), g["b"], off, nbr, base, roots, roots, np.full(n, 20, np.int32), False, 5, 1, dmax + 3)
    want = np.zeros((n, 20, 64), dtype=np.int64)
    for w in range(n * 20):
        L = res["path_len"][w]
        want[w // 20, w % 20, :L] = res["paths"][w, :L] + 1
    assert np.array_equal(got["paths"].astype(np.int64), want)

    dis = orc.Discriminator(g["E"], 1e-3)
    rs = np.random.RandomState(0)
    u, v = rs.randint(0, n, 256), rs.randint(0, n, 256)
    lab = (rs.rand(256) < 0.5).astype(np.float32)
    dis.d_step(u, v, lab, 1e-5)
    assert np.allclose(got["E"], dis.E, rtol=1e-5, atol=1e-6) and np.allclose(got["b"], dis.b, rtol=1e-5, atol=1e-6)


def test_shard_roots_properties():
    sys.path.insert(0, ROOT)
    from graphgan_amd import parallel
    roots = np.arange(1000) * 3
    for world in (1, 2, 3, 8):
        parts = [parallel.shard_roots(roots, r, world) for r in range(world)]
        assert sorted(np.concatenate(parts).tolist()) == roots.tolist()
        w = np.random.RandomState(world).pareto(1.5, len(roots)) + 1
        parts = [parallel.shard_roots(roots, r, world, weights=w) for r in range(world)]
        assert sorted(np.concatenate(parts).tolist()) == roots.tolist()
        loads = [w[np.isin(roots, p)].sum() for p in parts]
        assert max(loads) <= min(loads) + w.max() + 1e-6


class _ReplicaEngine(object):
    """Stand-in for the HIP engine on one rank of a 2-rank GraphGAN.train(): it owns a small "model" vector, every
    optimizer step all-reduces a deterministic gradient of the rows this rank contributes (over gloo, like the engine
    does over RCCL) and applies the sum -- a step count mismatch between the ranks hangs the collective, a rank that
    skips its empty steps ends with a different replica."""

    def __init__(self, emb_g, emb_d, **kw):
        import torch.distributed as dist
        self.dist = dist
        self.kw = kw
        self.E = [np.array(emb_g, np.float32), np.array(emb_d, np.float32)]
        self.n_node, self.n_emb = self.E[0].shape
        self.model = [np.zeros(64), np.zeros(64)]
        self.steps = [0, 0]
        self.calls = []
        self.tree_roots = None
        self.rows = [0, 0]

    def tree_bytes_estimate(self, n_roots): return 8.0 * n_roots * (self.n_node + 1)
    def set_profiling(self, k): pass
    def set_graph_csr(self, rowptr, col): self.deg = np.diff(rowptr)
    def comm_unique_id(self): return b"x" * 128
    def comm_init(self, uid, rank, world): self.calls.append(("comm_init", uid, rank, world)); self.rank, self.world = rank, world
    def build_trees(self, roots, **kw): self.tree_roots = [int(r) for r in roots]
    def save_trees(self, path): pass
    def load_state(self, path): pass
    def save_state(self, path): open(path, "w").write("x")

    def prepare_d(self, slots, seed, stream, fetch=True):
        self.rows[1] = int(2 * self.deg[np.asarray(self.tree_roots)[np.asarray(slots)]].sum())  # 2 * deg rows per root
        return self.rows[1]

    def prepare_g_begin(self, slots, n_sample, seed, stream): pass   # (a head start on the device: no collective, no result)
    def counters(self): return {k: 0 for k in ("walks", "hops", "rows_scored", "d_pairs", "g_pairs", "d_steps", "g_steps", "bfs_trees", "bfs_kernel_ms", "walk_reruns")}

    def prepare_g(self, slots, n_sample, seed, stream, fetch=True):
        self.rows[0] = 37 * len(slots) + 5 * self.rank
        return self.rows[0]

    def _pass(self, which, starts, batch):
        import torch
        for s in starts:
            g = np.zeros(64)
            if s >= 0:
                n = min(batch, self.rows[which] - s)
                assert n > 0
                g[(s // batch) % 64] += n * (1 + self.rank)
            t = torch.from_numpy(g)
            self.dist.all_reduce(t)
            self.model[which] = self.model[which] * 0.9 + t.numpy()
            self.steps[which] += 1

    def d_pass(self, starts, batch): self._pass(1, starts, batch)
    def g_pass(self, starts, batch): self._pass(0, starts, batch)
    def get_embeddings(self, which): return self.E[which]
    def get_bias(self, which): return np.zeros(self.n_node, np.float32)
    def edge_scores(self, which, u, v): return np.arange(len(u), dtype=np.float64)

    def write_embeddings(self, which, path):
        self.calls.append(("write_embeddings", which))
        open(path, "w").write("%d\t%d\n" % (self.n_node, self.n_emb))


def _trainer_worker(rank, world, port, base, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), GG_CTL_BACKEND="gloo")
    import torch
    import torch.distributed as dist
    from graphgan_amd import engine as eng_mod, graph_gan
    from tests.test_gpu_e2e import make_cfg
    eng_mod.Engine = _ReplicaEngine
    cfg = make_cfg(base, n_epochs=2, n_epochs_dis=2, n_epochs_gen=2, dis_interval=1, gen_interval=2, engine_seed=3, update_ratio=0.3,
                   batch_size_dis=64, batch_size_gen=64)
    g = graph_gan.GraphGAN(cfg)
    assert (g.rank, g.world) == (rank, world) and g.engine.calls[0][0] == "comm_init" and g.engine.calls[0][2:] == (rank, world)
    g.train()
    e = g.engine
    # gather what the other rank ended with
    state = torch.from_numpy(np.concatenate([e.model[0], e.model[1], np.array(e.steps, dtype=np.float64)]))
    both = [torch.zeros_like(state) for _ in range(world)]
    dist.all_gather(both, state)
    own = np.zeros(g.n_node)
    own[g.root_nodes] = 1
    t = torch.from_numpy(own)
    dist.all_reduce(t)
    wrote = len([c for c in e.calls if c[0] == "write_embeddings"])
    np.savez(out % rank, a=both[0].numpy(), b=both[1].numpy(), cover=t.numpy(), wrote=wrote, n_roots=len(g.root_nodes),
             deg_load=float(e.deg[g.root_nodes].sum()))
    dist.destroy_process_group()


def test_two_rank_trainer_keeps_replicas_identical(tmp_path):
    """GraphGAN.train() on two ranks (gloo) against an engine stand-in whose optimizer steps are collectives: the roots are
    partitioned (degree balanced), both ranks issue the same number of steps in every pass although their batch lists
    differ in length (padding with empty steps), the replicas end identical, only rank 0 writes files, and
    update_ratio < 1 selects roots independently of the sharding."""
    import torch.multiprocessing as mp
    from tests.test_gpu_e2e import write_reference_layout
    base = str(tmp_path)
    d, n, graph = write_reference_layout(base)
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_trainer_worker, args=(2, _free_port(), base, out), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert np.array_equal(r0["cover"], np.ones(n))                        # a partition of the roots
    assert r0["n_roots"] + r1["n_roots"] == n and abs(r0["deg_load"] - r1["deg_load"]) <= 80  # balanced by degree (max degree 74)
    for r in (r0, r1):
        assert np.array_equal(r["a"], r["b"])                             # identical replicas, identical step counts
    assert np.array_equal(r0["a"], r1["a"]) and r0["a"][-1] > 10 and r0["a"][-2] > 10
    assert r0["wrote"] == 2 * 3 and r1["wrote"] == 0                       # before training + after each of the 2 epochs, rank 0 only


class _BatchedReplicaEngine(_ReplicaEngine):
    """The same stand-in when the trees cannot be resident (every estimate exceeds the budget): the trainer runs its prepares
    over root batches -- epoch_add WITHOUT a collective (the ranks' batch counts differ), epoch_commit WITH one (the max of the
    ranks' row counts, as gg_epoch_commit does): a rank that commits once more or once less than the other hangs the test."""

    def tree_bytes_estimate(self, n_roots): return 1e15 * max(1, n_roots)

    def epoch_begin(self, reset_d=True, reset_g=True):
        if not hasattr(self, "adds"):
            self.adds, self.commits, self.ep_rows = 0, 0, [0, 0]
        if reset_d: self.ep_rows[1] = 0
        if reset_g: self.ep_rows[0] = 0

    def epoch_add(self, roots, do_d, do_g, n_sample, seed, stream_d, stream_g):
        assert len(roots) > 0
        self.adds += 1
        if do_d: self.ep_rows[1] += int(2 * self.deg[np.asarray(roots)].sum())
        if do_g: self.ep_rows[0] += 37 * len(roots) + 5 * self.rank
        return self.ep_rows[1], self.ep_rows[0]

    def epoch_commit(self, which_d):
        import torch
        which = 1 if which_d else 0
        t = torch.tensor([float(self.ep_rows[which])], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)   # (the engine's exchange_count_max: once per commit, not per batch)
        self.commits += 1
        self.rows[which] = self.ep_rows[which]
        return self.rows[which]


def _batched_trainer_worker(rank, world, port, base, out, update_ratio):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), GG_CTL_BACKEND="gloo")
    import torch
    import torch.distributed as dist
    from graphgan_amd import engine as eng_mod, graph_gan
    from tests.test_gpu_e2e import make_cfg
    eng_mod.Engine = _BatchedReplicaEngine
    cfg = make_cfg(base, n_epochs=2, n_epochs_dis=2, n_epochs_gen=2, dis_interval=1, gen_interval=1, engine_seed=3, update_ratio=update_ratio,
                   batch_size_dis=4096, batch_size_gen=4096, engine_batch_roots=900)
    g = graph_gan.GraphGAN(cfg)
    assert not g._all_resident and g._batch_roots == 900
    g.train()
    e = g.engine
    state = torch.from_numpy(np.concatenate([e.model[0], e.model[1], np.array(e.steps + [e.commits], dtype=np.float64)]))
    both = [torch.zeros_like(state) for _ in range(world)]
    dist.all_gather(both, state)
    np.savez(out % rank, a=both[0].numpy(), b=both[1].numpy(), adds=e.adds, commits=e.commits, n_roots=len(g.root_nodes))
    dist.destroy_process_group()


@pytest.mark.parametrize("update_ratio", [1.0, 0.3])
def test_two_rank_trainer_over_root_batches(tmp_path, update_ratio):
    """The root-batched epoch (trees not resident: gg_epoch_*) on two ranks: the ranks add different numbers of root batches
    (their shares and draws differ) without any collective in between, commit equally often, issue the same number of steps in
    every pass, and end with identical replicas (graph_gan.py::_prepare_root_batches; reference schedule :144-176)."""
    import torch.multiprocessing as mp
    from tests.test_gpu_e2e import write_reference_layout
    base = str(tmp_path)
    write_reference_layout(base)
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_batched_trainer_worker, args=(2, _free_port(), base, out, update_ratio), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    for r in (r0, r1):
        assert np.array_equal(r["a"], r["b"])                 # identical replicas, step counts and commit counts
    assert np.array_equal(r0["a"], r1["a"]) and r0["a"][-2] > 0 and r0["a"][-3] > 0
    assert r0["commits"] == r1["commits"] and r0["commits"] >= 4
    if update_ratio >= 1:
        # every prepare adds ceil(own roots / 900) batches; with update_ratio >= 1 the last D prepare of an outer epoch also samples
        # the first G epoch's pairs (one BFS per root for both), so an outer epoch has 2 D prepares + 1 G-only prepare
        for r in (r0, r1):
            assert r["adds"] == 2 * 3 * -(-int(r["n_roots"]) // 900)
    else:
        assert r0["adds"] > 0 and r1["adds"] > 0


class _ScoreEngine(object):
    """Engine.all_score_reduce on numpy (the stand-in of a rank's GPU: every rank holds the full table)."""

    def __init__(self, E, b):
        self.E, self.b, self.n_node = E, b, len(b)

    def all_score_reduce(self, rows, precision="bf16", logsumexp=True):
        S = self.E[np.asarray(rows)] @ self.E.T + self.b[None, :]          # generator.py:21
        m = S.max(1)
        return dict(max=m.astype(np.float32), argmax=S.argmax(1).astype(np.int32),
                    logsumexp=(m + np.log(np.exp(S - m[:, None]).sum(1))).astype(np.float32) if logsumexp else None, kernel_ms=1.0 + len(rows))


def _allpairs_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), GG_CTL_BACKEND="gloo")
    import torch.distributed as dist
    from graphgan_amd import parallel
    ctl = parallel.Control()
    rs = np.random.RandomState(4)
    E, b = rs.randn(301, 16).astype(np.float32), rs.randn(301).astype(np.float32)
    eng = _ScoreEngine(E, b)
    full = parallel.all_score_reduce_sharded(ctl, eng)                      # every node, 301 rows over 3 ranks: 101 + 101 + 99
    some = parallel.all_score_reduce_sharded(ctl, eng, rows=np.array([7, 300], np.int32), logsumexp=False)   # fewer rows than ranks
    if rank == 0:
        np.savez(out, max=full["max"], argmax=full["argmax"], lse=full["logsumexp"], ms=full["kernel_ms"], smax=some["max"], sarg=some["argmax"])
    ctl.barrier()
    dist.destroy_process_group()


def test_all_pairs_consumer_sharded_over_ranks(tmp_path):
    """BASELINE.json configs[4]: the all-pairs evaluation over the GPUs of a node = rows sharded over the ranks (each holds the
    full table, so a row's max / argmax / log-sum-exp over all columns needs no merge), results concatenated in rank order:
    three gloo ranks give exactly the single-rank result, also when there are fewer rows than ranks."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ap.npz")
    mp.spawn(_allpairs_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    got = np.load(out)
    rs = np.random.RandomState(4)
    E, b = rs.randn(301, 16).astype(np.float32), rs.randn(301).astype(np.float32)
    want = _ScoreEngine(E, b).all_score_reduce(np.arange(301))
    assert np.array_equal(got["max"], want["max"]) and np.array_equal(got["argmax"], want["argmax"]) and np.array_equal(got["lse"], want["logsumexp"])
    assert got["ms"] == 1.0 + 101                                            # the slowest rank's kernel time
    w2 = _ScoreEngine(E, b).all_score_reduce(np.array([7, 300]), logsumexp=False)
    assert np.array_equal(got["smax"], w2["max"]) and np.array_equal(got["sarg"], w2["argmax"])


def _socket_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.pop("GG_CTL_BACKEND", None)
    from graphgan_amd import parallel
    ctl = parallel.Control()  # the default control plane: a TCP star of the ranks, no torch.distributed
    assert ctl.group is not None and ctl.dist is None and (ctl.rank, ctl.world) == (rank, world)
    uid = ctl.broadcast_bytes(bytes(range(128)) if rank == 0 else b"", 128)
    s = ctl.sum(np.arange(5, dtype=np.float64) * (rank + 1))
    m = ctl.max(10.0 * rank)
    cat = ctl.all_gather_concat(np.full(rank + 1, rank, dtype=np.int32))
    ctl.barrier()
    ctl2 = parallel.Control()  # a second control object of the process shares the connection
    assert ctl2.group is ctl.group and ctl2.max(1.0) == 1.0
    np.savez(out % rank, uid=np.frombuffer(uid, np.uint8), s=s, m=m, cat=cat)


def test_socket_control_plane_three_ranks(tmp_path):
    """parallel.Control without torch.distributed (north star: "no PyTorch-ROCm needed"): broadcast of the 128-byte RCCL id,
    sum, max, ragged all-gather and barrier over three processes."""
    import multiprocessing as mp
    out = str(tmp_path / "rank%d.npz")
    port = _free_port()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_socket_worker, args=(r, 3, port, out)) for r in range(3)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    for r in range(3):
        got = np.load(out % r)
        assert np.array_equal(got["uid"], np.arange(128, dtype=np.uint8))
        assert np.array_equal(got["s"], np.arange(5) * 6.0) and got["m"] == 20.0
        assert np.array_equal(got["cat"], np.array([0, 1, 1, 2, 2, 2], dtype=np.int32))
