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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
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
