"""Host-side multi-GPU logic (one process per GPU; no reference counterpart -- the reference
is a single ``tf.Session``, graph_gan.py:57-61).

Data path: walks shard by root with no collective (the walk RNG is keyed by root id, so any
partition gives the same walks); the gradient exchange is an RCCL all-reduce inside
``libgraphgan_hip.so`` (``gg_comm_init``).  This module only (a) partitions roots, (b) carries
the 128-byte RCCL unique id between processes over ``torch.distributed`` (gloo, CPU) and
(c) provides the barrier / max-over-ranks used for timing.
"""
from __future__ import annotations

import os

import numpy as np


def env_rank():
    """(rank, world, local_rank) as exported by ``torch.distributed.run``."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_roots(roots, rank, world, weights=None):
    """Roots owned by ``rank``.  Without weights: round-robin (root i -> rank i mod world, the
    SURVEY section 8e rule).  With weights (e.g. degree ~ D-mode walks per root): greedy
    longest-processing-time balancing, deterministic on every rank.  Every root lands on exactly
    one rank; order within a rank follows the input order."""
    roots = np.asarray(roots)
    if world <= 1:
        return roots.copy()
    if weights is None:
        return roots[rank::world].copy()
    weights = np.asarray(weights, dtype=np.float64)
    order = np.argsort(-weights, kind="stable")
    load = np.zeros(world)
    owner = np.empty(len(roots), dtype=np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += weights[i] + 1e-9
    return roots[owner == rank].copy()


class Control:
    """Control plane over torch.distributed (gloo): never touches device memory."""

    def __init__(self, rank=None, world=None, backend="gloo"):
        r, w, lr = env_rank()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.local_rank = lr
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    def broadcast_bytes(self, payload, n, src=0):
        if self.dist is None:
            return payload
        import torch
        buf = torch.zeros(n, dtype=torch.uint8)
        if self.rank == src:
            buf = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone()
        self.dist.broadcast(buf, src)
        return bytes(buf.numpy().tobytes())

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def sum(self, values):
        values = np.asarray(values, dtype=np.float64)
        if self.dist is None:
            return values
        import torch
        t = torch.from_numpy(values.copy())
        self.dist.all_reduce(t)
        return t.numpy()

    def max(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def connect_engine(self, engine):
        """RCCL communicator for ``engine``: rank 0 creates the unique id, everybody joins."""
        if self.world <= 1:
            return
        uid = engine.comm_unique_id() if self.rank == 0 else b""
        uid = self.broadcast_bytes(uid, 128)
        engine.comm_init(uid, self.rank, self.world)
