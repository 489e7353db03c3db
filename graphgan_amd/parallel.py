"""Host-side multi-GPU logic (one process per GPU; no reference counterpart -- the reference
is a single ``tf.Session``, graph_gan.py:57-61).

Data path: walks shard by root with no collective (the walk RNG is keyed by root id, so any
partition gives the same walks); the gradient exchange is an RCCL all-reduce inside
``libgraphgan_hip.so`` (``gg_comm_init``).  This module only (a) partitions roots, (b) carries
the 128-byte RCCL unique id between the processes and (c) provides the barrier / max-over-ranks used for
timing -- over a TCP star of the node's ranks (standard library; ``torch.distributed`` / gloo stays
available as ``Control(backend="gloo")`` or GG_CTL_BACKEND=gloo for launchers that bring it).
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


class _SocketGroup:
    """The ranks of one node as a star over TCP: rank 0 listens on MASTER_ADDR : (MASTER_PORT + 29, or GG_CTL_PORT), the others
    connect; one primitive -- every rank contributes a picklable object, every rank gets the list of all of them in rank order --
    carries the few hundred bytes the control plane ever moves (the 128-byte RCCL id, three scalars per timing, the 12 bytes per
    row of the sharded all-pairs evaluation).  No torch: the engine's only tensors are its embedding tables."""

    _live = {}

    @classmethod
    def get(cls, rank, world):
        import socket
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("GG_CTL_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 29))
        key = (addr, port, rank, world)
        if key not in cls._live:
            cls._live[key] = cls(socket, addr, port, rank, world)
        return cls._live[key]

    def __init__(self, socket, addr, port, rank, world):
        import time
        self.rank, self.world, self.peers, self.sock = rank, world, {}, None
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(300.0)
            while len(self.peers) < world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self.peers[self._recv(c)] = c
            srv.close()
        else:
            deadline = time.time() + 300.0
            while True:
                try:
                    self.sock = socket.create_connection((addr, port), timeout=300.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._send(self.sock, rank)

    @staticmethod
    def _send(c, obj):
        import pickle
        import struct
        b = pickle.dumps(obj, protocol=4)
        c.sendall(struct.pack("<Q", len(b)) + b)

    @staticmethod
    def _recv(c):
        import pickle
        import struct

        def take(n):
            out = bytearray()
            while len(out) < n:
                chunk = c.recv(min(1 << 20, n - len(out)))
                if not chunk:
                    raise ConnectionError("control plane: a rank closed its connection")
                out += chunk
            return bytes(out)
        return pickle.loads(take(struct.unpack("<Q", take(8))[0]))

    def all_gather(self, obj):
        if self.rank == 0:
            items = [obj] + [self._recv(self.peers[r]) for r in range(1, self.world)]
            for r in range(1, self.world):
                self._send(self.peers[r], items)
            return items
        self._send(self.sock, obj)
        return self._recv(self.sock)


class Control:
    """Control plane of a multi-GPU run: never touches device memory.  backend "socket" (default): a TCP star of the node's
    ranks, standard library only; "gloo": torch.distributed, for launchers that already initialised it."""

    def __init__(self, rank=None, world=None, backend=None):
        r, w, lr = env_rank()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.local_rank = lr
        self.dist = None
        self.group = None
        backend = backend or os.environ.get("GG_CTL_BACKEND", "socket")
        if self.world > 1:
            if backend == "gloo":
                import torch.distributed as dist
                if not dist.is_initialized():
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                    dist.init_process_group(backend, rank=self.rank, world_size=self.world)
                self.dist = dist
            else:
                self.group = _SocketGroup.get(self.rank, self.world)

    def broadcast_bytes(self, payload, n, src=0):
        if self.group is not None:
            return bytes(self.group.all_gather(bytes(payload) if self.rank == src else b"")[src])
        if self.dist is None:
            return payload
        import torch
        buf = torch.zeros(n, dtype=torch.uint8)
        if self.rank == src:
            buf = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone()
        self.dist.broadcast(buf, src)
        return bytes(buf.numpy().tobytes())

    def barrier(self):
        if self.group is not None:
            self.group.all_gather(None)
        elif self.dist is not None:
            self.dist.barrier()

    def sum(self, values):
        values = np.asarray(values, dtype=np.float64)
        if self.group is not None:
            return np.sum(np.stack(self.group.all_gather(values)), axis=0)
        if self.dist is None:
            return values
        import torch
        t = torch.from_numpy(values.copy())
        self.dist.all_reduce(t)
        return t.numpy()

    def max(self, value):
        if self.group is not None:
            return float(max(self.group.all_gather(float(value))))
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather_concat(self, arr):
        """Concatenation, in rank order, of every rank's 1-D array (lengths may differ)."""
        arr = np.ascontiguousarray(arr)
        if self.group is not None:
            return np.concatenate(self.group.all_gather(arr))
        if self.dist is None:
            return arr.copy()
        import torch
        n = torch.tensor([len(arr)], dtype=torch.int64)
        ns = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        self.dist.all_gather(ns, n)
        m = max(int(x.item()) for x in ns)
        pad = np.zeros(max(m, 1), dtype=arr.dtype)
        pad[: len(arr)] = arr
        t = torch.from_numpy(pad)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.concatenate([o.numpy()[: int(k.item())] for o, k in zip(out, ns)])

    def connect_engine(self, engine):
        """RCCL communicator for ``engine``: rank 0 creates the unique id, everybody joins."""
        if self.world <= 1:
            return
        uid = engine.comm_unique_id() if self.rank == 0 else b""
        uid = self.broadcast_bytes(uid, 128)
        engine.comm_init(uid, self.rank, self.world)


def all_score_reduce_sharded(ctl, engine, rows=None, precision="bf16", logsumexp=True):
    """The all-pairs evaluation of BASELINE.json configs[4] ("8 x MI355X, all-pairs eval as MFMA bf16 GEMM") over the ranks of
    one node: ``generator.all_score`` (reference generator.py:21) streamed through the fused max / argmax / log-sum-exp
    consumer (``Engine.all_score_reduce``), ROWS sharded.  Every rank holds a full replica of the table, so the reduction of a
    row over all N columns is formed on one rank -- no partial (max, argmax, lse) triples to merge -- and the ranks' results
    are concatenated over the control plane (12 bytes per row).  rows = None: every node.  Returns the same dict as
    ``Engine.all_score_reduce`` for ALL requested rows, in their order, on every rank; ``kernel_ms`` = the slowest rank's."""
    n_rows = engine.n_node if rows is None else len(rows)
    all_rows = np.arange(engine.n_node, dtype=np.int32) if rows is None else np.ascontiguousarray(rows, dtype=np.int32)
    per = (n_rows + ctl.world - 1) // max(ctl.world, 1)
    mine = all_rows[ctl.rank * per: min((ctl.rank + 1) * per, n_rows)]
    if len(mine):
        res = engine.all_score_reduce(mine, precision=precision, logsumexp=logsumexp)
    else:
        res = dict(max=np.zeros(0, np.float32), argmax=np.zeros(0, np.int32), logsumexp=np.zeros(0, np.float32) if logsumexp else None, kernel_ms=0.0)
    out = dict(max=ctl.all_gather_concat(res["max"]), argmax=ctl.all_gather_concat(res["argmax"]),
               logsumexp=ctl.all_gather_concat(res["logsumexp"]) if logsumexp else None, kernel_ms=ctl.max(res["kernel_ms"]))
    assert len(out["max"]) == n_rows
    return out
