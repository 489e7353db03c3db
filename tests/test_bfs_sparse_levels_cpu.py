"""The claim behind the sparse levels of the GPU BFS (graphgan_amd/csrc/bfs_gpu.hip), checked on the CPU in
plain Python against the reference's FIFO BFS (graph_gan.py:84-108):

  * only a FATHER -- the first queue node adjacent to an unseen node -- appends anything while a level is popped, so popping ANY
    superset S of the level's fathers, in queue order, appends the same nodes, in the same order, at the same edges;
  * the set the kernel builds -- the level's ranks cut into buckets, every unseen node marks the neighbours it has in the FIRST
    bucket it hits -- is such a superset;
  * the child ranges of the nodes outside S are empty: cstart over the level is the running maximum of the values S's nodes get.

The GPU parity tests (tests/test_gpu_walk.py, sparse levels forced on every level) check the kernel; this checks the idea."""
import numpy as np
import pytest


def random_graph(rs, n, m, with_loops=True):
    """adjacency lists in 'file order' with multi-edges (and self-loops): what read_edges / edges_to_csr produce"""
    adj = [[] for _ in range(n)]
    for _ in range(m):
        a, b = int(rs.randint(n)), int(rs.randint(n))
        if a == b and not with_loops:
            continue
        adj[a].append(b)
        adj[b].append(a)
    return adj


def fifo_bfs(adj, root):
    """reference order (graph_gan.py:84-108): pop order, father rank and adjacency index of the appending edge, cstart"""
    order, father, edge_idx = [root], [-1], [-1]
    seen = {root}
    cstart = [1]  # cstart[i + 1] = end of the children of rank i; cstart[0] = 1
    head = 0
    while head < len(order):
        v = order[head]
        for k, w in enumerate(adj[v]):
            if w not in seen:
                seen.add(w)
                order.append(w)
                father.append(head)
                edge_idx.append(k)
        cstart.append(len(order))
        head += 1
    return order, father, edge_idx, cstart


def levels_of(order, father):
    depth = [0] * len(order)
    for i in range(1, len(order)):
        depth[i] = depth[father[i]] + 1
    bounds = [0]
    for i in range(1, len(order)):
        if depth[i] != depth[i - 1]:
            bounds.append(i)
    bounds.append(len(order))
    return bounds  # level l = ranks [bounds[l], bounds[l + 1])


def pop_subset(adj, order_so_far, S_ranks):
    """pop only the ranks in S (ascending) of the current level: appended nodes, their father ranks / edge indices, and the child
    range END of every popped rank"""
    seen = set(order_so_far)
    new, fa, ei, end = [], [], [], {}
    for r in sorted(S_ranks):
        v = order_so_far[r]
        for k, w in enumerate(adj[v]):
            if w not in seen:
                seen.add(w)
                new.append(w)
                fa.append(r)
                ei.append(k)
        end[r] = len(order_so_far) + len(new)
    return new, fa, ei, end


def marked_superset(adj, order, lo, hi, n_buckets):
    """the kernel's S: ranks [lo, hi) in n_buckets buckets; every unseen node marks its neighbours in the first bucket it hits"""
    rank_of = {v: i for i, v in enumerate(order)}  # (nodes on the queue so far; the level is its tail)
    bs = max(1, -(-(hi - lo) // n_buckets))
    marked = set()
    n = len(adj)
    for x in range(n):
        if x in rank_of:
            continue
        in_level = [rank_of[u] for u in adj[x] if u in rank_of and lo <= rank_of[u] < hi]
        if not in_level:
            continue
        first = min((r - lo) // bs for r in in_level)
        marked.update(r for r in in_level if (r - lo) // bs == first)
    return marked


@pytest.mark.parametrize("seed", range(12))
def test_popping_a_superset_of_the_fathers_gives_the_reference_tree(seed):
    rs = np.random.RandomState(seed)
    n = int(rs.randint(30, 220))
    adj = random_graph(rs, n, int(n * rs.uniform(0.8, 4.0)))
    root = int(rs.randint(n))
    order, father, edge_idx, cstart = fifo_bfs(adj, root)
    bounds = levels_of(order, father)
    for l in range(len(bounds) - 1):
        lo, hi = bounds[l], bounds[l + 1]
        nxt_lo = hi
        nxt_hi = bounds[l + 2] if l + 2 < len(bounds) else hi
        fathers = set(father[nxt_lo:nxt_hi])
        assert all(lo <= f < hi for f in fathers)
        for n_buckets in (1, 3, 16):
            S = marked_superset(adj, order[:hi], lo, hi, n_buckets)
            assert fathers <= S <= set(range(lo, hi))                      # the kernel's set contains every father
            if rs.rand() < 0.5 and hi - lo > len(S):                        # ... and any larger set does as well
                S = S | set(int(r) for r in rs.randint(lo, hi, size=3))
            new, fa, ei, end = pop_subset(adj, order[:hi], S)
            assert new == order[nxt_lo:nxt_hi] and fa == father[nxt_lo:nxt_hi] and ei == edge_idx[nxt_lo:nxt_hi]
            # child ranges: the popped nodes' ends, a running maximum over the level for everyone else (start value: the level's end)
            run, got = hi, []
            for r in range(lo, hi):
                run = max(run, end.get(r, 0))
                got.append(run)
            assert got == cstart[lo + 1:hi + 1]
