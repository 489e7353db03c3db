"""Property tests (hypothesis) of the oracle's building blocks -- the arithmetic spec pieces the HIP
kernels are compared against must themselves be right for every input, not only for the fixtures."""
import ctypes

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import graphgan_oracle as orc


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 2**64 - 1), st.integers(0, 2**32 - 1), st.integers(0, 2**32 - 1), st.integers(0, 2**32 - 1), st.integers(0, 2**32 - 1))
def test_uniform53_python_equals_c(seed, stream, root, walk, hop):
    lib = orc.c_oracle()
    m = lib.orc_uniform53(ctypes.c_uint64(seed), stream, root, walk, hop)
    assert m == orc.uniform53(seed, stream, root, walk, hop) and 0 <= m < 2**53


@settings(max_examples=100, deadline=None)
@given(st.lists(st.floats(-30, 30, width=32), min_size=1, max_size=40), st.integers(0, 2**53 - 1))
def test_integer_choice_is_the_inverse_cdf(scores, m):
    """S3/S5: with exact integer weights the picked index is the first j whose inclusive prefix exceeds
    floor(m W / 2^53) -- equal to searchsorted(cdf, u, 'right') computed in exact rational arithmetic."""
    lib = orc.c_oracle()
    sc = np.array(scores, dtype=np.float32)
    mx = sc.max()
    w = [int(lib.orc_weight(ctypes.c_float(lib.orc_expf(ctypes.c_float(float(np.float32(x - mx))))))) for x in sc]
    assert max(w) == 2**40 and all(0 <= x <= 2**40 for x in w)
    W = sum(w)
    t = (m * W) >> 53
    prefix = np.cumsum(np.array(w, dtype=object))
    want = next(j for j, c in enumerate(prefix) if c > t)
    # exact rational statement: first j with C_j / W > m / 2^53
    assert all(prefix[j] * 2**53 <= m * W for j in range(want)) and prefix[want] * 2**53 > m * W
    # the C walk oracle on a star graph with these scores picks the same leaf
    k = len(w)
    n = k + 1
    E = np.zeros((n, 4), dtype=np.float32)
    E[0, 0] = 1.0
    E[1:, 0] = 0.0
    bias = np.concatenate([[0.0], sc]).astype(np.float32)  # score(0 -> j) = 0 + bias[j]
    rowptr = np.concatenate([[0, k], k + 1 + np.arange(k)]).astype(np.int64)
    col = np.concatenate([1 + np.arange(k), np.zeros(k)]).astype(np.int32)
    off, nbr, base, dmax = orc.c_build_trees(n, rowptr, col, np.array([0], dtype=np.int32))
    seed = 12345
    res = orc.c_walk_sample(E, bias, off, nbr, base, np.array([0], np.int32), np.array([0], np.int32), np.array([1], np.int32),
                            False, seed, 0, 8)
    m0 = orc.uniform53(seed, 0, 0, 0, 0)
    t0 = (m0 * W) >> 53
    want0 = next(j for j, c in enumerate(prefix) if c > t0)
    assert res["paths"][0, 1] == 1 + want0


@settings(max_examples=60, deadline=None)
@given(st.integers(2, 40), st.integers(0, 10**6))
def test_tree_builder_invariants(n, seed):
    rs = np.random.RandomState(seed)
    m = rs.randint(n - 1, 3 * n)
    a, b = rs.randint(0, n, m), rs.randint(0, n, m)
    src = np.concatenate([a, b])
    dst = np.concatenate([b, a]).astype(np.int32)
    order = np.argsort(src, kind="stable")
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=rowptr[1:])
    col = np.ascontiguousarray(dst[order])
    roots = np.arange(n, dtype=np.int32)
    off, nbr, base, dmax = orc.c_build_trees(n, rowptr, col, roots)
    graph = {v: col[rowptr[v]:rowptr[v + 1]].tolist() for v in range(n)}
    ref = orc.trees_to_csr(orc.construct_trees(graph, list(range(n))), list(range(n)), n)
    assert np.array_equal(off, ref[0]) and np.array_equal(nbr, ref[1]) and np.array_equal(base, ref[2])
    for r in range(n):
        lst = nbr[base[r]:base[r + 1]]
        reached = {v for v in range(n) if off[r, v + 1] > off[r, v]}
        assert len(lst) == 2 * len(reached) - 1 and r in reached
        children = [x for v in reached for x in lst[off[r, v] + 1: off[r, v + 1]]]
        assert sorted(children) == sorted(reached - {r})  # every reached node except the root is a child exactly once
        for v in reached:
            f = lst[off[r, v]]
            assert f == r if v == r else (f in reached and v in lst[off[r, f] + 1: off[r, f + 1]].tolist())


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(0, 1000), min_size=1, max_size=30), st.integers(1, 4))
def test_window_pairs_count_and_content(path, window):
    pairs = orc.pairs_from_path(path, window)
    L = len(path) - 1
    want = sum(min(i + window + 1, L) - max(i - window, 0) - 1 for i in range(L)) if L > 0 else 0
    assert len(pairs) == want
    lib = orc.c_oracle()
    p = np.array(path, dtype=np.int32)
    a = np.zeros(2 * window * len(p) + 4, dtype=np.int32)
    b = np.zeros_like(a)
    n = lib.orc_pairs_from_path(p.ctypes.data, len(p), window, a.ctypes.data, b.ctypes.data)
    assert np.stack([a[:n], b[:n]], 1).tolist() == pairs


def _search16(pf, thr):
    """The 16-ary search of level_advance_kernel (graphgan_amd/csrc/walk_sample.hip), restated: first j with
    pf[j] > thr for a non-decreasing pf whose last element exceeds thr; 15 pivots per round."""
    lo, n = 0, len(pf)
    while n > 1:
        step = (n + 15) >> 4
        seg = 0
        for i in range(1, 16):
            idx = i * step - 1
            if idx < n - 1 and pf[lo + idx] <= thr:
                seg += 1
        lo += seg * step
        n = min(step, n - seg * step)
    return lo


@settings(max_examples=300, deadline=None)
@given(st.lists(st.integers(min_value=0, max_value=5), min_size=1, max_size=700), st.data())
def test_16ary_prefix_search_equals_searchsorted(weights, data):
    """Spec S5 picks the first j whose inclusive prefix sum exceeds the threshold = searchsorted(side='right');
    the GPU finds it with a 16-ary search whose rounds are independent loads.  Zero weights (ties in the prefix
    sums) and every length class (1 round <= 16, 2 rounds <= 256, 3 rounds) are covered."""
    pf = np.cumsum(np.asarray(weights, dtype=np.uint64) + (np.arange(len(weights)) == len(weights) - 1).astype(np.uint64))
    thr = data.draw(st.integers(min_value=0, max_value=int(pf[-1]) - 1))
    assert _search16([int(x) for x in pf], thr) == int(np.searchsorted(pf, thr, side="right"))
