// tree_builder.cpp -- H1: host BFS-tree construction (north-star: "the BFS-tree construction
// stays on the host").  Replaces GraphGAN.construct_trees / construct_trees_with_mp
// (reference src/GraphGAN/graph_gan.py:63-108) and the pickle cache (:31-46).
//
// Internal form: the BFS-ORDER tree.  A FIFO BFS appends the children of the node it pops
// next to each other, in adjacency order, so the queue itself -- order[rank] = node id -- together with
// cstart[rank] = rank of the first child describes the whole tree: the reference's tree[v] is
// [father] ++ order[cstart[i] .. cstart[i+1]) for v = order[i].  Self-loops and already-used nodes are skipped
// exactly like the reference does (:101-103).
// The C ABI hands trees over in the reference's own shape (per node the list [father, child_0, ...], lists in
// node-id order behind an offsets row): order_to_lists / lists_to_order convert.
// Roots are independent: they are striped over std::threads.
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "gg_internal.h"

namespace gg {

int32_t host_bfs_order(const int64_t *rowptr, const int32_t *col, int32_t root, int32_t *order, int32_t *cstart,
                       std::vector<uint32_t> &stamp, uint32_t &epoch, int32_t *depth_out, int32_t *max_children_out) {
    if (++epoch == 0) {  // stamp wrap-around
        std::fill(stamp.begin(), stamp.end(), 0u);
        epoch = 1;
    }
    const uint32_t ep = epoch;
    int32_t head = 0, tail = 1, level_end = 1, depth = 0, max_children = 0;
    order[0] = root;
    stamp[root] = ep;
    while (head < tail) {
        if (head == level_end) {  // the first node of the next level is about to be popped
            level_end = tail;
            ++depth;
        }
        const int32_t cur = order[head];
        cstart[head] = tail;
        for (int64_t e = rowptr[cur]; e < rowptr[cur + 1]; ++e) {
            const int32_t sub = col[e];
            if (stamp[sub] != ep) {
                stamp[sub] = ep;
                order[tail++] = sub;
            }
        }
        if (tail - cstart[head] > max_children) max_children = tail - cstart[head];
        ++head;
    }
    cstart[tail] = tail;
    if (depth_out) *depth_out = depth;
    if (max_children_out) *max_children_out = max_children;
    return tail;
}

// Component size of every node (one sweep), so that each root's node count C_r and therefore the tree bases
// are known before any per-root BFS runs.
void component_sizes(int n, const int64_t *rowptr, const int32_t *col, std::vector<int32_t> &comp_size) {
    std::vector<int32_t> label(n, -1), stack;
    std::vector<int32_t> sizes;
    for (int v = 0; v < n; ++v) {
        if (label[v] >= 0) continue;
        const int id = (int)sizes.size();
        int cnt = 0;
        stack.push_back(v);
        label[v] = id;
        while (!stack.empty()) {
            const int x = stack.back();
            stack.pop_back();
            ++cnt;
            for (int64_t e = rowptr[x]; e < rowptr[x + 1]; ++e) {
                const int y = col[e];
                if (label[y] < 0) { label[y] = id; stack.push_back(y); }
            }
        }
        sizes.push_back(cnt);
    }
    comp_size.resize(n);
    for (int v = 0; v < n; ++v) comp_size[v] = sizes[label[v]];
}

void order_to_lists(int32_t n, int32_t C, const int32_t *order, const int32_t *cstart, const uint32_t *q3, int32_t *off, int32_t *nbr) {
    // list length of v = order[i]: 1 + #children; unreached nodes have empty lists
    std::vector<int32_t> rank(n, -1);
    for (int32_t i = 0; i < C; ++i) rank[order[i]] = i;
    int32_t run = 0;
    for (int32_t v = 0; v < n; ++v) {
        off[v] = run;
        if (rank[v] >= 0) run += 1 + (cstart[rank[v] + 1] - cstart[rank[v]]);
    }
    off[n] = run;
    const int32_t root = order[0];
    nbr[off[root]] = root;  // tree[root] = [root, child...] (graph_gan.py:96)
    for (int32_t i = 0; i < C; ++i) {
        const int32_t v = order[i];
        int32_t *lst = nbr + off[v];
        for (int32_t j = cstart[i]; j < cstart[i + 1]; ++j) {
            lst[1 + (j - cstart[i])] = order[j];
            nbr[off[order[j]]] = v;  // the child's father entry
        }
    }
    if (q3)  // D-mode removed the father entry of these depth-1 children (graph_gan.py:258-259): shown as -1
        for (int32_t j = cstart[0]; j < cstart[1]; ++j)
            if (q3[(j - 1) >> 5] >> ((j - 1) & 31) & 1u) nbr[off[order[j]]] = -1;
}

int32_t lists_to_order(int32_t n, int32_t root, const int32_t *off, const int32_t *nbr, int32_t *order, int32_t *cstart,
                       uint32_t *q3, int32_t q3_words, int32_t *depth_out, int32_t *max_children_out) {
    const int32_t total = off[n];
    if (total <= 0 || off[root + 1] - off[root] < 1) return -1;
    const int32_t C = (total + 1) / 2;  // 2C - 1 entries
    if (2 * C - 1 != total) return -1;
    int32_t head = 0, tail = 1, level_end = 1, depth = 0, max_children = 0;
    order[0] = root;
    while (head < tail) {
        if (head == level_end) { level_end = tail; ++depth; }
        const int32_t v = order[head];
        const int32_t b = off[v], e = off[v + 1];
        if (e - b < 1) return -1;
        cstart[head] = tail;
        for (int32_t k = b + 1; k < e; ++k) {
            const int32_t c = nbr[k];
            if (c < 0 || c >= n || tail >= C) return -1;
            order[tail++] = c;
        }
        if (e - b - 1 > max_children) max_children = e - b - 1;
        if (depth == 1 && nbr[b] < 0) {  // a removed father entry (only depth-1 children ever lose theirs)
            const int32_t bit = head - 1;
            if (!q3 || (bit >> 5) >= q3_words) return -1;
            q3[bit >> 5] |= 1u << (bit & 31);
        }
        ++head;
    }
    if (tail != C) return -1;
    cstart[C] = C;
    if (depth_out) *depth_out = depth;
    if (max_children_out) *max_children_out = max_children;
    return C;
}

}  // namespace gg

// gg_host_build_trees: reference-shaped tree lists of the given roots (see include/graphgan_hip.h).
extern "C" int64_t gg_host_build_trees(int32_t n_node, const int64_t *rowptr, const int32_t *col, const int32_t *roots,
                                       int32_t n_roots, int32_t *off, int32_t *nbr, int64_t *nbr_base, int64_t cap,
                                       int32_t n_threads, int32_t *max_depth_out) {
    if (n_node <= 0 || !rowptr || (!col && rowptr[n_node] > 0) || n_roots < 0 || (n_roots && !roots) || !nbr_base)
        return gg::fail(nullptr, GG_EINVAL, "gg_host_build_trees: bad argument");
    for (int r = 0; r < n_roots; ++r)
        if (roots[r] < 0 || roots[r] >= n_node) return gg::fail(nullptr, GG_EINVAL, "gg_host_build_trees: root %d out of range", roots[r]);
    std::vector<int32_t> cs;
    gg::component_sizes(n_node, rowptr, col, cs);
    int64_t total = 0;
    for (int r = 0; r < n_roots; ++r) {
        nbr_base[r] = total;
        total += 2 * (int64_t)cs[roots[r]] - 1;
    }
    nbr_base[n_roots] = total;
    if (!nbr) return total;
    if (!off) return gg::fail(nullptr, GG_EINVAL, "gg_host_build_trees: off is NULL");
    if (cap < total) return gg::fail(nullptr, GG_ECAPACITY, "gg_host_build_trees: cap %lld < %lld entries", (long long)cap, (long long)total);
    if (n_threads < 1) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    if (n_threads > n_roots) n_threads = n_roots > 0 ? n_roots : 1;
    std::atomic<int> next(0);
    std::vector<int> md(n_threads, 0);
    auto work = [&](int tid) {
        std::vector<uint32_t> stamp(n_node, 0u);
        uint32_t epoch = 0;
        std::vector<int32_t> order(n_node), cstart(n_node + 1);
        for (;;) {
            const int r = next.fetch_add(1);
            if (r >= n_roots) break;
            int32_t depth = 0;
            const int32_t C = gg::host_bfs_order(rowptr, col, roots[r], order.data(), cstart.data(), stamp, epoch, &depth, nullptr);
            gg::order_to_lists(n_node, C, order.data(), cstart.data(), nullptr, off + (int64_t)r * (n_node + 1), nbr + nbr_base[r]);
            if (depth > md[tid]) md[tid] = depth;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    int a = 0;
    for (int t = 0; t < n_threads; ++t) a = std::max(a, md[t]);
    if (max_depth_out) *max_depth_out = a;
    return total;
}
