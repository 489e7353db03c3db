// tree_builder.cpp -- H1: host BFS-tree construction (north-star: "the BFS-tree construction
// stays on the host").  Replaces GraphGAN.construct_trees / construct_trees_with_mp
// (reference src/GraphGAN/graph_gan.py:63-108) and the pickle cache (:31-46).
//
// Output is the tree CSR of DESIGN.md section 2: for root slot r and node v the list
// [father, child_0, child_1, ...] (root: [root, child...]) in the reference's order -- FIFO BFS,
// children in adjacency (file) order, self-loops and already-used nodes skipped -- stored in
// node-id order so that one (off[v], off[v+1]) pair addresses it.
// Roots are independent: they are striped over std::threads.
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "gg_internal.h"

namespace gg {

struct TreeScratch {
    std::vector<uint32_t> stamp;
    std::vector<int32_t> father, queue, cnt, depth;
    uint32_t epoch = 0;
    explicit TreeScratch(int n) : stamp(n, 0), father(n), queue(n), cnt(n), depth(n) {}
};

// BFS from root; returns number of reached nodes (queue holds them in pop order).
static int bfs_one(int n, const int64_t *rowptr, const int32_t *col, int root, TreeScratch &s, int &max_depth) {
    if (++s.epoch == 0) {  // stamp wrap-around
        std::fill(s.stamp.begin(), s.stamp.end(), 0u);
        s.epoch = 1;
    }
    const uint32_t ep = s.epoch;
    int qh = 0, qt = 0;
    s.queue[qt++] = root;
    s.stamp[root] = ep;
    s.father[root] = root;
    s.depth[root] = 0;
    s.cnt[root] = 1;
    while (qh < qt) {
        const int cur = s.queue[qh++];
        const int dc = s.depth[cur] + 1;
        for (int64_t e = rowptr[cur]; e < rowptr[cur + 1]; ++e) {
            const int sub = col[e];
            if (s.stamp[sub] != ep) {
                s.stamp[sub] = ep;
                s.father[sub] = cur;
                s.depth[sub] = dc;
                s.cnt[sub] = 1;
                s.cnt[cur] += 1;
                s.queue[qt++] = sub;
                if (dc > max_depth) max_depth = dc;
            }
        }
    }
    (void)n;
    return qt;
}

// Component size of every node (one sweep), so that each root's entry count 2*|comp|-1 and
// therefore nbr_base[] are known before any per-root BFS runs.
static void component_sizes(int n, const int64_t *rowptr, const int32_t *col, std::vector<int32_t> &comp_size) {
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

int64_t host_tree_sizes(int32_t n, const int64_t *rowptr, const int32_t *col, const int32_t *roots, int32_t n_roots,
                        int64_t *nbr_base) {
    std::vector<int32_t> cs;
    component_sizes(n, rowptr, col, cs);
    int64_t run = 0;
    for (int r = 0; r < n_roots; ++r) {
        nbr_base[r] = run;
        run += 2 * (int64_t)cs[roots[r]] - 1;
    }
    nbr_base[n_roots] = run;
    return run;
}

// Fill off / nbr for roots [r0, r1) given nbr_base (absolute).  off rows and nbr are indexed
// relative to (off_row0, nbr_origin) so that callers can build batch-local buffers.
void host_fill_trees(int32_t n, const int64_t *rowptr, const int32_t *col, const int32_t *roots, int32_t r0, int32_t r1,
                     const int64_t *nbr_base, int32_t *off, int32_t off_row0, int32_t *nbr, int64_t nbr_origin,
                     int32_t n_threads, int32_t *max_depth_out, int32_t *max_list_out) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > r1 - r0) n_threads = r1 - r0 > 0 ? r1 - r0 : 1;
    std::atomic<int> next(r0);
    std::vector<int> md(n_threads, 0), ml(n_threads, 0);
    auto work = [&](int tid) {
        TreeScratch s(n);
        std::vector<int32_t> fill(n);
        int max_depth = 0, max_list = 0;
        for (;;) {
            const int r = next.fetch_add(1);
            if (r >= r1) break;
            const int root = roots[r];
            const int reached = bfs_one(n, rowptr, col, root, s, max_depth);
            int32_t *o = off + (int64_t)(r - off_row0) * (n + 1);
            int32_t *nb = nbr + (nbr_base[r] - nbr_origin);
            const uint32_t ep = s.epoch;
            int32_t run = 0;
            for (int v = 0; v < n; ++v) {
                o[v] = run;
                if (s.stamp[v] == ep) {
                    const int c = s.cnt[v];
                    nb[run] = s.father[v];
                    fill[v] = 1;
                    run += c;
                    if (c > max_list) max_list = c;
                }
            }
            o[n] = run;
            for (int i = 1; i < reached; ++i) {
                const int v = s.queue[i], f = s.father[v];
                nb[o[f] + fill[f]] = v;
                fill[f] += 1;
            }
        }
        md[tid] = max_depth;
        ml[tid] = max_list;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    int a = 0, b = 0;
    for (int t = 0; t < n_threads; ++t) { if (md[t] > a) a = md[t]; if (ml[t] > b) b = ml[t]; }
    if (max_depth_out && a > *max_depth_out) *max_depth_out = a;
    if (max_list_out && b > *max_list_out) *max_list_out = b;
}

}  // namespace gg

extern "C" int64_t gg_host_build_trees(int32_t n_node, const int64_t *rowptr, const int32_t *col, const int32_t *roots,
                                       int32_t n_roots, int32_t *off, int32_t *nbr, int64_t *nbr_base, int64_t cap,
                                       int32_t n_threads, int32_t *max_depth_out) {
    if (n_node <= 0 || !rowptr || (!col && rowptr[n_node] > 0) || n_roots < 0 || (n_roots && !roots) || !nbr_base)
        return gg::fail(nullptr, GG_EINVAL, "gg_host_build_trees: bad argument");
    for (int r = 0; r < n_roots; ++r)
        if (roots[r] < 0 || roots[r] >= n_node) return gg::fail(nullptr, GG_EINVAL, "gg_host_build_trees: root %d out of range", roots[r]);
    const int64_t total = gg::host_tree_sizes(n_node, rowptr, col, roots, n_roots, nbr_base);
    if (!nbr) return total;
    if (!off) return gg::fail(nullptr, GG_EINVAL, "gg_host_build_trees: off is NULL");
    if (cap < total) return gg::fail(nullptr, GG_ECAPACITY, "gg_host_build_trees: cap %lld < %lld entries", (long long)cap, (long long)total);
    int32_t md = 0, ml = 0;
    gg::host_fill_trees(n_node, rowptr, col, roots, 0, n_roots, nbr_base, off, 0, nbr, 0, n_threads, &md, &ml);
    if (max_depth_out) *max_depth_out = md;
    return total;
}
