// synth.cpp -- synthetic power-law graphs for the benchmark configurations
// (BASELINE.json configs[2..4]; recipe SURVEY.md section 8d): Barabasi-Albert preferential
// attachment with m edges per new node (-> E ~ m*N undirected edges, one component,
// degree exponent ~3), node ids randomly permuted so that id order != age order, edge order
// shuffled (the reference's adjacency order is file order, src/utils.py:12-47).
// Host only; deterministic for given seeds (splitmix64).
#include <algorithm>
#include <vector>

#include "gg_internal.h"

namespace {
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    // unbiased enough for synthetic data: 64-bit multiply-shift
    uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
};
}  // namespace

extern "C" int64_t gg_synth_powerlaw(int32_t n_node, int32_t m, uint64_t seed_graph, uint64_t seed_perm,
                                     int32_t *edges_out, int64_t n_edges_cap) {
    if (n_node <= 0 || m <= 0 || n_node <= m) return gg::fail(nullptr, GG_EINVAL, "gg_synth_powerlaw: need n_node > m > 0");
    const int64_t n_edges = (int64_t)(n_node - m) * m;
    if (!edges_out) return n_edges;
    if (n_edges_cap < n_edges) return gg::fail(nullptr, GG_ECAPACITY, "gg_synth_powerlaw: cap %lld < %lld", (long long)n_edges_cap, (long long)n_edges);
    SplitMix64 rg(seed_graph), rp(seed_perm);
    std::vector<int32_t> rep;  // every edge endpoint once: sampling from it = degree-proportional
    rep.reserve((size_t)2 * n_edges);
    std::vector<int32_t> tgt(m);
    int64_t ne = 0;
    for (int t = m; t < n_node; ++t) {
        if (t == m) {
            for (int i = 0; i < m; ++i) tgt[i] = i;
        } else {
            for (int i = 0; i < m;) {
                const int32_t c = rep[rg.below(rep.size())];
                bool dup = false;
                for (int k = 0; k < i; ++k) dup |= (tgt[k] == c);
                if (!dup) tgt[i++] = c;
            }
        }
        for (int i = 0; i < m; ++i) {
            edges_out[2 * ne] = t;
            edges_out[2 * ne + 1] = tgt[i];
            ++ne;
            rep.push_back(tgt[i]);
            rep.push_back(t);
        }
    }
    // permute ids
    std::vector<int32_t> perm(n_node);
    for (int i = 0; i < n_node; ++i) perm[i] = i;
    for (int i = n_node - 1; i > 0; --i) std::swap(perm[i], perm[rp.below((uint64_t)i + 1)]);
    for (int64_t e = 0; e < 2 * ne; ++e) edges_out[e] = perm[edges_out[e]];
    // shuffle edge order
    for (int64_t e = ne - 1; e > 0; --e) {
        const int64_t j = (int64_t)rp.below((uint64_t)e + 1);
        std::swap(edges_out[2 * e], edges_out[2 * j]);
        std::swap(edges_out[2 * e + 1], edges_out[2 * j + 1]);
    }
    return ne;
}
