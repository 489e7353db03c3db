// edge_cache.hip -- index structures of the walk sampler's edge-score cache (gg_internal.h, walk_sample.hip).
//
//   g_rev[e]   for graph edge e = (u -> v): an index e' of adj(v) with col[e'] == u (the undirected graph lists every edge
//              in both adjacencies, utils.py:36-37).  A walk standing on v that came from its tree father u finds the
//              score of its father candidate, s(v, u) (generator.py:21), at es[g_rev[t_edge[rank(v)]]].
//   t_edge[i]  for the node of BFS rank i of a root's tree: the CSR index of the edge (father -> node) the reference's BFS
//              appends it at (graph_gan.py:98-107: the FIRST occurrence of the node in its father's adjacency).  The GPU
//              BFS writes it in the same pass as the queue; trees that arrive in another way -- host builder, gg_set_trees,
//              tree cache file -- get it from derive_tree_edges below.
#include "gg_internal.h"

namespace gg {

// one thread per edge: scan adj(v) for u.  Sum over edges of deg(target) reads -- milliseconds for the 10^7-edge graph,
// once per gg_set_graph_csr.  An edge without a reverse (a directed CSR handed to the C ABI) gets -1 and clears `ok`.
__global__ __launch_bounds__(256) void reverse_edges_kernel(const int64_t *rowptr, const int32_t *col, int32_t n_node, int64_t nnz, int32_t *rev, int32_t *ok) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    int lo = 0, hi = n_node;  // u = the row of e: rowptr[u] <= e < rowptr[u + 1] (last such row: empty rows share offsets)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rowptr[mid] <= e) lo = mid; else hi = mid;
    }
    const int u = lo, v = col[e];
    int32_t r = -1;
    for (int64_t p = rowptr[v], pe = rowptr[v + 1]; p < pe; ++p)
        if (col[p] == u) { r = (int32_t)p; break; }
    rev[e] = r;
    if (r < 0) *ok = 0;
}

// does some adjacency list hold a node twice?  e is the FIRST occurrence of its target in its list iff rev[rev[e]] == e (rev names
// first occurrences); the lazy trees' resolution (walk_sample.hip) skips its first-occurrence tests on a graph without duplicates
__global__ __launch_bounds__(256) void multi_edge_kernel(const int32_t *rev, int64_t nnz, int32_t *multi) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int32_t r = rev[e];
    if (r >= 0 && rev[r] != (int32_t)e) *multi = 1;
}

// one thread per tree node (slot r, rank i): match its children -- consecutive ranks, in adjacency order -- against
// adj(order[i]) from the left; the first occurrence of each child is the edge the BFS appended it at.
__global__ __launch_bounds__(256) void tree_edges_kernel(const int32_t *order, const int32_t *cstart, const int64_t *base, int32_t n_roots,
                                                        const int64_t *rowptr, const int32_t *col, int32_t *edge, int32_t *ok) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= base[n_roots]) return;
    int lo = 0, hi = n_roots;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (base[mid] <= j) lo = mid; else hi = mid;
    }
    const int r = lo;
    const int64_t b = base[r];
    const int i = (int)(j - b);
    if (i == 0) edge[b] = -1;  // the root has no father edge
    const int32_t *cs = cstart + b + r;
    int c = cs[i];
    const int cend = cs[i + 1];
    if (c >= cend) return;
    const int v = order[j];
    int want = order[b + c];
    for (int64_t p = rowptr[v], pe = rowptr[v + 1]; p < pe; ++p) {
        if (col[p] == want) {
            edge[b + c] = (int32_t)p;
            if (++c == cend) return;
            want = order[b + c];
        }
    }
    *ok = 0;  // a child that is no graph neighbour of its father: uploaded lists that are no subgraph of the resident graph
}

int compute_reverse_edges(gg_ctx *ctx) {
    if (!ctx->g_rev || ctx->g_nnz == 0) return GG_OK;
    int32_t *ok = (int32_t *)(ctx->dev_ctr + 2001);
    int32_t h_ok = 1;
    GG_HIP(ctx, hipMemcpy(ok, &h_ok, sizeof(int32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(reverse_edges_kernel, dim3((unsigned)cdiv(ctx->g_nnz, 256)), dim3(256), 0, ctx->stream, ctx->g_rowptr, ctx->g_col, ctx->n_node,
                       ctx->g_nnz, ctx->g_rev, ok);
    GG_HIP(ctx, hipGetLastError());
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GG_HIP(ctx, hipMemcpy(&h_ok, ok, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (!h_ok) {  // not a symmetric adjacency: the cache cannot serve father candidates -> every distribution scores privately
        (void)hipFree(ctx->g_rev);
        ctx->g_rev = nullptr;
        return GG_OK;
    }
    int32_t h_multi = 0;
    GG_HIP(ctx, hipMemcpy(ok, &h_multi, sizeof(int32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(multi_edge_kernel, dim3((unsigned)cdiv(ctx->g_nnz, 256)), dim3(256), 0, ctx->stream, ctx->g_rev, ctx->g_nnz, ok);
    GG_HIP(ctx, hipGetLastError());
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GG_HIP(ctx, hipMemcpy(&h_multi, ok, sizeof(int32_t), hipMemcpyDeviceToHost));
    ctx->g_multi = h_multi != 0;
    return GG_OK;
}

int derive_tree_edges(gg_ctx *ctx) {
    ctx->t_edge_valid = false;
    if (!ctx->g_rowptr || !ctx->t_edge || ctx->n_tree_roots <= 0 || ctx->tree_nodes <= 0 || ctx->g_nnz >= (1ll << 31)) return GG_OK;
    int32_t *ok = (int32_t *)(ctx->dev_ctr + 2001);
    int32_t h_ok = 1;
    GG_HIP(ctx, hipMemcpy(ok, &h_ok, sizeof(int32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(tree_edges_kernel, dim3((unsigned)cdiv(ctx->tree_nodes, 256)), dim3(256), 0, ctx->stream, ctx->t_order, ctx->t_cstart, ctx->t_base,
                       ctx->n_tree_roots, ctx->g_rowptr, ctx->g_col, ctx->t_edge, ok);
    GG_HIP(ctx, hipGetLastError());
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GG_HIP(ctx, hipMemcpy(&h_ok, ok, sizeof(int32_t), hipMemcpyDeviceToHost));
    ctx->t_edge_valid = h_ok != 0;
    return GG_OK;
}

}  // namespace gg
