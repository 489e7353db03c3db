// bfs_gpu.hip -- BFS-tree construction on the GPU: the same trees as
// GraphGAN.construct_trees (reference src/GraphGAN/graph_gan.py:84-108) -- FIFO BFS, children in
// adjacency (file) order, self-loops / already-used nodes skipped -- for a batch of roots at once,
// written straight into the context's tree CSR without touching the host.
//
// The reference's queue order is reproduced level by level.  Let F_d be the nodes of depth d in
// pop order.  A node w of depth d+1 is appended by the FIRST frontier node (in pop order) that
// lists it, at that node's FIRST edge to it; so
//   claim : cand[w] = min over frontier edges (rank(v) << 32 | position of the edge in adj(v))
//   count : v's children = its edges that won the claim (in adjacency order)
//   scan  : exclusive scan of the child counts over F_d in pop order
//   write : F_{d+1}[base(v) + i] = i-th child of v; father, child index recorded
// After the last level: list length of v = 1 + #children, exclusive scan over node ids gives the
// tree CSR offsets, and every node drops itself into its father's list at its child index.
// All roots of a batch run in the same launches (grid.y = root); one small read-back per level.
#include <algorithm>
#include <vector>

#include "gg_internal.h"

namespace gg {

struct BfsArgs {
    int n_node, n_batch;
    const int64_t *rowptr;
    const int32_t *col;
    int32_t *father;               // [B][N]  -1 = not reached
    unsigned long long *cand;      // [B][N]  claim key, ~0 = unclaimed
    int32_t *queue;                // [B][N]  nodes in pop order
    int32_t *qcnt;                 // [B][N]  children of the node at queue position q
    int64_t *qbase;                // [B][N]  exclusive scan of qcnt over the current frontier
    int32_t *ccnt;                 // [B][N]  children per node id
    int32_t *cidx;                 // [B][N]  index of a node among its father's children
    int32_t *lo, *hi;              // [B]     current frontier = queue[lo, hi)
    int32_t *next_hi;              // [B]
};

// one wavefront per frontier node: claim the unvisited neighbours
__global__ __launch_bounds__(256) void bfs_claim_kernel(const BfsArgs a) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int lo = a.lo[b], hi = a.hi[b];
    const int64_t o = (int64_t)b * a.n_node;
    for (int q = lo + blockIdx.x * 4 + (threadIdx.x >> 6); q < hi; q += gridDim.x * 4) {
        const int v = a.queue[o + q];
        const int64_t e0 = a.rowptr[v], e1 = a.rowptr[v + 1];
        const unsigned long long rank = (unsigned long long)(q - lo) << 32;
        for (int64_t e = e0 + lane; e < e1; e += 64) {
            const int w = a.col[e];
            if (a.father[o + w] < 0) atomicMin(&a.cand[o + w], rank | (unsigned long long)(e - e0));
        }
    }
}

// one wavefront per frontier node.  WRITE == 0: count the edges that won their claim;
// WRITE == 1: append those children to the next frontier in adjacency order.
template <int WRITE>
__global__ __launch_bounds__(256) void bfs_children_kernel(const BfsArgs a) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int lo = a.lo[b], hi = a.hi[b];
    const int64_t o = (int64_t)b * a.n_node;
    for (int q = lo + blockIdx.x * 4 + (threadIdx.x >> 6); q < hi; q += gridDim.x * 4) {
        const int v = a.queue[o + q];
        const int64_t e0 = a.rowptr[v], e1 = a.rowptr[v + 1];
        const unsigned long long rank = (unsigned long long)(q - lo) << 32;
        const int64_t base = WRITE ? (int64_t)hi + a.qbase[o + q] : 0;
        int run = 0;
        for (int64_t eb = e0; eb < e1; eb += 64) {
            const int64_t e = eb + lane;
            int w = -1;
            bool child = false;
            if (e < e1) {
                w = a.col[e];
                // not reached before this level (in the write pass father[w] is only ever set by the one
                // edge whose key matches, i.e. by this very lane) and claimed by exactly this edge
                child = a.father[o + w] < 0 && a.cand[o + w] == (rank | (unsigned long long)(e - e0));
            }
            const unsigned long long bal = __ballot(child);
            if (WRITE && child) {
                const int idx = run + __popcll(bal & ((1ull << lane) - 1ull));
                a.queue[o + base + idx] = w;
                a.father[o + w] = v;
                a.cidx[o + w] = idx;
            }
            run += __popcll(bal);
        }
        if (!WRITE && lane == 0) {
            a.qcnt[o + q] = run;
            a.ccnt[o + v] = run;
        }
    }
}

// one block per root: exclusive scan of qcnt over the frontier [lo, hi) -> qbase, next_hi = hi + total
__global__ __launch_bounds__(1024) void bfs_scan_frontier_kernel(const BfsArgs a) {
    __shared__ int64_t wave_tot[16];
    __shared__ int64_t carry_sh;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lo = a.lo[b], hi = a.hi[b];
    const int64_t o = (int64_t)b * a.n_node;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    for (int base = lo; base < hi; base += 1024) {
        const int q = base + threadIdx.x;
        const int64_t v = q < hi ? a.qcnt[o + q] : 0;
        int64_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        int64_t pre = carry_sh, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i < wv) pre += wave_tot[i];
            tot += wave_tot[i];
        }
        if (q < hi) a.qbase[o + q] = pre + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_sh += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.next_hi[b] = hi + (int)carry_sh;
}

__global__ void bfs_advance_kernel(const BfsArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.n_batch) return;
    a.lo[b] = a.hi[b];
    a.hi[b] = a.next_hi[b];
}

__global__ void bfs_seed_kernel(const BfsArgs a, const int32_t *roots) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.n_batch) return;
    const int64_t o = (int64_t)b * a.n_node;
    const int r = roots[b];
    a.father[o + r] = r;
    a.queue[o] = r;
    a.lo[b] = 0;
    a.hi[b] = 1;
}

// one block per root: tree-CSR offsets = exclusive scan over node ids of (reached ? 1 + #children : 0)
__global__ __launch_bounds__(1024) void bfs_offsets_kernel(const BfsArgs a, int32_t *t_off /* rows of this batch */, int32_t *max_list) {
    __shared__ int64_t wave_tot[16];
    __shared__ int64_t carry_sh;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t o = (int64_t)b * a.n_node;
    int32_t *off = t_off + (int64_t)b * (a.n_node + 1);
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    int mx = 0;
    for (int base = 0; base < a.n_node; base += 1024) {
        const int v = base + threadIdx.x;
        const int64_t len = (v < a.n_node && a.father[o + v] >= 0) ? 1 + a.ccnt[o + v] : 0;
        mx = max(mx, (int)len);
        int64_t inc = len;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int64_t t = __shfl_up(inc, s, 64);
            if (lane >= s) inc += t;
        }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        int64_t pre = carry_sh, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i < wv) pre += wave_tot[i];
            tot += wave_tot[i];
        }
        if (v < a.n_node) off[v] = (int32_t)(pre + inc - len);
        __syncthreads();
        if (threadIdx.x == 0) carry_sh += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[a.n_node] = (int32_t)carry_sh;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) mx = max(mx, __shfl_xor(mx, s, 64));
    if (lane == 0) atomicMax(max_list, mx);
}

// every reached node writes its father slot and drops itself into its father's list
__global__ void bfs_fill_kernel(const BfsArgs a, const int32_t *t_off, int32_t *t_nbr, const int64_t *t_base /* of this batch */,
                                const int32_t *roots) {
    const int b = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= a.n_node) return;
    const int64_t o = (int64_t)b * a.n_node;
    const int f = a.father[o + v];
    if (f < 0) return;
    const int32_t *off = t_off + (int64_t)b * (a.n_node + 1);
    int32_t *nb = t_nbr + t_base[b];
    nb[off[v]] = f;
    if (v != roots[b]) nb[off[f] + 1 + a.cidx[o + v]] = v;
}

}  // namespace gg

using namespace gg;

extern "C" int gg_build_trees_device(gg_ctx *ctx, const int32_t *roots, int32_t n_roots) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, ctx->g_rowptr && !ctx->h_rowptr.empty(), GG_EINVAL, "gg_build_trees_device: call gg_set_graph_csr first");
    GG_CHECK(ctx, n_roots >= 0 && (roots || n_roots == 0), GG_EINVAL, "gg_build_trees_device: bad roots");
    const int n = ctx->n_node;
    for (int r = 0; r < n_roots; ++r) GG_CHECK(ctx, roots[r] >= 0 && roots[r] < n, GG_EINVAL, "gg_build_trees_device: root %d out of range", roots[r]);
    GG_HIP(ctx, hipSetDevice(ctx->device));
    // entries per root = 2 * |component| - 1: component sizes from one host sweep
    std::vector<int64_t> base(n_roots + 1);
    host_tree_sizes(n, ctx->h_rowptr.data(), ctx->h_col.data(), roots, n_roots, base.data());
    int rc = alloc_trees(ctx, roots, n_roots, base.data());
    if (rc != GG_OK) return rc;
    if (n_roots == 0) return GG_OK;

    // batch size: 36 bytes of working set per (root, node), at most ~8 GiB
    int B = (int)std::min<int64_t>(n_roots, std::max<int64_t>(1, (8ll << 30) / (36ll * n)));
    if (B > 65535) B = 65535;
    DevBuf father, cand, queue, qcnt, qbase, ccnt, cidx, misc;
    const size_t bn = (size_t)B * n;
    GG_HIP(ctx, father.reserve(4 * bn));
    GG_HIP(ctx, cand.reserve(8 * bn));
    GG_HIP(ctx, queue.reserve(4 * bn));
    GG_HIP(ctx, qcnt.reserve(4 * bn));
    GG_HIP(ctx, qbase.reserve(8 * bn));
    GG_HIP(ctx, ccnt.reserve(4 * bn));
    GG_HIP(ctx, cidx.reserve(4 * bn));
    GG_HIP(ctx, misc.reserve(sizeof(int32_t) * (3 * (size_t)B + 8)));
    BfsArgs a{};
    a.n_node = n;
    a.rowptr = ctx->g_rowptr;
    a.col = ctx->g_col;
    a.father = father.as<int32_t>();
    a.cand = cand.as<unsigned long long>();
    a.queue = queue.as<int32_t>();
    a.qcnt = qcnt.as<int32_t>();
    a.qbase = qbase.as<int64_t>();
    a.ccnt = ccnt.as<int32_t>();
    a.cidx = cidx.as<int32_t>();
    a.lo = misc.as<int32_t>();
    a.hi = a.lo + B;
    a.next_hi = a.hi + B;
    int32_t *d_maxlist = a.next_hi + B;
    GG_HIP(ctx, hipMemsetAsync(d_maxlist, 0, sizeof(int32_t), ctx->stream));
    std::vector<int32_t> h_lo(B), h_hi(B);
    int max_depth = 0;
    auto cleanup = [&]() {
        father.release(); cand.release(); queue.release(); qcnt.release(); qbase.release(); ccnt.release(); cidx.release(); misc.release();
    };
    for (int r0 = 0; r0 < n_roots; r0 += B) {
        const int nb = std::min(B, n_roots - r0);
        a.n_batch = nb;
        const size_t cur = (size_t)nb * n;
        GG_HIP(ctx, hipMemsetAsync(a.father, 0xFF, 4 * cur, ctx->stream));
        GG_HIP(ctx, hipMemsetAsync(a.cand, 0xFF, 8 * cur, ctx->stream));
        GG_HIP(ctx, hipMemsetAsync(a.ccnt, 0, 4 * cur, ctx->stream));
        hipLaunchKernelGGL(bfs_seed_kernel, dim3(cdiv(nb, 256)), dim3(256), 0, ctx->stream, a, ctx->t_root + r0);
        int64_t max_front = 1;
        for (int depth = 0;; ++depth) {
            int gx = (int)std::min<int64_t>((max_front + 3) / 4, 4096);
            if (gx < 1) gx = 1;
            hipLaunchKernelGGL(bfs_claim_kernel, dim3(gx, nb), dim3(256), 0, ctx->stream, a);
            hipLaunchKernelGGL(bfs_children_kernel<0>, dim3(gx, nb), dim3(256), 0, ctx->stream, a);
            hipLaunchKernelGGL(bfs_scan_frontier_kernel, dim3(nb), dim3(1024), 0, ctx->stream, a);
            hipLaunchKernelGGL(bfs_children_kernel<1>, dim3(gx, nb), dim3(256), 0, ctx->stream, a);
            hipLaunchKernelGGL(bfs_advance_kernel, dim3(cdiv(nb, 256)), dim3(256), 0, ctx->stream, a);
            GG_HIP(ctx, hipMemcpyAsync(h_lo.data(), a.lo, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, ctx->stream));
            GG_HIP(ctx, hipMemcpyAsync(h_hi.data(), a.hi, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, ctx->stream));
            hipError_t e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { cleanup(); return fail(ctx, GG_EHIP, "gg_build_trees_device: %s", hipGetErrorString(e)); }
            max_front = 0;
            for (int b = 0; b < nb; ++b) max_front = std::max<int64_t>(max_front, h_hi[b] - h_lo[b]);
            if (max_front == 0) break;
            max_depth = std::max(max_depth, depth + 1);
        }
        int32_t *off = ctx->t_off + (size_t)r0 * (n + 1);
        hipLaunchKernelGGL(bfs_offsets_kernel, dim3(nb), dim3(1024), 0, ctx->stream, a, off, d_maxlist);
        hipLaunchKernelGGL(bfs_fill_kernel, dim3(cdiv(n, 256), nb), dim3(256), 0, ctx->stream, a, off, ctx->t_nbr, ctx->t_base + r0,
                           ctx->t_root + r0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { cleanup(); return fail(ctx, GG_EHIP, "gg_build_trees_device: %s", hipGetErrorString(e)); }
    }
    int32_t ml = 0;
    hipError_t e = hipMemcpy(&ml, d_maxlist, sizeof(int32_t), hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(ctx, GG_EHIP, "gg_build_trees_device: %s", hipGetErrorString(e));
    ctx->tree_max_depth = max_depth;
    ctx->tree_max_list = ml;
    return GG_OK;
}
