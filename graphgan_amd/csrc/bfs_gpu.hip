// bfs_gpu.hip -- BFS-tree construction on the GPU: the same trees as
// GraphGAN.construct_trees (reference src/GraphGAN/graph_gan.py:84-108) -- FIFO BFS, children in
// adjacency (file) order, self-loops / already-used nodes skipped -- written straight into the
// context's BFS-order tree arrays (gg_internal.h) without touching the host.
//
// One WORKGROUP per root, one root per CU at a time (persistent grid, roots drawn from a ticket counter).
// The reference's BFS is sequential: pop v, scan adj(v) in order, append every node not seen before.  The
// edges it inspects form ONE STREAM -- (pop order of v, position in adj(v)) -- and a node is appended by the
// FIRST edge of that stream that reaches it.  The workgroup replays that stream 8 192 edges at a time:
//
//   * the "seen" set is a BITMAP IN LDS (1 bit per node: 122 KB for 10^6 nodes, of the CU's 160 KB), so
//     the test that dominates a BFS -- 20 M of them per tree of the 1M-node / 10M-edge graph -- never leaves the
//     CU.  (The round-1 kernels kept 36 B of state per (root, node) in HBM for 238 roots at once: every test
//     was a random HBM access, 4.9 GB of traffic per tree.)  Graphs whose bitmap exceeds the LDS use a per-
//     workgroup bitmap in global memory (L2 / MALL resident) through the same code.
//   * a chunk = 1 024 threads x 8 consecutive stream positions.  Test (all threads) | barrier | claim: the
//     unseen targets set their bit with an LDS atomic-or; exactly one edge per new node sees the bit clear
//     (the hardware winner), the others (duplicates INSIDE the chunk: rare) append {node, position} to a short
//     LDS list | barrier | every hardware winner takes the smallest stream position among its own and the
//     list's entries for its node -- the edge the sequential BFS would have appended it at -- and marks that
//     position in an 8 192-bit LDS mask | barrier | the marked positions are compacted in stream order (wave
//     scan + 16 wave totals) onto the queue, and the thread holding the LAST edge of a queue node records
//     where that node's children end: cstart[] comes out of the same pass.  If the duplicate list overflows
//     (many edges of one chunk into the same few new nodes) the chunk is resolved through a per-workgroup
//     atomic-min key array in global memory instead -- exact as well, just slower.
//   * the queue IS the output (t_order), cstart the second output: no offsets scan, no fill pass.
//
// Traffic per tree: the adjacency (80 MB, shared by all 256 concurrent roots: L2 / MALL hits), 16 B of row
// pointers per node, and 8 B per node of output -- the only part that has to reach HBM.
#include <algorithm>
#include <vector>

#include "gg_internal.h"

namespace gg {

constexpr int BFS_T = 1024;              // threads per workgroup
constexpr int BFS_WAVES = BFS_T / 64;
constexpr int BFS_U = 8;                 // consecutive stream positions per thread and chunk
constexpr int BFS_CH = BFS_T * BFS_U;    // edges per chunk
constexpr int BFS_NB = BFS_T;            // queue nodes per batch (one per thread)
constexpr int BFS_TE_CAP = 32768;        // edges per batch (the batch -> node map below has TE_CAP / U entries)
constexpr int BFS_BIG = 1024;            // a node with more edges forms a batch of its own (trivial map)
constexpr int BFS_LCAP = 512;            // in-chunk duplicate list

struct BfsArgs {
    int n_node, n_roots;
    const int64_t *rowptr;
    const uint32_t *rowptr32;  // the same offsets in 4 bytes (< 2^31 entries): 4 MB for 10^6 nodes stay in an XCD's L2, the 8 MB of rowptr do not
    const int32_t *col;
    const int32_t *roots;    // [n_roots] device (t_root)
    const int64_t *base;     // [n_roots + 1] device (t_base)
    int32_t *order;          // t_order
    int32_t *cstart;         // t_cstart
    int32_t *edge;           // t_edge: CSR index of the edge a node was appended at (father -> node)
    unsigned int *ticket;    // next root to take
    int32_t *stats;          // [0] max depth, [1] longest list (1 + most children), [2] error flag
    uint32_t *gbitmap;       // [grid][bm_words]  (graphs too large for the LDS bitmap)
    uint32_t *gkey;          // [grid][n_node]    all-ones between uses (duplicate-list overflow path)
    int bm_words;
    int exp;                   // GG_BFS_EXPERIMENT: timing ablations (results are then WRONG): 1 = no cstart stores, 2 = synthetic targets instead of adjacency loads, 4 = no queue stores beyond level 1, 8 = no early exit when the component is complete, 64 = streaming stores for t_edge, 128 = nothing (the instrumented instance as its own baseline): results stay right with 8 / 64 / 128
    unsigned long long *prof;  // GG_BFS_PROFILE: [32] shader-clock cycles of wave 0 per phase + event counts (NULL: off)
    // sparse levels (bfs_order2_kernel with the bitmap in LDS; see "SPARSE LEVEL" there): per-workgroup scratch
    int sp_k;                  // a level is expanded through its candidate fathers when unseen nodes * sp_k <= level nodes (< 0: never)
    int sp_min;                // ... and the level has at least this many nodes
    int sp_buckets;            // rank buckets of a level (passes over the unseen nodes)
    int sp_cap;                // unseen nodes the scratch holds
    int32_t *sp_ux;            // [grid][2][sp_cap]   unseen nodes not served yet
    unsigned long long *sp_ui; // [grid][2][sp_cap]   their first CSR entry | degree << 32
    unsigned long long *sp_s;  // [grid][n_node]   the candidate fathers, ascending: queue rank | node id << 32
    uint32_t *sp_vis;          // [grid][bm_words] the visited bitmap while the LDS holds a bucket's
    uint32_t *sp_mark;         // [grid][bm_words] candidate fathers by node id (all-zero between uses)
    unsigned long long *sp_mask;  // [grid][n_node / 64 + 32] the same by rank: one ballot per 64 ranks of the level
    // whole trees outside the slots' own segments (the arena of lazy builds) / LAZY trees (bfs_order2_kernel<.., .., true>)
    const int32_t *expect;     // [n_roots] nodes of the root's component (NULL: base[r + 1] - base[r])
    const int32_t *slot_ids;   // [n_roots] tree slot of launch item r (NULL: r): offsets the cstart row, indexes the lz_* rows
    const int32_t *lz_limit;   // [n_roots] LAZY: node limit of the exact part (>= the component: the tree is built whole)
    int4 *lz_info;             // [slots]   LAZY: {first rank without a children list, exact ranks, level of those ranks, capacity of the segment}
    uint2 *lz_bm;              // [slots][bm_words] LAZY: {visited word, members below the word}
    int32_t *lz_rank;          // [tree nodes] LAZY: BFS rank by member index, at base[r]
    int32_t *lz_cursor;        // [slots]   LAZY: first free rank of the pool
    // LAZY: the BFS runs in a per-workgroup scratch tree; what it keeps -- the exact ranks, the built lists, a pool behind them --
    // gets its place in the tree arrays from ONE cursor when the size is known (a slot costs what its tree needs, not a worst case)
    int32_t *scr_order, *scr_edge, *scr_cstart;  // [grid][scr_cap] ([grid][scr_cap + 1])
    int scr_cap;
    const int32_t *lz_pool;    // [n_roots] pool entries behind the exact ranks of a lazy slot
    unsigned long long *lz_alloc;  // the cursor (entries)
    long long lz_alloc_end;    // entries the segments may take; a slot that does not fit raises lz_flag (whole tree in the arena) and stats[4]
    int64_t *base_w;           // [slots]   LAZY: t_base, written here
    int32_t *lz_flag;          // [slots]
    int lz_easy, lz_limit_easy;  // levels 1 .. lz_easy are expanded while they fit lz_limit_easy (>= the root's own limit)
    int lz_ablate;             // GG_LZ_ABLATE (timing only, results WRONG): 1 = no rank scatter, 2 = no copy out of the scratch tree, 4 = no visited index; 8 (results right) = the ranks' indices from the index in global memory instead of LDS
};

__device__ __forceinline__ int lanes_below(unsigned long long m) {  // popcount of m restricted to the lanes below this one
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// What bounds the kernel (GG_BFS_PROFILE, rocprof SQ counters, GG_BFS_EXPERIMENT ablations): instruction issue.  The
// adjacency loads return within 3 % of the time (replacing them by computed targets saves a quarter), stores cost
// nothing; 16 wavefronts share 4 SIMDs and one scalar unit, and every edge costs a map step, a load, a bit test and
// its share of the compaction.  So the per-edge path is kept short: 8 consecutive stream positions per thread whose
// owner is found with ONE LDS read (a per-batch map position / 8 -> batch node) and then followed along, per-position
// flags in bit masks, ballot + mbcnt compaction (no LDS shuffles), and chunks that discover nothing -- half of all
// chunks, most of the last two levels -- leave after one barrier.  (Measured and not kept: lane-consecutive positions
// for fully coalesced loads, and a per-wave node-start bitmap instead of the map -- same speed or slower: more
// instructions per edge, and the loads were never the limit.)
// INSTR: the phase clocks (GG_BFS_PROFILE) and the ablation switches (GG_BFS_EXPERIMENT) are compiled into a second
// instance only -- tested at run time in the production kernel they cost 10 % (85.7 -> 95.7 us per tree).
template <bool LDS_BM, bool INSTR>
__global__ __launch_bounds__(BFS_T) void bfs_order_kernel(const BfsArgs a) {
    extern __shared__ uint32_t lds_bm[];           // [bm_words] when LDS_BM
    __shared__ int32_t eoff[BFS_NB + 1];           // exclusive scan of the batch's degrees
    __shared__ uint32_t e0s[BFS_NB];               // first edge of each batch node
    __shared__ uint16_t emap[BFS_TE_CAP / BFS_U];  // batch node that owns stream position U * g
    __shared__ int32_t L_w[BFS_LCAP], L_pos[BFS_LCAP];
    __shared__ uint32_t winbits[BFS_CH / 32];
    __shared__ int32_t wave_cnt[BFS_WAVES], wave_tot[BFS_WAVES];
    __shared__ int32_t Lcount, s_root, s_max, s_cap, s_fb, s_any[3];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t *const bm = LDS_BM ? lds_bm : a.gbitmap + (size_t)blockIdx.x * a.bm_words;
    uint32_t *const gkey = a.gkey + (size_t)blockIdx.x * a.n_node;

    auto seen = [&](int w) -> bool {
        if (LDS_BM) return (bm[w >> 5] >> (w & 31)) & 1u;
        return (__hip_atomic_load(&bm[w >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (w & 31)) & 1u;
    };

    for (;;) {
        if (tid == 0) {
            s_root = (int)atomicAdd(a.ticket, 1u);
            Lcount = 0;
            s_max = 0;
            s_any[0] = s_any[1] = s_any[2] = 0;
        }
        for (int i = tid; i < BFS_CH / 32; i += BFS_T) winbits[i] = 0u;
        __syncthreads();
        const int r = s_root;
        if (r >= a.n_roots) return;
        const int root = a.roots[r];
        int32_t *const order = a.order + a.base[r];
        int32_t *const cstart = a.cstart + a.base[r] + r;
        int32_t *const tedge = a.edge + a.base[r];
        const int expect = (int)(a.base[r + 1] - a.base[r]);
        for (int i = tid; i < a.bm_words; i += BFS_T) bm[i] = 0u;
        __syncthreads();
        if (tid == 0) {
            bm[root >> 5] = 1u << (root & 31);
            order[0] = root;
            tedge[0] = -1;
            cstart[0] = 1;
            if (a.rowptr[root + 1] == a.rowptr[root]) cstart[1] = 1;  // isolated root: no edge ever closes its (empty) child range
        }
        __syncthreads();

        int head = 0, tail = 1, level_end = 1, depth = 0;
        // phase clocks of wave 0 (time to the barrier that ends the phase = what the whole workgroup waited for)
        unsigned long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        long long tprev = (INSTR && a.prof) ? (long long)clock64() : 0;
#define BFS_TICK(k)                                             \
    if (INSTR && a.prof) {                                      \
        const long long tn = (long long)clock64();              \
        pc[k] += (unsigned long long)(tn - tprev);              \
        tprev = tn;                                             \
    }
        int pf_q = -1;          // queue index whose node this thread has prefetched for the NEXT batch
        uint32_t pf_e0 = 0;
        int pf_deg = 0;
        unsigned chunk_no = 0;  // rotates the "any new node" flags
        while (head < tail) {
            if (head == level_end) {  // the next level starts: everything up to `tail` belongs to it
                level_end = tail;
                ++depth;
            }
            const int navail = min(BFS_NB, level_end - head);
            // ---- the batch's nodes: first edge and degree (prefetched during the previous batch when the queue was that far)
            uint32_t e0 = 0;
            int deg = 0;
            if (tid < navail) {
                if (pf_q == head + tid) {
                    e0 = pf_e0;
                    deg = pf_deg;
                } else {
                    const int v = __hip_atomic_load(&order[head + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int64_t b = a.rowptr[v];
                    e0 = (uint32_t)b;
                    deg = (int)(a.rowptr[v + 1] - b);
                }
            }
            int inc = deg;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            if (lane == 63) wave_tot[wv] = inc;
            if (tid == 0) { s_cap = 0; s_fb = BFS_NB; }
            __syncthreads();
            BFS_TICK(0)  // node info + scan
            int pre = 0;
#pragma unroll
            for (int i = 0; i < BFS_WAVES; ++i)
                if (i < wv) pre += wave_tot[i];
            const int incl = pre + inc, excl = incl - deg;
            // batch = the longest prefix of the available nodes that has at most TE_CAP edges and no big node,
            // or one big node alone
            const bool capok = tid < navail && incl <= BFS_TE_CAP;
            const unsigned long long capbal = __ballot(capok);
            if (lane == 0 && capbal) atomicAdd(&s_cap, (int)__popcll(capbal));
            if (tid < navail && deg > BFS_BIG) atomicMin(&s_fb, tid);
            __syncthreads();
            const int fb = s_fb;
            const bool single = fb == 0;
            const int nb = single ? 1 : min(fb, s_cap);
            if (tid < nb) {
                eoff[tid] = excl;
                e0s[tid] = e0;
                if (tid == nb - 1) eoff[nb] = incl;
                if (!single) {
                    const int g1 = (incl + BFS_U - 1) / BFS_U;
                    for (int g = (excl + BFS_U - 1) / BFS_U; g < g1; ++g) emap[g] = (uint16_t)tid;
                }
            }
            // ---- prefetch the next batch's nodes (queue entries below `tail` are final)
            {
                const int q2 = head + nb + tid;
                pf_q = -1;
                if (q2 < tail) {
                    const int v2 = __hip_atomic_load(&order[q2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int64_t b2 = a.rowptr[v2];
                    pf_e0 = (uint32_t)b2;
                    pf_deg = (int)(a.rowptr[v2 + 1] - b2);
                    pf_q = q2;
                }
            }
            __syncthreads();
            BFS_TICK(1)  // batch formation, map fill, prefetch issue
            pc[8] += 1;
            const int TE = eoff[nb];

            // ---- the batch's edge stream, BFS_CH positions at a time
            for (int P = 0; P < TE; P += BFS_CH, ++chunk_no) {
                const int p0 = P + BFS_U * tid;
                int w[BFS_U];
                uint32_t vmask = 0, lmask = 0, cmask = 0;  // valid / last edge of its node / unseen target
                int qfirst = 0;
                if (p0 < TE) {
                    int idx = single ? 0 : (int)emap[p0 / BFS_U];
                    qfirst = idx;
                    uint32_t e = e0s[idx] + (uint32_t)(p0 - eoff[idx]);
                    int nextb = eoff[idx + 1];
#pragma unroll
                    for (int j = 0; j < BFS_U; ++j) {
                        const int p = p0 + j;
                        w[j] = 0;
                        if (p < TE) {
                            if (p == nextb) {  // every batch node has at least one edge: at most one boundary per step
                                ++idx;
                                e = e0s[idx];
                                nextb = eoff[idx + 1];
                            }
                            w[j] = (INSTR && (a.exp & 2)) ? (int)((e * 2654435761u) % (uint32_t)a.n_node) : a.col[e];
                            ++e;
                            vmask |= 1u << j;
                            if (p + 1 == nextb) lmask |= 1u << j;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < BFS_U; ++j)
                        if (((vmask >> j) & 1u) && !seen(w[j])) cmask |= 1u << j;
                }
                const int slot = (int)(chunk_no % 3u);
                if (__ballot(cmask != 0u) && lane == 0) s_any[slot] = 1;
                __syncthreads();  // every test before any set: a later edge must not hide an earlier one
                BFS_TICK(2)  // map, adjacency loads, tests
                pc[9] += 1;
                if (tid == 0) s_any[(slot + 2) % 3] = 0;  // the previous chunk's flag: everyone is past reading it
                if (!s_any[slot]) {
                    // nothing new in this chunk: every node that ends here has its children end at `tail`
                    uint32_t lm = lmask;
                    while (lm) {
                        const int j = __ffs(lm) - 1;
                        lm &= lm - 1;
                        if (!(INSTR && (a.exp & 1))) cstart[head + qfirst + __popc(lmask & ((1u << j) - 1u)) + 1] = tail;
                    }
                    BFS_TICK(3)  // chunk without new nodes
                    pc[10] += 1;
                    continue;
                }
                uint32_t hmask = 0;  // this edge cleared -> set the bit (the hardware winner among the chunk's edges to its node)
#pragma unroll
                for (int j = 0; j < BFS_U; ++j) {
                    if ((cmask >> j) & 1u) {
                        const uint32_t bit = 1u << (w[j] & 31);
                        const uint32_t old = atomicOr(&bm[w[j] >> 5], bit);
                        if (old & bit) {  // another edge of this chunk reaches the same new node
                            const int li = atomicAdd(&Lcount, 1);
                            if (li < BFS_LCAP) { L_w[li] = w[j]; L_pos[li] = p0 + j - P; }
                        } else {
                            hmask |= 1u << j;
                        }
                    }
                }
                __syncthreads();
                BFS_TICK(4)  // claim
                const int nL = Lcount;
                if (nL > 0) pc[11] += 1;
                uint32_t mine = hmask;  // positions that append their target; without in-chunk duplicates: the winners themselves
                if (nL > 0) {
                    if (nL <= BFS_LCAP) {
                        // the sequential BFS appends a node at the FIRST edge that reaches it: smallest position among the
                        // hardware winner's and the duplicates'
                        if (hmask) {
                            int eff[BFS_U];
#pragma unroll
                            for (int j = 0; j < BFS_U; ++j) eff[j] = p0 + j - P;
                            for (int i = 0; i < nL; ++i) {
                                const int lw = L_w[i], lp = L_pos[i];
#pragma unroll
                                for (int j = 0; j < BFS_U; ++j)
                                    if (lw == w[j]) eff[j] = min(eff[j], lp);
                            }
#pragma unroll
                            for (int j = 0; j < BFS_U; ++j)
                                if ((hmask >> j) & 1u) atomicOr(&winbits[eff[j] >> 5], 1u << (eff[j] & 31));
                        }
                    } else {
                        // too many in-chunk duplicates for the list: smallest position per node through the key array
#pragma unroll
                        for (int j = 0; j < BFS_U; ++j)
                            if ((cmask >> j) & 1u) atomicMin(&gkey[w[j]], (uint32_t)(p0 + j - P));
                        __syncthreads();
#pragma unroll
                        for (int j = 0; j < BFS_U; ++j)
                            if (((cmask >> j) & 1u) && atomicMin(&gkey[w[j]], 0xFFFFFFFFu) == (uint32_t)(p0 + j - P))
                                atomicOr(&winbits[(p0 + j - P) >> 5], 1u << ((p0 + j - P) & 31));
                        __syncthreads();
#pragma unroll
                        for (int j = 0; j < BFS_U; ++j)
                            if ((cmask >> j) & 1u) __hip_atomic_store(&gkey[w[j]], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __syncthreads();
                    BFS_TICK(5)  // duplicate resolution
                    mine = (winbits[(BFS_U * tid) >> 5] >> ((BFS_U * tid) & 31)) & ((1u << BFS_U) - 1u);
                }
                // ---- the appending positions, in stream order, are the next queue entries
                int before = 0, wtotal = 0;
#pragma unroll
                for (int j = 0; j < BFS_U; ++j) {
                    const unsigned long long bal = __ballot((mine >> j) & 1u);
                    before += lanes_below(bal);
                    wtotal += (int)__popcll(bal);
                }
                if (lane == 0) wave_cnt[wv] = wtotal;
                __syncthreads();
                BFS_TICK(6)  // in-wave compaction
                int wpre = 0, total = 0;
#pragma unroll
                for (int i = 0; i < BFS_WAVES; ++i) {
                    const int c = wave_cnt[i];
                    if (i < wv) wpre += c;
                    total += c;
                }
                int rank = tail + wpre + before;
                int q = head + qfirst + 1;
                if (mine | lmask) {
                    // the CSR index of this thread's stream positions again (as in the load loop): an appended node records
                    // the edge it was appended at -- the edge-score cache of the walk sampler is indexed by it
                    int idx2 = qfirst;
                    uint32_t e2 = e0s[idx2] + (uint32_t)(p0 - eoff[idx2]);
                    int nextb2 = eoff[idx2 + 1];
#pragma unroll
                    for (int j = 0; j < BFS_U; ++j) {
                        if (p0 + j < TE && p0 + j == nextb2) {
                            ++idx2;
                            e2 = e0s[idx2];
                            nextb2 = eoff[idx2 + 1];
                        }
                        if ((mine >> j) & 1u) {
                            if (rank < expect) { order[rank] = w[j]; tedge[rank] = (int32_t)e2; }
                            ++rank;
                        }
                        ++e2;
                        if ((lmask >> j) & 1u) {  // children of this queue node end here
                            if (!(INSTR && (a.exp & 1))) cstart[q] = rank;
                            ++q;
                        }
                    }
                }
                tail += total;
                if (nL > 0) {  // everyone has read its bits (barrier above); the next set is two barriers away
                    if (tid < BFS_CH / 32) winbits[tid] = 0u;
                    if (tid == 0) Lcount = 0;
                }
                BFS_TICK(7)  // queue / cstart stores issued
                if (tail > expect) break;  // more nodes than the component sweep promised: the graph is not the one set (uniform)
            }
            if (tail > expect) break;
            head += nb;
            __syncthreads();
            // Every node of the root's component is on the queue (its size is known from the component sweep): no edge of
            // the nodes still waiting can discover anything.  On a small-world graph that is the bulk of the stream -- the
            // last two levels hold most nodes and found nothing in half of all chunks -- so their adjacency is not read at
            // all: they are leaves of the tree (their child ranges are empty, at the end of the queue).
            if (tail == expect && head < tail && !(INSTR && (a.exp & 8))) {
                for (int i = head + tid; i < expect; i += BFS_T) cstart[i + 1] = expect;
                if (level_end < tail) ++depth;  // the nodes behind level_end are one level deeper than the one being popped
                __syncthreads();                // (the child-count sweep below reads these entries)
                break;
            }
        }
        if (INSTR && a.prof && tid == 0)
            for (int k = 0; k < 12; ++k) atomicAdd(&a.prof[k], pc[k]);
#undef BFS_TICK

        // ---- per-root results: node count check, depth, longest list (1 + most children)
        int mc = 0;
        if (tail == expect)
            for (int i = tid; i < tail; i += BFS_T) mc = max(mc, cstart[i + 1] - cstart[i]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mc = max(mc, __shfl_xor(mc, off, 64));
        if (lane == 0) atomicMax(&s_max, mc);
        __syncthreads();
        if (tid == 0) {
            if (tail != expect) a.stats[2] = 1;
            atomicMax(&a.stats[0], depth);
            atomicMax(&a.stats[1], s_max + 1);
        }
        __syncthreads();
    }
}


// ============================================================================================================================
// Round 4: the same BFS as a SCAN phase and a CLAIM phase per window (bfs_order2_kernel, the default; GG_BFS_V1=1 runs the
// chunk kernel above).  Measured on the chunk kernel and on the first version of this one (profiles/r4_bfs_notes.txt): a unit of
// ~8 000 stream positions costs ~17 000 cycles either way -- a third of it memory round trips that a workgroup alone on its CU
// cannot hide (every __syncthreads also waits for the stores and prefetches in flight), the rest ~550 instructions per thread of
// per-unit bookkeeping (two block scans, a candidate list, a child-count scan).  So, per window of up to 1 024 queue nodes /
// 4 096 quads (= 16 384 stream positions):
//   * SCAN: a thread takes quads of 4 consecutive adjacency entries (one 16-byte load each, all of a thread's loads in flight
//     together), tests the four targets against the visited bitmap in LDS and remembers the unseen ones (CANDIDATES) in a 4-bit
//     mask per quad -- the targets stay in its registers.  Nothing is written; a window without a candidate (most windows
//     of the last levels) ends here.
//   * CLAIM, by the thread that scanned: atomic-or on the bitmap -- the hardware winner among the window's edges into the same
//     new node; in-window duplicates register their smallest position per node in an LDS hash and the winner takes the minimum of it and its own, i.e. the edge the
//     sequential BFS appends the node at (graph_gan.py:101-107), exactly as in the chunk kernel, overflow path included.  The
//     appending positions are bits of a mask over the window's positions (position = 4 * quad + entry: stream order); ONE
//     popcount prefix over its 512 words gives every appended node its queue rank AND every window node the end of its child
//     range (cstart): no candidate list, no per-node counters, no second scan.
//   * barriers wait for LDS only; the one global dependency -- queue entries written by earlier windows and read back as the
//     next nodes -- is fenced when the reader gets within reach of entries written since the last fence (once per level).
// Same outputs, bit for bit (tests/test_gpu_walk.py::test_gpu_bfs_builds_the_reference_trees runs both kernels).
constexpr int B2_T = 1024;
constexpr int B2_WAVES = B2_T / 64;
constexpr int B2_NB = 1024;              // queue nodes per window
constexpr int B2_SLOTS = 4096;           // quads per window
constexpr int B2_POS = 4 * B2_SLOTS;     // stream positions per window
constexpr int B2_WORDS = B2_POS / 32;    // words of the position mask (one per thread of the first 8 wavefronts)
constexpr int B2_NARROW = 32;           // a node of up to this many quads writes its own quad -> node entries; the wavefront fills a wider one's together (~100 cycles per such node: with 8, the hub-rich levels 2-3 spent thousands of cycles per window there)
constexpr int B2_HASH = 1024;            // in-window duplicates: LDS hash node -> smallest duplicate position ...
constexpr int B2_DCAP = 768;             // ... for up to this many duplicates per window (more: the key array in global memory)

typedef int32_t int4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load from a 4-byte aligned address
typedef uint32_t uint2u __attribute__((ext_vector_type(2), aligned(4)));  // 8-byte load from a 4-byte aligned address

// inclusive prefix sum over the wavefront with data-parallel-primitive moves (no LDS round trips: six dependent ds_bpermute made a
// scan ~700 cycles, and a window has two): shifts by 1, 2, 4, 8 inside the rows of 16 lanes, then lane 15 of row 0 / 2 broadcast into
// row 1 / 3 and lane 31 into rows 2-3 (the sequence of LLVM's own wave64 scan for this ISA family)
__device__ __forceinline__ int wave_incl_scan(int v, int /*lane*/) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

// workgroup barrier that waits for this wavefront's LDS traffic only: global loads / stores in flight stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// loads of words this workgroup (or an L2 atomic) wrote earlier in the same launch: served by the L2, never by a stale L1 line
__device__ __forceinline__ int ldi(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ldu(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ldq(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

typedef uint32_t __attribute__((address_space(3))) lds_u32_t;
typedef int32_t __attribute__((address_space(3))) lds_i32_t;

// SPARSE LEVEL of bfs_order2_kernel (see there): the candidate fathers S of the level [lo, hi) of one root's queue, ascending, into
// `slist`; returns their number, or -1 if the unseen nodes do not fit the scratch (nothing has changed then).  On return the LDS
// bitmap `bm` is the visited bitmap again, cstart[lo + 1 .. hi] is zero, and wtot[16] / wtot[17] hold the unseen nodes listed and
// the bucket passes made (profile).  A function of its own (not inlined): its registers must not compete with the window loop's
// (inlined, the kernel spilled 84-172 VGPRs, some inside the window loop).  Called by all 1 024 threads of the workgroup.
// Every loop keeps several independent loads in flight per thread -- a workgroup alone on its CU hides no latency by itself:
// written as plain one-load-per-iteration loops the search cost 9 M cycles per tree, more than the level popped whole.
template <bool INSTR>
__device__ __forceinline__ int sparse_fathers(unsigned long long *sph, lds_u32_t *bm, lds_i32_t *wtot, lds_i32_t *s_live, const uint32_t *__restrict__ rowptr32,
                                           const int32_t *__restrict__ col, const int32_t *order, int32_t *cstart, int32_t *ux, unsigned long long *ui,
                                           unsigned long long *slist, uint32_t *vis, uint32_t *mark, unsigned long long *smask, int n_node, int W, int sp_cap,
                                           int NBK, int lo, int hi, int Un) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int F = hi - lo;
    long long tprev = INSTR ? (long long)clock64() : 0;  // phase clocks (GG_BFS_PROFILE): list, cstart zeros, bucket bitmap, scan, restore, rank masks, S list
#define SP_TICK(k)                                  \
    if (INSTR) {                                    \
        const long long tn = (long long)clock64();  \
        sph[k] += (unsigned long long)(tn - tprev); \
        tprev = tn;                                 \
    }
    const uint32_t lastmask = (n_node & 31) ? (1u << (n_node & 31)) - 1u : 0xffffffffu;
    // unseen nodes: count (the bitmap is saved on the way), list
    int zc = 0;
    for (int i = tid; i < W; i += B2_T) {
        const uint32_t wd = bm[i];
        vis[i] = wd;
        uint32_t z = ~wd;
        if (i == W - 1) z &= lastmask;
        zc += (int)__popc(z);
    }
    const int zinc = wave_incl_scan(zc, lane);
    if (lane == 63) wtot[wv] = zinc;
    lds_barrier();
    int zpre = 0, nU = 0;
#pragma unroll
    for (int i = 0; i < B2_WAVES; ++i) {
        const int c = wtot[i];
        if (i < wv) zpre += c;
        nU += c;
    }
    if (nU > sp_cap) return -1;  // (other components' nodes are listed too: they never hit)
    {
        int o = zpre + zinc - zc;
        for (int i = tid; i < W; i += B2_T) {
            uint32_t z = ~bm[i];
            if (i == W - 1) z &= lastmask;
            while (z) {
                const int b = __ffs((int)z) - 1;
                z &= z - 1u;
                ux[o++] = (i << 5) + b;
            }
        }
    }
    __syncthreads();
    for (int idx0 = tid; idx0 < nU; idx0 += 4 * B2_T) {  // their adjacency ranges
        int xs[4];
        uint2u rp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xs[u] = idx0 + u * B2_T < nU ? ldi(&ux[idx0 + u * B2_T]) : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (xs[u] >= 0) rp[u] = *reinterpret_cast<const uint2u *>(rowptr32 + xs[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (xs[u] >= 0) ui[idx0 + u * B2_T] = (unsigned long long)rp[u].x | ((unsigned long long)(rp[u].y - rp[u].x) << 32);
    }
    SP_TICK(0)
    for (int i = lo + 1 + tid; i <= hi; i += B2_T) cstart[i] = 0;
    if (tid == 0) *s_live = 0;
    __syncthreads();
    SP_TICK(1)
    const int bs = (F + NBK - 1) / NBK;
    int jdone = 0;
    int L = nU;  // unseen nodes not served yet: a dense list, rewritten by every pass
    int32_t *cx = ux, *nx = ux + sp_cap;
    unsigned long long *ci = ui, *ni = ui + sp_cap;
    auto probe = [&](int w) -> bool {
        const uint32_t bit = 1u << (w & 31);
        if (!(bm[w >> 5] & bit)) return false;
        (void)__hip_atomic_fetch_or(&mark[w >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    };
    for (int j = 0; j < NBK; ++j) {
        if (L <= nU - Un) break;  // only other components' nodes are left
        for (int i = tid; i < W; i += B2_T) bm[i] = 0u;
        lds_barrier();
        const int b0 = lo + j * bs, b1 = min(b0 + bs, hi);
        for (int i0 = b0 + tid; i0 < b1; i0 += 8 * B2_T) {  // the bucket's nodes into the LDS bitmap
            int vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = i0 + u * B2_T < b1 ? ldi(&order[i0 + u * B2_T]) : -1;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (vv[u] >= 0) (void)__hip_atomic_fetch_or(&bm[vv[u] >> 5], 1u << (vv[u] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        lds_barrier();
        SP_TICK(2)
        // Four lanes per list entry, one quad each (the entry's adjacency is one or two cache lines: a wave's load touches ~20
        // lines; with a lane per entry it touched 64, and the CU's L1 path -- 2-3 cycles per line of a fully divergent load --
        // was the whole scan); four entries per group and round in flight, the next round's entries under them (eight: registers spill, 3.4 -> 5 M cycles).
        {
            const int sub = tid & 3, grp = tid >> 2;
            constexpr int NG = B2_T / 4, NE = 4;
            int pxs[NE];
            unsigned long long pis[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int id = grp + e * NG;
                pxs[e] = id < L ? ldi(&cx[id]) : -1;
                pis[e] = id < L ? ldq(&ci[id]) : 0ull;
            }
            for (int idx = grp; idx < L; idx += NE * NG) {
                int xs[NE], dg[NE];
                uint32_t e0[NE];
                unsigned long long is[NE];
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    xs[e] = pxs[e];
                    is[e] = pis[e];
                    e0[e] = (uint32_t)is[e];
                    dg[e] = xs[e] >= 0 ? (int)(is[e] >> 32) : 0;
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int id = idx + (NE + e) * NG;
                    pxs[e] = id < L ? ldi(&cx[id]) : -1;
                    pis[e] = id < L ? ldq(&ci[id]) : 0ull;
                }
                int4u w4[NE];
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    if (4 * sub < dg[e]) w4[e] = *reinterpret_cast<const int4u *>(col + e0[e] + (uint32_t)(4 * sub));
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    int hit = 0;
                    int c = dg[e] - 4 * sub;
                    if (c > 0) {
                        hit |= probe(w4[e].x) ? 1 : 0;
                        if (c > 1) hit |= probe(w4[e].y) ? 1 : 0;
                        if (c > 2) hit |= probe(w4[e].z) ? 1 : 0;
                        if (c > 3) hit |= probe(w4[e].w) ? 1 : 0;
                    }
                    for (int q = 16 + 4 * sub; q < dg[e]; q += 16) {  // (more than 16 neighbours: this lane's further quads)
                        const int4u t4 = *reinterpret_cast<const int4u *>(col + e0[e] + (uint32_t)q);
                        c = dg[e] - q;
                        hit |= probe(t4.x) ? 1 : 0;
                        if (c > 1) hit |= probe(t4.y) ? 1 : 0;
                        if (c > 2) hit |= probe(t4.z) ? 1 : 0;
                        if (c > 3) hit |= probe(t4.w) ? 1 : 0;
                    }
                    hit |= __builtin_amdgcn_mov_dpp(hit, 0xB1, 0xf, 0xf, true);  // quad_perm:[1,0,3,2] (the four lanes of a group are active together)
                    hit |= __builtin_amdgcn_mov_dpp(hit, 0x4E, 0xf, 0xf, true);  // quad_perm:[2,3,0,1]
                    if (sub == 0 && xs[e] >= 0 && !hit) {  // not served: onto the next pass's list
                        const int p = __hip_atomic_fetch_add(s_live, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        nx[p] = xs[e];
                        ni[p] = is[e];
                    }
                }
            }
        }
        __syncthreads();
        L = *s_live;
        { int32_t *const t0 = cx; cx = nx; nx = t0; }
        { unsigned long long *const t1 = ci; ci = ni; ni = t1; }
        jdone = j + 1;
        lds_barrier();  // (everyone has read the count)
        if (tid == 0) *s_live = 0;
        SP_TICK(3)
    }
    // the marks into LDS (and cleared in global memory): S by rank is 0.75 M bit tests in rank order
    for (int i0 = tid; i0 < W; i0 += 8 * B2_T) {
        uint32_t vw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vw[u] = i0 + u * B2_T < W ? ldu(&mark[i0 + u * B2_T]) : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * B2_T < W) {
                bm[i0 + u * B2_T] = vw[u];
                if (vw[u]) mark[i0 + u * B2_T] = 0u;
            }
    }
    lds_barrier();
    SP_TICK(4)
    // S by rank: the level's ranks up to the last bucket used, a contiguous share per wavefront
    const long long cend_l = (long long)lo + (long long)jdone * bs;
    const int cend = cend_l < (long long)hi ? (int)cend_l : hi;
    const int per = ((cend - lo + 64 * B2_WAVES - 1) / (64 * B2_WAVES)) * 64;
    const int r0 = lo + wv * per, r1 = min(r0 + per, cend);
    int scount = 0;
    for (int base = r0; base < r1; base += 512) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + 64 * u + lane;
            v[u] = i < r1 ? ldi(&order[i]) : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool bit = v[u] >= 0 && ((bm[v[u] >> 5] >> (v[u] & 31)) & 1u);
            const unsigned long long m = __ballot(bit);
            if (base + 64 * u < r1) {
                if (lane == 0) smask[(base + 64 * u - lo) >> 6] = m;
                scount += (int)__popcll(m);
            }
        }
    }
    if (lane == 0) wtot[wv] = scount;
    __syncthreads();  // the masks have landed; every mark has been read
    SP_TICK(5)
    int spre = 0, stot = 0;
#pragma unroll
    for (int i = 0; i < B2_WAVES; ++i) {
        const int c = wtot[i];
        if (i < wv) spre += c;
        stot += c;
    }
    // the visited bitmap back into LDS
    for (int i0 = tid; i0 < W; i0 += 8 * B2_T) {
        uint32_t vw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vw[u] = i0 + u * B2_T < W ? ldu(&vis[i0 + u * B2_T]) : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * B2_T < W) bm[i0 + u * B2_T] = vw[u];
    }
    int pos = spre;
    for (int base = r0; base < r1; base += 512) {  // (eight masks requested together)
        unsigned long long ms[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ms[u] = base + 64 * u < r1 ? ldq(&smask[(base + 64 * u - lo) >> 6]) : 0ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if ((ms[u] >> lane) & 1ull) slist[pos + lanes_below(ms[u])] = (unsigned long long)(base + 64 * u + lane);
            pos += (int)__popcll(ms[u]);
        }
    }
    if (tid == 0) { wtot[16] = nU; wtot[17] = jdone; }
    __syncthreads();
    for (int p0 = tid; p0 < stot; p0 += 8 * B2_T) {  // ... and their node ids beside the ranks (what the windows' prefetch starts from)
        int rk[8], vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) rk[u] = p0 + u * B2_T < stot ? (int)ldq(&slist[p0 + u * B2_T]) : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = rk[u] >= 0 ? ldi(&order[rk[u]]) : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (rk[u] >= 0) slist[p0 + u * B2_T] = (unsigned long long)(uint32_t)rk[u] | ((unsigned long long)(uint32_t)vv[u] << 32);
    }
    __syncthreads();
    SP_TICK(6)
#undef SP_TICK
    return stot;
}

// INSTR: the phase clocks (GG_BFS_PROFILE), event counts and ablation switches (GG_BFS_EXPERIMENT) live in a second instance only.
// LAZY (round 6): the tree is built exactly only through the last level whose expansion is KNOWN to fit the slot's node limit
// (nodes so far + adjacency entries of the level's nodes <= limit: every entry could append a node); the children lists of that
// level and below are resolved by the walks that need them (walk_sample.hip, "LAZY").  The kernel then leaves what resolution
// needs: the visited set of the exact levels with a popcount index, and the BFS rank of every member by that index.
template <bool LDS_BM, bool INSTR, bool LAZY = false>
__global__ __launch_bounds__(B2_T) void bfs_order2_kernel(const BfsArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_bm[];  // [bm_words, rounded up to 16] when LDS_BM
    __shared__ uint32_t e0s[B2_NB];                   // first CSR entry of each window node (of the segment, for a partial node)
    __shared__ uint16_t soff[B2_NB + 2];              // first quad of each window node; [nb] = quads of the window
    __shared__ uint16_t degs[B2_NB];                  // adjacency entries of each window node (of the segment)
    __shared__ uint32_t scratch[B2_SLOTS / 2];        // quad -> window node (uint16 x 4 096)
    __shared__ int32_t tkey[B2_HASH], tpos[B2_HASH];  // duplicates of the window: node -> smallest position (open addressing; -1 / INT_MAX when free)
    __shared__ uint32_t winbits[B2_WORDS + 1];        // positions that append a node ([B2_WORDS] stays 0)
    __shared__ int32_t wpre[B2_WORDS + 1];            // exclusive popcount prefix over those words; [B2_WORDS] = total
    __shared__ int32_t wtot[2][B2_WAVES];
    __shared__ int32_t s_root, s_any, s_Lcount, s_cap, s_max, s_deg, s_live;
    __shared__ uint32_t s_e0;
    __shared__ long long s_off;
    uint16_t *const emap = reinterpret_cast<uint16_t *>(scratch);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t *const bm = LDS_BM ? lds_bm : a.gbitmap + (size_t)blockIdx.x * a.bm_words;
    uint32_t *const gkey = a.gkey + (size_t)blockIdx.x * a.n_node;
    // SPARSE LEVEL scratch of this workgroup (LDS_BM only)
    unsigned long long *const slist = a.sp_s + (size_t)blockIdx.x * a.n_node;

    auto seen = [&](int w) -> bool {
        if (LDS_BM) return (bm[w >> 5] >> (w & 31)) & 1u;
        return (__hip_atomic_load(&bm[w >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (w & 31)) & 1u;
    };

    for (;;) {
        if (tid == 0) {
            s_root = (int)atomicAdd(a.ticket, 1u);
            s_any = 0;
            s_Lcount = 0;
            s_max = 0;
        }
        for (int i = tid; i <= B2_WORDS; i += B2_T) winbits[i] = 0u;
        for (int i = tid; i < B2_HASH; i += B2_T) { tkey[i] = -1; tpos[i] = 0x7fffffff; }
        __syncthreads();
        const int r = s_root;
        if (r >= a.n_roots) return;
        const int root = a.roots[r];
        const int slot = a.slot_ids ? a.slot_ids[r] : r;
        int32_t *const order = LAZY ? a.scr_order + (size_t)blockIdx.x * a.scr_cap : a.order + a.base[r];
        int32_t *const cstart = LAZY ? a.scr_cstart + (size_t)blockIdx.x * (a.scr_cap + 1) : a.cstart + a.base[r] + slot;
        int32_t *const tedge = LAZY ? a.scr_edge + (size_t)blockIdx.x * a.scr_cap : a.edge + a.base[r];
        const int expect = a.expect ? a.expect[r] : (int)(a.base[r + 1] - a.base[r]);
        const int limit = LAZY ? a.lz_limit[r] : expect;
        for (int i = tid; i < a.bm_words; i += B2_T) bm[i] = 0u;
        __syncthreads();
        if (tid == 0) {
            bm[root >> 5] = 1u << (root & 31);
            order[0] = root;
            tedge[0] = -1;
            cstart[0] = 1;
        }
        __syncthreads();

        int head = 0, tail = 1, level_end = 1, depth = 0;
        int fenced = 1;          // queue entries below this index were written before the last full fence
        unsigned long long st_win = 0, st_fence = 0, st_cand = 0, st_slots = 0, st_empty = 0, st_dup = 0, st_key = 0;  // GG_BFS_PROFILE
        unsigned long long st_sp = 0, st_spu = 0, st_sps = 0, st_spp = 0;  // sparse levels, unseen nodes listed, candidate fathers, passes
        unsigned long long sph[7] = {0, 0, 0, 0, 0, 0, 0};  // phases of the father search
        unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // wave 0's clock per phase: setup, scan, claim, duplicates, prefix, append; [8] sparse-level build, [9] its cstart fill
        long long tprev = INSTR ? (long long)clock64() : 0;
#define B2_TICK(k)                                  \
    if (INSTR) {                                    \
        const long long tn = (long long)clock64();  \
        ph[k] += (unsigned long long)(tn - tprev);  \
        tprev = tn;                                 \
    }
        int seg_a = 0;           // > 0: the node at `head` is being scanned in segments, this many entries are done
        int pf_q = -1;           // queue index whose node this thread has prefetched
        // THE PREFETCH of the next window's nodes is a chain node id -> row offsets -> degree (a sparse level's list carries the id
        // beside the rank).  The compiler waits for a load where its value is first USED: requested and used in one place (rounds
        // 2-4) the chain cost every window two memory round trips, ~5 k of its ~37 k cycles.  So each link is requested at one point
        // of the window and used at a later one, where it has long arrived: the id behind the window choice, the row offsets behind
        // the scan barrier, the degree in front of the append (claim and prefix touch LDS only; the append's stores come later).
        // This only works with the explicit vmcnt(0) behind the scan barrier: without it the compiler's wait insertion carries
        // "maybe still in flight" marks of the scan's loads into the loops of the claim and waits there for EVERYTHING in flight.
        // pf_stage: 2 = the id is requested, 3 = the row offsets, 0 = pf_e0 / pf_deg hold them.
        int pf_v = 0, pf_stage = 0;
        unsigned long long pf_sl = 0ull;
        uint2u pf_rp = {0u, 0u};
        uint32_t pf_e0 = 0;
        int pf_deg = 0;
        // SPARSE LEVEL.  Late levels of a small-world graph pop most of the graph to discover a few nodes (1M / 10M bench
        // graph: 0.5-0.85 M nodes, 10-15 M adjacency entries, for 0.3-200 k unseen nodes).  Only a FATHER -- the first queue
        // node adjacent to some unseen node -- can append anything, and popping any superset S of the fathers, in queue order,
        // through the same windows appends the same nodes at the same edges: a node of S that is nobody's first neighbour finds
        // its unseen neighbours taken.  S is found from the unseen side: the level's ranks are cut into buckets; per bucket the
        // LDS bitmap holds the bucket's nodes (the visited bitmap waits in global memory), every unseen node not yet served scans
        // its adjacency and marks the neighbours it has in the bucket -- its father is among the marks of the FIRST bucket it
        // hits, and it is served.  The marked nodes in rank order are S; nodes outside S have empty child ranges (cstart: the
        // running maximum over the level).  Cost: the unseen nodes' adjacencies ~2 times + the adjacencies of S, instead of the
        // whole level's.  `head` indexes S while smode is set.
        bool smode = false;
        int s_cnt = 0, s_lo = 0, s_hi = 0;
        int my_rk = 0;           // queue rank of the window node this thread holds
        auto fill_level = [&](int lo, int hi) {  // cstart[lo + 1 .. hi]: zeros (nodes outside S, fathers not reached) -> the running maximum
            // (a contiguous share per wavefront, eight groups of 64 entries requested together: one group per iteration was a memory
            // round trip + six LDS permutes per 64 entries, ~3 M cycles for a level of 0.85 M nodes -- and invisible in the phase
            // clocks when the early exit calls it)
            __syncthreads();
            const int per = ((hi - lo + 64 * B2_WAVES - 1) / (64 * B2_WAVES)) * 64;
            const int r0 = lo + 1 + wv * per, r1 = min(r0 + per, hi + 1);
            int mx = 0;
            for (int base = r0; base < r1; base += 512) {
                int vals[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) vals[u] = base + 64 * u + lane < r1 ? ldi(&cstart[base + 64 * u + lane]) : 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) mx = max(mx, vals[u]);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
            if (lane == 0) wtot[0][wv] = mx;
            lds_barrier();
            int carry = hi;  // the level's children start where the level ends
#pragma unroll
            for (int i = 0; i < B2_WAVES; ++i)
                if (i < wv) carry = max(carry, wtot[0][i]);
            for (int base = r0; base < r1; base += 512) {
                int vals[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) vals[u] = base + 64 * u + lane < r1 ? ldi(&cstart[base + 64 * u + lane]) : 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int val = vals[u];  // inclusive running maximum over the wavefront (values >= 0: 0 is the identity), then the carry
                    val = max(val, __builtin_amdgcn_update_dpp(0, val, 0x111, 0xf, 0xf, false));
                    val = max(val, __builtin_amdgcn_update_dpp(0, val, 0x112, 0xf, 0xf, false));
                    val = max(val, __builtin_amdgcn_update_dpp(0, val, 0x114, 0xf, 0xf, false));
                    val = max(val, __builtin_amdgcn_update_dpp(0, val, 0x118, 0xf, 0xf, false));
                    val = max(val, __builtin_amdgcn_update_dpp(0, val, 0x142, 0xa, 0xf, false));
                    val = max(val, __builtin_amdgcn_update_dpp(0, val, 0x143, 0xc, 0xf, false));
                    val = max(val, carry);
                    if (base + 64 * u + lane < r1) cstart[base + 64 * u + lane] = val;
                    carry = __builtin_amdgcn_readlane(val, 63);
                }
            }
            __syncthreads();
        };
        // The window loop proper is the inner loop; the cold steps between runs of windows -- the father search of a sparse
        // level (a call), the cstart fill behind one, the early exit -- sit in the outer loop, where their registers (and the call's
        // save area) do not reach into the windows.
        int why = 0;  // why the window loop stopped: 0 queue empty / error, 1 a sparse level starts, 2 S is through, 3 every node is on the queue
        for (;;) {
          why = 0;
          for (;;) {
            if (!smode) {
                if (head >= tail) break;
                if (head == level_end) {  // the next level starts: everything up to `tail` belongs to it
                    level_end = tail;
                    ++depth;
                    const int F = tail - head, Un = expect - tail;
                    if (LAZY && a.lz_pool[r] > 0) {  // (a lazy candidate: the host gave it a pool)
                        // may this level be expanded?  Only if even an append per adjacency entry fits the limit.
                        __syncthreads();  // (the level's queue entries have landed)
                        fenced = tail;
                        uint32_t dsum = 0;
                        for (int i0 = head + tid; i0 < tail; i0 += 8 * B2_T) {
                            int vv[8];
                            uint2u rp[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) vv[u] = i0 + u * B2_T < tail ? ldi(&order[i0 + u * B2_T]) : -1;
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (vv[u] >= 0) rp[u] = *reinterpret_cast<const uint2u *>(a.rowptr32 + vv[u]);
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (vv[u] >= 0) dsum += rp[u].y - rp[u].x;
                        }
                        dsum = (uint32_t)wave_incl_scan((int)dsum, lane);
                        if (lane == 63) wtot[0][wv] = (int32_t)dsum;
                        lds_barrier();
                        unsigned long long D = 0;
#pragma unroll
                        for (int i = 0; i < B2_WAVES; ++i) D += (unsigned long long)(uint32_t)wtot[0][i];
                        lds_barrier();  // (wtot is reused by the windows)
                        // (the first lz_easy levels may take up to lz_limit_easy: a tree that stops before its third level sends its
                        // walks through two levels of hub-heavy resolution and, mostly, to a whole rebuild)
                        if ((unsigned long long)tail + D > (unsigned long long)(depth <= a.lz_easy ? max(limit, a.lz_limit_easy) : limit)) {
                            why = 4;
                            break;
                        }
                    }
                    if (LDS_BM && a.sp_k >= 0 && Un > 0 && F >= a.sp_min && (long long)Un * a.sp_k <= (long long)F) {
                        why = 1;
                        break;
                    }
                }
            } else if (head >= s_cnt) {  // S is through: the level is complete
                why = 2;
                break;
            }
            // queue entries are read back from global memory (this window's nodes, the next window's prefetch): entries
            // written since the last fence must have landed first
            if (!smode && fenced < tail && head + 2 * B2_NB > fenced) {
                __syncthreads();
                fenced = tail;
                if (INSTR) ++st_fence;
            }
            int nb = 0;      // complete nodes of this window; 0 = one node, entries [seg_a, seg_a + seg_len)
            int seg_len = 0;
            if (seg_a == 0) {
                const int navail = min(B2_NB, (smode ? s_cnt : level_end) - head);
                uint32_t e0 = 0;
                int deg = 0;
                if (tid < navail) {
                    if (pf_q == head + tid) {
                        if (pf_stage == 2) {  // (only behind a mode switch)
                            pf_rp = *reinterpret_cast<const uint2u *>(a.rowptr32 + (smode ? (int)(pf_sl >> 32) : pf_v));
                            pf_stage = 3;
                        }
                        if (pf_stage == 3) {  // (a window without a candidate ends before the chain does)
                            pf_e0 = pf_rp.x;
                            pf_deg = (int)(pf_rp.y - pf_rp.x);
                            pf_stage = 0;
                        }
                        e0 = pf_e0;
                        deg = pf_deg;
                        my_rk = smode ? (int)(uint32_t)pf_sl : head + tid;
                    } else {
                        const unsigned long long sl = smode ? ldq(&slist[head + tid]) : 0ull;
                        my_rk = smode ? (int)(uint32_t)sl : head + tid;
                        const int v = smode ? (int)(sl >> 32) : ldi(&order[my_rk]);
                        const uint2u rp = *reinterpret_cast<const uint2u *>(a.rowptr32 + v);
                        e0 = rp.x;
                        deg = (int)(rp.y - rp.x);
                    }
                }
                const int q = min((deg + 3) >> 2, B2_SLOTS + 1);  // (a node above the window size ends the window whatever its size)
                const int incq = wave_incl_scan(q, lane);
                if (lane == 63) wtot[0][wv] = incq;
                if (tid == 0) { s_cap = 0; s_e0 = e0; s_deg = deg; }
                lds_barrier();
                B2_TICK(0)
                int preq = 0;
#pragma unroll
                for (int i = 0; i < B2_WAVES; ++i)
                    if (i < wv) preq += wtot[0][i];
                const int inclq = preq + incq, excq = inclq - q;
                // window = the longest prefix of the available nodes whose quads fit
                const unsigned long long okb = __ballot(tid < navail && inclq <= B2_SLOTS);
                if (lane == 0 && okb) atomicAdd(&s_cap, (int)__popcll(okb));
                lds_barrier();
                B2_TICK(1)
                nb = s_cap;
                if (nb > 0) {
                    if (tid < nb) {
                        e0s[tid] = e0;
                        soff[tid] = (uint16_t)excq;
                        degs[tid] = (uint16_t)deg;
                        if (tid == nb - 1) soff[nb] = (uint16_t)inclq;
                        if (q <= B2_NARROW)
                            for (int k = 0; k < q; ++k) emap[excq + k] = (uint16_t)tid;
                    }
                    unsigned long long wide = __ballot(tid < nb && q > B2_NARROW);  // their quads are filled by the whole wavefront
                    while (wide) {
                        const int l = __ffsll((long long)wide) - 1;
                        wide &= wide - 1;
                        const int qq = __builtin_amdgcn_readlane(q, l), ss = __builtin_amdgcn_readlane(excq, l);
                        for (int k = lane; k < qq; k += 64) emap[ss + k] = (uint16_t)((wv << 6) + l);
                    }
                    // the next window's nodes: the first link of the prefetch chain (queue entries below `fenced` have landed)
                    const int q2 = head + nb + tid;
                    pf_q = -1;
                    pf_stage = 0;
                    if (q2 < (smode ? s_cnt : fenced)) {
                        if (smode) pf_sl = ldq(&slist[q2]);
                        else pf_v = ldi(&order[q2]);
                        pf_q = q2;
                        pf_stage = 2;
                    }
                }
            }
            if (nb == 0) {  // one node alone (it exceeds the window, or is being continued): the segment [seg_a, seg_a + seg_len)
                seg_len = min(s_deg - seg_a, B2_POS);
                const int qs = (seg_len + 3) >> 2;
                for (int k = tid; k < qs; k += B2_T) emap[k] = 0;
                if (tid == 0) {
                    e0s[0] = s_e0 + (uint32_t)seg_a;
                    soff[0] = 0;
                    soff[1] = (uint16_t)qs;
                    degs[0] = (uint16_t)seg_len;
                }
            }
            lds_barrier();
            B2_TICK(2)
            const int nbn = nb > 0 ? nb : 1;  // window nodes
            const int T = (int)soff[nbn];
            if (INSTR) ++st_win;
            if (INSTR) st_slots += (unsigned long long)T;

            // ---------------- SCAN: quads tid, tid + 1024, ...; position of entry k of quad s = 4 s + k (stream order)
            int4u w4[4];
            uint32_t we[4];     // CSR index of the quad's first entry
            uint32_t cand[4];   // entries whose target is unseen
            // Straight-line on purpose: quad -> node -> CSR index for all four quads, then the four loads back to back, UNCONDITIONALLY
            // (a thread without a quad re-reads entry 0).  With the load inside `if (s < T)` the compiler merged the loaded registers
            // with the not-loaded case behind every branch -- a copy, hence an s_waitcnt vmcnt(0) behind EACH load: a thread's four
            // loads were four memory round trips per window instead of one (rounds 3-4; found in the disassembly).
            bool okq[4];
            uint32_t cnt4[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int s = tid + it * B2_T;
                okq[it] = s < T;
                const int i = (int)emap[okq[it] ? s : 0];
                const int j0 = okq[it] ? 4 * (s - (int)soff[i]) : 0;
                cnt4[it] = (uint32_t)((int)degs[i] - j0);  // >= 1 for a real quad
                we[it] = okq[it] ? e0s[i] + (uint32_t)j0 : 0u;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (INSTR && (a.exp & 2)) {  // (ablation: computed targets instead of the adjacency loads)
                    const uint32_t hx = we[it] * 2654435761u;
                    w4[it].x = (int)(hx % (uint32_t)a.n_node); w4[it].y = (int)((hx >> 3) % (uint32_t)a.n_node);
                    w4[it].z = (int)((hx >> 5) % (uint32_t)a.n_node); w4[it].w = (int)((hx >> 7) % (uint32_t)a.n_node);
                } else {
                    w4[it] = *reinterpret_cast<const int4u *>(a.col + we[it]);
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                cand[it] = !okq[it] ? 0u : cnt4[it] >= 4u ? 15u : ((1u << cnt4[it]) - 1u);  // valid entries for now
                if (!okq[it]) we[it] = 0xffffffffu;  // (no quad)
            }
            uint32_t anyc = 0;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (we[it] != 0xffffffffu) {
                    // four independent LDS reads per quad (entries behind the node's end test its first target again: a valid id)
                    const uint32_t valid = cand[it];
                    const int wx = w4[it].x, wy = (valid & 2u) ? w4[it].y : wx, wz = (valid & 4u) ? w4[it].z : wx, ww = (valid & 8u) ? w4[it].w : wx;
                    const uint32_t sx = seen(wx) ? 1u : 0u, sy = seen(wy) ? 2u : 0u, sz = seen(wz) ? 4u : 0u, sw = seen(ww) ? 8u : 0u;
                    const uint32_t c = valid & ~(sx | sy | sz | sw);
                    cand[it] = c;
                    anyc |= c;
                }
            }
            if (__ballot(anyc != 0u) && lane == 0) s_any = 1;
            B2_TICK(3)
            lds_barrier();  // every test before any set: a later edge must not hide an earlier one
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), stated where the compiler sees it (free: the scan has just used its loads): see "THE PREFETCH"
            if (pf_stage == 2) {  // second link: the row offsets (the id arrived under the scan)
                pf_rp = *reinterpret_cast<const uint2u *>(a.rowptr32 + (smode ? (int)(pf_sl >> 32) : pf_v));
                pf_stage = 3;
            }
            B2_TICK(4)

            if (!s_any) {
                // nothing new: every node that ends in this window has its children end at `tail`
                if (INSTR) ++st_empty;
                if (nb > 0) {
                    if (tid < nb) cstart[my_rk + 1] = tail;
                    head += nb;
                } else {
                    seg_a += seg_len;
                    if (seg_a == s_deg) {
                        if (tid == 0) cstart[my_rk + 1] = tail;
                        head += 1;
                        seg_a = 0;
                    }
                }
            } else {
                // ---------------- CLAIM, by the thread that scanned
                uint32_t won[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    won[it] = 0u;
                    uint32_t c = cand[it];
                    while (c) {
                        const int k = __ffs((int)c) - 1;
                        c &= c - 1u;
                        const int w = k == 0 ? w4[it].x : k == 1 ? w4[it].y : k == 2 ? w4[it].z : w4[it].w;
                        const uint32_t bit = 1u << (w & 31);
                        const uint32_t old = atomicOr(&bm[w >> 5], bit);
                        if (old & bit) {  // another edge of this window reaches the same new node: smallest such position per node
                            const int li = atomicAdd(&s_Lcount, 1);
                            if (li < B2_DCAP) {
                                uint32_t h = ((uint32_t)w * 2654435761u) >> 22;
                                for (;;) {
                                    const int k0 = atomicCAS(&tkey[h], -1, w);
                                    if (k0 == -1 || k0 == w) {
                                        atomicMin(&tpos[h], 4 * (tid + it * B2_T) + k);
                                        break;
                                    }
                                    h = (h + 1) & (B2_HASH - 1);
                                }
                            }
                        } else {
                            won[it] |= 1u << k;
                        }
                    }
                }
                lds_barrier();
                B2_TICK(5)
                const int nL = s_Lcount;
                if (nL <= B2_DCAP) {
                    // the sequential BFS appends a node at the FIRST edge that reaches it: smallest position among the hardware
                    // winner's and the duplicates'
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        uint32_t c = won[it];
                        while (c) {
                            const int k = __ffs((int)c) - 1;
                            c &= c - 1u;
                            int eff = 4 * (tid + it * B2_T) + k;
                            if (nL > 0 && !(INSTR && (a.exp & 16))) {
                                const int w = k == 0 ? w4[it].x : k == 1 ? w4[it].y : k == 2 ? w4[it].z : w4[it].w;
                                uint32_t h = ((uint32_t)w * 2654435761u) >> 22;
                                for (;;) {
                                    const int k0 = tkey[h];
                                    if (k0 == w) { eff = min(eff, tpos[h]); break; }
                                    if (k0 == -1) break;
                                    h = (h + 1) & (B2_HASH - 1);
                                }
                            }
                            atomicOr(&winbits[eff >> 5], 1u << (eff & 31));
                        }
                    }
                } else {
                    // too many duplicates for the list: smallest position per node through the key array (all-ones between uses)
                    if (INSTR) ++st_key;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        uint32_t c = cand[it];
                        while (c) {
                            const int k = __ffs((int)c) - 1;
                            c &= c - 1u;
                            const int w = k == 0 ? w4[it].x : k == 1 ? w4[it].y : k == 2 ? w4[it].z : w4[it].w;
                            atomicMin(&gkey[w], (uint32_t)(4 * (tid + it * B2_T) + k));
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        uint32_t c = cand[it];
                        while (c) {
                            const int k = __ffs((int)c) - 1;
                            c &= c - 1u;
                            const int w = k == 0 ? w4[it].x : k == 1 ? w4[it].y : k == 2 ? w4[it].z : w4[it].w;
                            const uint32_t p = (uint32_t)(4 * (tid + it * B2_T) + k);
                            if (atomicMin(&gkey[w], 0xFFFFFFFFu) == p) atomicOr(&winbits[p >> 5], 1u << (p & 31));
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        uint32_t c = cand[it];
                        while (c) {
                            const int k = __ffs((int)c) - 1;
                            c &= c - 1u;
                            const int w = k == 0 ? w4[it].x : k == 1 ? w4[it].y : k == 2 ? w4[it].z : w4[it].w;
                            __hip_atomic_store(&gkey[w], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
                if (INSTR && nL > 0) ++st_dup;
                lds_barrier();
                // place of every appending position in stream order = popcount prefix over the mask
                const int wcnt = tid < B2_WORDS ? (int)__popc(winbits[tid]) : 0;
                const int winc = wave_incl_scan(wcnt, lane);
                if (lane == 63) wtot[1][wv] = winc;
                lds_barrier();
                int wpr = 0, total = 0;
#pragma unroll
                for (int i = 0; i < B2_WORDS / 64; ++i) {
                    const int c = wtot[1][i];
                    if (i < wv) wpr += c;
                    total += c;
                }
                if (tid < B2_WORDS) wpre[tid] = wpr + winc - wcnt;
                if (tid == 0) wpre[B2_WORDS] = total;
                lds_barrier();
                if (pf_stage == 3) {  // third link: the degree (the offsets arrived under claim and prefix; the append's stores are not waited for)
                    pf_e0 = pf_rp.x;
                    pf_deg = (int)(pf_rp.y - pf_rp.x);
                    pf_stage = 0;
                }
                B2_TICK(6)
                if (INSTR) st_cand += (unsigned long long)total;
                // append: every candidate (hardware winner or duplicate) whose position carries the bit
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    uint32_t c = cand[it];
                    while (c) {
                        const int k = __ffs((int)c) - 1;
                        c &= c - 1u;
                        const int p = 4 * (tid + it * B2_T) + k;
                        const uint32_t word = winbits[p >> 5];
                        if ((word >> (p & 31)) & 1u) {
                            const int rank = tail + wpre[p >> 5] + (int)__popc(word & ((1u << (p & 31)) - 1u));
                            if (rank < expect) {
                                order[rank] = k == 0 ? w4[it].x : k == 1 ? w4[it].y : k == 2 ? w4[it].z : w4[it].w;
                                if (INSTR && (a.exp & 64)) __builtin_nontemporal_store((int32_t)(we[it] + (uint32_t)k), &tedge[rank]);  // (ablation: streaming stores for the array nobody reads back)
                                else tedge[rank] = (int32_t)(we[it] + (uint32_t)k);
                            }
                        }
                    }
                }
                // children of window node i end behind the appends of the positions below its last quad's end
                if (nb > 0) {
                    if (tid < nb) {
                        const int P = 4 * (int)soff[tid + 1];
                        cstart[my_rk + 1] = tail + wpre[P >> 5] + (int)__popc(winbits[P >> 5] & ((1u << (P & 31)) - 1u));
                    }
                    tail += total;
                    head += nb;
                } else {
                    tail += total;
                    seg_a += seg_len;
                    if (seg_a == s_deg) {
                        if (tid == 0) cstart[my_rk + 1] = tail;
                        head += 1;
                        seg_a = 0;
                    }
                }
                lds_barrier();  // everyone has read the mask and the prefix
                if (nL > 0) { tkey[tid] = -1; tpos[tid] = 0x7fffffff; }  // (B2_HASH == B2_T entries: one each)
                if (tid < B2_WORDS) winbits[tid] = 0u;
                if (tid == 0) { s_any = 0; s_Lcount = 0; }
                if (tail > expect) break;  // more nodes than the component sweep promised: the graph is not the one set (uniform)
            }
            lds_barrier();
            B2_TICK(7)
            // Every node of the root's component is on the queue (its size is known from the component sweep): no edge of the
            // nodes still waiting can discover anything -- they are leaves of the tree (empty child ranges at the end of the queue).
            if (tail == expect && (smode || head < tail) && !(INSTR && (a.exp & 8))) {
                why = 3;
                break;
            }
          }
          if (why == 0 || why == 4) break;
          if (why == 1) {
              __syncthreads();  // the level's queue entries have landed: the search reads them back
              fenced = tail;
              const int got = sparse_fathers<INSTR>(sph, (lds_u32_t *)bm, (lds_i32_t *)&wtot[0][0], (lds_i32_t *)&s_live, a.rowptr32, a.col, order, cstart,
                                             a.sp_ux + (size_t)blockIdx.x * a.sp_cap * 2, a.sp_ui + (size_t)blockIdx.x * a.sp_cap * 2, slist,
                                             a.sp_vis + (size_t)blockIdx.x * a.bm_words, a.sp_mark + (size_t)blockIdx.x * a.bm_words,
                                             a.sp_mask + (size_t)blockIdx.x * ((size_t)a.n_node / 64 + 32), a.n_node, a.bm_words, a.sp_cap, a.sp_buckets,
                                             head, tail, expect - tail);
              if (got >= 0) {
                  smode = true;
                  s_cnt = got;
                  s_lo = head;
                  s_hi = tail;
                  head = 0;
                  pf_q = -1;
                  if (INSTR) {
                      ++st_sp;
                      st_spu += (unsigned long long)wtot[1][0];
                      st_spp += (unsigned long long)wtot[1][1];
                      st_sps += (unsigned long long)got;
                  }
              }  // (else: the level is popped whole -- level_end has moved on, the test above does not fire again)
              B2_TICK(8)
              continue;
          }
          if (why == 2) {
              fill_level(s_lo, s_hi);
              B2_TICK(9)
              head = s_hi;
              smode = false;
              pf_q = -1;
              continue;
          }
          // why == 3.  Every node of the root's component is on the queue (its size is known from the component sweep): no edge of
          // the nodes still waiting can discover anything -- they are leaves of the tree (empty child ranges at the end of the queue).
          int first_open = head;  // first queue rank whose children may not be final
          if (smode) {            // fathers not reached (and the one a segmented scan is inside) end at `expect` like everything behind them
              fill_level(s_lo, s_hi);
              first_open = head < s_cnt ? (int)ldq(&slist[head]) : s_hi;
              smode = false;
          }
          for (int i = first_open + tid; i < expect; i += B2_T) cstart[i + 1] = expect;
          if (level_end < tail) ++depth;  // the nodes behind level_end are one level deeper than the one being popped
          break;
        }
        __syncthreads();  // (all queue / cstart stores have landed: the child-count sweep below reads cstart)
        if (INSTR && a.prof && tid == 0) {
            atomicAdd(&a.prof[0], st_win); atomicAdd(&a.prof[1], st_fence); atomicAdd(&a.prof[2], st_cand); atomicAdd(&a.prof[3], st_slots);
            atomicAdd(&a.prof[4], st_empty); atomicAdd(&a.prof[5], st_dup); atomicAdd(&a.prof[6], st_key); atomicAdd(&a.prof[7], st_sp);
            for (int k = 0; k < 10; ++k) atomicAdd(&a.prof[8 + k], ph[k]);
            atomicAdd(&a.prof[18], st_spu); atomicAdd(&a.prof[19], st_sps); atomicAdd(&a.prof[20], st_spp);
            for (int k = 0; k < 7; ++k) atomicAdd(&a.prof[21 + k], sph[k]);
        }
#undef B2_TICK
        const bool lazy_stop = LAZY && why == 4;  // ranks [head, tail) are level `depth`: exact queue entries without children lists
        if (LAZY) {
            // the slot's place in the tree arrays: exact ranks (+ pool), built lists; then the copy out of the scratch tree
            const int n_keep = lazy_stop ? tail : expect, n_built = lazy_stop ? head : expect;
            const int seg = n_keep + 1 + (lazy_stop ? a.lz_pool[r] : 0);  // (+ 1: the row of first-child ranks has one entry more than ranks)
            const bool complete = tail == expect || lazy_stop;
            if (tid == 0) {
                long long off = complete ? (long long)atomicAdd(a.lz_alloc, (unsigned long long)seg) : -1;
                if (off >= 0 && off + seg > a.lz_alloc_end) off = -1;
                s_off = off;
            }
            __syncthreads();  // (also: every queue / cstart store of the scratch tree has landed)
            const long long off = s_off;
            if (off >= 0 && !(a.lz_ablate & 2)) {
                int32_t *const o_out = a.order + off, *const e_out = a.edge + off, *const c_out = a.cstart + off;
                for (int i0 = tid; i0 < n_keep; i0 += 4 * B2_T) {
                    int vo[4], ve[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u * B2_T;
                        vo[u] = i < n_keep ? ldi(&order[i]) : 0;
                        ve[u] = i < n_keep ? ldi(&tedge[i]) : 0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u * B2_T;
                        if (i < n_keep) { o_out[i] = vo[u]; e_out[i] = ve[u]; }
                    }
                }
                for (int i = tid; i <= n_built; i += B2_T) c_out[i] = ldi(&cstart[i]);
            }
            if (lazy_stop && off >= 0 && !(a.lz_ablate & 4)) {
                // the visited set of the exact levels with its popcount index: {word, members below the word}
                uint2 *const zb = a.lz_bm + (size_t)slot * a.bm_words;
                int32_t *const zr = a.lz_rank + off;
                const int W = a.bm_words, wpt = (W + B2_T - 1) / B2_T;
                const int w0 = min(tid * wpt, W), w1 = min(w0 + wpt, W);
                int cnt = 0;
                for (int i = w0; i < w1; ++i) cnt += (int)__popc(LDS_BM ? bm[i] : ldu(&bm[i]));
                const int inc = wave_incl_scan(cnt, lane);
                if (lane == 63) wtot[0][wv] = inc;
                lds_barrier();
                int run = inc - cnt;
#pragma unroll
                for (int i = 0; i < B2_WAVES; ++i)
                    if (i < wv) run += wtot[0][i];
                // (members below every 16th word stay in LDS -- the windows' quad map is free now -- for the ranks' indices below)
                const bool coarse = LDS_BM && (W + 15) / 16 <= B2_SLOTS / 2 && !(a.lz_ablate & 8);
                for (int i = w0; i < w1; ++i) {
                    const uint32_t wd = LDS_BM ? bm[i] : ldu(&bm[i]);
                    zb[i] = make_uint2(wd, (uint32_t)run);
                    if (coarse && (i & 15) == 0) scratch[i >> 4] = (uint32_t)run;
                    run += (int)__popc(wd);
                }
                __syncthreads();  // (the index has landed: the ranks below read it back)
                // rank of every member by its index.  A member's index is as good as random: written straight to memory every 4-byte
                // store is a read-modify-write of its own line -- 25 of the kernel's 46 ms per 16 384 roots (timing ablation,
                // GG_LZ_ABLATE=1; the PMC counters charge the kernel 17 MB of HBM traffic per root).  So: the indices once, in queue
                // order, into the scratch row of first-child ranks (copied out above: free); then the index space through the LDS
                // -- the bitmap's words are in `zb` now -- in blocks of W entries, each block one coalesced pass over the indices
                // and one coalesced flush.  (Looking the index up again in every block pass instead: 61 ms.)
                int32_t *const idxs = cstart;
                for (int i0 = tid; i0 < tail && !(a.lz_ablate & 1); i0 += 8 * B2_T) {
                    int vv[8];
                    unsigned long long wq[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) vv[u] = i0 + u * B2_T < tail ? ldi(&order[i0 + u * B2_T]) : -1;
                    if (coarse) {
                        // the index from LDS: members below the node's 16-word block + popcounts of the block's words in front of its
                        // own (four 16-byte reads).  (From the index in global memory this was 88 k random 8-byte reads per root --
                        // 2.8 MB of sectors, more than everything else the block moves.)
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (vv[u] < 0) continue;
                            const int w = vv[u] >> 5, nb = w & 15;
                            const uint4 *const b4 = reinterpret_cast<const uint4 *>(bm + (w & ~15));
                            const uint4 q0 = b4[0], q1 = b4[1], q2 = b4[2], q3 = b4[3];
                            const uint32_t ww[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
                            int idx = (int)scratch[w >> 4];
                            uint32_t wd = 0u;
#pragma unroll
                            for (int k = 0; k < 16; ++k) {
                                idx += k < nb ? (int)__popc(ww[k]) : 0;
                                wd = k == nb ? ww[k] : wd;
                            }
                            idxs[i0 + u * B2_T] = idx + (int)__popc(wd & ((1u << (vv[u] & 31)) - 1u));
                        }
                        continue;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) wq[u] = vv[u] >= 0 ? ldq(reinterpret_cast<const unsigned long long *>(zb + (vv[u] >> 5))) : 0ull;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (vv[u] < 0) continue;
                        const int idx = (int)(wq[u] >> 32) + (int)__popc((uint32_t)wq[u] & ((1u << (vv[u] & 31)) - 1u));
                        if (LDS_BM) idxs[i0 + u * B2_T] = idx;
                        else zr[idx] = i0 + u * B2_T;
                    }
                }
                if (LDS_BM && !(a.lz_ablate & 1)) {
                    __syncthreads();
                    for (int blk0 = 0; blk0 < tail; blk0 += W) {
                        for (int i0 = tid; i0 < tail; i0 += 8 * B2_T) {
                            int ix[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) ix[u] = i0 + u * B2_T < tail ? ldi(&idxs[i0 + u * B2_T]) : -1;
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (ix[u] >= blk0 && ix[u] < blk0 + W) bm[ix[u] - blk0] = (uint32_t)(i0 + u * B2_T);
                        }
                        __syncthreads();
                        for (int j = tid; j < min(W, tail - blk0); j += B2_T) zr[blk0 + j] = (int32_t)bm[j];
                        __syncthreads();
                    }
                }
            }
            if (tid == 0) {
                if (off >= 0) {
                    a.base_w[slot] = off;
                    a.lz_info[slot] = lazy_stop ? make_int4(head, tail, depth, seg) : make_int4(expect, expect, depth, seg);
                    a.lz_cursor[slot] = lazy_stop ? tail : expect;
                    if (lazy_stop) { atomicMax(&a.stats[3], 1024 - depth); atomicMax(&a.stats[5], depth); }  // (smallest lazy level of the launch: 1024 - stats[3]; deepest: stats[5])
                } else if (complete) {
                    // no room among the segments: the slot gets its whole tree in the arena before anything walks on it
                    a.base_w[slot] = 0;
                    a.lz_info[slot] = make_int4(1, 1, 0, 1);
                    a.lz_cursor[slot] = 1;
                    a.lz_flag[slot] = 1;
                    atomicAdd(&a.stats[4], 1);
                }
            }
            __syncthreads();  // (the scratch tree is free for the next root)
        }
        // ---- per-root results: node count check, depth, longest list (1 + most children)
        int mc = 0;
        const int n_lists = lazy_stop ? head : tail;  // ranks whose children lists are built
        if ((tail == expect || lazy_stop) && !(INSTR && (a.exp & 32)) && !LAZY)  // (LAZY: a resolved list is as long as a degree; the host takes the graph's largest)
            for (int i0 = tid; i0 < n_lists; i0 += 8 * B2_T) {  // (eight positions per thread in flight: 977 one-load iterations otherwise)
                int c0[8], c1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * B2_T;
                    c0[u] = i < n_lists ? cstart[i] : 0;
                    c1[u] = i < n_lists ? cstart[i + 1] : 0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) mc = max(mc, c1[u] - c0[u]);
            }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mc = max(mc, __shfl_xor(mc, off, 64));
        if (lane == 0) atomicMax(&s_max, mc);
        __syncthreads();
        if (tid == 0) {
            if (tail != expect && !lazy_stop) a.stats[2] = 1;
            atomicMax(&a.stats[0], lazy_stop ? depth + 2 : depth);  // (walks on a lazy tree end two levels below the exact ones, or raise the slot's flag)
            atomicMax(&a.stats[1], s_max + 1);
        }
        __syncthreads();
    }
}

}  // namespace gg

using namespace gg;

namespace gg {

// Scratch, kernel choice, launch and read-back shared by the three builds: whole trees into the slots' segments, lazy trees
// (gg_build_trees_device), whole trees of single slots into the arena (lazy_fallback_rebuild).  The caller has set the roots /
// base / output pointers of `a`; stats = {deepest level, longest list, error, 1024 - smallest lazy level}.
static int bfs_run(gg_ctx *ctx, BfsArgs &a, int n_items, bool lazy, int32_t *stats /*[6]*/) {
    const int n = ctx->n_node;
    const int grid = std::min<int>(n_items, ctx->n_cus);
    const int bm_words = (n + 31) / 32;
    // static LDS of the kernel (eoff, e0s, duplicate list, mask, counters) ~ 21.5 KB; the CU has 160 KB
    const bool v1 = getenv("GG_BFS_V1") != nullptr && !lazy && !a.expect;  // the chunk kernel of rounds 2-3 (kept for A/B timing and as a second witness in the tests)
    hipFuncAttributes fa;
    GG_HIP(ctx, hipFuncGetAttributes(&fa, v1 ? (const void *)bfs_order_kernel<true, false> : (const void *)bfs_order2_kernel<true, false>));
    const size_t lds_total = 160 * 1024;
    const bool lds_bm = !getenv("GG_BFS_GLOBAL_BITMAP") && fa.sharedSizeBytes + (size_t)bm_words * 4 <= lds_total;

    // scratch kept in the context: an epoch over non-resident roots calls this once per root batch
    DevBuf &gkey = ctx->bfs_key, &gbm = ctx->bfs_bm, &misc = ctx->bfs_misc, &sp = ctx->bfs_sparse;
    if (!ctx->bfs_rowptr32_valid) {  // 4-byte offsets for the kernels here (nnz < 2^31 was checked above)
        if (ctx->bfs_rowptr32.reserve(sizeof(uint32_t) * ((size_t)n + 2)) != hipSuccess)
            return fail(ctx, GG_ENOMEM, "gg_build_trees_device: %zu bytes for the 4-byte row offsets", sizeof(uint32_t) * ((size_t)n + 2));
        std::vector<uint32_t> r32((size_t)n + 2);
        for (int i = 0; i <= n; ++i) r32[i] = (uint32_t)ctx->h_rowptr[i];
        r32[(size_t)n + 1] = r32[n];  // (8-byte loads at the last node read one word further)
        GG_HIP(ctx, hipMemcpy(ctx->bfs_rowptr32.p, r32.data(), sizeof(uint32_t) * r32.size(), hipMemcpyHostToDevice));
        ctx->bfs_rowptr32_valid = true;
    }
    const size_t key_bytes_before = gkey.bytes;
    hipError_t e = gkey.reserve(sizeof(uint32_t) * (size_t)grid * n);
    if (e == hipSuccess && !lds_bm) e = gbm.reserve(sizeof(uint32_t) * (size_t)grid * bm_words);
    constexpr size_t misc_bytes = sizeof(int32_t) * 8 + sizeof(unsigned long long) * 32;
    if (e == hipSuccess) e = misc.reserve(misc_bytes);
    if (e != hipSuccess) return fail(ctx, GG_ENOMEM, "gg_build_trees_device: scratch: %s", hipGetErrorString(e));
    // sparse levels (bfs_order2_kernel, LDS bitmap): GG_BFS_SPARSE=0 switches them off; _K / _MIN / _BUCKETS tune the choice
    int sp_k = -1, sp_min = 0, sp_buckets = 16, sp_cap = 0;
    size_t sp_off[6] = {0, 0, 0, 0, 0, 0};
    if (!v1 && lds_bm && !(getenv("GG_BFS_SPARSE") && atoi(getenv("GG_BFS_SPARSE")) == 0)) {
        sp_k = getenv("GG_BFS_SPARSE_K") ? std::max(0, atoi(getenv("GG_BFS_SPARSE_K"))) : 3;
        sp_min = getenv("GG_BFS_SPARSE_MIN") ? std::max(1, atoi(getenv("GG_BFS_SPARSE_MIN"))) : 8192;
        sp_buckets = getenv("GG_BFS_SPARSE_BUCKETS") ? std::min(1024, std::max(1, atoi(getenv("GG_BFS_SPARSE_BUCKETS")))) : 16;
        sp_cap = sp_k > 0 ? n / sp_k + 64 : n;  // unseen nodes * k <= level nodes <= n
        if (sp_cap > n) sp_cap = n;
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t sizes[6] = {sizeof(int32_t) * (size_t)grid * sp_cap * 2, sizeof(unsigned long long) * (size_t)grid * sp_cap * 2, sizeof(unsigned long long) * (size_t)grid * n,
                                 sizeof(uint32_t) * (size_t)grid * bm_words, sizeof(uint32_t) * (size_t)grid * bm_words,
                                 sizeof(unsigned long long) * (size_t)grid * ((size_t)n / 64 + 32)};
        size_t tot = 0;
        for (int i = 0; i < 6; ++i) { sp_off[i] = tot; tot += up(sizes[i]); }
        e = sp.reserve(tot);
        if (e != hipSuccess) {  // (not essential: without the scratch every level is popped whole)
            (void)hipGetLastError();
            sp_k = -1;
        }
    }
    a.n_node = n;
    a.n_roots = n_items;
    a.rowptr = ctx->g_rowptr;
    a.rowptr32 = ctx->bfs_rowptr32.as<uint32_t>();
    a.col = ctx->g_col;
    a.ticket = misc.as<unsigned int>();
    a.stats = misc.as<int32_t>() + 1;
    a.gbitmap = gbm.as<uint32_t>();
    a.gkey = gkey.as<uint32_t>();
    a.bm_words = bm_words;
    const bool prof = getenv("GG_BFS_PROFILE") != nullptr && !lazy;
    a.exp = (getenv("GG_BFS_EXPERIMENT") && !lazy) ? atoi(getenv("GG_BFS_EXPERIMENT")) : 0;
    a.prof = prof ? (unsigned long long *)(misc.as<int32_t>() + 8) : nullptr;
    a.sp_k = sp_k;
    a.sp_min = sp_min;
    a.sp_buckets = sp_buckets;
    a.sp_cap = sp_cap;
    if (sp_k >= 0) {
        char *const b = (char *)sp.p;
        a.sp_ux = (int32_t *)(b + sp_off[0]);
        a.sp_ui = (unsigned long long *)(b + sp_off[1]);
        a.sp_s = (unsigned long long *)(b + sp_off[2]);
        a.sp_vis = (uint32_t *)(b + sp_off[3]);
        a.sp_mark = (uint32_t *)(b + sp_off[4]);
        a.sp_mask = (unsigned long long *)(b + sp_off[5]);
        (void)hipMemsetAsync(a.sp_mark, 0, sizeof(uint32_t) * (size_t)grid * bm_words, ctx->stream);  // (the kernel leaves it clean; a launch that failed may not have)
    }
    (void)hipMemsetAsync(misc.p, 0, misc_bytes, ctx->stream);
    if (gkey.bytes != key_bytes_before)  // new allocation: all-ones over ALL of it (reserve over-allocates: a later, somewhat larger grid reuses
        (void)hipMemsetAsync(gkey.p, 0xFF, gkey.bytes, ctx->stream);  // the buffer without passing here again); the kernel restores every word it uses
    (void)hipEventRecord(ctx->ev0, ctx->stream);
    if (!v1) {
        const size_t dyn = lds_bm ? (size_t)((bm_words + 15) & ~15) * 4 : 0;  // (rounded up: the lazy finalize reads 16-word blocks)
        const bool instr = prof || a.exp != 0;
        const void *fn = lazy ? (lds_bm ? (const void *)bfs_order2_kernel<true, false, true> : (const void *)bfs_order2_kernel<false, false, true>)
                         : lds_bm ? (instr ? (const void *)bfs_order2_kernel<true, true> : (const void *)bfs_order2_kernel<true, false>)
                                  : (instr ? (const void *)bfs_order2_kernel<false, true> : (const void *)bfs_order2_kernel<false, false>);
        if (lds_bm) {
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            if (e != hipSuccess) return fail(ctx, GG_EHIP, "gg_build_trees_device: %zu bytes of LDS: %s", dyn, hipGetErrorString(e));
        }
        if (lazy && lds_bm) hipLaunchKernelGGL((bfs_order2_kernel<true, false, true>), dim3(grid), dim3(B2_T), dyn, ctx->stream, a);
        else if (lazy) hipLaunchKernelGGL((bfs_order2_kernel<false, false, true>), dim3(grid), dim3(B2_T), 0, ctx->stream, a);
        else if (lds_bm && instr) hipLaunchKernelGGL((bfs_order2_kernel<true, true>), dim3(grid), dim3(B2_T), dyn, ctx->stream, a);
        else if (lds_bm) hipLaunchKernelGGL((bfs_order2_kernel<true, false>), dim3(grid), dim3(B2_T), dyn, ctx->stream, a);
        else if (instr) hipLaunchKernelGGL((bfs_order2_kernel<false, true>), dim3(grid), dim3(B2_T), 0, ctx->stream, a);
        else hipLaunchKernelGGL((bfs_order2_kernel<false, false>), dim3(grid), dim3(B2_T), 0, ctx->stream, a);
    } else if (lds_bm) {
        const size_t dyn = (size_t)((bm_words + 15) & ~15) * 4;
        const bool instr = prof || a.exp != 0;
        const void *fn = instr ? (const void *)bfs_order_kernel<true, true> : (const void *)bfs_order_kernel<true, false>;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess) return fail(ctx, GG_EHIP, "gg_build_trees_device: %zu bytes of LDS: %s", dyn, hipGetErrorString(e));
        if (instr) hipLaunchKernelGGL((bfs_order_kernel<true, true>), dim3(grid), dim3(BFS_T), dyn, ctx->stream, a);
        else hipLaunchKernelGGL((bfs_order_kernel<true, false>), dim3(grid), dim3(BFS_T), dyn, ctx->stream, a);
    } else if (prof || a.exp != 0) {
        hipLaunchKernelGGL((bfs_order_kernel<false, true>), dim3(grid), dim3(BFS_T), 0, ctx->stream, a);
    } else {
        hipLaunchKernelGGL((bfs_order_kernel<false, false>), dim3(grid), dim3(BFS_T), 0, ctx->stream, a);
    }
    (void)hipEventRecord(ctx->ev1, ctx->stream);
    for (int i = 0; i < 6; ++i) stats[i] = 0;
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(stats, a.stats, sizeof(int32_t) * 6, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && prof && !v1) {
        const int n_roots = n_items;
        unsigned long long pc[32];
        if (hipMemcpy(pc, a.prof, sizeof(pc), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[bfs2 profile] %d roots, grid %d: per root %.0f windows (%.0f without a candidate, %.0f with in-window duplicates, %.0f through the key array), "
                    "%.1f full fences, %.0f nodes appended, %.0f quads scanned\n", n_roots, grid, (double)pc[0] / n_roots, (double)pc[4] / n_roots, (double)pc[5] / n_roots,
                    (double)pc[6] / n_roots, (double)pc[1] / n_roots, (double)pc[2] / n_roots, (double)pc[3] / n_roots);
            fprintf(stderr, "[bfs2 profile] sparse levels (k %d, min %d, %d buckets): %.2f per root, %.0f unseen nodes listed, %.1f bucket passes, %.0f candidate fathers per root\n",
                    sp_k, sp_min, sp_buckets, (double)pc[7] / n_roots, (double)pc[18] / n_roots, (double)pc[20] / n_roots, (double)pc[19] / n_roots);
            const char *names[10] = {"setup: node info + scan", "window choice", "quad map", "scan: loads + tests", "scan barrier", "claim", "duplicates + prefix",
                                     "append + cstart (empty windows: all behind the scan)", "sparse level: father search", "sparse level: cstart fill"};
            double tot = 0;
            for (int k = 0; k < 10; ++k) tot += (double)pc[8 + k];
            fprintf(stderr, "[bfs2 profile] wave-0 clock per root %.0f, per window %.0f:", tot / n_roots, tot / (double)(pc[0] ? pc[0] : 1));
            for (int k = 0; k < 10; ++k) fprintf(stderr, " %s %.1f%%", names[k], 100.0 * (double)pc[8 + k] / (tot > 0 ? tot : 1));
            fprintf(stderr, "\n");
            const char *snames[7] = {"unseen list + adjacency ranges", "cstart zeros", "bucket bitmaps", "scans", "marks into LDS", "rank masks", "S list + bitmap restore"};
            double stot = 0;
            for (int k = 0; k < 7; ++k) stot += (double)pc[21 + k];
            fprintf(stderr, "[bfs2 profile] father search, wave-0 clock per root %.0f:", stot / n_roots);
            for (int k = 0; k < 7; ++k) fprintf(stderr, " %s %.1f%%", snames[k], 100.0 * (double)pc[21 + k] / (stot > 0 ? stot : 1));
            fprintf(stderr, "\n");
        }
    }
    if (e == hipSuccess && prof && v1) {
        unsigned long long pc[16];
        if (hipMemcpy(pc, a.prof, sizeof(pc), hipMemcpyDeviceToHost) == hipSuccess) {
            const char *names[8] = {"nodes+scan", "batch form", "map+load+test", "empty chunk", "claim", "dup resolve", "compaction", "stores"};
            double tot = 0;
            for (int k = 0; k < 8; ++k) tot += (double)pc[k];
            fprintf(stderr, "[bfs profile] %d roots, grid %d: batches %llu chunks %llu (empty %llu, with dups %llu); wave-0 shader-clock share:", n_items, grid, pc[8], pc[9], pc[10], pc[11]);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.1f%%", names[k], 100.0 * (double)pc[k] / (tot > 0 ? tot : 1));
            fprintf(stderr, "; cycles per chunk %.0f\n", tot / (double)(pc[9] ? pc[9] : 1));
        }
    }
    float ms = 0.f;
    if (e == hipSuccess) (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    if (e != hipSuccess) return fail(ctx, GG_EHIP, "gg_build_trees_device: %s", hipGetErrorString(e));
    if (a.exp) fprintf(stderr, "[bfs experiment %d] kernel %.1f ms for %d roots (results invalid)\n", a.exp, ms, n_items);
    GG_CHECK(ctx, stats[2] == 0 || (a.exp & ~(8 | 64 | 128)), GG_EINVAL, "gg_build_trees_device: a BFS reached a different number of nodes than the component sweep of the graph");
    ctx->ctr.bfs_kernel_ms += ms;
    ctx->ctr.bfs_trees += n_items;
    return GG_OK;
}

// Is the next build lazy?  gg_set_tree_mode / GG_TREE_LAZY decide; by default (-1) the trees of an epoch's root batches
// (gg_epoch_add: built, walked twice, dropped) on graphs from GG_LZ_AUTO_NODES (2^18) nodes on -- trees that stay resident for many
// steps (gg_build_trees_device called directly) are built whole: every step would resolve its lists anew.
// Lazy trees need what the walks' resolution reads: a symmetric adjacency (g_rev), degrees that fit the 20 count bits of a pair.
bool lazy_build_wanted(const gg_ctx *ctx) {
    int mode = ctx->tree_mode;
    if (const char *e = getenv("GG_TREE_LAZY")) mode = atoi(e);
    if (mode < 0) {
        const long long auto_nodes = getenv("GG_LZ_AUTO_NODES") ? atoll(getenv("GG_LZ_AUTO_NODES")) : (1ll << 18);
        mode = (ctx->in_epoch_add && ctx->n_node >= auto_nodes) ? 1 : 0;
    }
    return mode > 0 && !ctx->lz_force_whole && ctx->g_rev && ctx->g_max_deg < 0xFFFFF && !getenv("GG_BFS_V1");
}

// Node limit of a slot's exact part: a level is expanded only while (nodes so far + adjacency entries of the level) fit it.  The
// default -- the node count -- stops in front of the level that holds the bulk of a small-world graph; components up to
// `whole_max` nodes are built whole.  gg_set_tree_mode's node_cap (tests) sets both.
static void lazy_limits(const gg_ctx *ctx, int64_t *cap, int64_t *whole_max) {
    const int64_t c = ctx->lz_cap;
    // (default 3/8 of the nodes: on the 10^6-node bench graph the node count itself lets 7 % of the roots expand their third level
    // -- up to 900 000 entries each, as dear as whole trees: 78 instead of 44 ms per 16 384 roots -- while 3/8 stops 2.4 % one level
    // early; those get whole trees right away, below)
    *cap = c > 0 ? c : getenv("GG_LZ_CAP") ? std::max<int64_t>(2, atoll(getenv("GG_LZ_CAP"))) : std::max<int64_t>(65536, (int64_t)ctx->n_node * 3 / 8);
    *whole_max = c > 0 ? c : 65536;
}

static int build_trees_lazy(gg_ctx *ctx, const int32_t *roots, int32_t n_roots) {
    const int n = ctx->n_node;
    if (ctx->h_comp_size.empty()) component_sizes(n, ctx->h_rowptr.data(), ctx->h_col.data(), ctx->h_comp_size);
    int64_t cap, whole_max;
    lazy_limits(ctx, &cap, &whole_max);
    // Two limits: a level is expanded while (nodes so far + its adjacency entries) fit `cap` -- 3/8 of the nodes: stops in front of
    // the level that holds the bulk of a small-world graph --, but the first two levels below the root's children may take up to
    // the node count: 2.4 % of the bench roots have a second level whose expansion exceeds 3/8 N; stopped there they resolve
    // through two levels of hubs and mostly end as whole trees (19 ms per 16 384 roots), expanded they cost a fraction of one.
    // (Raising `cap` itself to N lets 7 % of the roots expand their THIRD level, up to 900 000 entries each: 78 instead of 44 ms.)
    const int easy_levels = ctx->lz_cap > 0 ? 0 : (getenv("GG_LZ_EASY") ? atoi(getenv("GG_LZ_EASY")) : 2);
    const int64_t easy_cap = std::max<int64_t>(cap, n);
    // per root: node limit of the exact part, pool behind it.  The pool takes what a resolution reserves -- one entry per candidate
    // of the node's adjacency -- for the walks a root can have, ~3 lazy hops each.
    std::vector<int32_t> limit(n_roots), expect(n_roots), pool(n_roots);
    int64_t max_c = 1, max_limit = 1, sure = 0;
    for (int r = 0; r < n_roots; ++r) {
        const int64_t C = ctx->h_comp_size[roots[r]], deg = ctx->h_rowptr[roots[r] + 1] - ctx->h_rowptr[roots[r]];
        expect[r] = (int32_t)C;
        max_c = std::max(max_c, C);
        if (C <= whole_max || deg + 1 > cap) {  // (a root whose own children exceed the limit: whole as well)
            limit[r] = (int32_t)C;
            pool[r] = 0;
        } else {
            const int64_t pool_env = getenv("GG_LZ_POOL") ? atoll(getenv("GG_LZ_POOL")) : 0;
            limit[r] = (int32_t)std::min<int64_t>(cap, C);
            if (easy_levels > 0) max_limit = std::max<int64_t>(max_limit, std::min<int64_t>(easy_cap, C));  // (the scratch tree holds what the easy levels may append)
            pool[r] = (int32_t)(pool_env > 0 ? pool_env : std::min<int64_t>(std::max<int64_t>(16384, 192 * (deg + 64)), std::max<int64_t>(cap, 16384)));
        }
        max_limit = std::max<int64_t>(max_limit, limit[r]);
        sure += (int64_t)limit[r] + pool[r] + 1;
    }
    // entries for all segments: what the slots can need at most, but no more than an average of a sixth of the nodes (+ pool) per
    // slot -- a slot that finds no room is rebuilt whole in the arena right away
    const int64_t avg_env = getenv("GG_LZ_AVG") ? atoll(getenv("GG_LZ_AVG")) : 0;
    const int64_t avg = avg_env > 0 ? avg_env : std::max<int64_t>(65536, n / 6) + 16384;
    const int64_t budget = std::max<int64_t>(1, std::min<int64_t>(sure, (int64_t)n_roots * avg));
    // the arena: whole trees of the slots whose walks need more than a lazy tree gives (lazy_fallback_rebuild)
    const int64_t arena_roots = getenv("GG_LZ_ARENA") ? atoll(getenv("GG_LZ_ARENA")) : std::min<int64_t>(1024, std::max<int64_t>(8, n_roots / 16));
    const int64_t arena = arena_roots * (max_c + 1);
    std::vector<int64_t> zeros(n_roots, 0);
    int rc = alloc_trees(ctx, roots, n_roots, zeros.data(), nullptr, budget + arena);
    if (rc != GG_OK) return rc;
    if (n_roots == 0) return GG_OK;
    ctx->arena_next = budget;
    ctx->arena_end = budget + arena;
    const int bm_words = (n + 31) / 32;
    const int grid = std::min<int>(n_roots, ctx->n_cus);
    const size_t scr = (size_t)max_limit + 1;
    const size_t pair_before = ctx->lz_pair.bytes;
    hipError_t e = ctx->lz_info.reserve(sizeof(int4) * (size_t)n_roots);
    if (e == hipSuccess) e = ctx->lz_pair.reserve(sizeof(unsigned long long) * (size_t)ctx->t_cap_nodes);
    if (e == hipSuccess) e = ctx->lz_rank.reserve(sizeof(int32_t) * (size_t)ctx->t_cap_nodes);
    if (e == hipSuccess) e = ctx->lz_bm.reserve(sizeof(uint2) * (size_t)n_roots * bm_words);
    if (e == hipSuccess) e = ctx->lz_cursor.reserve(sizeof(int32_t) * (size_t)n_roots);
    if (e == hipSuccess) e = ctx->lz_flag.reserve(sizeof(int32_t) * (size_t)n_roots);
    if (e == hipSuccess) e = ctx->lz_limit.reserve(sizeof(int32_t) * (size_t)n_roots * 2);  // limits, pools
    if (e == hipSuccess) e = ctx->lz_expect.reserve(sizeof(int32_t) * (size_t)n_roots);
    if (e == hipSuccess) e = ctx->lz_list.reserve(sizeof(int32_t) * 4 * (size_t)n_roots + 64);  // (walk launches: items listed + claims per item; a launch may name a slot twice)
    if (e == hipSuccess) e = ctx->lz_scratch.reserve(sizeof(int32_t) * (size_t)grid * (3 * scr + 1) + 64);
    if (e != hipSuccess) return fail(ctx, GG_ENOMEM, "gg_build_trees_device: lazy-tree arrays: %s", hipGetErrorString(e));
    // a pair is valid iff it carries the build's stamp: nothing to initialise per build; cleared when the 12 bits wrap (and when new)
    ctx->lz_stamp = ctx->lz_stamp % 4095u + 1u;
    if (ctx->lz_stamp == 1u || ctx->lz_pair.bytes != pair_before) GG_HIP(ctx, hipMemsetAsync(ctx->lz_pair.p, 0, ctx->lz_pair.bytes, ctx->stream));
    GG_HIP(ctx, hipMemsetAsync(ctx->lz_flag.p, 0, sizeof(int32_t) * (size_t)n_roots, ctx->stream));
    GG_HIP(ctx, hipMemsetAsync(ctx->lz_list.p, 0, ctx->lz_list.bytes, ctx->stream));
    GG_HIP(ctx, hipMemsetAsync(ctx->dev_ctr + 1500, 0, sizeof(unsigned long long) * 10, ctx->stream));  // resolution statistics of this build (walk_sample.hip, lz_ctr) + [8] the segment cursor
    GG_HIP(ctx, hipMemcpyAsync(ctx->lz_limit.p, limit.data(), sizeof(int32_t) * (size_t)n_roots, hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, hipMemcpyAsync(ctx->lz_limit.as<int32_t>() + n_roots, pool.data(), sizeof(int32_t) * (size_t)n_roots, hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, hipMemcpyAsync(ctx->lz_expect.p, expect.data(), sizeof(int32_t) * (size_t)n_roots, hipMemcpyHostToDevice, ctx->stream));
    BfsArgs a{};
    a.roots = ctx->t_root;
    a.base = ctx->t_base;
    a.base_w = ctx->t_base;
    a.order = ctx->t_order;
    a.cstart = ctx->t_cstart;
    a.edge = ctx->t_edge;
    a.expect = ctx->lz_expect.as<int32_t>();
    a.lz_limit = ctx->lz_limit.as<int32_t>();
    a.lz_pool = ctx->lz_limit.as<int32_t>() + n_roots;
    a.lz_info = ctx->lz_info.as<int4>();
    a.lz_bm = ctx->lz_bm.as<uint2>();
    a.lz_rank = ctx->lz_rank.as<int32_t>();
    a.lz_cursor = ctx->lz_cursor.as<int32_t>();
    a.lz_flag = ctx->lz_flag.as<int32_t>();
    a.lz_alloc = ctx->dev_ctr + 1508;
    a.lz_alloc_end = budget;
    a.scr_cap = (int)scr;
    a.lz_ablate = getenv("GG_LZ_ABLATE") ? atoi(getenv("GG_LZ_ABLATE")) : 0;
    a.lz_easy = easy_levels;
    a.lz_limit_easy = (int)easy_cap;
    a.scr_order = ctx->lz_scratch.as<int32_t>();
    a.scr_edge = a.scr_order + (size_t)grid * scr;
    a.scr_cstart = a.scr_edge + (size_t)grid * scr;
    int32_t stats[6];
    rc = bfs_run(ctx, a, n_roots, /*lazy=*/true, stats);  // (copies and memsets above: same stream, done when it returns)
    if (rc != GG_OK) return rc;
    ctx->t_lazy = true;
    ctx->tree_entries = 0;
    ctx->lz_min_level = stats[3] > 0 ? 1024 - stats[3] : 0x7fffffff;
    ctx->tree_max_depth = stats[0];
    ctx->tree_max_list = ctx->g_max_deg + 1;  // (a resolved list holds up to the node's degree)
    ctx->t_edge_valid = true;
    // Slots that stopped a level (or more) before the batch's deepest exact level: their walks resolve through two levels of
    // hubs and, more often than not, leave the resolvable levels -- the launch would be voided and repeated for them.  When they
    // are few they get their whole trees now (gg_set_tree_mode with a node_cap: tests keep every slot as built).
    int early = 0;
    if (ctx->lz_cap == 0 && !getenv("GG_LZ_NO_EARLY") && stats[5] > ctx->lz_min_level && ctx->lz_min_level < 0x7fffffff) {
        std::vector<int4> info((size_t)n_roots);
        GG_HIP(ctx, hipMemcpy(info.data(), ctx->lz_info.p, sizeof(int4) * (size_t)n_roots, hipMemcpyDeviceToHost));
        std::vector<int32_t> flag((size_t)n_roots, 0);
        for (int r = 0; r < n_roots; ++r)
            if (info[r].x < info[r].y && info[r].z < stats[5]) { flag[r] = 1; ++early; }
        if (early > 0 && early <= std::min<int64_t>(arena_roots, n_roots / 8)) {
            std::vector<int32_t> cur((size_t)n_roots);
            GG_HIP(ctx, hipMemcpy(cur.data(), ctx->lz_flag.p, sizeof(int32_t) * (size_t)n_roots, hipMemcpyDeviceToHost));
            for (int r = 0; r < n_roots; ++r) flag[r] |= cur[r];
            GG_HIP(ctx, hipMemcpy(ctx->lz_flag.p, flag.data(), sizeof(int32_t) * (size_t)n_roots, hipMemcpyHostToDevice));
        } else {
            early = 0;
        }
    }
    if (stats[4] > 0 || early > 0) {  // (and the slots that found no room among the segments): whole trees in the arena, now
        int rebuilt = 0;
        rc = lazy_fallback_rebuild(ctx, &rebuilt);
        if (rc == GG_ECAPACITY) rc = lazy_rebuild_whole(ctx);
        if (rc != GG_OK) return rc;
        if (ctx->t_lazy) {  // the smallest exact level of what is still lazy
            ctx->lz_min_level = early > 0 ? stats[5] : ctx->lz_min_level;
        }
    }
    return GG_OK;
}

// the rebuilt slots' new bases and "no lazy ranks" records, in one launch (a synchronous 8-byte copy per slot was ~8 ms per batch)
__global__ void lazy_move_slots_kernel(const int32_t *slots, const int64_t *base, const int32_t *expect, int m, int64_t *t_base, int4 *lz_info) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    t_base[slots[i]] = base[i];
    lz_info[slots[i]] = make_int4(expect[i], expect[i], 0, 0);
}

// Whole trees for the slots whose lazy tree did not suffice (lz_flag raised by the walks of the launch that has just finished),
// built into the arena; the slots' bases move there.  Returns the number of slots rebuilt in *n_out; GG_ECAPACITY when the arena
// is full (the caller rebuilds the batch whole).
int lazy_fallback_rebuild(gg_ctx *ctx, int *n_out) {
    *n_out = 0;
    const int R = ctx->n_tree_roots;
    std::vector<int32_t> flag(R);
    GG_HIP(ctx, hipMemcpy(flag.data(), ctx->lz_flag.p, sizeof(int32_t) * (size_t)R, hipMemcpyDeviceToHost));
    std::vector<int32_t> slots, roots, expect;
    std::vector<int64_t> base;
    int64_t next = ctx->arena_next;  // (lazy builds: a slot's row of first-child ranks starts at its base -- no shift by the slot)
    for (int s = 0; s < R; ++s) {
        if (!flag[s]) continue;
        const int64_t C = ctx->h_comp_size[ctx->h_troot[s]];
        if (next + C + 1 > ctx->arena_end) return fail(ctx, GG_ECAPACITY, "lazy trees: the arena of whole trees is full");
        slots.push_back(s);
        roots.push_back(ctx->h_troot[s]);
        expect.push_back((int32_t)C);
        base.push_back(next);
        next += C + 1;
    }
    const int m = (int)slots.size();
    if (m == 0) return GG_OK;
    base.push_back(next);
    // (one scratch allocation kept in the context: [zeros | roots | expect | slots] int32 then [base] int64)
    DevBuf &scr = ctx->lz_fb_scratch;
    const size_t i32n = 4 * (size_t)m + 4;
    hipError_t e = scr.reserve(sizeof(int32_t) * i32n + sizeof(int64_t) * ((size_t)m + 2));
    if (e != hipSuccess) return fail(ctx, GG_ENOMEM, "lazy trees: %s", hipGetErrorString(e));
    int32_t *const d_zero = scr.as<int32_t>(), *const d_roots_p = d_zero + m, *const d_expect_p = d_roots_p + m, *const d_slots_p = d_expect_p + m;
    int64_t *const d_base_p = reinterpret_cast<int64_t *>(scr.as<int32_t>() + ((i32n + 1) & ~(size_t)1));
    auto release = [&]() {};
    GG_HIP(ctx, hipMemsetAsync(d_zero, 0, sizeof(int32_t) * m, ctx->stream));
    GG_HIP(ctx, hipMemcpyAsync(d_roots_p, roots.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, hipMemcpyAsync(d_expect_p, expect.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, hipMemcpyAsync(d_slots_p, slots.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, hipMemcpyAsync(d_base_p, base.data(), sizeof(int64_t) * (m + 1), hipMemcpyHostToDevice, ctx->stream));
    BfsArgs a{};
    a.roots = d_roots_p;
    a.base = d_base_p;
    a.order = ctx->t_order;
    a.cstart = ctx->t_cstart;
    a.edge = ctx->t_edge;
    a.expect = d_expect_p;
    a.slot_ids = d_zero;  // (all zero: the rows are not shifted)
    int32_t stats[6];
    int rc = bfs_run(ctx, a, m, /*lazy=*/false, stats);
    release();
    if (rc != GG_OK) return rc;
    // the slots now point into the arena and have no lazy ranks
    hipLaunchKernelGGL(lazy_move_slots_kernel, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream, d_slots_p, d_base_p, d_expect_p, m, ctx->t_base, ctx->lz_info.as<int4>());
    GG_HIP(ctx, hipMemsetAsync(ctx->lz_flag.p, 0, sizeof(int32_t) * (size_t)R, ctx->stream));
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->arena_next = next;
    ctx->tree_max_depth = std::max(ctx->tree_max_depth, stats[0]);
    ctx->tree_max_list = std::max(ctx->tree_max_list, stats[1]);
    ctx->lz_fallback_roots += m;
    ctx->lz_fallback_rounds += 1;
    *n_out = m;
    return GG_OK;
}

// The arena is full: the resident lazy batch is rebuilt as whole trees (the Q3 bits of the slots survive).
int lazy_rebuild_whole(gg_ctx *ctx) {
    const int R = ctx->n_tree_roots;
    const std::vector<int32_t> roots = ctx->h_troot;
    std::vector<uint32_t> q3((size_t)std::max<int64_t>(ctx->h_q3off[R], 1));
    GG_HIP(ctx, hipMemcpy(q3.data(), ctx->t_q3, sizeof(uint32_t) * (size_t)ctx->h_q3off[R], hipMemcpyDeviceToHost));
    ctx->lz_pair.release();
    ctx->lz_rank.release();
    ctx->lz_bm.release();
    ctx->lz_force_whole = true;
    const int rc = gg_build_trees_device(ctx, roots.data(), R);
    ctx->lz_force_whole = false;
    if (rc != GG_OK) return rc;
    if (ctx->h_q3off[R]) GG_HIP(ctx, hipMemcpy(ctx->t_q3, q3.data(), sizeof(uint32_t) * (size_t)ctx->h_q3off[R], hipMemcpyHostToDevice));
    ctx->lz_fallback_roots += R;
    ctx->lz_fallback_rounds += 1;
    return GG_OK;
}

}  // namespace gg

extern "C" int gg_build_trees_device(gg_ctx *ctx, const int32_t *roots, int32_t n_roots) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, ctx->g_rowptr && !ctx->h_rowptr.empty(), GG_EINVAL, "gg_build_trees_device: call gg_set_graph_csr first");
    GG_CHECK(ctx, n_roots >= 0 && (roots || n_roots == 0), GG_EINVAL, "gg_build_trees_device: bad roots");
    const int n = ctx->n_node;
    for (int r = 0; r < n_roots; ++r) GG_CHECK(ctx, roots[r] >= 0 && roots[r] < n, GG_EINVAL, "gg_build_trees_device: root %d out of range", roots[r]);
    GG_CHECK(ctx, ctx->g_nnz < (1ll << 31), GG_EINVAL, "gg_build_trees_device: %lld adjacency entries (limit 2^31 - 1)", (long long)ctx->g_nnz);
    GG_HIP(ctx, hipSetDevice(ctx->device));
    ctx->t_lazy = false;
    ctx->lz_min_level = 0x7fffffff;
    if (lazy_build_wanted(ctx)) return build_trees_lazy(ctx, roots, n_roots);
    int rc = alloc_trees(ctx, roots, n_roots, nullptr, nullptr);  // node counts from the (cached) component sweep
    if (rc != GG_OK) return rc;
    if (n_roots == 0) return GG_OK;
    BfsArgs a{};
    a.roots = ctx->t_root;
    a.base = ctx->t_base;
    a.order = ctx->t_order;
    a.cstart = ctx->t_cstart;
    a.edge = ctx->t_edge;
    int32_t stats[6];
    rc = bfs_run(ctx, a, n_roots, /*lazy=*/false, stats);
    if (rc != GG_OK) return rc;
    ctx->tree_max_depth = stats[0];
    ctx->tree_max_list = stats[1];
    ctx->t_edge_valid = (a.exp & ~(8 | 64 | 128)) == 0;
    return GG_OK;
}
