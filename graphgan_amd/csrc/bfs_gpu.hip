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
    int exp;                   // GG_BFS_EXPERIMENT: timing ablations (results are then WRONG): 1 = no cstart stores, 2 = synthetic targets instead of adjacency loads, 4 = no queue stores beyond level 1, 8 = no early exit when the component is complete (results stay right)
    unsigned long long *prof;  // GG_BFS_PROFILE: [16] shader-clock cycles of wave 0 per phase + event counts (NULL: off)
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
constexpr bool B2_PF_TWO_STAGE = false;  // row pointers of the prefetched nodes issued behind the scan instead of right behind their ids (measured: 48.7 vs 47.2 us per tree)
constexpr int B2_HASH = 1024;            // in-window duplicates: LDS hash node -> smallest duplicate position ...
constexpr int B2_DCAP = 768;             // ... for up to this many duplicates per window (more: the key array in global memory)

typedef int32_t int4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load from a 4-byte aligned address

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// workgroup barrier that waits for this wavefront's LDS traffic only: global loads / stores in flight stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool LDS_BM>
__global__ __launch_bounds__(B2_T) void bfs_order2_kernel(const BfsArgs a) {
    extern __shared__ uint32_t lds_bm[];              // [bm_words] when LDS_BM
    __shared__ uint32_t e0s[B2_NB];                   // first CSR entry of each window node (of the segment, for a partial node)
    __shared__ uint16_t soff[B2_NB + 2];              // first quad of each window node; [nb] = quads of the window
    __shared__ uint16_t degs[B2_NB];                  // adjacency entries of each window node (of the segment)
    __shared__ uint32_t scratch[B2_SLOTS / 2];        // quad -> window node (uint16 x 4 096)
    __shared__ int32_t tkey[B2_HASH], tpos[B2_HASH];  // duplicates of the window: node -> smallest position (open addressing; -1 / INT_MAX when free)
    __shared__ uint32_t winbits[B2_WORDS + 1];        // positions that append a node ([B2_WORDS] stays 0)
    __shared__ int32_t wpre[B2_WORDS + 1];            // exclusive popcount prefix over those words; [B2_WORDS] = total
    __shared__ int32_t wtot[2][B2_WAVES];
    __shared__ int32_t s_root, s_any, s_Lcount, s_cap, s_max, s_deg;
    __shared__ uint32_t s_e0;
    uint16_t *const emap = reinterpret_cast<uint16_t *>(scratch);

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
        int32_t *const order = a.order + a.base[r];
        int32_t *const cstart = a.cstart + a.base[r] + r;
        int32_t *const tedge = a.edge + a.base[r];
        const int expect = (int)(a.base[r + 1] - a.base[r]);
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
        unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // wave 0's clock per phase: setup, scan, claim, duplicates, prefix, append
        long long tprev = a.prof ? (long long)clock64() : 0;
#define B2_TICK(k)                                  \
    if (a.prof) {                                   \
        const long long tn = (long long)clock64();  \
        ph[k] += (unsigned long long)(tn - tprev);  \
        tprev = tn;                                 \
    }
        int seg_a = 0;           // > 0: the node at `head` is being scanned in segments, this many entries are done
        int pf_q = -1;           // queue index whose node this thread has prefetched
        int pf_v = 0, pf_stage = 0;
        uint32_t pf_e0 = 0;
        int pf_deg = 0;
        while (head < tail) {
            if (head == level_end) {  // the next level starts: everything up to `tail` belongs to it
                level_end = tail;
                ++depth;
            }
            // queue entries are read back from global memory (this window's nodes, the next window's prefetch): entries
            // written since the last fence must have landed first
            if (fenced < tail && head + 2 * B2_NB > fenced) {
                __syncthreads();
                fenced = tail;
                ++st_fence;
            }
            int nb = 0;      // complete nodes of this window; 0 = one node, entries [seg_a, seg_a + seg_len)
            int seg_len = 0;
            if (seg_a == 0) {
                const int navail = min(B2_NB, level_end - head);
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
                        if (q <= 8)
                            for (int k = 0; k < q; ++k) emap[excq + k] = (uint16_t)tid;
                    }
                    unsigned long long wide = __ballot(tid < nb && q > 8);  // their quads are filled by the whole wavefront
                    while (wide) {
                        const int l = __ffsll((long long)wide) - 1;
                        wide &= wide - 1;
                        const int qq = __shfl(q, l, 64), ss = __shfl(excq, l, 64);
                        for (int k = lane; k < qq; k += 64) emap[ss + k] = (uint16_t)((wv << 6) + l);
                    }
                    // prefetch the next window's nodes, stage 1: their ids (queue entries below `fenced` have landed).  Stage 2 --
                    // the row pointers, which need the id -- is issued behind the scan, when the id has long arrived: issued
                    // here, the dependent pair stalled every thread for a memory round trip per window (a third of the kernel).
                    const int q2 = head + nb + tid;
                    pf_q = -1;
                    if (q2 < fenced) {
                        pf_v = __hip_atomic_load(&order[q2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pf_q = q2;
                        pf_stage = 1;
                        if (!B2_PF_TWO_STAGE) {
                            const int64_t b2 = a.rowptr[pf_v];
                            pf_e0 = (uint32_t)b2;
                            pf_deg = (int)(a.rowptr[pf_v + 1] - b2);
                            pf_stage = 0;
                        }
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
            ++st_win;
            st_slots += (unsigned long long)T;

            // ---------------- SCAN: quads tid, tid + 1024, ...; position of entry k of quad s = 4 s + k (stream order)
            int4u w4[4];
            uint32_t we[4];     // CSR index of the quad's first entry
            uint32_t cand[4];   // entries whose target is unseen
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int s = tid + it * B2_T;
                cand[it] = 0u;
                we[it] = 0xffffffffu;  // (no quad)
                if (s < T) {
                    const int i = (int)emap[s];
                    const int j0 = 4 * (s - (int)soff[i]);
                    const int cnt = (int)degs[i] - j0;  // >= 1
                    we[it] = e0s[i] + (uint32_t)j0;
                    if (a.exp & 2) {  // (ablation: computed targets instead of the adjacency loads)
                        const uint32_t hx = we[it] * 2654435761u;
                        w4[it].x = (int)(hx % (uint32_t)a.n_node); w4[it].y = (int)((hx >> 3) % (uint32_t)a.n_node);
                        w4[it].z = (int)((hx >> 5) % (uint32_t)a.n_node); w4[it].w = (int)((hx >> 7) % (uint32_t)a.n_node);
                    } else {
                        w4[it] = *reinterpret_cast<const int4u *>(a.col + we[it]);
                    }
                    cand[it] = cnt >= 4 ? 15u : ((1u << cnt) - 1u);  // valid entries for now
                }
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
            if (pf_stage == 1) {  // prefetch, stage 2: the row pointers of the next window's nodes
                const int64_t b2 = a.rowptr[pf_v];
                pf_e0 = (uint32_t)b2;
                pf_deg = (int)(a.rowptr[pf_v + 1] - b2);
                pf_stage = 0;
            }
            B2_TICK(4)

            if (!s_any) {
                // nothing new: every node that ends in this window has its children end at `tail`
                ++st_empty;
                if (nb > 0) {
                    if (tid < nb) cstart[head + tid + 1] = tail;
                    head += nb;
                } else {
                    seg_a += seg_len;
                    if (seg_a == s_deg) {
                        if (tid == 0) cstart[head + 1] = tail;
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
                            if (nL > 0 && !(a.exp & 16)) {
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
                    ++st_key;
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
                if (nL > 0) ++st_dup;
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
                B2_TICK(6)
                st_cand += (unsigned long long)total;
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
                                tedge[rank] = (int32_t)(we[it] + (uint32_t)k);
                            }
                        }
                    }
                }
                // children of window node i end behind the appends of the positions below its last quad's end
                if (nb > 0) {
                    if (tid < nb) {
                        const int P = 4 * (int)soff[tid + 1];
                        cstart[head + tid + 1] = tail + wpre[P >> 5] + (int)__popc(winbits[P >> 5] & ((1u << (P & 31)) - 1u));
                    }
                    tail += total;
                    head += nb;
                } else {
                    tail += total;
                    seg_a += seg_len;
                    if (seg_a == s_deg) {
                        if (tid == 0) cstart[head + 1] = tail;
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
            if (tail == expect && head < tail && !(a.exp & 8)) {
                for (int i = head + tid; i < expect; i += B2_T) cstart[i + 1] = expect;
                if (level_end < tail) ++depth;  // the nodes behind level_end are one level deeper than the one being popped
                break;
            }
        }
        __syncthreads();  // (all queue / cstart stores have landed: the child-count sweep below reads cstart)
        if (a.prof && tid == 0) {
            atomicAdd(&a.prof[0], st_win); atomicAdd(&a.prof[1], st_fence); atomicAdd(&a.prof[2], st_cand); atomicAdd(&a.prof[3], st_slots);
            atomicAdd(&a.prof[4], st_empty); atomicAdd(&a.prof[5], st_dup); atomicAdd(&a.prof[6], st_key);
            for (int k = 0; k < 8; ++k) atomicAdd(&a.prof[8 + k], ph[k]);
        }
#undef B2_TICK
        // ---- per-root results: node count check, depth, longest list (1 + most children)
        int mc = 0;
        if (tail == expect && !(a.exp & 32))
            for (int i = tid; i < tail; i += B2_T) mc = max(mc, cstart[i + 1] - cstart[i]);
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

}  // namespace gg

using namespace gg;

extern "C" int gg_build_trees_device(gg_ctx *ctx, const int32_t *roots, int32_t n_roots) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, ctx->g_rowptr && !ctx->h_rowptr.empty(), GG_EINVAL, "gg_build_trees_device: call gg_set_graph_csr first");
    GG_CHECK(ctx, n_roots >= 0 && (roots || n_roots == 0), GG_EINVAL, "gg_build_trees_device: bad roots");
    const int n = ctx->n_node;
    for (int r = 0; r < n_roots; ++r) GG_CHECK(ctx, roots[r] >= 0 && roots[r] < n, GG_EINVAL, "gg_build_trees_device: root %d out of range", roots[r]);
    GG_CHECK(ctx, ctx->g_nnz < (1ll << 31), GG_EINVAL, "gg_build_trees_device: %lld adjacency entries (limit 2^31 - 1)", (long long)ctx->g_nnz);
    GG_HIP(ctx, hipSetDevice(ctx->device));
    int rc = alloc_trees(ctx, roots, n_roots, nullptr, nullptr);  // node counts from the (cached) component sweep
    if (rc != GG_OK) return rc;
    if (n_roots == 0) return GG_OK;

    const int grid = std::min<int>(n_roots, ctx->n_cus);
    const int bm_words = (n + 31) / 32;
    // static LDS of the kernel (eoff, e0s, duplicate list, mask, counters) ~ 21.5 KB; the CU has 160 KB
    const bool v1 = getenv("GG_BFS_V1") != nullptr;  // the chunk kernel of rounds 2-3 (kept for A/B timing and as a second witness in the tests)
    hipFuncAttributes fa;
    GG_HIP(ctx, hipFuncGetAttributes(&fa, v1 ? (const void *)bfs_order_kernel<true, false> : (const void *)bfs_order2_kernel<true>));
    const size_t lds_total = 160 * 1024;
    const bool lds_bm = !getenv("GG_BFS_GLOBAL_BITMAP") && fa.sharedSizeBytes + (size_t)bm_words * 4 <= lds_total;

    // scratch kept in the context: an epoch over non-resident roots calls this once per root batch
    DevBuf &gkey = ctx->bfs_key, &gbm = ctx->bfs_bm, &misc = ctx->bfs_misc;
    auto cleanup = [&]() {};
    const size_t key_bytes_before = gkey.bytes;
    hipError_t e = gkey.reserve(sizeof(uint32_t) * (size_t)grid * n);
    if (e == hipSuccess && !lds_bm) e = gbm.reserve(sizeof(uint32_t) * (size_t)grid * bm_words);
    if (e == hipSuccess) e = misc.reserve(sizeof(int32_t) * 8 + sizeof(unsigned long long) * 16);
    if (e != hipSuccess) { cleanup(); return fail(ctx, GG_ENOMEM, "gg_build_trees_device: scratch: %s", hipGetErrorString(e)); }
    BfsArgs a{};
    a.n_node = n;
    a.n_roots = n_roots;
    a.rowptr = ctx->g_rowptr;
    a.col = ctx->g_col;
    a.roots = ctx->t_root;
    a.base = ctx->t_base;
    a.order = ctx->t_order;
    a.cstart = ctx->t_cstart;
    a.edge = ctx->t_edge;
    a.ticket = misc.as<unsigned int>();
    a.stats = misc.as<int32_t>() + 1;
    a.gbitmap = gbm.as<uint32_t>();
    a.gkey = gkey.as<uint32_t>();
    a.bm_words = bm_words;
    const bool prof = getenv("GG_BFS_PROFILE") != nullptr;
    a.exp = getenv("GG_BFS_EXPERIMENT") ? atoi(getenv("GG_BFS_EXPERIMENT")) : 0;
    a.prof = prof ? (unsigned long long *)(misc.as<int32_t>() + 8) : nullptr;
    (void)hipMemsetAsync(misc.p, 0, sizeof(int32_t) * 8 + sizeof(unsigned long long) * 16, ctx->stream);
    if (gkey.bytes != key_bytes_before)  // new allocation: all-ones; the kernel restores every word it uses
        (void)hipMemsetAsync(gkey.p, 0xFF, sizeof(uint32_t) * (size_t)grid * n, ctx->stream);
    (void)hipEventRecord(ctx->ev0, ctx->stream);
    if (!v1) {
        const size_t dyn = lds_bm ? (size_t)bm_words * 4 : 0;
        if (lds_bm) {
            e = hipFuncSetAttribute((const void *)bfs_order2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            if (e != hipSuccess) { cleanup(); return fail(ctx, GG_EHIP, "gg_build_trees_device: %zu bytes of LDS: %s", dyn, hipGetErrorString(e)); }
            hipLaunchKernelGGL((bfs_order2_kernel<true>), dim3(grid), dim3(B2_T), dyn, ctx->stream, a);
        } else {
            hipLaunchKernelGGL((bfs_order2_kernel<false>), dim3(grid), dim3(B2_T), 0, ctx->stream, a);
        }
    } else if (lds_bm) {
        const size_t dyn = (size_t)bm_words * 4;
        const bool instr = prof || a.exp != 0;
        const void *fn = instr ? (const void *)bfs_order_kernel<true, true> : (const void *)bfs_order_kernel<true, false>;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess) { cleanup(); return fail(ctx, GG_EHIP, "gg_build_trees_device: %zu bytes of LDS: %s", dyn, hipGetErrorString(e)); }
        if (instr) hipLaunchKernelGGL((bfs_order_kernel<true, true>), dim3(grid), dim3(BFS_T), dyn, ctx->stream, a);
        else hipLaunchKernelGGL((bfs_order_kernel<true, false>), dim3(grid), dim3(BFS_T), dyn, ctx->stream, a);
    } else if (prof || a.exp != 0) {
        hipLaunchKernelGGL((bfs_order_kernel<false, true>), dim3(grid), dim3(BFS_T), 0, ctx->stream, a);
    } else {
        hipLaunchKernelGGL((bfs_order_kernel<false, false>), dim3(grid), dim3(BFS_T), 0, ctx->stream, a);
    }
    (void)hipEventRecord(ctx->ev1, ctx->stream);
    int32_t stats[3] = {0, 0, 0};
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(stats, a.stats, sizeof(stats), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && prof && !v1) {
        unsigned long long pc[16];
        if (hipMemcpy(pc, a.prof, sizeof(pc), hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[bfs2 profile] %d roots, grid %d: per root %.0f windows (%.0f without a candidate, %.0f with in-window duplicates, %.0f through the key array), "
                    "%.1f full fences, %.0f nodes appended, %.0f quads scanned\n", n_roots, grid, (double)pc[0] / n_roots, (double)pc[4] / n_roots, (double)pc[5] / n_roots,
                    (double)pc[6] / n_roots, (double)pc[1] / n_roots, (double)pc[2] / n_roots, (double)pc[3] / n_roots);
        {
            const char *names[8] = {"setup: node info + scan", "window choice", "quad map", "scan: loads + tests", "scan barrier", "claim", "duplicates + prefix", "append + cstart (empty windows: all behind the scan)"};
            double tot = 0;
            for (int k = 0; k < 8; ++k) tot += (double)pc[8 + k];
            fprintf(stderr, "[bfs2 profile] wave-0 clock per window %.0f:", tot / (double)(pc[0] ? pc[0] : 1));
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.1f%%", names[k], 100.0 * (double)pc[8 + k] / (tot > 0 ? tot : 1));
            fprintf(stderr, "\n");
        }
    }
    if (e == hipSuccess && prof && v1) {
        unsigned long long pc[16];
        if (hipMemcpy(pc, a.prof, sizeof(pc), hipMemcpyDeviceToHost) == hipSuccess) {
            const char *names[8] = {"nodes+scan", "batch form", "map+load+test", "empty chunk", "claim", "dup resolve", "compaction", "stores"};
            double tot = 0;
            for (int k = 0; k < 8; ++k) tot += (double)pc[k];
            fprintf(stderr, "[bfs profile] %d roots, grid %d: batches %llu chunks %llu (empty %llu, with dups %llu); wave-0 shader-clock share:", n_roots, grid, pc[8], pc[9], pc[10], pc[11]);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.1f%%", names[k], 100.0 * (double)pc[k] / (tot > 0 ? tot : 1));
            fprintf(stderr, "; cycles per chunk %.0f\n", tot / (double)(pc[9] ? pc[9] : 1));
        }
    }
    float ms = 0.f;
    if (e == hipSuccess) (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    cleanup();
    if (e != hipSuccess) return fail(ctx, GG_EHIP, "gg_build_trees_device: %s", hipGetErrorString(e));
    if (a.exp) fprintf(stderr, "[bfs experiment %d] kernel %.1f ms for %d roots (results invalid)\n", a.exp, ms, n_roots);
    GG_CHECK(ctx, stats[2] == 0 || (a.exp & ~8), GG_EINVAL, "gg_build_trees_device: a BFS reached a different number of nodes than the component sweep of the graph");
    ctx->tree_max_depth = stats[0];
    ctx->tree_max_list = stats[1];
    ctx->t_edge_valid = (a.exp & ~8) == 0;
    ctx->ctr.bfs_kernel_ms += ms;
    ctx->ctr.bfs_trees += n_roots;
    return GG_OK;
}
