// walk_sample.hip -- K1: the generator's graph-softmax walk sampler on gfx950.
//
// Replaces GraphGAN.sample (reference src/GraphGAN/graph_gan.py:225-270) together with the
// all_score fetch it performs per call (:238; generator.py:21, score = g_cur . g_j + b[j]),
// utils.softmax (src/utils.py:131-133) and np.random.choice (:262).
//
// Structure: all walks of a launch advance one hop at a time.
//   level_advance_kernel        one thread per walk: finish hop h-1 (Philox uniform, threshold, search in the owner's
//                               prefix sums -- or, for <= 16 candidates on a scored node, the distribution evaluated in
//                               registers from the edge-score cache -- path append, termination), prepare hop h (tree list +
//                               the reference's hop rules; a leaf ends the walk right there; in-workgroup dedup of identical
//                               (root, node) distributions; where the scores come from: gather / score the node / private;
//                               chunk offsets, chunk descriptors, task lists)
//   level_score_kernel          one 16-lane group per <= 16-row chunk -- the candidates of a private (root, node)
//                               distribution, or 16 neighbours of a node whose whole adjacency goes into the edge-score
//                               cache once for all roots: rows as float4 (256 B contiguous per row per load), fmaf chain,
//                               xor butterfly (spec S1), + bias.  Dominant, HBM-bound.
//   level_weights_kernel        max, exact fixed-point weights, uint64 prefix sums (S2, S3), once per distribution with more
//                               than 16 candidates: 16-lane groups up to 256, workgroups for hubs; scores from the task's
//                               private region or gathered from the cache through the tree's edge indices
//   walk_sample_kernel          "finisher": one wavefront per walk runs the remaining hops
//                               (GG_WALK_LEVELS < tree depth + 2; = 0: the whole walk)
// Scores follow S1 and everything after them is exact integer arithmetic, so neither the decomposition nor WHO scored a
// shared node can change a sampled node.  Algorithmic bytes of the score kernel: 4(d+2) per candidate row +
// 4d+12 per scoring task (one current row).
#include <algorithm>

#include "gg_arith.h"
#include "gg_internal.h"

namespace gg {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int SCORE_CAP = 1024;  // fp32 scores per wave kept in LDS (finisher)
constexpr int CHUNK = 16;        // neighbours per score work item (= one 16-lane group pass: little divergence between the 4 groups of a wave)

struct WalkArgs {
    const float *E;
    const float *bias;
    int32_t n_node, ld, nchunk;  // nchunk = ld / 4 float4 chunks per row
    const int32_t *t_root;
    const int32_t *t_order;   // BFS-order trees (gg_internal.h): node id of rank i of slot r at t_base[r] + i
    const int32_t *t_cstart;  // first child rank of rank i at t_base[r] + r + i; children = consecutive ranks
    const int64_t *t_base;
    uint32_t *t_q3;           // removed-father bits of the roots' children (quirk Q3), row r at word t_q3off[r]
    const int64_t *t_q3off;
    const int32_t *slots;
    const int64_t *walk_ptr;  // [n_slots + 1]
    int32_t n_slots;
    int64_t total_walks;
    int32_t for_d;
    uint64_t seed;
    uint32_t stream;
    int32_t *samples, *paths, *path_len;
    int32_t stride;
    int32_t *status;       // [n_slots]
    int32_t *first_child;  // [total_walks]  D-mode: rank (>= 1) of the depth-1 child whose father entry this walk removes
    int32_t *abort_walk;   // [n_slots]      D-mode: smallest walk index that hit a leaf child
    float *scratch;        // per-wave score rows for k > SCORE_CAP (finisher)
    int64_t scratch_stride;
    unsigned long long *ctr;  // [0] hops [1] nbr_reads [2] alive walks [3] error flag [4] ticket [5] rows scored
    // walk state carried between levels / into the finisher
    int32_t *st_cur, *st_prev, *st_len, *st_alive;
    int32_t *st_rank;      // BFS rank of st_cur in its root's tree
    int4 *st_const;        // {item, slot, root, walk index inside the root}: fixed per walk, ONE load per hop
    int4 *st_const2;       // {tree base of the slot (t_base), Q3 row of the slot (t_q3off)} as lo / hi words: no slot -> base hop on the chain
    int32_t level;         // hop index handled by this launch (level kernels) / first hop (finisher)
    // per-level tasks
    int64_t *lv_beg;       // absolute index in t_order of the first CHILD candidate
    int32_t *lv_k;         // candidates (father entry included); bit 31: candidate 0 is the father (= the walk's previous node)
    int32_t *lv_chunks;    // 16-candidate chunks this walk owns (0 for non-owners / dead walks)
    int64_t *lv_coff;      // first chunk of the score / prefix region this walk samples from (its owner's)
    float *lv_scores;      // [CHUNK * total chunks]
    int4 *lv_chunk_desc;   // [total chunks] {cur node, rows | flags | offset bits 32..47, offset bits 0..31, father id} of chunk c
    uint64_t *lv_prefix;   // [CHUNK * chunks of the launch(es)] inclusive prefix sums of the fixed-point weights; regions are addressed
                           // by GLOBAL chunk offsets: launch base + chunks of the earlier levels + offset inside the level
    int64_t *lv_pfx;       // [total_walks] global chunk offset of the prefix sums this walk samples from
    int64_t cap_total;     // chunks the prefix buffer holds
    // Distribution cache: the generator's tables do not change between the D-mode and the G-mode walks
    // of a step, so a (root slot, node rank, father flag) distribution the D launch evaluated is NOT evaluated again by
    // the G launch: D owners register {key -> global prefix offset, k} in a hash table, G walks look their node up.
    int32_t dc_mode;       // 0 = off, 1 = register (D launch), 2 = look up (G launch)
    ulonglong2 *dc_tab;    // {key, value} entries: one 16-byte read per probe
    uint32_t dc_mask;      // table size - 1 (power of two)
    const int64_t *dc_words;  // [0] global chunk offset this launch starts at, [1] chunks in the buffer after the D launch
    int32_t *lv_big;       // [lv_big_cap] task lists of the weights kernel, per level: hub tasks from the front, small multi-chunk tasks from the back
    // A launch may run as two halves of its walks on two streams (run_levels): each half has its own walk range, task
    // list, chunk buffers, prefix region and level counters `lc`; flags, finisher list and finisher counters are shared.
    unsigned long long *lc;  // this half's counter block: indices CTR_ALIVE and up
    int64_t w0, w_end;       // walks [w0, w_end) of the launch
    int64_t lv_big_cap;
    int32_t *fin_list;       // [total_walks] walks still alive behind the last streamed level (CTR_FIN entries)
    // Edge-score cache (gg_internal.h): the score of graph edge e = (u -> col[e]) is the same for every root, so a node's
    // adjacency is scored ONCE per generator state (stamped in es_stamp[u]) and every (root, u) distribution gathers its
    // candidates' scores through the tree's edge indices: child rank i -> es[t_edge[i]], father -> es[g_rev[t_edge[rank(u)]]].
    const int64_t *rowptr;
    const int32_t *col;
    const int32_t *t_edge;
    const int32_t *rev;
    float *es;
    long long *es_stamp;     // tick of the score kernel that fills the node's adjacency scores (gg_internal.h)
    long long es_now;        // tick of THIS level's score kernel of this half
    long long es_valid_from; // older stamps predate the generator's current tables
    int32_t es_mode;         // 0 = off, 1 = a stale node is scored whole when the asking distribution needs most of it, 2 = always
    int32_t es_ratio, es_hub;
    int32_t *lv_fe;          // [total_walks] es index of the father candidate's score (gather tasks with a father entry)
    int32_t exp;             // GG_WALK_EXPERIMENT: timing ablations (results are then WRONG): 1 / 2 = the weights kernel skips its big / small tasks, 64 / 32 = runs them twice (results stay right), 8 = per-level row counts
    // LAZY trees (gg_internal.h "LAZY trees", bfs_gpu.hip): slot r is exact through level L_r; the children list of a rank >= lzs_r
    // is resolved by the first walk that stands on it ("LAZY RESOLUTION" below) and appended to the slot's pool
    int32_t lazy;
    const int4 *lz_info;          // [slots] {lzs: first rank without a built children list, exact ranks, their level L, capacity of the segment}
    unsigned long long *lz_pair;  // at t_base[r] + rank: start << 32 | count << 12 | stamp
    const int32_t *lz_rank;       // at t_base[r] + member index: BFS rank
    const uint2 *lz_bm;           // [slots][lz_words] {visited word, members below the word}
    int32_t lz_words;
    int32_t *lz_cursor;           // [slots] first free rank of the pool
    int32_t *lz_flag;             // [slots] the walks need the whole tree of this slot
    uint32_t lz_stamp;
    int32_t g_multi;              // some adjacency list holds a node twice: first-occurrence tests needed
    int32_t *lz_list;             // [n_slots] launch items with claimed walks at this level, then [n_slots] claims per item (zero between launches)
    int32_t *t_order_w, *t_edge_w;  // the tree arrays, writable (pool appends)
    int32_t lz_coop_min;          // adjacencies longer than this are resolved by the whole workgroup (GG_LZ_COOP_MIN)
    int32_t lz_budget;            // 256-entry scan batches one list may cost; a list that needs more sends its slot to the whole-tree rebuild (bounds a launch's tail)
    unsigned long long *lz_ctr;   // statistics: [depth] lists resolved at depth 0 / 1 / 2, [3] candidates judged, [4] 16-entry scan rounds, [5] most rounds of one list
};

// ---- cross-lane moves without an LDS round trip (DPP): shifts / rotations inside a row of 16 lanes, row broadcasts across rows.
// (On the product path of every walk kernel since round 5: the bit-exact walk tests -- tests/test_gpu_walk.py, all seven
// decompositions -- run through them.  row_bcast:15 / :31 and row_newbcast exist on gfx9-class targets only.)
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "walk_sample.hip uses gfx9 DPP row broadcasts: build for gfx950 (ARCH in the Makefile)"
#endif
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v) {  // both halves moved by the same control; lanes without a source get 0
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t group16_incl_scan_u64_dpp(uint64_t v) {  // inclusive prefix sum inside every row of 16 lanes
    v += dpp_u64<0x111>(v);  // row_shr:1
    v += dpp_u64<0x112>(v);  // row_shr:2
    v += dpp_u64<0x114>(v);  // row_shr:4
    v += dpp_u64<0x118>(v);  // row_shr:8
    return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, int /*lane*/) {
    v = group16_incl_scan_u64_dpp(v);
    v += dpp_u64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_u64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false)); }
__device__ __forceinline__ float group16_max_f32(float v) {  // maximum over every row of 16 lanes, in all its lanes (rotations: every lane has a source)
    v = fmaxf(v, dpp_f32<0x128>(v));  // row_ror:8
    v = fmaxf(v, dpp_f32<0x124>(v));
    v = fmaxf(v, dpp_f32<0x122>(v));
    v = fmaxf(v, dpp_f32<0x121>(v));
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = group16_max_f32(v);
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

__device__ __forceinline__ int find_item(const int64_t *walk_ptr, int n_slots, int64_t w) {
    int lo = 0, hi = n_slots;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (walk_ptr[mid] <= w) lo = mid; else hi = mid;
    }
    return lo;
}

// Scores of up to 64 candidates ids[0..nblock) against the row of `cur` (spec S1), handed to
// store(j, score).  The whole wave participates; group q = lane >> 4 handles candidates q, q+4, ...
template <int NCH, class Store>
__device__ __forceinline__ float score_block(const WalkArgs &a, const float4 (&gc)[NCH], const int32_t *ids, int first_id, int nblock,
                                             int lane, Store store) {
    const int t = lane & 15, q = lane >> 4;
    // first_id >= 0: candidate 0 of this block is the father entry (not stored in the children range ids[1..])
    const int myid = (lane < nblock) ? ((lane == 0 && first_id >= 0) ? first_id : ids[lane]) : 0;
    float mx = -INFINITY;
    for (int s = 0; s < nblock; s += 16) {
        float4 y[4][NCH];
        int id[4];
        bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int jj = s + u * 4 + q;
            valid[u] = jj < nblock;
            id[u] = __shfl(myid, jj & 63, 64);
            const float4 *const row = (const float4 *)(a.E + (int64_t)id[u] * a.ld);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = t + 16 * c;
                y[u][c] = (valid[u] && ch < a.nchunk) ? row[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                acc = __builtin_fmaf(gc[c].x, y[u][c].x, acc);
                acc = __builtin_fmaf(gc[c].y, y[u][c].y, acc);
                acc = __builtin_fmaf(gc[c].z, y[u][c].z, acc);
                acc = __builtin_fmaf(gc[c].w, y[u][c].w, acc);
            }
            acc = acc + __shfl_xor(acc, 8, 64);
            acc = acc + __shfl_xor(acc, 4, 64);
            acc = acc + __shfl_xor(acc, 2, 64);
            acc = acc + __shfl_xor(acc, 1, 64);
            if (valid[u]) {
                const float sc = acc + a.bias[id[u]];
                mx = fmaxf(mx, sc);
                if (t == 0) store(s + u * 4 + q, sc);
            }
        }
    }
    return mx;
}

template <int NCH>
__device__ __forceinline__ void load_cur_row(const WalkArgs &a, int cur, int lane, float4 (&gc)[NCH]) {
    const int t = lane & 15;
    const float4 *const crow = (const float4 *)(a.E + (int64_t)cur * a.ld);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = t + 16 * c;
        gc[c] = (ch < a.nchunk) ? crow[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// Exact inverse-CDF pick among k scores sbuf(0..k) (spec S2, S3, S5); all lanes return idx.
template <class Load>
__device__ __forceinline__ int sample_index(Load sbuf, int k, float mx, uint64_t m53, int lane) {
    int idx = 0;
    if (k <= 64) {
        const uint64_t wgt = (lane < k) ? weight_fix40(exp_spec(sbuf(lane) - mx)) : 0ull;
        const uint64_t C = wave_incl_scan_u64(wgt, lane);
        const uint64_t W = __shfl(C, 63, 64);
        const uint64_t thr = threshold(m53, W);
        const unsigned long long bal = __ballot(C > thr);
        idx = __ffsll((long long)bal) - 1;
    } else {
        uint64_t part = 0;
        for (int jj = lane; jj < k; jj += 64) part += weight_fix40(exp_spec(sbuf(jj) - mx));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
        if (part == 0ull) return -1;  // no weight at all: non-finite scores (the caller raises the error flag)
        const uint64_t thr = threshold(m53, part);
        uint64_t carry = 0;
        for (int j0 = 0; j0 < k; j0 += 64) {
            const int jj = j0 + lane;
            const uint64_t wgt = (jj < k) ? weight_fix40(exp_spec(sbuf(jj) - mx)) : 0ull;
            const uint64_t C = carry + wave_incl_scan_u64(wgt, lane);
            const unsigned long long bal = __ballot(C > thr);
            if (bal) { idx = j0 + __ffsll((long long)bal) - 1; break; }
            carry = __shfl(C, 63, 64);
        }
    }
    return idx;
}

// ------------------------------------------------------------------------------------------
// LAZY RESOLUTION (round 6): children lists the BFS did not build, identical to the ones it would have built.
//
// Slot r is exact through level L: the visited set V (levels <= L) with the BFS rank of every member.  The FIFO BFS
// (graph_gan.py:96-107) appends a node when the first queue node adjacent to it is popped, at the first occurrence of
// the node in that adjacency; a level's queue order is (rank of the father, index of the appending edge).  So for a walk
// standing on `cur`:
//   depth 0 (cur in level L, rank known): w in adj(cur) is a child iff w is not in V (then all its neighbours in V are in
//           level L) and no neighbour of w in V has a smaller rank than cur;
//   depth 1 (cur in level L + 1, a pool node; its key = (rank of its father f, index of the edge f -> cur)): x in adj(cur) is
//           a child iff x is not in V, has no neighbour in V (else it is in level L + 1) and no neighbour y of x that is in
//           level L + 1 -- y has a neighbour in V; its key = (smallest rank among them, index of that father's edge to y,
//           through the reverse-edge index) -- has a smaller key;
//   depth 2 (cur in level L + 2): only "is it a leaf": a neighbour that is not in V, has no neighbour in V and no neighbour
//           in level L + 1 would be a child -- then the slot asks for its whole tree (lz_flag; walk_finalize rebuilds it in
//           the arena and reruns the launch); so does a pool that is full.
// Children are entries of adj(cur) in adjacency order, first occurrences only: the list the BFS builds, bit for bit.
// One WAVEFRONT per node: 64 adjacency entries of cur per round; the adjacencies of the candidates among them (and, below cur's
// own level, of THEIR neighbours) are tested as one flat stream of entries, 256 at a time (lz_flat_scan).
// ------------------------------------------------------------------------------------------
constexpr unsigned long long LZ_CLAIMED = 0xFFFFFull;  // count field of a pair whose list is being resolved
constexpr int CTR_LZ_FB = 2;                           // ctr[2]: a slot raised its flag: the launch is void, the host rebuilds the flagged slots whole and repeats it
__device__ __forceinline__ unsigned long long lz_make(int start, unsigned long long count, uint32_t stamp) {
    return ((unsigned long long)(uint32_t)start << 32) | (count << 12) | (unsigned long long)stamp;
}
// where the visited words of the slot come from: the index in global memory (any kernel), or the copy a workgroup of the
// resolve kernel holds in LDS for the slot it is working on -- a list costs ~1 000 membership tests, each a random sector of a
// 250 KB index of ITS slot; 16 384 slots' indices share no cache line: from HBM that was the whole run time of the kernel
struct LzBitsGlobal {
    const uint2 *bm;
    __device__ __forceinline__ uint32_t operator()(int i) const { return bm[i].x; }
    // member index of node u (its visited word wx): members below its word + below its bit
    __device__ __forceinline__ int index(int u, uint32_t wx) const { return (int)bm[u >> 5].y + __popc(wx & ((1u << (u & 31)) - 1u)); }
};
struct LzBitsLds {
    const uint32_t *w;  // [lz_words] visited words in LDS
    const uint2 *bm;    // (the index in global memory: members below a word.  Measured and dropped: members below every 16th word in
                        // LDS as well + popcounts of up to 15 LDS words -- 72 instead of 61 ms per 16 384 roots: the LDS reads cost more than the one load)
    __device__ __forceinline__ uint32_t operator()(int i) const { return w[i]; }
    __device__ __forceinline__ int index(int u, uint32_t wx) const { return (int)bm[u >> 5].y + __popc(wx & ((1u << (u & 31)) - 1u)); }
};
template <class Bits>
__device__ __forceinline__ bool lz_in(const Bits &bits, int x) { return (bits(x >> 5) >> (x & 31)) & 1u; }
// Per-wavefront LDS workspace of a resolution.
struct LzWork {
    int32_t pre[64], q0[64];          // inner segments: inclusive prefix of their lengths, first CSR entry
    int32_t xpre[64], xq0[64];        // outer segments (the candidates whose neighbours become pairs)
    unsigned long long best[64];      // per pair: smallest (rank << 32 | CSR entry) over the pair node's neighbours in V
    unsigned long long mask[2];       // per candidate lane: [0] ruled out / touches V, [1] has a neighbour in level L + 1
};
__device__ __forceinline__ void wave_lds_sync() {  // LDS writes (stores and atomics, through the generic workspace pointer) of this wavefront's lanes are visible to its other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (a wavefront-scope fence emits no wait at all: with the statistics' atomics gone, reads overtook the writes they depend on)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
// element `flat` of the concatenation of 64 segments (inclusive length prefix in LDS): its segment
__device__ __forceinline__ int lz_owner(const int32_t *pre, int flat) {
    int lo = 0, hi = 63;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int mid = (lo + hi) >> 1;
        if (pre[mid] > flat) hi = mid; else lo = mid + 1;
    }
    return lo;
}
// Every lane brings one segment of adjacency (first CSR entry, length; 0 = none).  ALL entries of all 64 segments are tested
// against V, 64 x U at a time with the U batches' loads in flight together -- a list's latency no longer grows with the number
// or the degrees of its candidates (judged one 16-entry round at a time, a list was a chain of ~70 memory round trips):
//   MODE 0  owners with a neighbour in V whose rank is below `below`  -> bit in ws->mask[0]
//   MODE 1  owners with any neighbour in V                            -> bit in ws->mask[0]
//   MODE 2  per owner the smallest (rank << 32 | CSR entry) over its neighbours in V -> ws->best[owner]
// The caller initialises mask / best.  `work` counts the batches (the list's budget).
template <int MODE, class Bits>
__device__ __forceinline__ void lz_flat_scan(const WalkArgs &a, const Bits &bits, const uint2 *bm, const int32_t *rk, LzWork *ws, int lane, int my_q0, int my_len,
                                             int below, int &work) {
    constexpr int U = 4;
    const int incl = wave_incl_scan_i32(my_len);
    ws->pre[lane] = incl;
    ws->q0[lane] = my_q0;
    const int T = __builtin_amdgcn_readlane(incl, 63);
    wave_lds_sync();
    for (int f0 = 0; f0 < T && work <= a.lz_budget; f0 += 64 * U) {
        int own[U], e[U], u[U];
        uint32_t wx[U], pr[U];
        bool vis[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int flat = f0 + 64 * k + lane;
            own[k] = -1;
            e[k] = 0;
            if (flat < T) {
                own[k] = lz_owner(ws->pre, flat);
                e[k] = ws->q0[own[k]] + (flat - (own[k] ? ws->pre[own[k] - 1] : 0));
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) u[k] = own[k] >= 0 ? a.col[e[k]] : 0;
#pragma unroll
        for (int k = 0; k < U; ++k) {
            wx[k] = own[k] >= 0 ? bits(u[k] >> 5) : 0u;
            vis[k] = (wx[k] >> (u[k] & 31)) & 1u;
        }
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (vis[k]) atomicOr(&ws->mask[0], 1ull << own[k]);
        } else {
#pragma unroll
            for (int k = 0; k < U; ++k) pr[k] = vis[k] ? (uint32_t)bits.index(u[k], wx[k]) : 0u;
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (vis[k]) {
                    const int r = rk[pr[k]];
                    if (MODE == 0) {
                        if (r < below) atomicOr(&ws->mask[0], 1ull << own[k]);
                    } else {
                        atomicMin(&ws->best[own[k]], ((unsigned long long)(uint32_t)r << 32) | (unsigned long long)(uint32_t)e[k]);
                    }
                }
            }
        }
        work += 1;
    }
    wave_lds_sync();
}

// Resolve the children list of (slot, rank) = node `cur` on tree level `level` (its father on the walk: `prev`) and publish it;
// the caller has claimed the pair.  Returns the published pair.  All 64 lanes take part; everything returned is wave-uniform.
// FENCE: readers of the pair may run in the SAME kernel (the finisher's walks): the list is made visible before the pair that names
// it.  The resolve kernel's readers are later kernels -- and an agent-scope release on this chip writes the XCD's L2 back
// (buffer_wbl2) and invalidates it: per list, that was ~60 % of the resolve kernel's time.
// COOP (the resolve kernel's big lists, adjacencies of > LZ_COOP_MIN entries): ALL wavefronts of the workgroup call this for the
// same list; wavefront wv judges the 64-entry chunks wv, wv + NW, .. (each privately, as above), the children masks meet in LDS,
// and every wavefront writes the children of its own chunks behind the prefix of the masks' popcounts.  A walk through a hub
// costs its list one round of chunks per NW instead of one chunk after the other on ONE wavefront while the item's other 15 wait:
// in a 12 ms launch of the resolve kernel an item (root) took 187 us, most of it such a list.
constexpr int LZ_COOP_CHUNKS = 256, LZ_COOP_MIN = 256;  // (adjacencies up to 16 384 entries; longer ones stay with one wavefront)
struct LzCoop {
    unsigned long long kids[LZ_COOP_CHUNKS];
    int work, fb, start;
};
template <bool FENCE, class Bits, bool COOP = false>
__device__ __forceinline__ unsigned long long lazy_resolve_wave(const WalkArgs &a, const Bits &bits, LzWork *ws, int slot, int64_t tbase, int rank, int cur, int prev,
                                                                int level, int lane, int wv = 0, int nw = 1, LzCoop *sh = nullptr) {
    // Every argument is wave-uniform; as scalars the loops below branch on scalar conditions.
    // NO "if (lane == 0)" BRANCHES in this function or around its call: inlined into the resolve kernel's loop over the listed
    // walks, the lane-0-only store of the pair at its end was merged with the loop's latch -- lane 0 LEFT the loop after its first
    // list while lanes 1-63 went round for ever on entry 0 (EXEC = ...fffe at the function's entry, its LDS stores missing; found
    // on the GPU, in the disassembly).  Uniform stores are made by every lane (same address, same value: one store), counters
    // are added to with (lane == 0 ? value : 0).
    slot = __builtin_amdgcn_readfirstlane(slot);
    rank = __builtin_amdgcn_readfirstlane(rank);
    cur = __builtin_amdgcn_readfirstlane(cur);
    prev = __builtin_amdgcn_readfirstlane(prev);
    level = __builtin_amdgcn_readfirstlane(level);
    tbase = ((int64_t)__builtin_amdgcn_readfirstlane((int)(tbase >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tbase);
    const int4 info = a.lz_info[slot];
    const int depth = level - info.z, seg = info.w;
    const uint2 *const bm = a.lz_bm + (size_t)slot * a.lz_words;
    const int32_t *const rk = a.lz_rank + tbase;
    unsigned long long *const pair = a.lz_pair + tbase + rank;
    const int64_t e0 = a.rowptr[cur], e1 = a.rowptr[cur + 1];
    bool fallback = depth < 0 || depth > 2;
    // the key of cur inside its level
    int my_rank = rank, my_edge = 0;
    if (depth == 1) {
        const uint2 wd = bm[prev >> 5];
        my_rank = rk[wd.y + __popc(wd.x & ((1u << (prev & 31)) - 1u))];  // (the father of a level L + 1 node is in V)
        my_edge = a.t_edge[tbase + rank];
    }
    auto candidate = [&](int64_t e, int &x) -> bool {  // entry e of adj(cur): a node outside V, at its first occurrence in the list
        x = -1;
        if (e >= e1) return false;
        x = a.col[e];
        if (x == cur || lz_in(bits, x)) return false;
        return !a.g_multi || (a.exp & 32768) || a.rev[a.rev[e]] == (int32_t)e;  // (exp 16384 / 32768 / 65536: timing ablations of the resolution -- no pair stage / no first-occurrence tests / no scans; results WRONG)
    };
    // candidates of the whole adjacency: what the list can hold at most -> its place in the pool
    int x0 = -1, ncand = 0;
    bool c0 = false;
    if (!fallback && !COOP) {
        c0 = candidate(e0 + lane, x0);
        ncand = (int)__popcll(__ballot(c0));
        for (int64_t b = e0 + 64; b < e1; b += 64) {
            int x;
            ncand += (int)__popcll(__ballot(candidate(b + lane, x)));
        }
    }
    int start = 0;
    if (!COOP && depth <= 1 && ncand > 0) {
        start = __builtin_amdgcn_readfirstlane(atomicAdd(&a.lz_cursor[slot], lane == 0 ? ncand : 0));
        if ((long long)start + ncand > (long long)seg) fallback = true;  // the pool is full
    }
    int count = 0, work = 0;
    // one 64-entry chunk of adj(cur) (lane's entry: candidate c, node x; m = ballot of c): which entries are children
    auto judge = [&](int x, bool c, unsigned long long m) -> unsigned long long {
            // the candidates' adjacencies, one segment per lane
            int64_t xq = 0, xe = 0;
            if (c) { xq = a.rowptr[x]; xe = a.rowptr[x + 1]; }
            ws->mask[0] = 0ull;
            ws->mask[1] = 0ull;
            wave_lds_sync();
            unsigned long long kids;
            if (depth == 0) {
                // a child unless a neighbour in V precedes cur
                if (!(a.exp & 65536)) lz_flat_scan<0>(a, bits, bm, rk, ws, lane, (int)xq, (int)(xe - xq), my_rank, work);
                kids = m & ~ws->mask[0];
            } else {
                // candidates that touch V are in cur's own level; the others are one level below cur
                if (!(a.exp & 65536)) lz_flat_scan<1>(a, bits, bm, rk, ws, lane, (int)xq, (int)(xe - xq), 0, work);
                const unsigned long long below_m = (a.exp & 16384) ? 0ull : m & ~ws->mask[0];
                wave_lds_sync();  // (every lane has read the mask)
                ws->mask[0] = 0ull;
                // their neighbours y (PAIRS (x, y), 64 at a time): is y in level L + 1, and with which key?
                const bool xb = (below_m >> lane) & 1ull;
                const int xincl = wave_incl_scan_i32(xb ? (int)(xe - xq) : 0);
                ws->xpre[lane] = xincl;
                ws->xq0[lane] = (int)xq;
                const int P = __builtin_amdgcn_readlane(xincl, 63);
                wave_lds_sync();
                for (int p0 = 0; p0 < P && work <= a.lz_budget; p0 += 64) {
                    const int pi = p0 + lane;
                    int xl = -1, y = -1;
                    int64_t yq = 0, ye = 0;
                    if (pi < P) {
                        xl = lz_owner(ws->xpre, pi);
                        y = a.col[ws->xq0[xl] + (pi - (xl ? ws->xpre[xl - 1] : 0))];
                        if (y == cur) y = -1;  // (cur itself: level L + 1 with cur's own key at depth 1, not in level L + 1 at depth 2)
                    }
                    if (y >= 0) { yq = a.rowptr[y]; ye = a.rowptr[y + 1]; }
                    ws->best[lane] = ~0ull;
                    wave_lds_sync();
                    lz_flat_scan<2>(a, bits, bm, rk, ws, lane, (int)yq, (int)(ye - yq), 0, work);
                    const unsigned long long best = ws->best[lane];
                    if (y >= 0 && best != ~0ull) {  // y is in level L + 1
                        atomicOr(&ws->mask[1], 1ull << xl);
                        if (depth == 1) {
                            const int ry = (int)(best >> 32);
                            if (ry < my_rank || (ry == my_rank && a.rev[(int)(uint32_t)best] < my_edge)) atomicOr(&ws->mask[0], 1ull << xl);  // y precedes cur in the queue
                        }
                    }
                    wave_lds_sync();
                }
                if (depth == 1) {
                    kids = below_m & ~ws->mask[0];
                } else {
                    kids = 0ull;
                    if (below_m & ~ws->mask[1]) fallback = true;  // a neighbour two levels below the exact ones' children: cur is no leaf
                }
            }
            return kids;
    };
    if (!COOP) {
        if (!fallback && ncand > 0) {
            for (int64_t b = e0; b < e1 && work <= a.lz_budget; b += 64) {
                int x = x0;
                bool c = c0;
                if (b != e0) c = candidate(b + lane, x);
                const unsigned long long m = __ballot(c);
                if (m) {
                    const unsigned long long kids = judge(x, c, m);
                    if ((kids >> lane) & 1ull) {
                        const int64_t at = tbase + start + count + (int)__popcll(kids & ((1ull << lane) - 1ull));
                        a.t_order_w[at] = x;
                        a.t_edge_w[at] = (int32_t)(b + lane);
                    }
                    count += (int)__popcll(kids);
                    wave_lds_sync();  // (the masks are read: the next round resets them)
                }
            }
        }
    } else {
        // this wavefront's chunks; every value that decides a branch below is workgroup-uniform or private to the wavefront
        const int n_chunks = (int)((e1 - e0 + 63) >> 6);
        if (!fallback) {
            for (int ch = wv; ch < n_chunks && work <= a.lz_budget; ch += nw) {
                int x;
                const bool c = candidate(e0 + 64 * (int64_t)ch + lane, x);
                const unsigned long long m = __ballot(c);
                unsigned long long kids = 0ull;
                if (m) {
                    kids = judge(x, c, m);
                    wave_lds_sync();
                }
                sh->kids[ch] = kids;  // (every lane: the same word)
            }
        }
        atomicAdd(&sh->work, lane == 0 ? work : 0);
        if (fallback || work > a.lz_budget) sh->fb = 1;
        __syncthreads();
        work = sh->work;
        fallback = sh->fb != 0;
        // prefix of the chunks' children (every wavefront for itself: lane l sums chunks 4 l .. 4 l + 3)
        int mine[4], tot4 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mine[k] = 4 * lane + k < n_chunks ? (int)__popcll(sh->kids[4 * lane + k]) : 0;
            tot4 += mine[k];
        }
        const int incl = wave_incl_scan_i32(tot4);
        count = __builtin_amdgcn_readlane(incl, 63);
        if (!fallback && depth <= 1 && count > 0) {
            if (wv == 0) {
                const int st = __builtin_amdgcn_readfirstlane(atomicAdd(&a.lz_cursor[slot], lane == 0 ? count : 0));
                sh->start = st;  // (every lane: the same word)
            }
            __syncthreads();
            start = sh->start;
            if ((long long)start + count > (long long)seg) fallback = true;  // the pool is full
            if (!fallback) {
                ws->pre[lane] = incl - tot4;  // children in front of chunk 4 lane
                wave_lds_sync();
                for (int ch = wv; ch < n_chunks; ch += nw) {
                    const unsigned long long kids = sh->kids[ch];
                    if (kids) {
                        int x;
                        (void)candidate(e0 + 64 * (int64_t)ch + lane, x);
                        int off = ws->pre[ch >> 2];
                        for (int k = 0; k < (ch & 3); ++k) off += (int)__popcll(sh->kids[(ch & ~3) + k]);
                        if ((kids >> lane) & 1ull) {
                            const int64_t at = tbase + start + off + (int)__popcll(kids & ((1ull << lane) - 1ull));
                            a.t_order_w[at] = x;
                            a.t_edge_w[at] = (int32_t)(e0 + 64 * (int64_t)ch + lane);
                        }
                    }
                }
                wave_lds_sync();
            }
        }
    }
    // a list this expensive (a hub's adjacency judged through hubs) is not worth resolving: the slot gets its whole tree instead
    fallback = fallback || work > a.lz_budget;
    if (fallback) {
        count = 0;
        a.lz_flag[slot] = 1;
        // the launch is void -- the host rebuilds the flagged slots whole and repeats it -- but it runs to its end: every slot
        // whose walks need the whole tree is found in ONE pass (cut short, each repeat would find one level's worth)
        a.ctr[CTR_LZ_FB] = 1ull;
    }
    if (a.lz_ctr) {  // (GG_LZ_STATS=1: same-address atomics of every list of the launch -- ~40 ms per 16 384 roots)
        const unsigned long long one = (lane == 0 && wv == 0) ? 1ull : 0ull;
        atomicAdd(&a.lz_ctr[depth < 0 ? 0 : depth > 2 ? 2 : depth], one);
        atomicAdd(&a.lz_ctr[3], one * (unsigned long long)ncand);
        atomicAdd(&a.lz_ctr[4], one * (unsigned long long)work);
        atomicMax(&a.lz_ctr[5], (unsigned long long)work);
        atomicMax(&a.lz_ctr[6], (unsigned long long)(e1 - e0));
        if (COOP) atomicAdd(&a.lz_ctr[7], one);  // lists resolved by a whole workgroup
    }
    const unsigned long long v = lz_make(start, (unsigned long long)count, a.lz_stamp);
    if (FENCE) __threadfence();  // the list before the pair that names it
    __hip_atomic_store(pair, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (every lane: the same word)
    return v;
}

// The children range of a rank >= lzs for a whole wavefront (the finisher): resolved on the spot if nobody has, waited for if
// another wave is at it (that wave is running: claims are only ever taken by running waves).
__device__ __forceinline__ unsigned long long lazy_children_wave(const WalkArgs &a, LzWork *ws, int slot, int64_t tbase, int rank, int cur, int prev, int level, int lane) {
    unsigned long long *const pair = a.lz_pair + tbase + rank;
    for (int spins = 0;; ++spins) {
        const unsigned long long v = __hip_atomic_load(pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(v & 0xFFFull) == a.lz_stamp) {
            if (((v >> 12) & 0xFFFFFull) != LZ_CLAIMED) {
                __threadfence();
                return v;
            }
            if (spins > (1 << 20)) {  // (a claim nobody is working on -- cannot happen; if it does, the slot gets its whole tree)
                if (lane == 0) { a.lz_flag[slot] = 1; a.ctr[CTR_LZ_FB] = 1ull; }
                return lz_make(0, 0ull, a.lz_stamp);
            }
            __builtin_amdgcn_s_sleep(16);
            continue;
        }
        unsigned long long old = v;
        if (lane == 0) old = atomicCAS(pair, v, lz_make(0, LZ_CLAIMED, a.lz_stamp));
        old = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(old >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)old, 0, 64);
        if (old == v) return lazy_resolve_wave<true>(a, LzBitsGlobal{a.lz_bm + (size_t)slot * a.lz_words}, ws, slot, tbase, rank, cur, prev, level, lane);
    }
}

// ------------------------------------------------------------------------------------------
// Level kernels (streaming front end)
// ------------------------------------------------------------------------------------------
constexpr int SINGLE_CHUNK = 0x100;  // descriptor flag: the task has one chunk -> the score kernel finishes its prefix sums itself
// Persistent grid of the score kernel: 6 of the 8 wave slots per SIMD.  The two free slots let kernels of the other
// stream (the discriminator update that runs beside the generator's walks) start at once instead of queueing behind
// a launch that holds the whole chip for hundreds of microseconds; the stream of rows in flight is deep enough
// either way.  GG_SCORE_BLOCKS overrides.
static int score_blocks() {
    static const int v = [] {
        const char *e = getenv("GG_SCORE_BLOCKS");
        int b = e ? atoi(e) : 256 * 6;
        if (b < 8) b = 8;
        return b - b % 8;
    }();
    return v;
}
#ifndef GG_BIG_TASK
#define GG_BIG_TASK 256
#endif
constexpr int BIG_TASK = GG_BIG_TASK;   // owner tasks with more candidates get a whole workgroup for their prefix sums
constexpr int CTR_NONFINITE = 6; // ctr[6]: a distribution had total weight 0 (non-finite generator scores) -> GG_EINVAL
constexpr int CTR_FIN = 7;      // ctr[7]: walks still alive behind the last streamed level = entries of the finisher's walk list
constexpr int CTR_ALIVE = 8;    // ctr[CTR_ALIVE + level]: walks alive when hop `level` was set up
constexpr int CTR_BIG = 72;     // ctr[CTR_BIG + level]:   owner tasks of hop `level` that need the weights kernel: big ones (k > BIG_TASK) in
                                //                         the low 32 bits, small multi-chunk ones in the high 32 bits (one atomic hands out both)
constexpr int CTR_CHUNKS = 136; // ctr[CTR_CHUNKS + level]: chunks of hop `level`, two index spaces handed out by one atomic: low 32 bits = PREFIX
                                //                         chunks (every owner: ceil(k / 16), addresses lv_prefix), high 32 bits = SCORE chunks (descriptors /
                                //                         lv_scores: private owners ceil(k / 16), node scorings ceil(deg / 16), gather-only owners none)
constexpr int CTR_ROWS = 200;   // ctr[CTR_ROWS + (block & 63)]: candidate rows scored by this launch (spread words, folded by the host)
constexpr int CTR_HOPS_V = 264;  // ctr[CTR_HOPS_V + (block & 63)]: hop counts of the level pipeline, spread over 64 words
constexpr int CTR_READS_V = 328; // (same-address atomics serialise at ~12 ns each); summed by the host
constexpr int CTR_DISTS = 392;   // ctr[CTR_DISTS + (block & 63)]: (root, node) distributions set up by the level pipeline (owners), spread words
constexpr int CTR_TINY = 456;    // (64 spare words)
constexpr int CTR_GATHER = 520;  // ctr[CTR_GATHER + (block & 63)]: owner distributions that gather their scores from the edge-score cache, spread words
constexpr int CTR_NODES = 584;   // ctr[CTR_NODES + (block & 63)]: nodes whose adjacency this launch scored into the cache, spread words
constexpr int CTR_BASE = 648;    // ctr[CTR_BASE + level]: global chunk offset of the first PREFIX chunk of hop `level` (launch base + the prefix chunks
                                 //                        of the earlier hops): [0] by the reset kernel, [level + 1] by the score kernel of `level`
constexpr int CTR_WORDS = 720;
static_assert(CTR_WORDS == gg_ctx::CTR_WORDS, "counter layout");
constexpr int MAX_LEVELS = 64;
constexpr int LVK_GATHER = 1 << 30;  // lv_k flag: the distribution's scores come from the edge-score cache
constexpr int LVK_NODE = 1 << 29;    // lv_k flag: ... and this owner scores the node's adjacency (its score chunks are node chunks)
constexpr int LVK_SELF = 1 << 28;    // lv_k flag: <= 16 candidates on a scored node: the WALK gathers the scores itself when it samples (no owner, no task, no prefix sums)
constexpr int LVK_MASK = (1 << 28) - 1;

// Global chunk offset of the first prefix chunk of hop a.level (one word; it used to be a loop over the earlier levels'
// counters -- up to `level` dependent reads at the head of every level kernel).
__device__ __forceinline__ int64_t level_chunk_base(const WalkArgs &a) { return (int64_t)a.lc[CTR_BASE + a.level]; }

__device__ __forceinline__ unsigned long long dc_key(int slot, int rank, int hf) {
    return ((unsigned long long)(unsigned)slot << 32) | ((unsigned long long)(unsigned)rank << 1) | (unsigned long long)hf;
}

// Descriptor of score chunk c: two int4 {cur, rows | flags, offset bits 0..31, offset bits 32..63} {father id, prefix chunk
// bits 0..31, bits 32..63, 0}.  Private chunk i of a k-candidate distribution: candidate j of the distribution is the father
// (j == 0, only if hf) or the child order[beg_abs + j - hf]; `offset` is the t_order index of the chunk's candidate 0 (for
// the first chunk of a list with a father entry that is ONE BEFORE the first child -- never dereferenced: lane 0 takes the
// father id from the descriptor).  Node chunk i of node cur: its graph neighbours col[offset .. offset + rows), scores to
// es[offset ..].  `prefix chunk` (single-chunk private distributions): where the score kernel writes the prefix sums.
constexpr int DESC_HAS_FATHER = 0x200;
constexpr int DESC_NODE = 0x400;
__device__ __forceinline__ void write_chunk_desc(int4 *desc, int64_t c, int cur, int k, int hf, int father, int64_t beg_abs, int i, int64_t pfx) {
    const int64_t o = beg_abs + (int64_t)i * CHUNK - hf;
    const int flags = (k <= CHUNK ? SINGLE_CHUNK : 0) | ((hf && i == 0) ? DESC_HAS_FATHER : 0);
    desc[2 * c] = make_int4(cur, min(CHUNK, k - i * CHUNK) | flags, (int)(o & 0xffffffffll), (int)(o >> 32));
    desc[2 * c + 1] = make_int4(father, (int)(pfx & 0xffffffffll), (int)(pfx >> 32), 0);
}
__device__ __forceinline__ void write_node_desc(int4 *desc, int64_t c, int cur, int deg, int64_t e0, int i) {
    const int64_t o = e0 + (int64_t)i * CHUNK;
    desc[2 * c] = make_int4(cur, min(CHUNK, deg - i * CHUNK) | DESC_NODE, (int)(o & 0xffffffffll), (int)(o >> 32));
    desc[2 * c + 1] = make_int4(-1, 0, 0, 0);
}

// One thread per walk (a wave = 64 consecutive walks), fused per hop boundary:
//   do_sample: finish hop (level-1) -- Philox uniform, threshold, binary search in the owner's
//              prefix sums (first j with C_j > floor(m W / 2^53), spec S4/S5), path append,
//              termination (next == previous);
//   do_setup : prepare hop (level) -- tree list of (root, cur) with the reference's hop rules
//              (root-only-children, Q2 abort, Q3 father removal) and the in-workgroup dedup: walks of
//              one root standing on the same node need the SAME distribution -> one owner.
//              do_setup == 2 (the launch's last advance): the hop rules only -- a walk that has just reached a leaf
//              steps back and ends right there, like in every other round -- no distributions are set up.
// The kernel is ONE CHAIN OF DEPENDENT RANDOM READS per walk at two to three wavefronts per SIMD: its run time is the length
// of that chain (a level with 1 800 live walks took as long as one with 160 000: ~35 us = 16 round trips).  So every load
// is issued as early as its address is known, independent of the branches around it -- per-walk state before the liveness
// test, the tree row / graph row / stamp of the picked node before the dedup decides who needs them -- the per-walk
// constants include the tree base and the Q3 row (no slot -> base hop), the level's prefix base is one word (written by the
// previous score kernel) instead of a loop over the earlier levels' counters, the dedup is an LDS hash instead of a
// backward scan (up to 255 dependent LDS reads for a hub root's walks), and the block's returning atomics go out from
// three lanes at once instead of one after the other: 6-7 round trips.
__device__ __forceinline__ uint32_t dc_hash(unsigned long long key, uint32_t mask) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & mask; }

// LAZY (trees exact through a level only, see "LAZY RESOLUTION"): on a level where a walk may stand on a node without a built
// children list the kernel runs as two launches around the resolve kernel -- do_setup == 0: finish hop level-1 only, store the
// walk state, claim the picked node's list if nobody has resolved it (the claimed walks go to lz_list); do_sample == 2: the
// walk state is read back and hop `level` is set up as always, its children range taken from the node's pair.
constexpr int CTR_LZ = 456;  // lc[CTR_LZ + level]: walks listed for the resolve kernel of `level` (the spare words of CTR_TINY)
template <bool LAZY>
__global__ __launch_bounds__(256) void level_advance_kernel(const WalkArgs a, const int do_sample, const int do_setup, const int write_desc,
                                                            const int64_t cap_chunks) {
    __shared__ int blk_skip;
    __shared__ long long blk_lbase;
    __shared__ long long hk[512];   // dedup hash: key (item, cur) -> ...
    __shared__ int ho[512];         // ... smallest thread index that holds it
    __shared__ long long blk_coff[256];
    __shared__ unsigned char blk_self[256];
    const int tid = (int)threadIdx.x;
    const int64_t w = a.w0 + (int64_t)blockIdx.x * blockDim.x + tid;
    const int lane = tid & 63;
    const bool in_range = w < a.w_end;
    // ---- loads that need nothing but the walk index (issued before the block learns whether it runs at all)
    int alive_w = 1, kraw = 0, len = 0, cur0 = -1, prev0 = -1, fe0 = -1;
    int4 sc = make_int4(0, 0, 0, 0), sc2 = make_int4(0, 0, 0, 0);
    int64_t pfx0 = 0, beg0 = 0;
    int rank0 = 0;
    if (in_range && do_sample) {
        alive_w = a.st_alive[w];
        sc = a.st_const[w];
        sc2 = a.st_const2[w];
        len = a.st_len[w];
        cur0 = a.st_cur[w];
        prev0 = a.st_prev[w];
        if (!LAZY || do_sample == 1) {
            kraw = a.lv_k[w];
            pfx0 = a.lv_pfx[w];
            beg0 = a.lv_beg[w];
            if (a.es_mode) fe0 = a.lv_fe[w];
        } else {
            rank0 = a.st_rank[w];
        }
    }
    // Block-uniform early exit: other blocks of this very launch may raise flag 2 (speculative overflow below, a walk
    // still alive after the last level), so every thread testing the global word itself could split a workgroup in
    // front of the barriers further down.  One thread reads, all threads test the shared copy.
    if (tid == 0) {
        const unsigned long long flag = __atomic_load_n(&a.ctr[3], __ATOMIC_RELAXED);
        const unsigned long long prev_alive = a.level > 0 ? __atomic_load_n(&a.lc[CTR_ALIVE + a.level - 1], __ATOMIC_RELAXED) : 1ull;
        const unsigned long long lb = __atomic_load_n(&a.lc[CTR_BASE + a.level], __ATOMIC_RELAXED);
        blk_skip = (flag == 2ull ||        // an earlier level overflowed: the host reruns in sized mode
                    prev_alive == 0ull)   // every walk has finished: empty level
                       ? 1 : 0;
        blk_lbase = (long long)lb;         // global chunk offset of this hop's first prefix chunk (launch base + earlier hops)
    }
    hk[tid] = -1ll; hk[tid + 256] = -1ll;
    ho[tid] = 0x7fffffff; ho[tid + 256] = 0x7fffffff;
    __syncthreads();
    if (blk_skip) return;
    bool alive = false, sampled = false, forced = false, claimed = false;
    int item = 0, cur = -1, k = 0, hf = 0, father = -1, slot_w = 0, rank_w = 0, up_edge = -1;
    unsigned long long my_k = 0;
    int64_t beg_abs = 0;
    // graph row / stamp of the node the walk stands on: loaded for every live walk as soon as the node is known (only the
    // distribution's owner uses them, but waiting for the dedup first would put them behind it on the chain)
    int64_t e0 = 0, e1 = 0;
    long long st = 0;
    int fe = -1;
    if (in_range && alive_w != 0) {
        int slot, root, j, rank = 0;
        int64_t tbase, q3off;
        if (!do_sample) {
            item = find_item(a.walk_ptr, a.n_slots, w);
            slot = a.slots[item];
            root = a.t_root[slot];
            j = (int)(w - a.walk_ptr[item]);
            tbase = a.t_base[slot];
            q3off = a.t_q3off[slot];
            a.st_const[w] = make_int4(item, slot, root, j);
            a.st_const2[w] = make_int4((int)(tbase & 0xffffffffll), (int)(tbase >> 32), (int)(q3off & 0xffffffffll), (int)(q3off >> 32));
        } else {
            item = sc.x; slot = sc.y; root = sc.z; j = sc.w;
            tbase = ((int64_t)sc2.y << 32) | (unsigned)sc2.x;
            q3off = ((int64_t)sc2.w << 32) | (unsigned)sc2.z;
        }
        if (!do_sample) {  // level 0: start the walk
            alive = true;
            cur = root;
            a.st_prev[w] = -1;
            a.st_len[w] = 1;
            a.paths[w * (int64_t)a.stride] = root;
            if (a.for_d) a.first_child[w] = -1;
        } else if (LAZY && do_sample == 2) {  // hop level-1 was finished by the launch in front of the resolve kernel
            alive = true;
            cur = cur0;
            father = prev0;
            rank = rank0;
            len -= 1;  // (entries of the path before that hop's append, as the fused kernel counts below)
        } else {
            alive = true;
            sampled = true;
            const int kk = kraw & LVK_MASK, hf0 = (int)((unsigned)kraw >> 31);
            my_k = (unsigned long long)kk;
            int lo = 0;
            if (kk == 1) {
                // ONE candidate (a leaf of the tree: its list is [father]; or an only child): it is picked whatever its score
                // and the uniform are -- weight 2^40 of 2^40 -- so nothing was scored for it (at the deep levels of a
                // small-world tree most walks stand on leaves: 47 k one-row score chunks per level-4 launch of the bench).
                // (A non-finite score of such a candidate no longer raises GG_EINVAL by itself; any distribution with two
                // candidates still does.)
            } else if (kraw & LVK_SELF) {
                // <= 16 candidates on a node whose adjacency is in the edge-score cache: gather the scores through the tree's
                // edge indices and evaluate the distribution right here -- max, exact fixed-point weights, running sum, first
                // j whose sum exceeds the threshold (spec S2-S5: the same integers as a search in stored prefix sums)
                const int32_t *const edges = a.t_edge + beg0 - hf0;
                int e[CHUNK];
                float v[CHUNK];
#pragma unroll
                for (int i = 0; i < CHUNK; ++i) e[i] = (i < kk) ? ((i == 0 && hf0) ? fe0 : edges[i]) : -1;
#pragma unroll
                for (int i = 0; i < CHUNK; ++i) v[i] = (e[i] >= 0) ? a.es[e[i]] : -INFINITY;
                float mx = v[0];
#pragma unroll
                for (int i = 1; i < CHUNK; ++i) mx = fmaxf(mx, v[i]);
                uint64_t Wtot = 0;
#pragma unroll
                for (int i = 0; i < CHUNK; ++i) Wtot += (i < kk) ? weight_fix40(exp_spec(v[i] - mx)) : 0ull;
                if (Wtot == 0ull) a.ctr[CTR_NONFINITE] = 1ull;
                const uint64_t thr = threshold(uniform53(a.seed, a.stream, (uint32_t)root, (uint32_t)j, (uint32_t)(a.level - 1)), Wtot);
                uint64_t C = 0;
                int cnt = 0;  // candidates whose inclusive sum is <= thr: the pick is the first one above
#pragma unroll
                for (int i = 0; i < CHUNK; ++i) {
                    C += (i < kk) ? weight_fix40(exp_spec(v[i] - mx)) : 0ull;
                    cnt += (i < kk && C <= thr) ? 1 : 0;
                }
                lo = min(cnt, kk - 1);
            } else {
            const uint64_t *const pf = a.lv_prefix + pfx0 * CHUNK;
            // first j with C_j > thr by a 16-ary search: 15 independent pivot loads per round, ceil(log16 k)
            // dependent rounds instead of log2 k (a hub hop was a chain of 12+ dependent random reads); the
            // first round's pivots do not depend on the threshold and fly together with W = C_{k-1}
            int n = kk;  // invariant: the answer lies in [lo, lo + n) and C_{lo+n-1} > thr
            int step = (n + 15) >> 4;
            uint64_t piv[15];
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                const int idx = i * step - 1;
                piv[i - 1] = (idx < n - 1) ? pf[idx] : ~0ull;
            }
            const uint64_t Wtot = pf[kk - 1];
            if (Wtot == 0ull) a.ctr[CTR_NONFINITE] = 1ull;  // non-finite scores: the search below stays inside [0, kk), the host reports the error
            const uint64_t thr = threshold(uniform53(a.seed, a.stream, (uint32_t)root, (uint32_t)j, (uint32_t)(a.level - 1)), Wtot);
            int seg = 0;
#pragma unroll
            for (int i = 0; i < 15; ++i) seg += (piv[i] <= thr) ? 1 : 0;  // C is non-decreasing: a prefix of the pivots
            lo = seg * step;
            n = min(step, n - seg * step);
            while (n > 1) {
                step = (n + 15) >> 4;
#pragma unroll
                for (int i = 1; i < 16; ++i) {
                    const int idx = i * step - 1;
                    piv[i - 1] = (idx < n - 1) ? pf[lo + idx] : ~0ull;
                }
                seg = 0;
#pragma unroll
                for (int i = 0; i < 15; ++i) seg += (piv[i] <= thr) ? 1 : 0;
                lo += seg * step;
                n = min(step, n - seg * step);
            }
            }
            // Candidate 0 of a list with a father entry IS the previous node (walks only move down the tree until
            // their back-step), so the terminating condition (:264-266) needs no load: the walk ends iff it picked
            // that entry.  Otherwise the pick is a child: its rank follows from the index alone, and the setup of
            // the next hop (the cstart pair of that rank) does not wait for the node id.
            const bool back = hf0 && lo == 0;
            const int64_t pick = beg0 + lo - hf0;  // index in t_order (children only)
            const int nxt = back ? prev0 : a.t_order[pick];
            if (len >= a.stride) {
                a.ctr[3] = 1ull;
                a.path_len[w] = 0;
                a.samples[w] = -1;
                alive = false;
            } else {
                a.paths[w * (int64_t)a.stride + len] = nxt;
                if (back) {                   // next == previous: sample = cur
                    a.path_len[w] = len + 1;
                    a.samples[w] = cur0;
                    alive = false;
                } else {
                    a.st_len[w] = len + 1;
                    a.st_prev[w] = cur0;
                    father = cur0;
                    cur = nxt;
                    rank = (int)(pick - tbase);
                }
            }
        }
        int4 lzi = make_int4(0x7fffffff, 0, 0, 0);
        if (LAZY && alive) lzi = a.lz_info[slot];
        if (LAZY && alive && do_setup == 0 && rank >= lzi.x) {
            // the picked node has no built children list: claim it unless somebody has (resolved or claimed)
            unsigned long long *const pair = a.lz_pair + tbase + rank;
            const unsigned long long v = __hip_atomic_load(pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(v & 0xFFFull) != a.lz_stamp && atomicCAS(pair, v, lz_make(0, LZ_CLAIMED, a.lz_stamp)) == v) {
                claimed = true;
                // the first claim of a launch item lists the item for the resolve kernel (low half of the level's word: items listed)
            }
        }
        if (alive && do_setup) {
            const int32_t *const cs = a.t_cstart + tbase + (LAZY ? 0 : slot);  // (lazy builds place a slot's row at its base: the segments come from a cursor, in no slot order)
            int cbeg, cend;  // children of cur = ranks [cbeg, cend)
            if (LAZY && rank >= lzi.x) {
                const unsigned long long v = __hip_atomic_load(a.lz_pair + tbase + rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long cnt = (v >> 12) & 0xFFFFFull;
                cbeg = (int)(v >> 32);
                cend = cbeg + (int)cnt;
                if ((uint32_t)(v & 0xFFFull) != a.lz_stamp || cnt == LZ_CLAIMED) {  // (not resolved: cannot happen -- the slot gets its whole tree)
                    a.lz_flag[slot] = 1;
                    a.ctr[CTR_LZ_FB] = 1ull;
                    cbeg = cend = 0;
                }
            } else {
                cbeg = cs[rank];
                cend = cs[rank + 1];
            }
            unsigned q3w = 0u;
            if (a.level == 1) q3w = a.t_q3[q3off + ((rank - 1) >> 5)];
            if (a.es_mode) {
                if (a.level > 0) up_edge = a.t_edge[tbase + rank];  // the edge (father -> cur)
                e0 = a.rowptr[cur];
                e1 = a.rowptr[cur + 1];
                st = __hip_atomic_load(&a.es_stamp[cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (up_edge >= 0) fe = a.rev[up_edge];  // s(cur, father) sits at the reverse of that edge, inside adj(cur)
            }
            const int nchild = cend - cbeg;
            // list of cur (graph_gan.py:250): the root's is children only (tree[root][1:]); any other node's is
            // [father] ++ children unless D-mode removed the father entry of this depth-1 child earlier (Q3)
            hf = 1;
            if (a.level == 0) hf = 0;
            else if (a.level == 1 && ((q3w >> ((rank - 1) & 31)) & 1u)) hf = 0;
            k = nchild + hf;
            bool aborted = false;
            if (k == 0) {                          // "the tree only has a root" (:252-253)
                aborted = true;
                if (a.for_d) atomicMin(&a.abort_walk[item], 0);
                else a.status[item] = GG_ROOT_ABORTED;
            } else if (a.for_d && a.level == 1 && hf) {  // the list still starts with the root
                if (nchild == 0) {                 // node_neighbor == [root] (:255-257)
                    atomicMin(&a.abort_walk[item], j);
                    aborted = true;
                } else {                           // node_neighbor.remove(root) (:258-259), applied by the post-pass
                    a.first_child[w] = rank;
                    hf = 0;
                    k -= 1;
                }
            }
            if (aborted) {
                alive = false;
                k = 0;
                a.path_len[w] = 0;
                a.samples[w] = -1;
            }
            // A LEAF: the list is [father] alone, so the next hop is the back-step and the walk ends (:264-266) -- finished right
            // here instead of in another round of level kernels (at the deep levels of a small-world tree that is nearly every
            // walk: 59 k of the bench's walks were alive at hop 4 only to step back).  The hop's uniform is not needed (counter
            // RNG: nothing to consume); the hop and its one candidate are counted like a sampled hop.
            if (alive && k == 1 && hf) {
                const int len_now = a.level == 0 ? 1 : len + 1;  // entries in the path so far
                if (len_now >= a.stride) {
                    a.ctr[3] = 1ull;
                    a.path_len[w] = 0;
                    a.samples[w] = -1;
                } else {
                    a.paths[w * (int64_t)a.stride + len_now] = father;
                    a.path_len[w] = len_now + 1;
                    a.samples[w] = cur;
                }
                alive = false;
                k = 0;
                forced = true;
            }
            beg_abs = tbase + cbeg;
        }
        if (alive) {
            a.st_cur[w] = cur;
            a.st_rank[w] = rank;  // also behind the last streamed level: the finisher resumes from it
        }
        slot_w = slot;
        rank_w = rank;
        // the sync-free launch ran only as many levels as earlier launches needed: a walk that is
        // still going after the last one sends the launch to the sized rerun
        if (alive && do_setup != 1 && write_desc == 2) a.ctr[3] = 2ull;
    }
    if (in_range) a.st_alive[w] = alive ? (claimed ? 2 : 1) : 0;  // (2: the resolve kernel owes this walk's node its children list)
    {
        const unsigned long long bal = __ballot(sampled), fbal = __ballot(forced);  // hops sampled here + leaf back-steps finished here
        my_k += forced ? 1ull : 0ull;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) my_k += __shfl_xor(my_k, off, 64);
        if (lane == 0 && (bal | fbal)) {
            atomicAdd(&a.lc[CTR_HOPS_V + (blockIdx.x & 63)], (unsigned long long)(__popcll(bal) + __popcll(fbal)));
            atomicAdd(&a.lc[CTR_READS_V + (blockIdx.x & 63)], my_k);
        }
    }
    if (LAZY && do_setup == 0) return;  // (the resolve kernel and the set-up launch follow)
    if (do_setup != 1) {
        // behind the last streamed level: the walks that are still going, as a compact list for the finisher (it used to
        // draw one ticket per WALK, finished or not: 160 k same-address atomics, ~2 ms, to find a few thousand live walks)
        const unsigned long long abal = __ballot(alive);
        if (abal) {
            unsigned long long base = 0;
            if (lane == __ffsll((long long)abal) - 1) base = atomicAdd(&a.ctr[CTR_FIN], (unsigned long long)__popcll(abal));
            base = __shfl(base, __ffsll((long long)abal) - 1, 64);
            if (alive) a.fin_list[base + __popcll(abal & ((1ull << lane) - 1ull))] = (int32_t)w;
        }
        return;
    }
    // <= 16 candidates on a node whose adjacency is (or is being) scored into the edge-score cache: the walk evaluates its
    // distribution itself when it samples (LVK_SELF) -- no owner, no chunks, no prefix sums, nothing for the weights kernel
    // (and a distribution with ONE candidate needs nothing at all: see the sampling branch above)
    const bool self_early = alive && (k == 1 || (a.es_mode && k <= CHUNK && st >= a.es_valid_from && st <= a.es_now));
    // G launch: was this very distribution evaluated by the D launch of the step?  (Same slot, same node, same
    // father flag => same candidate list; same generator tables => same prefix sums, bit for bit.)
    bool cached = false;
    int64_t cached_off = 0;
    if (alive && a.dc_mode == 2 && !self_early) {
        const unsigned long long key = dc_key(slot_w, rank_w, hf);
        uint32_t h = dc_hash(key, a.dc_mask);
        for (int tries = 0; tries < 64; ++tries) {
            const ulonglong2 ent = a.dc_tab[h];  // {key, value}: one 16-byte read per probe
            if (ent.x == key) {
                if ((int)(ent.y & 0xffffffffull) == k) { cached = true; cached_off = (int64_t)(ent.y >> 32); }
                break;
            }
            if (ent.x == ~0ull) break;
            h = (h + 1) & a.dc_mask;
        }
    }
    // dedup inside the workgroup (256 consecutive walks): the walk with the smallest index among those with the same
    // (item, cur) owns the distribution -- an LDS hash (insert with compare-and-swap, owner with atomic-min).
    const long long key = (alive && !cached && !self_early) ? (((long long)item << 32) | (unsigned)cur) : -1ll;
    int hs = -1;
    if (key >= 0) {
        int s = (int)(((unsigned long long)key * 0x9E3779B97F4A7C15ull) >> 55);  // 9 bits
        for (;;) {
            const long long old = (long long)atomicCAS((unsigned long long *)&hk[s], ~0ull, (unsigned long long)key);
            if (old == -1ll || old == key) break;
            s = (s + 1) & 511;
        }
        atomicMin(&ho[s], tid);
        hs = s;
    }
    __syncthreads();
    const int owner = hs >= 0 ? ho[hs] : tid;
    const bool owns = alive && !cached && !self_early && owner == tid;
    // ---- where do this distribution's scores come from?  (edge-score cache, see WalkArgs)
    //   gather : adj(cur) has been scored since the generator last changed (this level by another root, an earlier level,
    //            or the D launch of the step) -> no rows at all, the weights kernel gathers k four-byte scores;
    //   node   : stale, and this distribution needs most of adj(cur) anyway (k * ratio >= deg; or a hub): the first owner to
    //            swing the stamp scores the WHOLE adjacency into the cache (node chunks), everybody else gathers;
    //   private: stale and k << deg (deep levels: most neighbours already belong to other subtrees): score the k
    //            candidates only, as before.
    // Which of two racing owners wins a node only decides WHO scores it: the scores (spec S1) and therefore the walks are
    // the same in every outcome.
    int mode = 0;
    const int deg = (int)(e1 - e0);
    if (owns && a.es_mode) {
        if (st >= a.es_valid_from) {
            // scored (or about to be, by a score kernel that is ordered before this level's weights kernel): gather.  A stamp
            // from the FUTURE belongs to the other half of a split launch (its later score kernel): score privately.
            mode = st <= a.es_now ? 1 : 0;
        } else if (a.es_mode == 2 || (int64_t)k * a.es_ratio >= (int64_t)deg || (a.es_hub > 0 && deg >= a.es_hub)) {
            const long long old = (long long)atomicCAS((unsigned long long *)&a.es_stamp[cur], (unsigned long long)st, (unsigned long long)a.es_now);
            mode = old == st ? 2 : ((old >= a.es_valid_from && old <= a.es_now) ? 1 : 0);
        }
    }
    // an owner that gathers <= 16 candidates needs nothing but (mode 2) the node chunks: its walks evaluate the distribution themselves
    const bool self_owner = owns && mode != 0 && k <= CHUNK;
    blk_self[tid] = self_owner ? 1 : 0;
    const int p_chunks = (owns && !self_owner) ? (k + CHUNK - 1) / CHUNK : 0;                      // prefix region
    const int s_chunks = !owns ? 0 : (mode == 0 ? p_chunks : (mode == 2 ? (deg + CHUNK - 1) / CHUNK : 0));  // score chunks
    const bool big = owns && k > BIG_TASK;
    const bool small = owns && p_chunks > 1 && !big;  // 16 < k <= BIG_TASK: a 16-lane group of the weights kernel (private single chunks are finished by the score kernel)
    // chunk offsets and task slots: in-wave exclusive scans, per-block totals through LDS, and the block's returning
    // atomics issued by three lanes AT ONCE (a single word serves only ~88 returning atomics per us, and three of them one
    // after the other were three round trips); the order of the blocks' regions in the buffers is irrelevant
    __shared__ int wv_pch[4], wv_sch[4], wv_big[4], wv_own[4], wv_small[4], wv_gat[4], wv_node[4], wv_alive[4], wv_p24[4];
    __shared__ unsigned long long blk_base[3];
    int inc_p = p_chunks, inc_s = s_chunks;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int op = __shfl_up(inc_p, off, 64), os = __shfl_up(inc_s, off, 64);
        if (lane >= off) { inc_p += op; inc_s += os; }
    }
    const int wv = tid >> 6;
    const unsigned long long big_bal = __ballot(big), small_bal = __ballot(small), p24_bal = __ballot(small && mode == 0 && p_chunks <= 4);
    const unsigned long long own_bal = __ballot(owns && mode != 1);  // tasks that read a current row: private owners + node scorings
    // distributions served from the cache: gather tasks of the weights kernel + walks that will gather themselves (counted per
    // walk: nothing dedups them; the walks behind a self-gathering owner are not counted)
    const unsigned long long gat_bal = __ballot((owns && mode != 0) || (self_early && k > 1)), node_bal = __ballot(owns && mode == 2), alive_bal = __ballot(alive);
    if (lane == 63) { wv_pch[wv] = inc_p; wv_sch[wv] = inc_s; }
    if (lane == 0) {
        wv_big[wv] = __popcll(big_bal); wv_small[wv] = __popcll(small_bal);
        wv_own[wv] = __popcll(own_bal); wv_gat[wv] = __popcll(gat_bal); wv_node[wv] = __popcll(node_bal); wv_alive[wv] = __popcll(alive_bal);
        wv_p24[wv] = __popcll(p24_bal);
    }
    __syncthreads();
    if (tid < 7) {
        auto tot = [&](const int *v) { return (unsigned long long)(v[0] + v[1] + v[2] + v[3]); };
        unsigned long long *word = nullptr;
        unsigned long long val = 0;
        if (tid == 0) { word = &a.lc[CTR_CHUNKS + a.level]; val = tot(wv_pch) | (tot(wv_sch) << 32); }
        else if (tid == 1) { word = &a.lc[CTR_BIG + a.level]; val = tot(wv_big) | (tot(wv_small) << 32); }
        else if (tid == 2) { word = &a.lc[CTR_TINY + a.level]; val = ((a.exp & 16) && !LAZY) ? tot(wv_p24) : 0; }  // (GG_WALK_EXPERIMENT & 16: small tasks that score their own 17..64 candidates -- what a "finish in the score kernel" wave task would take off the weights kernel's list; the word is the resolve ticket of lazy launches)
        else if (tid == 3) { word = &a.lc[CTR_DISTS + (blockIdx.x & 63)]; val = tot(wv_own); }
        else if (tid == 4) { word = &a.lc[CTR_GATHER + (blockIdx.x & 63)]; val = tot(wv_gat); }
        else if (tid == 5) { word = &a.lc[CTR_NODES + (blockIdx.x & 63)]; val = tot(wv_node); }
        else { word = &a.lc[CTR_ALIVE + a.level]; val = tot(wv_alive); }  // walks still alive at this hop
        if (tid < 3) blk_base[tid] = val ? atomicAdd(word, val) : 0ull;
        else if (val) atomicAdd(word, val);
    }
    __syncthreads();
    int pch_before = 0, sch_before = 0, big_before = 0, small_before = 0, blk_pch = 0, blk_sch = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < wv) { pch_before += wv_pch[i]; sch_before += wv_sch[i]; big_before += wv_big[i]; small_before += wv_small[i]; }
        blk_pch += wv_pch[i];
        blk_sch += wv_sch[i];
    }
    const int64_t base_p = (int64_t)(blk_base[0] & 0xffffffffull), base_s = (int64_t)(blk_base[0] >> 32);
    const int64_t coff_p_own = base_p + pch_before + inc_p - p_chunks;
    const int64_t coff_s = base_s + sch_before + inc_s - s_chunks;
    blk_coff[tid] = coff_p_own;
    __syncthreads();
    const int64_t coff_p = blk_coff[owner];  // non-owners sample from their owner's region
    const bool self = self_early || (alive && !cached && blk_self[owner]);
    const int64_t lbase = blk_lbase;
    const bool fits = base_s + blk_sch <= cap_chunks && lbase + base_p + blk_pch <= a.cap_total;
    if (write_desc == 1 && !fits && tid == 0) a.ctr[3] = 2ull;  // speculative capacity exceeded: the host reruns in sized mode
    if (in_range) {
        a.lv_beg[w] = beg_abs;
        a.lv_k[w] = k | (hf << 31) | (mode != 0 ? LVK_GATHER : 0) | (mode == 2 ? LVK_NODE : 0) | (self ? LVK_SELF : 0);
        a.lv_chunks[w] = s_chunks;
        a.lv_coff[w] = coff_s;
        a.lv_pfx[w] = cached ? cached_off : lbase + coff_p;
        if (mode != 0 || self) a.lv_fe[w] = hf ? fe : -1;
        if (owns && !self_owner && a.dc_mode == 1 && (fits || write_desc == 0)) {  // D launch: register the distribution for the G launch of the step (sized mode: the buffers are sized after this kernel)
            const unsigned long long key2 = dc_key(slot_w, rank_w, hf);
            uint32_t h = dc_hash(key2, a.dc_mask);
            for (int tries = 0; tries < 64; ++tries) {
                const unsigned long long old = atomicCAS(&a.dc_tab[h].x, ~0ull, key2);
                if (old == ~0ull) { a.dc_tab[h].y = ((unsigned long long)(lbase + coff_p) << 32) | (unsigned long long)(unsigned)k; break; }
                if (old == key2) break;  // another workgroup registered the same distribution: one copy is enough
                h = (h + 1) & a.dc_mask;
            }
        }
        // the task lists of the weights kernel: big tasks from the front of lv_big, small ones from its back
        if (big) a.lv_big[(blk_base[1] & 0xffffffffull) + big_before + __popcll(big_bal & ((1ull << lane) - 1ull))] = (int32_t)w;
        if (small) a.lv_big[a.lv_big_cap - 1 - (int64_t)((blk_base[1] >> 32) + small_before + __popcll(small_bal & ((1ull << lane) - 1ull)))] = (int32_t)w;
        if (write_desc == 1 && fits) {
            if (mode == 2)
                for (int i = 0; i < s_chunks; ++i) write_node_desc(a.lv_chunk_desc, coff_s + i, cur, deg, e0, i);
            else
                for (int i = 0; i < s_chunks; ++i) write_chunk_desc(a.lv_chunk_desc, coff_s + i, cur, k, hf, father, beg_abs, i, lbase + coff_p);
        }
    }
}

// chunk descriptors (one thread per walk; owners describe their chunks): everything the score
// kernel needs arrives with ONE 16-byte load per chunk instead of a chain of dependent loads
// (sized mode only: in the sync-free mode level_advance_kernel writes them itself)
__global__ void level_expand_kernel(const WalkArgs a) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.total_walks) return;
    const int n = a.lv_chunks[w];
    if (n == 0) return;
    const int64_t c0 = a.lv_coff[w];
    const int kraw = a.lv_k[w], cur = a.st_cur[w];
    const int k = kraw & LVK_MASK, hf = (int)((unsigned)kraw >> 31);
    if (kraw & LVK_NODE) {
        const int64_t e0 = a.rowptr[cur];
        const int deg = (int)(a.rowptr[cur + 1] - e0);
        for (int i = 0; i < n; ++i) write_node_desc(a.lv_chunk_desc, c0 + i, cur, deg, e0, i);
        return;
    }
    const int father = hf ? a.st_prev[w] : -1;
    const int64_t beg = a.lv_beg[w], pfx = a.lv_pfx[w];
    for (int i = 0; i < n; ++i) write_chunk_desc(a.lv_chunk_desc, c0 + i, cur, k, hf, father, beg, i, pfx);
}

// LAZY: the children lists the walks of this level were claimed for ("LAZY RESOLUTION").  One workgroup per launch item (a root
// slot and its walks) at a time, items from a ticket: the claimed walks of the item (st_alive == 2) are listed in LDS, the slot's
// visited words are copied into LDS (LDS_BITS: graphs up to ~1.2 M nodes; else the tests read the index in global memory), and
// the workgroup's wavefronts take one listed walk each until the list is empty -- except walks that stand on a node with a LONG
// adjacency (> lz_coop_min entries: a hub): those are set aside and resolved afterwards by ALL wavefronts together, chunk by
// chunk (lazy_resolve_wave<.., .., COOP>).  Runs whatever the launch's flags say: a claim that stayed unresolved would stall
// every later reader of the pair.
constexpr int LZ_T = 1024, LZ_LIST = 1024;
template <bool LDS_BITS>
__global__ __launch_bounds__(LZ_T) void lazy_resolve_kernel(const WalkArgs a) {
    extern __shared__ uint32_t lz_lds[];  // [lz_words] when LDS_BITS
    __shared__ int32_t s_list[LZ_LIST];
    __shared__ LzWork s_ws[LZ_T / 64];
    __shared__ int s_item, s_n, s_next, s_nbig;
    __shared__ int32_t s_big[LZ_LIST];  // listed walks whose node has a long adjacency: resolved by the whole workgroup (LzCoop)
    __shared__ LzCoop s_coop;
    const int tid = threadIdx.x, lane = tid & 63, wvi = __builtin_amdgcn_readfirstlane(tid >> 6);
    LzWork *const ws = &s_ws[tid >> 6];
    for (;;) {
        if (tid == 0) {
            // the level's word: items listed (low half, final: the claiming launch is through) | ticket (high half)
            int it = (int)atomicAdd(&a.lc[CTR_LZ + a.level], 1ull);
            if (it >= a.n_slots) it = -1;
            s_item = it;
            s_n = 0;
            s_next = 0;
            s_nbig = 0;
        }
        __syncthreads();
        const int item = s_item;
        if (item < 0) return;
        const int64_t w0 = a.walk_ptr[item], w1 = a.walk_ptr[item + 1];
        for (;;) {  // (rounds of LZ_LIST claimed walks; one round unless the root has thousands of walks)
            for (int64_t w = w0 + tid; w < w1; w += LZ_T)
                if (a.st_alive[w] == 2) {
                    const int i = atomicAdd(&s_n, 1);
                    if (i < LZ_LIST) {
                        s_list[i] = (int32_t)(w - w0);
                        a.st_alive[w] = 1;
                    }
                }
            __syncthreads();
            const int found = s_n, n = min(found, LZ_LIST);
            if (n > 0) {
                const int slot = a.slots[item];
                if (LDS_BITS && !(a.exp & 8192)) {  // (GG_WALK_EXPERIMENT & 8192 / 4096: timing ablations -- no copy / no resolution; results are WRONG)
                    const uint2 *const src = a.lz_bm + (size_t)slot * a.lz_words;
                    for (int i0 = tid; i0 < a.lz_words; i0 += 8 * LZ_T) {  // (eight loads in flight per thread: one at a time this copy was 31 memory round trips per slot)
                        uint32_t wv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) wv[u] = i0 + u * LZ_T < a.lz_words ? src[i0 + u * LZ_T].x : 0u;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (i0 + u * LZ_T < a.lz_words) lz_lds[i0 + u * LZ_T] = wv[u];
                    }
                    __syncthreads();
                }
                for (;;) {
                    const int i = __builtin_amdgcn_readfirstlane(atomicAdd(&s_next, lane == 0 ? 1 : 0));  // (a scalar: the loop branches on it; no lane-0 branch, see lazy_resolve_wave)
                    if (i >= n) break;
                    if (a.exp & 4096) continue;
                    const int64_t w = w0 + s_list[i];
                    const int cur = __builtin_amdgcn_readfirstlane(a.st_cur[w]);
                    const int64_t deg = a.rowptr[cur + 1] - a.rowptr[cur];
                    if (deg > a.lz_coop_min && deg <= 64 * LZ_COOP_CHUNKS && !(a.exp & 131072)) {  // (GG_WALK_EXPERIMENT & 131072: every list by one wavefront)
                        const int j = __builtin_amdgcn_readfirstlane(atomicAdd(&s_nbig, lane == 0 ? 1 : 0));
                        s_big[j] = s_list[i];  // (every lane: the same word)
                    } else {
                        const int4 sc2 = a.st_const2[w];
                        const int64_t tbase = ((int64_t)sc2.y << 32) | (unsigned)sc2.x;
                        if (LDS_BITS) (void)lazy_resolve_wave<false>(a, LzBitsLds{lz_lds, a.lz_bm + (size_t)slot * a.lz_words}, ws, slot, tbase, a.st_rank[w], cur, a.st_prev[w], a.level, lane);
                        else (void)lazy_resolve_wave<false>(a, LzBitsGlobal{a.lz_bm + (size_t)slot * a.lz_words}, ws, slot, tbase, a.st_rank[w], cur, a.st_prev[w], a.level, lane);
                    }
                }
                __syncthreads();
                const int nbig = s_nbig;
                for (int j = 0; j < nbig; ++j) {  // the long lists: all wavefronts on one list
                    if (tid == 0) { s_coop.work = 0; s_coop.fb = 0; s_coop.start = 0; }
                    __syncthreads();
                    const int64_t w = w0 + s_big[j];
                    const int4 sc2 = a.st_const2[w];
                    const int64_t tbase = ((int64_t)sc2.y << 32) | (unsigned)sc2.x;
                    if (LDS_BITS) (void)lazy_resolve_wave<false, LzBitsLds, true>(a, LzBitsLds{lz_lds, a.lz_bm + (size_t)slot * a.lz_words}, ws, slot, tbase, a.st_rank[w], a.st_cur[w], a.st_prev[w], a.level, lane, wvi, LZ_T / 64, &s_coop);
                    else (void)lazy_resolve_wave<false, LzBitsGlobal, true>(a, LzBitsGlobal{a.lz_bm + (size_t)slot * a.lz_words}, ws, slot, tbase, a.st_rank[w], a.st_cur[w], a.st_prev[w], a.level, lane, wvi, LZ_T / 64, &s_coop);
                    __syncthreads();
                }
            }
            __syncthreads();  // (everyone is through with the list and the LDS words)
            if (found <= LZ_LIST) break;
            if (tid == 0) { s_n = 0; s_next = 0; s_nbig = 0; }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ uint64_t group16_incl_scan_u64(uint64_t v, int /*t*/) { return group16_incl_scan_u64_dpp(v); }

// One 16-lane group per 16-candidate chunk (grid-stride): four independent chunks in flight per
// wavefront, so the short dependent chain (descriptor -> ids + current row -> neighbour rows) of
// the many small tasks is overlapped four-fold; rows are streamed UNROLL at a time per group
// (float4 per lane, 256 B contiguous per row per load), fmaf chain + xor butterfly (spec S1).
template <int NCH>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void level_score_kernel(const WalkArgs a, const int64_t cap_chunks) {
    constexpr int UNROLL = 4;
    __shared__ unsigned long long blk_rows;
    if (threadIdx.x == 0) blk_rows = 0;
    __syncthreads();
    const unsigned long long cw = a.lc[CTR_CHUNKS + a.level];
    const int64_t total_chunks = (int64_t)(cw >> 32);  // score chunks
    const int64_t lbase_now = level_chunk_base(a);
    // the next hop's prefix base (this hop's chunk counts are final: its advance kernel has finished)
    if (blockIdx.x == 0 && threadIdx.x == 0) a.lc[CTR_BASE + a.level + 1] = (unsigned long long)(lbase_now + (int64_t)(cw & 0xffffffffull));
    if (total_chunks > cap_chunks || lbase_now + (int64_t)(cw & 0xffffffffull) > a.cap_total) return;
    const int t = threadIdx.x & 15;
    const int nblk = gridDim.x;
    // consecutive chunks (same task / same root) -> consecutive logical blocks -> one XCD's L2.  (Giving every XCD
    // its own eighth of the chunk list instead measured 5 % slower: eight distant regions of the tree arrays at once.)
    const int lblock = (nblk % 8 == 0) ? (int)((blockIdx.x % 8) * (nblk / 8) + blockIdx.x / 8) : (int)blockIdx.x;
    const int64_t n_groups = (int64_t)nblk * (WAVES_PER_BLOCK * 4);
    unsigned long long rows = 0;
    for (int64_t c = (int64_t)lblock * (WAVES_PER_BLOCK * 4) + (threadIdx.x >> 4); c < total_chunks; c += n_groups) {
        const int4 d = a.lv_chunk_desc[2 * c], d2 = a.lv_chunk_desc[2 * c + 1];
        const int cur = d.x, nblock = d.y & 0xff;
        const bool single = (d.y & SINGLE_CHUNK) != 0, node = (d.y & DESC_NODE) != 0;
        const int64_t off = ((int64_t)d.w << 32) | (unsigned)d.z;
        // candidate ids: the children of (root, cur) in BFS order, or -- node chunk -- cur's graph neighbours
        const int32_t *const ids = (node ? a.col : a.t_order) + off;
        float *const out = node ? a.es + off : a.lv_scores + c * CHUNK;
        float4 gc[NCH];
        const float4 *const crow = (const float4 *)(a.E + (int64_t)cur * a.ld);
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) {
            const int ch = t + 16 * cc;
            gc[cc] = (ch < a.nchunk) ? crow[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // the chunk's ids with one coalesced load (candidate 0 of a list with a father entry comes with the descriptor)
        const int myid = (t < nblock) ? (((d.y & DESC_HAS_FATHER) && t == 0) ? d2.x : ids[t]) : -1;
        const float mybias = (t < nblock) ? a.bias[myid] : 0.f;  // ... and its biases with one 16-lane gather
        float mysc = 0.f;
        for (int j0 = 0; j0 < nblock; j0 += UNROLL) {
            float4 y[UNROLL][NCH];
            int id[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                id[u] = __shfl(myid, j0 + u, 16);
                const bool valid = id[u] >= 0;
                const float4 *const row = (const float4 *)(a.E + (int64_t)(valid ? id[u] : 0) * a.ld);
#pragma unroll
                for (int cc = 0; cc < NCH; ++cc) {
                    const int ch = t + 16 * cc;
                    y[u][cc] = (valid && ch < a.nchunk) ? row[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                float acc = 0.0f;
#pragma unroll
                for (int cc = 0; cc < NCH; ++cc) {
                    acc = __builtin_fmaf(gc[cc].x, y[u][cc].x, acc);
                    acc = __builtin_fmaf(gc[cc].y, y[u][cc].y, acc);
                    acc = __builtin_fmaf(gc[cc].z, y[u][cc].z, acc);
                    acc = __builtin_fmaf(gc[cc].w, y[u][cc].w, acc);
                }
                // the xor butterfly over the 16 lanes (spec S1) as four rotate-and-add DPP instructions instead of four LDS permutes:
                // bit-identical -- behind the step of width w every lane's value has period w inside its row of 16, so the partner
                // a rotation by w reaches holds the same value as lane ^ w, and fp32 addition is commutative
                acc = acc + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x128, 0xf, 0xf, false));  // row_ror:8
                acc = acc + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x124, 0xf, 0xf, false));  // row_ror:4
                acc = acc + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x122, 0xf, 0xf, false));  // row_ror:2
                acc = acc + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x121, 0xf, 0xf, false));  // row_ror:1
                if (t == j0 + u) mysc = acc + mybias;  // lane j keeps the score of candidate j
            }
        }
        if (single) {
            // the whole distribution sits in this group's lanes: max, exact fixed-point weights and their
            // inclusive prefix sums (spec S2, S3) right here -- no score round trip, no second kernel
            float mx = (t < nblock) ? mysc : -INFINITY;
            mx = group16_max_f32(mx);
            const uint64_t wgt = (t < nblock) ? weight_fix40(exp_spec(mysc - mx)) : 0ull;
            const uint64_t C = group16_incl_scan_u64(wgt, t);
            const int64_t pfx = ((int64_t)d2.z << 32) | (unsigned)d2.y;
            if (t < nblock) a.lv_prefix[pfx * CHUNK + t] = C;
        } else if (t < nblock) {
            out[t] = mysc;  // one coalesced 64-byte store per chunk (score region of the task, or the edge-score cache)
        }
        rows += (unsigned long long)nblock;
    }
    // one counter update per block, spread over 64 words: same-address atomics serialise at ~12 ns each, and
    // 2 x 2048 of them at the end of every launch were a 20 us tail on the short levels (the host folds the words)
    if (t == 0 && rows) atomicAdd(&blk_rows, rows);
    __syncthreads();
    if (threadIdx.x == 0 && blk_rows) atomicAdd(&a.lc[CTR_ROWS + (blockIdx.x & 63)], blk_rows);
    if ((a.exp & 8) && threadIdx.x == 0 && blk_rows) atomicAdd(&a.lc[CTR_TINY + a.level], blk_rows);  // GG_WALK_EXPERIMENT & 8: rows per LEVEL (same-address atomics: a ~20 us tail)
}

// Where the scores of owner walk w's distribution are: its private score region, or (gather) the edge-score cache through
// the tree's edge indices -- candidate j is the father (j == 0, only if hf: es[fe]) or child rank beg + j - hf: es[t_edge[..]].
struct TaskScores {
    const float *sc;        // private region (gather: unused)
    const int32_t *edges;   // t_edge + beg - hf (gather)
    int k, hf, fe;
    bool gather;
};
__device__ __forceinline__ TaskScores task_scores(const WalkArgs &a, const int64_t w) {
    TaskScores s;
    const int kraw = a.lv_k[w];
    s.k = kraw & LVK_MASK;
    s.hf = (int)((unsigned)kraw >> 31);
    s.gather = (kraw & LVK_GATHER) != 0;
    s.sc = a.lv_scores + a.lv_coff[w] * CHUNK;
    s.edges = a.t_edge + a.lv_beg[w] - s.hf;
    s.fe = (s.gather && s.hf) ? a.lv_fe[w] : -1;
    return s;
}

// Small owner tasks (k <= 16 * PER_LANE): one 16-lane group per walk -- max, exact fixed-point weights
// (spec S2, S3) and their inclusive prefix sums, computed once per (root, node).  The kernel is latency bound -- list entry
// -> task words -> (edge indices ->) scores -> store, a few thousand groups resident -- so a group works on NT tasks AT ONCE:
// every stage's loads of all NT tasks are in flight together (and all of a task's scores are fetched with independent loads,
// <= PER_LANE per lane).
template <int PER_LANE, int NT>
__device__ __forceinline__ void weights_small_tasks(const WalkArgs &a, const int32_t *list, const int64_t i0, const int64_t n_tasks, const int64_t stride_sign,
                                                    const int t) {
    int64_t w[NT];
    TaskScores ts[NT];
    uint64_t *pf[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) w[u] = (i0 + u < n_tasks) ? (int64_t)list[stride_sign * (i0 + u)] : -1;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int64_t ww = w[u] >= 0 ? w[u] : 0;  // (a padding slot repeats walk 0's loads; nothing is stored for it)
        ts[u] = task_scores(a, ww);
        pf[u] = a.lv_prefix + a.lv_pfx[ww] * CHUNK;
        if (w[u] < 0) ts[u].k = 0;
    }
    float v[NT][PER_LANE];
    int e[NT][PER_LANE];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int i = 0; i < PER_LANE; ++i) {
            const int jj = i * 16 + t;
            e[u][i] = (ts[u].gather && jj < ts[u].k) ? ((jj == 0 && ts[u].hf) ? ts[u].fe : ts[u].edges[jj]) : -1;
        }
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int i = 0; i < PER_LANE; ++i) {
            const int jj = i * 16 + t;
            v[u][i] = ts[u].gather ? (e[u][i] >= 0 ? a.es[e[u][i]] : -INFINITY) : (jj < ts[u].k ? ts[u].sc[jj] : -INFINITY);
        }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int k = ts[u].k;
        if (k == 0) continue;
        float mx = v[u][0];
#pragma unroll
        for (int i = 1; i < PER_LANE; ++i) mx = fmaxf(mx, v[u][i]);
        mx = group16_max_f32(mx);
        uint64_t carry = 0;
#pragma unroll
        for (int i = 0; i < PER_LANE; ++i) {
            if (i * 16 < k) {  // uniform inside the 16-lane group
                const int jj = i * 16 + t;
                const uint64_t wgt = (jj < k) ? weight_fix40(exp_spec(v[u][i] - mx)) : 0ull;
                const uint64_t C = carry + group16_incl_scan_u64(wgt, t);
                if (jj < k) pf[u][jj] = C;
                if (PER_LANE > 1) carry = dpp_u64<0x15F>(C);  // row_newbcast:15: the row's last lane to all its lanes
            }
        }
    }
}

constexpr int SMALL_BLOCKS = 2048;  // workgroups of the weights launch that serve the small-task list
constexpr int SMALL_NT = 1;  // tasks a 16-lane group has in flight
__device__ __forceinline__ void weights_small_blocks(const WalkArgs &a, const int block) {
    const int t = threadIdx.x & 15;
    const int64_t n_small = (int64_t)(a.lc[CTR_BIG + a.level] >> 32);
    for (int64_t i = ((int64_t)block * 16 + (threadIdx.x >> 4)) * SMALL_NT; i < n_small; i += (int64_t)SMALL_BLOCKS * 16 * SMALL_NT)
        weights_small_tasks<BIG_TASK / 16, SMALL_NT>(a, a.lv_big + a.lv_big_cap - 1, i, n_small, -1, t);  // (the small list grows down from the end of lv_big)
}
// Big owner tasks (hubs): one 256-thread workgroup per task, from the level's big-task list, in tiles of BIG_TILE candidates.
// A thread owns BIG_PT CONSECUTIVE candidates of the tile: the max and the scan need one block-wide combination each (two
// barrier pairs per tile; lane-strided rows of 256 cost a barrier pair per row, sixteen per tile), the thread's own prefix
// is a register loop, and it stores 64 contiguous bytes.  A task of up to one tile keeps its scores in registers between
// the max and the scan pass; larger ones read them twice.
#ifndef GG_BIG_PT
#define GG_BIG_PT 8
#endif
constexpr int BIG_PT = GG_BIG_PT;
constexpr int BIG_TILE = 256 * BIG_PT;
constexpr int BIG_BLOCKS = 2048;  // workgroups of the weights launch that serve the big-task list
__device__ __forceinline__ void weights_big_blocks(const WalkArgs &a) {
    __shared__ float red[4];
    __shared__ uint64_t wave_tot[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n_big = (int)(a.lc[CTR_BIG + a.level] & 0xffffffffull);
    for (int b = blockIdx.x; b < n_big; b += BIG_BLOCKS) {
        const int64_t w = a.lv_big[b];
        const TaskScores ts = task_scores(a, w);
        const int k = ts.k;
        uint64_t *const pf = a.lv_prefix + a.lv_pfx[w] * CHUNK;
        auto load_tile = [&](int j0, float (&v)[BIG_PT]) {  // candidates j0 + 16 * thread + [0, 16)
            const int jb = j0 + BIG_PT * (int)threadIdx.x;
            if (ts.gather) {
                int e[BIG_PT];
#pragma unroll
                for (int i = 0; i < BIG_PT; ++i) e[i] = (jb + i < k) ? ((jb + i == 0 && ts.hf) ? ts.fe : ts.edges[jb + i]) : -1;
#pragma unroll
                for (int i = 0; i < BIG_PT; ++i) v[i] = e[i] >= 0 ? a.es[e[i]] : -INFINITY;
            } else {
#pragma unroll
                for (int i = 0; i < BIG_PT; ++i) v[i] = (jb + i < k) ? ts.sc[jb + i] : -INFINITY;
            }
        };
        float v[BIG_PT];
        float mx = -INFINITY;
        for (int j0 = 0; j0 < k; j0 += BIG_TILE) {
            load_tile(j0, v);
#pragma unroll
            for (int i = 0; i < BIG_PT; ++i) mx = fmaxf(mx, v[i]);
        }
        mx = wave_max_f32(mx);
        __syncthreads();
        if (lane == 0) red[wv] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        uint64_t carry = 0;
        for (int j0 = 0; j0 < k; j0 += BIG_TILE) {
            if (k > BIG_TILE) load_tile(j0, v);  // (a single tile is still in registers)
            const int jb = j0 + BIG_PT * (int)threadIdx.x;
            uint64_t c[BIG_PT], run = 0;
#pragma unroll
            for (int i = 0; i < BIG_PT; ++i) {
                run += (jb + i < k) ? weight_fix40(exp_spec(v[i] - mx)) : 0ull;
                c[i] = run;
            }
            const uint64_t inc = wave_incl_scan_u64(run, lane);
            __syncthreads();
            if (lane == 63) wave_tot[wv] = inc;
            __syncthreads();
            uint64_t pre = carry + inc - run, tot = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < wv) pre += wave_tot[i];
                tot += wave_tot[i];
            }
#pragma unroll
            for (int i = 0; i < BIG_PT; ++i)
                if (jb + i < k) pf[jb + i] = pre + c[i];
            carry += tot;
        }
    }
}

// One launch per level for all task classes: the first BIG_BLOCKS workgroups walk the big-task list (they run
// longest, so they are dispatched first), the others take 16 small tasks each per round.  As back-to-back launches the
// classes cost the sum of their latency-bound run times; together, the longest.
__global__ __launch_bounds__(256) void level_weights_kernel(const WalkArgs a, const int64_t cap_chunks) {
    const unsigned long long cw = a.lc[CTR_CHUNKS + a.level];
    const int64_t pch = (int64_t)(cw & 0xffffffffull), sch = (int64_t)(cw >> 32);
    if (sch > cap_chunks || pch == 0 || level_chunk_base(a) + pch > a.cap_total) return;
    if (blockIdx.x < BIG_BLOCKS) {
        if (!(a.exp & 1)) weights_big_blocks(a);
        if (a.exp & 64) { __syncthreads(); weights_big_blocks(a); }    // timing ablation: the class twice (idempotent, results stay right)
    } else {
        if (!(a.exp & 2)) weights_small_blocks(a, (int)blockIdx.x - BIG_BLOCKS);
        if (a.exp & 32) weights_small_blocks(a, (int)blockIdx.x - BIG_BLOCKS);
    }
}

// ------------------------------------------------------------------------------------------
// Finisher: one wavefront per walk, from hop a.level to the end of the walk.
// ------------------------------------------------------------------------------------------
template <int NCH, bool LAZY>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void walk_sample_kernel(const WalkArgs a) {
    __shared__ float lds_scores[WAVES_PER_BLOCK][SCORE_CAP];
    __shared__ unsigned long long blk_ctr[2];

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 2) blk_ctr[threadIdx.x] = 0;
    __syncthreads();

    // XCD-aware block order: hardware block b runs on XCD b % 8; give each XCD a contiguous
    // range of logical blocks so that the walks of one root share one L2.
    const int nblk = gridDim.x;
    const int lblock = (nblk % 8 == 0) ? (int)((blockIdx.x % 8) * (nblk / 8) + blockIdx.x / 8) : (int)blockIdx.x;
    const int64_t n_waves = (int64_t)nblk * WAVES_PER_BLOCK;
    float *const sbuf_lds = lds_scores[wib];
    float *const sbuf_glb = a.scratch + ((int64_t)blockIdx.x * WAVES_PER_BLOCK + wib) * a.scratch_stride;
    const bool resume = a.level > 0;
    if (resume && a.ctr[3] == 2ull) return;  // speculative level buffers overflowed: the host reruns the launch

    unsigned long long my_hops = 0, my_reads = 0;

    // dynamic work distribution: waves pull the next walk from one ticket counter, so a wave that
    // drew a hub hop does not hold back a fixed share of the remaining walks (the first n_waves
    // walks keep the XCD-contiguous static assignment)
    const int64_t n_items = resume ? (int64_t)a.ctr[CTR_FIN] : a.total_walks;  // resume: the live walks listed by the last level_advance_kernel
    for (int64_t it = (int64_t)lblock * WAVES_PER_BLOCK + wib; it < n_items;) {
        const int64_t w = resume ? (int64_t)a.fin_list[it] : it;
        {
            const int item = find_item(a.walk_ptr, a.n_slots, w);
            const uint32_t j = (uint32_t)(w - a.walk_ptr[item]);
            const int slot = a.slots[item];
            const int root = a.t_root[slot];
            const int64_t tbase = a.t_base[slot];
            const int32_t *const cs = a.t_cstart + tbase + (LAZY ? 0 : slot);  // (lazy builds place a slot's row at its base: the segments come from a cursor, in no slot order)
            const int32_t *const order = a.t_order + tbase;
            int32_t *const path = a.paths + w * (int64_t)a.stride;
            const int lzs = LAZY ? a.lz_info[slot].x : 0x7fffffff;  // first rank without a built children list

            int cur = root, prev = -1, len = 1, rank = 0;
            uint32_t hop = 0;  // = depth of cur: a walk only moves down the tree until its back-step
            if (resume) {
                cur = a.st_cur[w];
                prev = a.st_prev[w];
                len = a.st_len[w];
                rank = a.st_rank[w];
                hop = (uint32_t)a.level;
            } else {
                if (lane == 0) path[0] = root;
                if (a.for_d && lane == 0) a.first_child[w] = -1;
            }
            bool aborted = false, overflow = false;

            for (;;) {
                int cbeg, cend;
                if (LAZY && rank >= lzs) {  // resolved on the spot if no walk has stood here before
                    const unsigned long long v = lazy_children_wave(a, reinterpret_cast<LzWork *>(sbuf_lds), slot, tbase, rank, cur, prev, (int)hop, lane);  // (the score slots are free between hops)
                    cbeg = (int)(v >> 32);
                    cend = cbeg + (int)((v >> 12) & 0xFFFFFull);
                } else {
                    cbeg = cs[rank];
                    cend = cs[rank + 1];
                }
                const int nchild = cend - cbeg;
                int hf = 1;                                 // list = [father] ++ children ...
                if (hop == 0) hf = 0;                       // ... tree[root][1:]  (graph_gan.py:250)
                else if (hop == 1 && ((a.t_q3[a.t_q3off[slot] + ((rank - 1) >> 5)] >> ((rank - 1) & 31)) & 1u)) hf = 0;  // father entry removed earlier (Q3)
                int k = nchild + hf;
                if (k == 0) { aborted = true; break; }      // "the tree only has a root" (:252-253)
                if (a.for_d && hop == 1 && hf) {
                    if (nchild == 0) {                      // node_neighbor == [root] (:255-257)
                        if (lane == 0) atomicMin(&a.abort_walk[item], (int)j);
                        aborted = true;
                        break;
                    }
                    if (lane == 0) a.first_child[w] = rank;  // node_neighbor.remove(root) (:258-259), applied by the post-pass
                    hf = 0;
                    k -= 1;
                }
                // candidate c: the father (= prev) if hf and c == 0, else the child order[cbeg + c - hf]
                const int32_t *const ids = order + cbeg - hf;
                float *const sbuf = (k <= SCORE_CAP) ? sbuf_lds : sbuf_glb;

                float4 gc[NCH];
                load_cur_row<NCH>(a, cur, lane, gc);
                float mx = -INFINITY;
                for (int j0 = 0; j0 < k; j0 += 64) {
                    const int nblock = min(64, k - j0);
                    mx = fmaxf(mx, score_block<NCH>(a, gc, ids + j0, (hf && j0 == 0) ? prev : -1, nblock, lane,
                                                    [&](int jj, float sc) { sbuf[j0 + jj] = sc; }));
                }
                mx = wave_max_f32(mx);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

                const uint64_t m53 = uniform53(a.seed, a.stream, (uint32_t)root, j, hop);
                int idx = sample_index([&](int jj) { return sbuf[jj]; }, k, mx, m53, lane);
                __builtin_amdgcn_wave_barrier();
                if (idx < 0) {  // total weight 0 <=> the scores are not finite (diverged generator): a diagnosable error, not a wild index
                    if (lane == 0) a.ctr[CTR_NONFINITE] = 1ull;
                    idx = 0;
                }
                const bool back = hf && idx == 0;
                const int nxt = back ? prev : ids[idx];
                my_hops += 1;
                my_reads += (unsigned long long)k;
                if (len >= a.stride) { overflow = true; break; }
                if (lane == 0) path[len] = nxt;
                len += 1;
                hop += 1;
                if (back) break;  // terminating condition (:264-266): next == previous, sample = cur
                prev = cur;
                cur = nxt;
                rank = cbeg + idx - hf;
            }

            if (lane == 0) {
                if (overflow) {
                    a.ctr[3] = 1ull;
                    a.path_len[w] = 0;
                    a.samples[w] = -1;
                } else if (aborted) {
                    a.path_len[w] = 0;
                    a.samples[w] = -1;
                    if (!a.for_d) a.status[item] = GG_ROOT_ABORTED;
                    else if (hop == 0) atomicMin(&a.abort_walk[item], 0);
                } else {
                    a.path_len[w] = len;
                    a.samples[w] = cur;
                }
            }
        }
        unsigned long long nw = 0;
        if (lane == 0) nw = atomicAdd(&a.ctr[4], 1ull);
        it = n_waves + (int64_t)__shfl(nw, 0, 64);
    }

    if (lane == 0) {
        atomicAdd(&blk_ctr[0], my_hops);
        atomicAdd(&blk_ctr[1], my_reads);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blk_ctr[0]) atomicAdd(&a.ctr[0], blk_ctr[0]);
        if (blk_ctr[1]) {
            atomicAdd(&a.ctr[1], blk_ctr[1]);
            atomicAdd(&a.ctr[5], blk_ctr[1]);  // the finisher scores every hop's rows itself
        }
    }
}

// D-mode post-pass: the reference walks sequentially and stops a root at its first aborting
// walk (graph_gan.py:255-257); only walks before it have mutated the tree (:258-259).
__global__ void walk_d_postpass_kernel(const WalkArgs a) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.total_walks || a.ctr[3] == 2ull || a.ctr[CTR_LZ_FB] != 0ull) return;  // the launch is being rerun, mutate nothing
    const int item = find_item(a.walk_ptr, a.n_slots, w);
    const int j = (int)(w - a.walk_ptr[item]);
    const int ab = a.abort_walk[item];
    const int slot = a.slots[item];
    if (j < ab) {
        const int c = a.first_child[w];  // rank (>= 1) of the depth-1 child this walk stood on
        if (c >= 1) atomicOr(&a.t_q3[a.t_q3off[slot] + ((c - 1) >> 5)], 1u << ((c - 1) & 31));
    }
    if (ab != 0x7fffffff) {
        a.samples[w] = -1;
        a.path_len[w] = 0;
        if (j == 0) a.status[item] = GG_ROOT_ABORTED;
    }
}

// Counter words of both halves, and the base of each half's prefix region: dc_words = {base 0, end 0, base 1, end 1} in
// global chunk offsets.  A launch that looks distributions up (append) starts behind what the D launch of the step left in
// the region(s) it uses; a single-stream launch uses one region behind everything.
__global__ void walk_reset_kernel(unsigned long long *ctr, int n_words, int64_t *dc_words, int append, int split, int64_t S, const unsigned long long *gen_bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // (the launch starts with the non-finite flag of the generator's tables: hops that score nothing -- one candidate, a leaf's
    // back-step -- cannot notice a diverged row themselves)
    if (i < n_words && i != CTR_BASE && i != CTR_WORDS + CTR_BASE) ctr[i] = (i == CTR_NONFINITE) ? (*gen_bad ? 1ull : 0ull) : 0ull;
    if (i == 0) {
        if (!append) dc_words[1] = dc_words[3] = 0;
        const int64_t e0 = dc_words[1], e1 = dc_words[3];
        if (split) {
            dc_words[0] = append ? e0 : 0;
            dc_words[2] = (append && e1 > S) ? e1 : S;
        } else {
            dc_words[0] = append ? (e0 > e1 ? e0 : e1) : 0;
        }
        ctr[CTR_BASE] = (unsigned long long)dc_words[0];                        // prefix base of hop 0 (first half)
        ctr[CTR_WORDS + CTR_BASE] = (unsigned long long)(split ? dc_words[2] : 0);  // ... second half of a split launch
    }
}

__global__ void walk_init_status_kernel(const WalkArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_slots) return;
    a.status[i] = (a.walk_ptr[i + 1] == a.walk_ptr[i]) ? GG_ROOT_EMPTY : GG_ROOT_OK;
    a.abort_walk[i] = 0x7fffffff;
}

static int reserve_level_buffers(gg_ctx *ctx, WalkArgs &a, int64_t chunks) {
    GG_HIP(ctx, ctx->lv_scores.reserve(sizeof(float) * CHUNK * (size_t)chunks + 256));
    GG_HIP(ctx, ctx->lv_chunk_owner.reserve(sizeof(int4) * 2 * (size_t)chunks + 256));
    a.lv_scores = ctx->lv_scores.as<float>();
    a.lv_chunk_desc = ctx->lv_chunk_owner.as<int4>();
    return GG_OK;
}

// The prefix sums of ALL hops of a launch (and, for the distribution cache, of the D launch before it) live in one
// buffer addressed by global chunk offsets.  Growing it keeps what is there (`keep`): the G launch samples from the
// D launch's regions.
static int reserve_prefix(gg_ctx *ctx, WalkArgs &a, int64_t chunks, bool keep) {
    const size_t need = sizeof(uint64_t) * CHUNK * (size_t)chunks + 256;
    if (need > ctx->lv_prefix.bytes) {
        if (keep && ctx->lv_prefix.p) {
            DevBuf bigger;
            GG_HIP(ctx, bigger.reserve(need));
            GG_HIP(ctx, hipMemcpyAsync(bigger.p, ctx->lv_prefix.p, ctx->lv_prefix.bytes, hipMemcpyDeviceToDevice, ctx->walk_stream));
            GG_HIP(ctx, hipStreamSynchronize(ctx->walk_stream));
            ctx->lv_prefix.release();
            ctx->lv_prefix = bigger;
        } else {
            GG_HIP(ctx, ctx->lv_prefix.reserve(need));
        }
    }
    a.lv_prefix = ctx->lv_prefix.as<uint64_t>();
    a.cap_total = (int64_t)((ctx->lv_prefix.bytes - 256) / (sizeof(uint64_t) * CHUNK));
    return GG_OK;
}

__global__ __launch_bounds__(256) void fill_ones_kernel(uint4 *p, int64_t n16) {
    const uint4 v = make_uint4(~0u, ~0u, ~0u, ~0u);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = v;
}

// end of a D launch that registered its distributions: chunks now in the prefix buffer = where the G launch appends
__global__ void dc_finish_kernel(const WalkArgs a, int64_t *words, int n_levels) {
    if (a.ctr[3] == 2ull || a.ctr[CTR_LZ_FB] != 0ull) { words[1] = 0; return; }  // the launch is being rerun
    words[1] = (int64_t)a.lc[CTR_BASE + n_levels];  // (written by the score kernel of hop n_levels - 1)
}

// Levels 0 .. n_levels-1 through the streaming kernels.  sized == true: one 16-byte read-back
// per level sizes the buffers exactly (and learns the capacity for later launches);
// sized == false: no host synchronisation at all, buffers hold ctx->lv_cap_chunks chunks and a
// level that needs more raises flag 2 (the caller reruns the launch in sized mode).
//
// A sync-free launch of enough walks runs as TWO HALVES of its walks on two streams (walk_stream and stream3): a level is
// advance -> score -> weights, and only the score kernel moves data -- the other two are chains of dependent loads, 60-110 us
// per level with the chip nearly idle.  With two halves the advance / weights kernels of one half run under the score kernel
// of the other (which leaves two wave slots per SIMD free for exactly that); the score kernels themselves are chained by
// events so that they never share the HBM with each other.  Each half has its own walk range, task lists, chunk buffers,
// prefix region [h * S, (h + 1) * S) and level counters; dedup, registration and sampling are per workgroup of 256
// consecutive walks either way, so the split (at a multiple of 256) cannot change a walk (tested bit-exact).
// MEASURED AND OFF BY DEFAULT (GG_WALK_SPLIT=1 enables it): on the bench workload the walk call takes 1.93 ms split against
// 1.81 ms unsplit, the step 4.62 against 4.42 ms -- the advance / weights rounds do hide (about -0.2 ms per call), but a
// half-size score launch runs at 0.55-0.59 of the HBM peak instead of 0.68-0.70 (the persistent grid's ramp-up and tail are
// paid twice per level, and the kernel shares the chip with the other half's latency-bound kernels), which costs more.

// the resolve kernel of one lazy level: a persistent grid, one workgroup per CU when the slot's visited words go to LDS
static void launch_lazy_resolve(gg_ctx *ctx, const WalkArgs &x, hipStream_t st) {
    const size_t lds = sizeof(uint32_t) * (size_t)x.lz_words;
    const bool no_lds = getenv("GG_LZ_NO_LDS") != nullptr;  // (tests: the path of graphs whose visited words do not fit the LDS)
    const size_t lds_static = sizeof(LzWork) * (LZ_T / 64) + 2 * sizeof(int32_t) * LZ_LIST + sizeof(LzCoop) + 256;  // (the kernel's __shared__ arrays)
    if (!no_lds && lds + lds_static <= 160 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void *)lazy_resolve_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - lds_static)) != hipSuccess) (void)hipGetLastError();
            attr_set = true;
        }
        hipLaunchKernelGGL(lazy_resolve_kernel<true>, dim3((unsigned)std::min<int64_t>(x.n_slots, ctx->n_cus)), dim3(LZ_T), lds, st, x);
    } else {
        hipLaunchKernelGGL(lazy_resolve_kernel<false>, dim3((unsigned)std::min<int64_t>(x.n_slots, 2 * ctx->n_cus)), dim3(LZ_T), 0, st, x);
    }
}

template <int NCH>
static int run_levels(gg_ctx *ctx, WalkArgs &a, int64_t total_walks, int n_levels, bool sized, bool finisher_follows, bool *any_alive) {
    const dim3 blk(WAVES_PER_BLOCK * 64);
    int64_t cap = sized ? 0 : ctx->lv_cap_chunks;
    const bool keep = a.dc_mode != 0;  // cached prefix regions must survive a growing buffer
    int64_t base_host = 0, cum = 0;    // sized mode: global chunk offset of the launch / chunks of its hops so far
    const bool split = ctx->w_split;   // decided by launch_walk_sample (the reset kernel already set up both halves' words)
    const int n_half = split ? 2 : 1;
    const int64_t S = ctx->lv_cap_total;  // prefix region of a half
    if (!sized) {
        int rc = reserve_level_buffers(ctx, a, cap * n_half);
        if (rc == GG_OK) rc = reserve_prefix(ctx, a, std::max<int64_t>(S, cap) * n_half, keep);
        if (rc != GG_OK) return rc;
    } else {
        if (a.dc_mode == 2) {  // where the D launch of the step stopped
            GG_HIP(ctx, hipMemcpyAsync(&base_host, a.dc_words, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->walk_stream));
            GG_HIP(ctx, hipStreamSynchronize(ctx->walk_stream));
        }
        int rc = reserve_prefix(ctx, a, std::max<int64_t>(base_host, 1), keep);
        if (rc != GG_OK) return rc;
    }
    // the halves: walk ranges split at a multiple of the workgroup size
    WalkArgs h[2] = {a, a};
    hipStream_t hs[2] = {ctx->walk_stream, ctx->stream3};
    if (split) {
        const int64_t mid = ((total_walks / 2 + 255) / 256) * 256;
        h[0].w0 = 0; h[0].w_end = mid;
        h[1].w0 = mid; h[1].w_end = total_walks;
        for (int k = 0; k < 2; ++k) {
            h[k].lc = ctx->dev_ctr + (size_t)k * CTR_WORDS;
            h[k].lv_big = a.lv_big + h[k].w0;
            h[k].lv_big_cap = h[k].w_end - h[k].w0;
            h[k].lv_scores = a.lv_scores + (size_t)k * cap * CHUNK;
            h[k].lv_chunk_desc = a.lv_chunk_desc + (size_t)2 * k * cap;  // two int4 per chunk (write_chunk_desc / write_node_desc)
            h[k].dc_words = a.dc_words + 2 * k;
            h[k].cap_total = std::min<int64_t>((k + 1) * S, a.cap_total);
        }
        GG_HIP(ctx, hipEventRecord(ctx->ev_fork, hs[0]));
        GG_HIP(ctx, hipStreamWaitEvent(hs[1], ctx->ev_fork, 0));
    }
    const bool all_levels = n_levels >= ctx->tree_max_depth + 2;
    // as many levels as earlier (sized) launches found walks alive in; the final advance raises flag 2 if a walk of THIS launch
    // is still going after them (the rerun learns the new depth).  (One more level "for slack" was three empty launches per call.)
    if (!sized && all_levels && !finisher_follows && ctx->lv_levels_learned > 0) n_levels = std::min(n_levels, ctx->lv_levels_learned);
    *any_alive = true;
    ctx->lv_ev_used = 0;
    int level = 0;
    for (; level < n_levels; ++level) {
        for (int k = 0; k < n_half; ++k) {
            WalkArgs &x = h[k];
            x.level = level;
            x.es_now = a.es_now + 2 * level + k;  // the tick of this half's score kernel of this level (MAX_LEVELS = 64: < 256 per launch)
            const unsigned wblocks = (unsigned)cdiv(x.w_end - x.w0, 256);
            if (!a.lazy) {
                hipLaunchKernelGGL(level_advance_kernel<false>, dim3(wblocks), dim3(256), 0, hs[k], x, level > 0 ? 1 : 0, 1, sized ? 0 : 1, cap);
            } else if (level == 0 || level < ctx->lz_min_level) {  // every node of these levels has its children list built
                hipLaunchKernelGGL(level_advance_kernel<true>, dim3(wblocks), dim3(256), 0, hs[k], x, level > 0 ? 1 : 0, 1, sized ? 0 : 1, cap);
            } else {  // finish hop level - 1 | resolve the lists of the picked nodes nobody has stood on before | set hop `level` up
                hipLaunchKernelGGL(level_advance_kernel<true>, dim3(wblocks), dim3(256), 0, hs[k], x, 1, 0, 0, cap);
                launch_lazy_resolve(ctx, x, hs[k]);
                hipLaunchKernelGGL(level_advance_kernel<true>, dim3(wblocks), dim3(256), 0, hs[k], x, 2, 1, sized ? 0 : 1, cap);
            }
            if (sized) {
                unsigned long long total_chunks = 0, alive = 0;
                GG_HIP(ctx, hipMemcpyAsync(&total_chunks, ctx->dev_ctr + CTR_CHUNKS + level, sizeof(total_chunks), hipMemcpyDeviceToHost, ctx->walk_stream));
                GG_HIP(ctx, hipMemcpyAsync(&alive, ctx->dev_ctr + CTR_ALIVE + level, sizeof(alive), hipMemcpyDeviceToHost, ctx->walk_stream));
                GG_HIP(ctx, hipStreamSynchronize(ctx->walk_stream));
                if (alive == 0) {  // every walk has finished
                    *any_alive = false;
                    ctx->w_levels_run = level + 1;
                    if (a.dc_mode == 1) hipLaunchKernelGGL(dc_finish_kernel, dim3(1), dim3(1), 0, ctx->walk_stream, x, ctx->dc_words.as<int64_t>(), level);  // (hop `level` set nothing up)
                    return GG_OK;
                }
                if (level + 1 > ctx->lv_levels_learned) ctx->lv_levels_learned = level + 1;
                const int64_t pch = (int64_t)(total_chunks & 0xffffffffull);  // prefix chunks of the level
                cap = (int64_t)(total_chunks >> 32);                           // score chunks
                // (one learned capacity for both index spaces; which nodes are scored whole varies a little from launch to
                // launch -- the stamp races -- so the margin is generous)
                const int64_t need = std::max(cap, pch);
                if (need + need / 2 + 4096 > ctx->lv_cap_chunks) ctx->lv_cap_chunks = need + need / 2 + 4096;
                int rc = reserve_level_buffers(ctx, x, cap);
                if (rc == GG_OK) rc = reserve_prefix(ctx, x, base_host + cum + pch, true);
                if (rc != GG_OK) return rc;
                cum += pch;
                if (base_host + cum + (base_host + cum) / 4 + 4096 > ctx->lv_cap_total) ctx->lv_cap_total = base_host + cum + (base_host + cum) / 4 + 4096;
                hipLaunchKernelGGL(level_expand_kernel, dim3(wblocks), dim3(256), 0, ctx->walk_stream, x);
            }
            int64_t blocks = sized ? (cap + WAVES_PER_BLOCK * 4 - 1) / (WAVES_PER_BLOCK * 4) : score_blocks();
            if (blocks > score_blocks()) blocks = score_blocks();
            if (blocks >= 8) blocks -= blocks % 8;
            if (blocks < 1) blocks = 1;
            // the score kernels of the two halves are chained: this one starts behind the other half's previous one
            if (split && (level > 0 || k == 1)) GG_HIP(ctx, hipStreamWaitEvent(hs[k], ctx->ev_score[1 - k], 0));
            hipEvent_t *ev = ctx->lv_ev + 4 * level + 2 * k;
            if (ctx->walk_timed) {
                if (!ev[0]) {
                    GG_HIP(ctx, hipEventCreate(&ev[0]));
                    GG_HIP(ctx, hipEventCreate(&ev[1]));
                }
                GG_HIP(ctx, hipEventRecord(ev[0], hs[k]));
            }
            hipLaunchKernelGGL(level_score_kernel<NCH>, dim3((unsigned)blocks), blk, 0, hs[k], x, cap);
            if (ctx->walk_timed) {
                GG_HIP(ctx, hipEventRecord(ev[1], hs[k]));
                ctx->lv_ev_used = level + 1;
            }
            if (split) GG_HIP(ctx, hipEventRecord(ctx->ev_score[k], hs[k]));
            const int64_t half_walks = x.w_end - x.w0;
            hipLaunchKernelGGL(level_weights_kernel, dim3((unsigned)(BIG_BLOCKS + std::min<int64_t>(SMALL_BLOCKS, cdiv(half_walks * 16, 256)))), dim3(256), 0, hs[k], x, cap);
        }
    }
    // finish the last prepared hop
    for (int k = 0; k < n_half; ++k) {
        WalkArgs &x = h[k];
        x.level = level;
        const unsigned wblocks = (unsigned)cdiv(x.w_end - x.w0, 256);
        const int wd_last = (!sized && all_levels && !finisher_follows) ? 2 : 0;
        if (!a.lazy) {
            hipLaunchKernelGGL(level_advance_kernel<false>, dim3(wblocks), dim3(256), 0, hs[k], x, 1, 2, wd_last, 0);
        } else if (level < ctx->lz_min_level) {
            hipLaunchKernelGGL(level_advance_kernel<true>, dim3(wblocks), dim3(256), 0, hs[k], x, 1, 2, wd_last, 0);
        } else {
            hipLaunchKernelGGL(level_advance_kernel<true>, dim3(wblocks), dim3(256), 0, hs[k], x, 1, 0, 0, 0);
            launch_lazy_resolve(ctx, x, hs[k]);
            hipLaunchKernelGGL(level_advance_kernel<true>, dim3(wblocks), dim3(256), 0, hs[k], x, 2, 2, wd_last, 0);
        }
        if (a.dc_mode == 1) hipLaunchKernelGGL(dc_finish_kernel, dim3(1), dim3(1), 0, hs[k], x, ctx->dc_words.as<int64_t>() + 2 * k, level);
    }
    if (split) {
        GG_HIP(ctx, hipEventRecord(ctx->ev_join, hs[1]));
        GG_HIP(ctx, hipStreamWaitEvent(hs[0], ctx->ev_join, 0));
    }
    a = h[0];  // (level, pointers of half 0: what the finisher resumes with -- it works on the shared lists)
    a.w0 = 0; a.w_end = total_walks;
    ctx->w_levels_run = level;
    GG_HIP(ctx, hipGetLastError());
    return GG_OK;
}

template <int NCH>
static int run_levels_and_finish(gg_ctx *ctx, WalkArgs &a, int64_t total_walks) {
    const dim3 blk(WAVES_PER_BLOCK * 64);
    // a path has at most tree depth + 2 entries, so depth + 1 hops finish every walk
    int n_levels = ctx->walk_levels < 0 ? 0 : ctx->walk_levels;
    const bool all_levels = n_levels >= ctx->tree_max_depth + 2;
    if (all_levels) n_levels = ctx->tree_max_depth + 2;
    if (n_levels > MAX_LEVELS) n_levels = MAX_LEVELS;
    bool any_alive = true, small_finisher = false;
    ctx->lv_ev_used = 0;
    ctx->w_levels_run = 0;
    ctx->w_fin_follows = true;
    ctx->w_net = false;
    if (n_levels > 0) {
        GG_HIP(ctx, ctx->st_cur.reserve(sizeof(int32_t) * total_walks));
        GG_HIP(ctx, ctx->st_prev.reserve(sizeof(int32_t) * total_walks));
        GG_HIP(ctx, ctx->st_len.reserve(sizeof(int32_t) * total_walks));
        GG_HIP(ctx, ctx->st_alive.reserve(sizeof(int32_t) * total_walks));
        GG_HIP(ctx, ctx->st_rank.reserve(sizeof(int32_t) * total_walks));
        GG_HIP(ctx, ctx->lv_beg.reserve(sizeof(int64_t) * total_walks));
        GG_HIP(ctx, ctx->lv_k.reserve(sizeof(int32_t) * total_walks));
        GG_HIP(ctx, ctx->lv_chunks.reserve(sizeof(int32_t) * total_walks));
        GG_HIP(ctx, ctx->lv_big.reserve(sizeof(int32_t) * total_walks));
        GG_HIP(ctx, ctx->lv_fe.reserve(sizeof(int32_t) * total_walks));
        GG_HIP(ctx, ctx->lv_coff.reserve(sizeof(int64_t) * (total_walks + 1)));
        GG_HIP(ctx, ctx->lv_pfx.reserve(sizeof(int64_t) * (total_walks + 1)));
        GG_HIP(ctx, ctx->st_item.reserve(sizeof(int4) * total_walks));
        GG_HIP(ctx, ctx->st_item2.reserve(sizeof(int4) * total_walks));
        a.st_cur = ctx->st_cur.as<int32_t>();
        a.st_prev = ctx->st_prev.as<int32_t>();
        a.st_len = ctx->st_len.as<int32_t>();
        a.st_alive = ctx->st_alive.as<int32_t>();
        a.st_rank = ctx->st_rank.as<int32_t>();
        a.st_const = ctx->st_item.as<int4>();
        a.st_const2 = ctx->st_item2.as<int4>();
        a.lv_beg = ctx->lv_beg.as<int64_t>();
        a.lv_k = ctx->lv_k.as<int32_t>();
        a.lv_chunks = ctx->lv_chunks.as<int32_t>();
        a.lv_big = ctx->lv_big.as<int32_t>();
        a.lv_fe = ctx->lv_fe.as<int32_t>();
        a.lv_coff = ctx->lv_coff.as<int64_t>();
        a.lv_pfx = ctx->lv_pfx.as<int64_t>();
        const bool sized = ctx->lv_cap_chunks == 0 || ctx->walk_force_sized;
        // The deep hops hold a few thousand walks but cost a full advance / score / weights round each (~60 us of pure
        // latency): stop streaming where the previous launch of this mode had fewer than fin_threshold walks left and let
        // the finisher (one wavefront per listed walk, to the end of the walk) take those.  Any split gives the same walks.
        bool early = false;
        if (all_levels && !sized && ctx->fin_threshold > 0) {
            const int64_t *prof = ctx->alive_prof[a.for_d ? 1 : 0];
            for (int l = 1; l < n_levels && l < MAX_LEVELS; ++l)
                if (prof[l] >= 0 && prof[l] < ctx->fin_threshold) { n_levels = l; early = true; break; }
        }
        // A sync-free launch streams as many levels as earlier launches found walks alive in.  A walk of THIS launch that is
        // still going behind them used to send the whole launch to a sized rerun (flag 2: a full walk call, with a host
        // round trip per level, inside the timed region whenever one walk went a hop deeper than any before it -- the
        // generator moves, so that never quite stops).  Now the last advance lists such walks and the per-walk finisher
        // takes them: a safety net that costs one near-empty launch when nobody is left (round 4).
        bool net = false;
        if (all_levels && !sized && !early && ctx->lv_levels_learned > 0 && ctx->lv_levels_learned < n_levels && !getenv("GG_WALK_NO_NET")) {
            n_levels = ctx->lv_levels_learned;
            net = true;
        }
        ctx->w_net = net;
        ctx->w_fin_follows = early || !all_levels || net;
        int rc = run_levels<NCH>(ctx, a, total_walks, n_levels, sized, early || net, &any_alive);
        if (rc != GG_OK) return rc;
        ctx->walk_used_speculation = !sized;
        if (all_levels && !early && !net) any_alive = false;
        if (net) small_finisher = true;
    }
    if (any_alive) {
        a.level = n_levels;
        int64_t blocks = (total_walks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
        const int64_t max_blocks = small_finisher ? 256 : 256 * 8;  // (the safety net expects a handful of walks)
        if (blocks > max_blocks) blocks = max_blocks;
        if (blocks >= 8) blocks -= blocks % 8;
        const int64_t need_stride = ctx->tree_max_list > SCORE_CAP ? ((ctx->tree_max_list + 63) / 64 * 64) : 0;
        GG_HIP(ctx, ctx->w_scratch.reserve((size_t)need_stride * blocks * WAVES_PER_BLOCK * sizeof(float) + 16));
        a.scratch = ctx->w_scratch.as<float>();
        a.scratch_stride = need_stride;
        if (a.lazy) hipLaunchKernelGGL((walk_sample_kernel<NCH, true>), dim3((unsigned)blocks), blk, 0, ctx->walk_stream, a);
        else hipLaunchKernelGGL((walk_sample_kernel<NCH, false>), dim3((unsigned)blocks), blk, 0, ctx->walk_stream, a);
    }
    GG_HIP(ctx, hipGetLastError());
    return GG_OK;
}

int launch_walk_sample(gg_ctx *ctx, int32_t n_slots, int64_t total_walks, int for_d, uint64_t seed, uint32_t stream,
                       int32_t stride) {
    WalkArgs a{};
    a.E = ctx->model[0].E;
    a.bias = ctx->model[0].b;
    a.n_node = ctx->n_node;
    a.ld = ctx->ld;
    a.nchunk = ctx->ld / 4;
    a.t_root = ctx->t_root;
    a.t_order = ctx->t_order;
    a.t_cstart = ctx->t_cstart;
    a.t_base = ctx->t_base;
    a.t_q3 = ctx->t_q3;
    a.t_q3off = ctx->t_q3off;
    a.slots = ctx->w_slots_buf().as<int32_t>();
    a.walk_ptr = ctx->w_ptr_buf().as<int64_t>();
    a.n_slots = n_slots;
    a.total_walks = total_walks;
    a.for_d = for_d;
    a.seed = seed;
    a.stream = stream;
    a.samples = ctx->w_samples.as<int32_t>();
    a.paths = ctx->w_paths.as<int32_t>();
    a.path_len = ctx->w_len.as<int32_t>();
    a.stride = stride;
    a.status = ctx->w_status.as<int32_t>();
    a.first_child = ctx->w_first.as<int32_t>();
    a.abort_walk = ctx->w_abort.as<int32_t>();
    a.ctr = ctx->dev_ctr;
    a.lc = ctx->dev_ctr;
    a.rowptr = ctx->g_rowptr;
    a.col = ctx->g_col;
    a.t_edge = ctx->t_edge;
    a.rev = ctx->g_rev;
    a.es = ctx->es;
    a.es_stamp = ctx->es_stamp;
    a.es_valid_from = ctx->es_valid_from;
    a.es_now = ctx->es_tick;  // base of the launch's ticks; run_levels adds 2 * level + half
    ctx->es_tick += 256;
    a.es_ratio = ctx->es_ratio_num;
    a.es_hub = ctx->es_hub;
    a.es_mode = 0;  // set below
    {
        static const int exp_env = getenv("GG_WALK_EXPERIMENT") ? atoi(getenv("GG_WALK_EXPERIMENT")) : 0;
        a.exp = exp_env;
    }
    a.w0 = 0;
    a.w_end = total_walks;
    a.lv_big_cap = total_walks;
    a.lazy = ctx->t_lazy ? 1 : 0;
    a.t_order_w = ctx->t_order;
    a.t_edge_w = ctx->t_edge;
    if (a.lazy) {
        if (ctx->lz_list.bytes < sizeof(int32_t) * 2 * (size_t)n_slots) {  // (a launch that names slots more than once)
            GG_HIP(ctx, ctx->lz_list.reserve(sizeof(int32_t) * 2 * (size_t)n_slots));
            GG_HIP(ctx, hipMemset(ctx->lz_list.p, 0, ctx->lz_list.bytes));
        }
        a.lz_info = ctx->lz_info.as<int4>();
        a.lz_pair = ctx->lz_pair.as<unsigned long long>();
        a.lz_rank = ctx->lz_rank.as<int32_t>();
        a.lz_bm = ctx->lz_bm.as<uint2>();
        a.lz_words = (ctx->n_node + 31) / 32;
        a.lz_cursor = ctx->lz_cursor.as<int32_t>();
        a.lz_flag = ctx->lz_flag.as<int32_t>();
        a.lz_stamp = ctx->lz_stamp;
        a.g_multi = ctx->g_multi ? 1 : 0;
        a.lz_list = ctx->lz_list.as<int32_t>();
        {
            const bool stats_env = getenv("GG_LZ_STATS") != nullptr;  // (read per launch: tests switch it)
            a.lz_ctr = stats_env ? ctx->dev_ctr + 1500 : nullptr;  // (behind the launches' counter words; zeroed by the build)
        }
        {
            static const int budget_env = getenv("GG_LZ_BUDGET") ? atoi(getenv("GG_LZ_BUDGET")) : 2048;
            a.lz_budget = budget_env > 0 ? budget_env : 0x7fffffff;
            const int coop_env = getenv("GG_LZ_COOP_MIN") ? atoi(getenv("GG_LZ_COOP_MIN")) : LZ_COOP_MIN;  // (read per launch: tests switch it)
            a.lz_coop_min = std::max(64, coop_env);
        }
    }
    GG_HIP(ctx, ctx->fin_list.reserve(sizeof(int32_t) * (size_t)(total_walks + 1)));
    a.fin_list = ctx->fin_list.as<int32_t>();
    // distribution cache: mode requested by gg_prepare_d (register) / gg_prepare_g (look up); anything else runs without it
    GG_HIP(ctx, ctx->dc_words.reserve(sizeof(int64_t) * 8));
    a.dc_words = ctx->dc_words.as<int64_t>();
    a.dc_mode = (ctx->walk_levels > 0 && total_walks > 0) ? ctx->dc_request : 0;
    if (a.dc_mode == 1) {
        size_t want = 1u << 16;
        while (want < (size_t)total_walks * 8 && want < (1u << 24)) want <<= 1;  // distributions of a launch <= a few per walk
        if (want > ctx->dc_size) {
            GG_HIP(ctx, ctx->dc_keys.reserve(sizeof(ulonglong2) * want));  // {key, value} entries
            ctx->dc_size = want;
        }
        // (hipMemsetAsync of these 16 MB took ~150 us on the walk stream; a plain store kernel takes a few)
        hipLaunchKernelGGL(fill_ones_kernel, dim3(1024), dim3(256), 0, ctx->walk_stream, (uint4 *)ctx->dc_keys.p, (int64_t)ctx->dc_size);
    }
    a.dc_tab = ctx->dc_keys.as<ulonglong2>();
    a.dc_mask = (uint32_t)(ctx->dc_size ? ctx->dc_size - 1 : 0);
    // One small kernel resets the launch's words: the base of its prefix regions (behind the D launch's regions when it
    // looks distributions up, else 0) and every counter word -- they belong to ONE launch (the host accumulates,
    // walk_finalize): hops / reads / rows, error flag [3], ticket [4], per-level counters and the spread words.  (As a
    // hipMemcpyAsync + hipMemsetAsync pair these were blit kernels with system-scope fences: the copy took ~130 us on the side
    // stream while the discriminator's gradient kernel was filling the L2 with atomics.)
    const bool will_size = ctx->lv_cap_chunks == 0 || ctx->walk_force_sized;
    ctx->w_split = ctx->split_enabled && ctx->walk_levels > 0 && !will_size && total_walks >= ctx->split_min_walks && !ctx->t_lazy;
    // the edge-score cache needs the trees' edge indices (and a symmetric adjacency: g_rev)
    if (ctx->es && ctx->es_stamp && ctx->g_rev && ctx->t_edge_valid && ctx->walk_levels > 0) a.es_mode = ctx->es_mode;
    hipLaunchKernelGGL(walk_reset_kernel, dim3(cdiv(2 * CTR_WORDS, 256)), dim3(256), 0, ctx->walk_stream, ctx->dev_ctr, 2 * (int)CTR_WORDS,
                       ctx->dc_words.as<int64_t>(), a.dc_mode == 2 ? 1 : 0, ctx->w_split ? 1 : 0, ctx->lv_cap_total, ctx->table_bad.as<unsigned long long>());
    // HIP events around every profile_every-th call (a rerun keeps the decision of the launch it repeats)
    if (!ctx->walk_force_sized) ctx->walk_timed = ctx->profile_every > 0 && (ctx->walk_call_index++ % ctx->profile_every) == 0;
    if (ctx->walk_timed && ctx->walk_stream != ctx->stream && ctx->profile_solo) {
        // a profiled launch is measured alone: it does not share the HBM with the discriminator update in flight
        GG_HIP(ctx, hipEventRecord(ctx->ev_main_mark, ctx->stream));
        GG_HIP(ctx, hipStreamWaitEvent(ctx->walk_stream, ctx->ev_main_mark, 0));
    }
    hipLaunchKernelGGL(walk_init_status_kernel, dim3(cdiv(n_slots, 256)), dim3(256), 0, ctx->walk_stream, a);
    if (total_walks == 0) return GG_OK;

    const int nch = (a.nchunk + 15) / 16;
    if (ctx->walk_timed) GG_HIP(ctx, hipEventRecord(ctx->ev0, ctx->walk_stream));
    int rc;
    if (nch <= 1) rc = run_levels_and_finish<1>(ctx, a, total_walks);
    else if (nch == 2) rc = run_levels_and_finish<2>(ctx, a, total_walks);
    else if (nch <= 4) rc = run_levels_and_finish<4>(ctx, a, total_walks);
    else if (nch <= 8) rc = run_levels_and_finish<8>(ctx, a, total_walks);
    else return fail(ctx, GG_EINVAL, "n_emb %d not supported (max 512)", ctx->n_emb);
    if (rc != GG_OK) return rc;
    if (ctx->walk_timed) GG_HIP(ctx, hipEventRecord(ctx->ev1, ctx->walk_stream));
    if (for_d)
        hipLaunchKernelGGL(walk_d_postpass_kernel, dim3(cdiv(total_walks, 256)), dim3(256), 0, ctx->walk_stream, a);
    GG_HIP(ctx, hipGetLastError());
    return GG_OK;
}

}  // namespace gg
