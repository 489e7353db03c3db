// gg_internal.h -- context layout and helpers shared by the translation units of
// libgraphgan_hip.so.  Nothing here crosses the C ABI (include/graphgan_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/graphgan_hip.h"

namespace gg {

// One growable device allocation.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) {
            (void)hipDeviceSynchronize();  // kernels of calls that returned early (gg_set_profiling) may still use it
            (void)hipFree(p);
        }
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T *as() const { return (T *)p; }
};

// One embedding model (generator or discriminator): table, bias, Adam slots.
struct Model {
    float *E = nullptr, *b = nullptr;      // [n_node * ld], [n_node]
    float *mE = nullptr, *vE = nullptr;    // Adam first / second moments of E
    float *mb = nullptr, *vb = nullptr;    // ... of b
    float b1p = 0.9f, b2p = 0.999f;        // beta powers (TF keeps them as fp32 variables)
    int64_t t = 0;                         // optimizer steps taken
    float lr = 1e-3f, lambda = 1e-5f;
};

}  // namespace gg

struct gg_ctx {
    int32_t n_node = 0, n_emb = 0, ld = 0;  // ld = n_emb rounded up to a multiple of 4 (zero padded)
    gg_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Side stream for the walks of gg_prepare_g: they read only the generator's tables and the trees, so they may
    // run beside whatever discriminator work (gg_d_pass: gradient kernel, replica exchange, optimizer) is still in
    // flight on `stream`; the pair / reward kernels behind the walk go back to `stream` (ev_walk_done).
    hipStream_t stream2 = nullptr;
    hipStream_t walk_stream = nullptr;   // stream of the walk launch in flight (stream or stream2)
    hipEvent_t ev_walk_done = nullptr;   // side-stream walk finished  -> `stream` waits
    hipEvent_t ev_gen_pass = nullptr;    // last generator update enqueued on `stream` -> the side stream waits
    hipEvent_t ev_main_mark = nullptr;   // profiled side-stream launches wait for all of `stream` (measured alone)
    bool gen_pass_recorded = false;
    bool gen_dirty = false;  // a generator update was enqueued and ev_gen_pass does not cover it yet (error path, single steps)
    hipEvent_t lv_ev[256] = {};  // per level and half: event pairs around level_score_kernel ([4 * level + 2 * half + {0, 1}])
    // two-half walk launches (walk_sample.hip, run_levels): second stream, fork / join events, the chain of the score kernels
    hipStream_t stream3 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_score[2] = {nullptr, nullptr};
    bool split_enabled = false; // GG_WALK_SPLIT=1 switches the two-half launches on (measured slower on the bench workload, see run_levels)
    int64_t split_min_walks = 32768;  // GG_WALK_SPLIT_MIN: smaller launches stay on one stream (their levels are latency bound end to end)
    bool w_split = false;       // the current launch runs as two halves
    gg::DevBuf fin_list;        // walks handed to the finisher
    gg::DevBuf table_bad;       // two device words: a table of the generator / discriminator holds a non-finite value (rescan_table_finite, optimizer kernels)
    int lv_ev_used = 0;
    gg::Model model[2];  // 0 = generator, 1 = discriminator (config.modes order)

    // dense gradient accumulators shared by D and G steps (zero between steps)
    float *gradE = nullptr, *gradb = nullptr;
    // touched-row bookkeeping for the lazy / sgd modes
    int32_t *touched = nullptr;     // [n_node] flag
    int32_t *touched_list = nullptr;  // [n_node]
    int32_t *touched_cnt = nullptr;   // [1]

    // graph CSR (utils.read_edges adjacency, list order)
    int64_t *g_rowptr = nullptr;
    int32_t *g_col = nullptr;
    int64_t g_nnz = 0;
    uint64_t g_hash = 0;  // fingerprint of the adjacency (tree cache validation)
    std::vector<int64_t> h_rowptr;  // host copies (tree builder, degrees)
    std::vector<int32_t> h_col;
    // Edge-score cache (walk_sample.hip): s(u, col[e]) = g_u . g_col[e] + b[col[e]] of graph edge e (generator.py:21) does not
    // depend on the ROOT whose tree a walk moves in, so the walks of all roots, levels and both launches of a step share
    // one copy: es[e], valid when es_stamp[u] says adj(u) was scored since the generator last changed.
    // Stamps are TICKS of a 64-bit clock the host advances: every score kernel of every walk launch has its own tick (a launch
    // reserves 256: tick = launch base + 2 * level + half), and es_stamp[u] = tick of the score kernel that fills adj(u)'s
    // scores.  A reader whose own score kernel has tick T may gather from u iff es_valid_from <= es_stamp[u] <= T: scored
    // since the generator last changed, by a kernel that is stream-ordered before the reader's weights kernel (the two
    // halves of a split launch chain their score kernels by events, so ticks order them).
    float *es = nullptr;          // [g_nnz]
    long long *es_stamp = nullptr;  // [n_node]
    int32_t *g_rev = nullptr;     // [g_nnz] index of the reverse edge (col[e] -> u): a walk's father candidate
    long long es_tick = 256;      // next free tick
    long long es_valid_from = 256;  // moved to es_tick by every generator update / table upload / aborted launch
    int32_t es_mode = 1;          // GG_ES_MODE: 0 = off (every distribution scores its own candidates), 1 = policy, 2 = always share
    int32_t es_ratio_num = 2;     // GG_ES_RATIO: a stale node is scored whole when k * ratio >= deg (k = candidates the asking root needs)
    int32_t es_hub = 0;           // GG_ES_HUB: ... or when deg >= hub (0 = off)

    // BFS trees of n_tree_roots root slots in BFS-ORDER form.  For slot r with C_r reached nodes:
    //   t_order [t_base[r] + i]       node id of BFS pop rank i (rank 0 = the root): the reference's queue
    //   t_cstart[t_base[r] + r + i]   first child rank of rank i (i = 0 .. C_r; cstart[0] = 1, cstart[C_r] = C_r):
    //                                 the children of rank i are the CONSECUTIVE ranks [cstart[i], cstart[i+1]) -- a FIFO
    //                                 BFS appends the children of one node next to each other, in adjacency order
    // so tree[v] of the reference = [father] ++ order[cstart[i] .. cstart[i+1]) and a walk (which only ever moves DOWN
    // until its terminating back-step) needs no father array: the father of its current node is its previous node.
    // 8 B per (root, node) instead of the 12 B of a node-id-indexed offsets array + lists, written front to back by the BFS.
    // Q3 (graph_gan.py:258-259: D-mode removes the father entry of visited depth-1 children for good): one bit per
    // (root, child of the root), t_q3[t_q3off[r] * 32 + (rank - 1)].
    int32_t n_tree_roots = 0, tree_max_depth = 0, tree_max_list = 0;
    int64_t tree_nodes = 0;      // sum of C_r
    int64_t t_cap_nodes = 0, t_cap_roots = 0, t_cap_q3 = 0;  // allocated capacity of the tree arrays (kept across rebuilds)
    int64_t tree_entries = 0;    // entries of the reference-shaped lists: sum of 2 C_r - 1 (gg_tree_info / gg_get_trees)
    int32_t *t_root = nullptr;   // [R] root node id of each slot
    int32_t *t_order = nullptr;  // [tree_nodes]
    int32_t *t_cstart = nullptr; // [tree_nodes + R]
    // t_edge[t_base[r] + i]: CSR index of the graph edge (father -> node) the BFS appended rank i at (-1 for rank 0); the
    // child's score against its father is then ONE gather from the edge-score cache below.  Written by the GPU BFS in
    // the same pass as t_order; derived by tree_edges_kernel for host-built / uploaded / cached trees.  Not valid
    // (t_edge_valid = false) for uploaded lists that are no subgraph of the resident graph: those walks score privately.
    int32_t *t_edge = nullptr;   // [tree_nodes]
    bool t_edge_valid = false;
    int64_t *t_base = nullptr;   // [R+1]
    uint32_t *t_q3 = nullptr;    // [t_q3off[R]] words
    int64_t *t_q3off = nullptr;  // [R+1] word offsets (ceil(deg(root) / 32) words per slot: children of the root <= its degree)
    std::vector<int32_t> h_troot;
    std::vector<int64_t> h_tbase, h_q3off;
    std::vector<int32_t> h_comp_size;  // |component| of every node (one host sweep per graph, cached): C_r before any BFS runs

    // LAZY trees (round 6; bfs_gpu.hip "LAZY", walk_sample.hip "LAZY"): gg_build_trees_device builds a root's tree exactly only
    // THROUGH A LEVEL L_r -- the last level whose expansion is known to fit the slot's node limit -- and the walks resolve the
    // children list of a deeper node the first time one of them stands on it.  Slot r then holds
    //   ranks [0, lzs)        levels < L_r: exact, children through t_cstart as always
    //   ranks [lzs, P)        level L_r: exact queue entries (t_order / t_edge) whose children lists are NOT built
    //   ranks [P, seg)        the POOL: children lists resolved on demand, appended by an atomic cursor (lz_cursor)
    // and for a rank >= lzs the children range is lz_pair[t_base[r] + rank] = start << 32 | count << 12 | stamp (valid iff its
    // stamp is the build's; count 0xFFFFF = being resolved).  Resolution needs, per slot, the visited set of levels <= L_r with
    // the BFS rank of every member: lz_bm[r][w] = {bitmap word w, members below word w} and lz_rank[t_base[r] + index] = rank,
    // index = members below the node (popcount) -- a node's rank is two loads, the second only for members.
    // A slot whose walks need more than that (a level L_r + 3 node, a full pool) raises lz_flag[r]: walk_finalize rebuilds those
    // roots whole in the ARENA behind the slots' segments (t_base[r] moves there, lz_info[r].x = its node count) and reruns.
    int32_t tree_mode = -1;            // gg_set_tree_mode / GG_TREE_LAZY: 0 = whole trees, 1 = lazy, -1 = lazy from GG_LZ_AUTO_NODES nodes on
    int64_t lz_cap = 0;                // node limit of a slot's exact part (0 = the node count, components up to 65 536 nodes whole; GG_LZ_CAP)
    bool t_lazy = false;               // the resident trees are lazy
    bool lz_force_whole = false;       // (the rebuild of a lazy batch as whole trees is under way)
    int32_t lz_min_level = 0x7fffffff; // smallest L_r of the resident lazy slots (levels below it need no resolve step)
    uint32_t lz_stamp = 0;             // stamp of the resident build (1 .. 4095; the pair array is cleared when it wraps)
    gg::DevBuf lz_info, lz_pair, lz_rank, lz_bm, lz_cursor, lz_flag, lz_limit, lz_expect, lz_list, lz_scratch, lz_fb_scratch;
    int64_t arena_next = 0, arena_end = 0;  // the arena of whole trees behind the segments: [arena_next, arena_end) is free
    int64_t lz_fallback_roots = 0, lz_fallback_rounds = 0, lz_resolved = 0;  // statistics (gg_lazy_stats)
    bool g_multi = true;               // the adjacency holds a node twice in some list (first-occurrence tests needed)
    int32_t g_max_deg = 0;

    // walk outputs (device resident)
    // walk inputs (slot list, walk offsets): one resident copy per mode (index 1 = D-mode, 0 = anything else) with the host
    // shadow that tells whether the next call brings the same lists -- the trainer alternates D and G calls over the same
    // roots, and two staged pageable H2D copies at the head of every call were ~40 us of its critical path
    gg::DevBuf w_slots_m[2], w_ptr_m[2];
    std::vector<int32_t> h_slots_m[2];
    std::vector<int64_t> h_ptr_m[2], h_ptr_new;
    int w_mode = 0;                       // the mode of the current call
    // walks alive per streamed level in the previous launch of each mode (-1 = unknown): where the next launch hands over
    // to the finisher (walk_sample.hip, run_levels_and_finish); GG_FIN_THRESHOLD, 0 = stream every level
    int64_t alive_prof[2][64];
    int64_t fin_threshold = 0;
    int w_levels_run = 0;
    bool w_fin_follows = false;
    bool w_net = false;   // the finisher ran as the safety net behind the learned number of levels (walk_sample.hip, run_levels_and_finish)
    const gg::DevBuf &w_slots_buf() const { return w_slots_m[w_mode]; }
    const gg::DevBuf &w_ptr_buf() const { return w_ptr_m[w_mode]; }
    gg::DevBuf w_samples, w_paths, w_len, w_status, w_first, w_abort, w_scratch;
    // level-synchronous front end of the walk sampler (walk_sample.hip): per-walk state + per-level tasks
    gg::DevBuf st_cur, st_prev, st_len, st_alive, st_rank, st_item, st_item2, lv_beg, lv_k, lv_chunks, lv_coff, lv_scores, lv_chunk_owner, lv_prefix, lv_big, lv_fe;
    gg::DevBuf lv_pfx, dc_keys, dc_vals, dc_words;  // distribution cache (walk_sample.hip): prefix offsets per walk, hash table, base words
    size_t dc_size = 0;                // hash table entries (power of two)
    int32_t dc_request = 0;            // mode of the NEXT walk launch: 0 off, 1 register (gg_prepare_d), 2 look up (gg_prepare_g)
    bool dc_valid = false;             // the D launch's distributions match the generator's current tables and the resident trees
    bool dc_enabled = true;            // GG_NO_DIST_CACHE=1 switches it off
    int64_t lv_cap_total = 0;          // learned capacity (chunks over all hops, D + G launch) of the prefix buffer
    int32_t lv_levels_learned = 0;     // hops earlier (sized) launches needed until every walk had finished
    int64_t lv_cap_chunks = 0;         // learned capacity (chunks per level) for the sync-free launches
    bool walk_force_sized = false;     // retry path after a speculative overflow
    bool walk_used_speculation = false;
    int32_t walk_levels = 64;  // hops handled by the streaming level kernels before the per-walk finisher (GG_WALK_LEVELS)
    int64_t w_total = 0;
    int32_t w_stride = 0, w_nslots = 0;
    int32_t w_uniform = -1;  // walks per root of the resident launch when every root has the same number (prepare_g), else -1
    struct { int32_t for_d; uint64_t seed; uint32_t stream; } w_args{};

    // pinned host mirror: [0, 2 * CTR_WORDS) the launch's device counters (two halves), [H_TOTAL] the row / pair count of a prepare call --
    // both arrive with asynchronous copies behind the kernels and ONE stream synchronisation (pageable destinations
    // would make every copy its own host round trip)
    static constexpr int CTR_WORDS = 720;  // counter words per half of a walk launch (walk_sample.hip)
    static constexpr int PIN_WORDS = 2048;
    static constexpr int H_TOTAL = 1960;
    static constexpr int H_ROWS = 1940;   // [H_ROWS + k]: touched-row count of pending pass timing k (copied behind its optimizer kernel)
    unsigned long long *h_pin = nullptr;  // [PIN_WORDS], hipHostMalloc
    // profiling (gg_set_profiling): HIP events around every profile_every-th walk call; 1 = every call and every pass
    // (passes then wait for their events), 0 = never; an event pair costs ~6 us of stream bubble on each side
    int32_t profile_every = 1;
    int64_t walk_call_index = 0;
    bool walk_timed = false;  // the launch in flight carries events
    bool profile_solo = true; // profiled side-stream walks wait for the main stream first (measured alone)
    // per-kernel timing of prepare / pass calls: event triples recorded on `stream`, harvested at the next host sync
    struct PendingTiming { int kind; int64_t units; int slot; bool has_rows; };  // kind 0 = reward, 1 = D pass, 2 = G pass
    hipEvent_t tm_ev[8][3] = {};
    std::vector<PendingTiming> tm_pending;
    int64_t pass_call_index[2] = {0, 0};
    int tm_cur = -1;  // event triple of the pass being enqueued (steps.hip records its middle event behind the gradient kernel)

    // prepared data
    gg::DevBuf d_center, d_neighbor, d_label, d_cnt, d_ptr;
    int64_t d_rows = 0;
    gg::DevBuf g_node1, g_node2, g_reward, g_cnt, g_ptr;
    int64_t g_pairs = 0;
    bool g_paths_valid = false;  // w_paths / g_ptr still describe the resident prepare_g data
    // gg_prepare_g_begin: a G-mode walk launch enqueued on the side stream and NOT yet joined / finalized; adopted by the next
    // gg_prepare_g with the same arguments, discarded (side stream drained) by anything else that needs the walk buffers, the
    // trees or the generator's tables
    bool g_begun = false;
    struct { int32_t n_slots = 0, n_sample = 0; uint64_t seed = 0; uint32_t stream = 0; } g_begun_args;
    gg::DevBuf touched_ptr;
    gg::DevBuf bfs_key, bfs_bm, bfs_misc, bfs_sparse, bfs_rowptr32;  // scratch of gg_build_trees_device (bfs_gpu.hip)
    bool bfs_rowptr32_valid = false;                                 // ... its 4-byte copy of the row offsets matches the graph set
    // epoch over root batches (epoch.hip): persistent Q3 bits of EVERY root -- words [q3s_off[v], q3s_off[v + 1]) for root node v,
    // ceil(deg(v) / 32) of them, bit (rank - 1) for the child of BFS rank `rank` as in t_q3 -- and the rows / pairs the batches
    // of the running epoch have produced so far
    gg::DevBuf q3_store, q3s_off;
    int64_t q3_words = 0;
    bool q3_store_ready = false;
    gg::DevBuf ep_center, ep_neighbor, ep_label, ep_node1, ep_node2, ep_reward;
    int64_t ep_rows = 0, ep_pairs = 0;
    std::vector<int32_t> ep_slots;     // 0, 1, 2, ... (the slots of a batch)
    bool in_epoch_add = false;         // gg_prepare_* inside gg_epoch_add: no per-batch replica collective (the ranks' batch counts differ)
    int n_cus = 256;                       // compute units of the device (gg_create; hipGetDeviceProperties costs milliseconds)
    gg::DevBuf scan_tmp, step_u, step_v, step_x;
    // staged generator gradient (steps.hip, run_path_step): per-row counts / segment offsets, per path node slot, row list,
    // the stage itself (gradient rows + bias gradients of the small rows, segment by segment), two total words
    gg::DevBuf sg_cnt, sg_off, sg_slot, sg_list, sg_rows, sg_bias, sg_tot, sg_key;  // sg_key: source of each stage row (sum order)
    bool g_pairs_filled = false;       // g_node1 / g_node2 hold the pairs of the resident G walks (prepare.hip, ensure_g_pairs)
    bool sg_cnt_dirty = false;         // a staged pass counted rows and has not (yet) applied them
    // The generator pass's staging INDEX (per-row counts, segment offsets, slots, small-row list, keys) depends on the G-mode walks
    // only, so it is built on the side stream right behind them (enqueue_path_slots, steps.hip) -- beside the reward kernels
    // of the main stream -- in buffers of its own (the discriminator pass uses sg_* at the same time); ev_slots_done orders the
    // gradient kernel behind it.
    gg::DevBuf sgp_cnt, sgp_off, sgp_slot, sgp_list, sgp_tot, sgp_key, sgp_scan;
    hipEvent_t ev_slots_done = nullptr;
    bool g_slots_ready = false;        // the index of the resident G-mode walks is enqueued / done
    bool sgp_cnt_clean = false;        // sgp_cnt is all zero
    int g_slots_unused = 0;            // early indexes in a row that no whole-walk generator pass consumed (2: stop building them)
    int64_t g_slots_walks = 0;
    int32_t g_slots_stride = 0;
    int32_t *sg_cnt_active = nullptr;  // the count array of the staged pass that is applying its hub rows (sg_active)
    bool sg_active = false;            // a staged G pass is applying its hub rows (apply_optimizer resets their counts)
    int sg_threshold = 64;             // GG_STAGE_T: rows with more staged gradients than this keep the atomic path; 0 = everything atomic

    // device-side counters of the walk launch in flight (zeroed at its start): [0]=hops [1]=nbr_reads [3]=error flag
    // [4]=ticket [5]=rows scored by the finisher; per-level and spread words from [8] on (walk_sample.hip)
    unsigned long long *dev_ctr = nullptr;
    gg_counters ctr{};

    // multi-GPU
    void *comm = nullptr;  // ncclComm_t
    bool comm_rsag = true;             // dense exchange as reduce-scatter + all-gather (GG_COMM_DENSE=allreduce: one all-reduce)
    size_t grad_elems_padded = 0;      // length of gradE (n_node * ld rounded up to a multiple of the world size)
    int64_t d_rows_max = 0, g_pairs_max = 0;  // max over ranks of the prepared rows / pairs (exchanged inside gg_prepare_*)
    int64_t step_bound = 0;            // pairs any rank can contribute to the step being enqueued -> capacity of its row packs
    int64_t comm_steps_sparse = 0, comm_steps_dense = 0, comm_bytes_sent = 0;  // gg_comm_stats
    int32_t rank = 0, world = 1;
    bool deterministic = true;   // atomic-free single-workgroup gradient kernel for batches <= 256 pairs (the reference's batch 64); GG_DETERMINISTIC=0: atomics
    float dense_exchange_ratio = 1.5f;  // sparse exchange only while (rows touched over all ranks) < ratio * n_node (GG_COMM_DENSE_RATIO)
    int32_t fake_world = 0;  // GG_COMM_FAKE_WORLD=k: exercise the k-rank exchange code on one GPU (every rank = this one)
    gg::DevBuf x_cnt, x_send_ids, x_send_rows, x_recv_ids, x_recv_rows;  // sparse gradient exchange
    gg::DevBuf x_nglob;  // pairs of all ranks in the generator step in flight (device word)
    gg::DevBuf x_own;    // owner-partitioned exchange: per-owner counts / fill cursors / the gathered count matrix (int64 words)
    int32_t owner_exchange = 1;       // GG_COMM_OWNER=0 switches the owner-partitioned sparse exchange off
    bool comm_bf16 = false;           // GG_COMM_BF16=1: the owner-partitioned exchange moves bf16 rows (fp32 sums at the owner; the reduced rows are rounded once more)
    int64_t owner_min_bound = 4096;   // GG_COMM_OWNER_MIN: steps of fewer pairs never try it (its two host round trips outweigh a minibatch)
    int64_t comm_steps_owner = 0;     // optimizer steps that took it (gg_comm_stats counts them with the sparse steps)

    std::string err;
};

namespace gg {

extern thread_local std::string g_last_error;

int fail(gg_ctx *ctx, int code, const char *fmt, ...);

#define GG_HIP(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return gg::fail(ctx, GG_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,         \
                            hipGetErrorString(e__));                                           \
    } while (0)

#define GG_CHECK(ctx, cond, code, ...)                        \
    do {                                                      \
        if (!(cond)) return gg::fail(ctx, code, __VA_ARGS__); \
    } while (0)

// exclusive scan of n int32 counts into n+1 int64 offsets (prepare.hip)
int device_exclusive_scan(gg_ctx *ctx, const int32_t *cnt, int64_t *ptr, int64_t n);
int ensure_g_pairs(gg_ctx *ctx);  // prepare.hip: (node_1, node_2) of the resident G walks, written on first use
int device_compact_flags(gg_ctx *ctx, const int32_t *flag, int64_t n, int32_t *list, int64_t *total_out);  // prepare.hip
int device_segment_rows(gg_ctx *ctx, const int32_t *cnt, int64_t n, int T, int32_t *off, int4 *list, int64_t *totals, hipStream_t stream = nullptr,
                        DevBuf *scratch = nullptr);  // prepare.hip (defaults: the main stream and its scan scratch)

// trees (gg_api.hip / tree_builder.cpp)
int alloc_trees(gg_ctx *ctx, const int32_t *roots, int32_t n_roots, const int64_t *node_counts, const int64_t *root_children, int64_t extra_nodes = 0);
bool lazy_build_wanted(const gg_ctx *ctx);            // bfs_gpu.hip: is the next gg_build_trees_device lazy?
int lazy_rebuild_whole(gg_ctx *ctx);                  // bfs_gpu.hip: the resident lazy batch again, as whole trees
int lazy_fallback_rebuild(gg_ctx *ctx, int *n_out);   // bfs_gpu.hip: whole trees (arena) for the slots whose walks raised lz_flag
void component_sizes(int n, const int64_t *rowptr, const int32_t *col, std::vector<int32_t> &comp_size);
// FIFO BFS of one root into BFS-order form; returns C (order[0..C), cstart[0..C]); scratch: n-entry stamp array + epoch
int32_t host_bfs_order(const int64_t *rowptr, const int32_t *col, int32_t root, int32_t *order, int32_t *cstart,
                       std::vector<uint32_t> &stamp, uint32_t &epoch, int32_t *depth_out, int32_t *max_children_out);
// BFS-order form (+ Q3 bits, may be NULL) of one root -> the reference-shaped lists: off[n+1], nbr[2C-1]
void order_to_lists(int32_t n, int32_t C, const int32_t *order, const int32_t *cstart, const uint32_t *q3, int32_t *off, int32_t *nbr);
// reference-shaped lists of one root -> BFS-order form; returns C or -1 if the lists are not a tree of `root`
int32_t lists_to_order(int32_t n, int32_t root, const int32_t *off, const int32_t *nbr, int32_t *order, int32_t *cstart,
                       uint32_t *q3, int32_t q3_words, int32_t *depth_out, int32_t *max_children_out);

// launchers
int walk_launch_async(gg_ctx *ctx, const int32_t *slots, const int32_t *n_walks, int32_t uniform_walks, int32_t n_slots,
                      int32_t for_d, uint64_t seed, uint32_t stream, int32_t stride, bool side_stream = false, bool defer_join = false);
void discard_begun_walk(gg_ctx *ctx);  // gg_api.hip: drain and forget a gg_prepare_g_begin launch nobody adopted
int walk_finalize(gg_ctx *ctx, bool *retried);
int timing_slot(gg_ctx *ctx);
int check_exchange_flag(gg_ctx *ctx);
void harvest_timings(gg_ctx *ctx);
int derive_tree_edges(gg_ctx *ctx);   // walk_sample.hip: t_edge from t_order / t_cstart and the resident graph (sets t_edge_valid)
int compute_reverse_edges(gg_ctx *ctx);  // walk_sample.hip: g_rev of the resident graph
int rescan_table_finite(gg_ctx *ctx, int which);  // gg_api.hip
int enqueue_path_slots(gg_ctx *ctx);  // steps.hip: staging index of the G-mode walks just enqueued, on the side stream
void generator_changed(gg_ctx *ctx);  // gg_api.hip: cached distributions and edge scores are stale  // after a synchronisation of ctx->stream: fold finished event triples into the counters
int walk_resident(gg_ctx *ctx, const int32_t *slots, const int32_t *n_walks, int32_t uniform_walks, int32_t n_slots,
                  int32_t for_d, uint64_t seed, uint32_t stream, int32_t stride);
int launch_walk_sample(gg_ctx *ctx, int32_t n_slots, int64_t total_walks, int for_d, uint64_t seed, uint32_t stream,
                       int32_t stride);
int run_step(gg_ctx *ctx, int which, const int32_t *d_u, const int32_t *d_v, const float *d_x, int32_t n);
int comm_allreduce_grads(gg_ctx *ctx);
int comm_allreduce_flags(gg_ctx *ctx);
int comm_allreduce_i64(gg_ctx *ctx, int64_t *buf, size_t count);
int comm_allreduce_max_i64(gg_ctx *ctx, int64_t *buf, size_t count);
size_t comm_dense_bytes(const gg_ctx *ctx);
int exchange_count_max(gg_ctx *ctx, int64_t local, int64_t *max_out);  // steps.hip: max over ranks of a prepared row / pair count
int comm_allgather(gg_ctx *ctx, const void *send, void *recv, size_t count, int elem_bytes);
bool comm_has_p2p(const gg_ctx *ctx);
int comm_exchange_v(gg_ctx *ctx, const float *send, const int64_t *send_off, const int64_t *send_cnt, float *recv, const int64_t *recv_off,
                    const int64_t *recv_cnt);
void comm_destroy(gg_ctx *ctx);

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace gg
