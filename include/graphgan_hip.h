/*
 * graphgan_hip.h -- C ABI of libgraphgan_hip.so, the MI355X (gfx950) engine for the
 * GraphGAN hot path.  Plain C, no torch / numpy types: pointers, sizes, scalars.
 *
 * The reference (hwwang55/GraphGAN) has no FFI: its "operator API" is the five
 * tf.Session.run call sites of src/GraphGAN/graph_gan.py plus the host-side sampler.
 * Every entry point below names the reference interface it replaces (file:line under
 * /root/reference).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every function returns int: 0 = GG_OK, negative = GG_E*; no exceptions / aborts cross
 *     the ABI; gg_last_error() gives the message of the last failing call
 *     (ctx == NULL: the last gg_create / host-only failure of the calling thread).
 *   - the caller owns every host buffer it passes (C-contiguous; int32 node ids - the
 *     reference's placeholders are tf.int32, generator.py:17-18; fp32 values; int64
 *     offsets).  Inputs are copied; outputs are written before the call returns.  By default
 *     every call is synchronous (it returns after the context's HIP stream is idle).  The two
 *     exceptions are opt-in: after gg_set_profiling(ctx, k) with k != 1, gg_d_pass / gg_g_pass
 *     return once their kernels are ENQUEUED (stream-ordered behind everything issued before;
 *     they have no host outputs) -- errors of such a pass surface at the next call that
 *     synchronises (gg_prepare_*, gg_get_*, gg_synchronize); and gg_prepare_g_begin, which only
 *     enqueues the walks of the gg_prepare_g that follows (errors surface in that call).
 *   - a gg_ctx owns all device memory (embedding tables, Adam slots, graph CSR, tree CSR,
 *     prepared sample buffers, scratch) and one HIP stream on one device; it is not
 *     re-entrant.  Multi-GPU = one process and one context per GPU (gg_comm_*).
 *   - per-root outcomes (the reference's ``return None, None``) are DATA (root_status),
 *     not errors.
 *   - there is no CPU fallback: without a gfx950 device gg_create fails with GG_EHIP.
 */
#ifndef GRAPHGAN_HIP_H
#define GRAPHGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GG_ABI_VERSION 7  /* round 6: lazy trees (gg_set_tree_mode, gg_lazy_stats, gg_get_lazy_trees); round 5: gg_comm_stats_ex; round 4: epoch over root batches (gg_epoch_*, gg_q3_*); round 3: gg_counters extended, gg_prepare_g_begin */

enum {
    GG_OK = 0,
    GG_EINVAL = -1,    /* bad argument / call order */
    GG_ECAPACITY = -2, /* caller capacity too small (e.g. path stride) */
    GG_EHIP = -3,      /* HIP runtime error, or no usable device */
    GG_ECOMM = -4,     /* RCCL error / librccl not loadable */
    GG_ENOMEM = -5,
    GG_EIO = -6
};

/* optimizer modes (a12) */
enum {
    GG_OPT_ADAM_DENSE = 0, /* TF1.8 sparse-apply semantics: m, v and var move over ALL rows each step */
    GG_OPT_ADAM_LAZY = 1,  /* only touched rows decay/move (scale mode) */
    GG_OPT_SGD = 2         /* var -= lr * grad on touched rows (north-star SGD variant) */
};

/* root_status values written by gg_walk_sample / gg_prepare_* */
enum {
    GG_ROOT_OK = 0,
    GG_ROOT_ABORTED = 1, /* reference: sample() returned (None, None), graph_gan.py:252-257 */
    GG_ROOT_EMPTY = 2    /* zero walks requested (D-mode, deg == 0): reference returns ([], []) */
};

/* hyper-parameters, names as in src/GraphGAN/config.py:4-25 */
typedef struct gg_config {
    float lr_gen;      /* config.py:9  */
    float lr_dis;      /* config.py:10 */
    float lambda_gen;  /* config.py:6  */
    float lambda_dis;  /* config.py:7  */
    float adam_beta1;  /* tf.train.AdamOptimizer default 0.9   */
    float adam_beta2;  /* 0.999 */
    float adam_eps;    /* 1e-8  */
    int32_t window_size; /* config.py:25 */
    int32_t optimizer;   /* GG_OPT_* */
    int32_t device;      /* HIP device ordinal */
    int32_t reserved[6];
} gg_config;

typedef struct gg_counters {
    int64_t walks;          /* walks completed since creation */
    int64_t hops;           /* sampled edges: one softmax-sample each (graph_gan.py:262-263) */
    int64_t nbr_reads;      /* sum over hops of k = tree neighbours scored */
    int64_t reward_pairs;   /* rows through gg_pair_reward / prepare_g */
    int64_t d_pairs;        /* rows through d steps */
    int64_t g_pairs;        /* rows through g steps */
    int64_t d_steps;
    int64_t g_steps;
    double last_kernel_ms;  /* HIP-event time of the last timed kernel region (walk / pass) */
    double walk_kernel_ms;  /* cumulative HIP-event time of the profiled gg_walk_sample / prepare walk launches */
    int64_t walk_launches;  /* ... and their number */
    int64_t rows_scored;    /* neighbour rows actually streamed: identical (root, node) distributions of one
                               launch are evaluated once and shared by the walks that need them */
    /* timing counters cover the PROFILED walk calls only (gg_set_profiling; by default every call) */
    double score_kernel_ms; /* cumulative HIP-event time of level_score_kernel (the dominant kernel) */
    int64_t score_launches;
    int64_t score_chunks;   /* 16-candidate work items those launches processed */
    int64_t score_rows;     /* neighbour rows those launches streamed */
    double bfs_kernel_ms;   /* cumulative HIP-event time of the BFS-tree kernel (gg_build_trees_device) */
    int64_t bfs_trees;      /* ... and the trees it built */
    int64_t score_dists;    /* (root, node) distributions the timed score launches evaluated (one current row each) */
    /* per-kernel HIP-event times of the PROFILED prepare / pass calls (same cadence as the walks), with the units they
     * processed: K2 pair_reward; K3 / K4 gradient kernel and K5 optimizer kernel of the discriminator / generator pass */
    double reward_kernel_ms;
    int64_t reward_pairs_timed;
    double d_grad_ms, d_opt_ms;
    int64_t d_pairs_timed, d_rows_timed;   /* pairs of those passes; table rows their optimizer kernels updated */
    double g_grad_ms, g_opt_ms;
    int64_t g_pairs_timed, g_rows_timed;
    int64_t d_passes_timed, g_passes_timed;
    int64_t g_walk_nodes_timed;            /* path nodes (= hops) of the gg_prepare_g calls counted in reward_pairs_timed */
    /* edge-score cache of the walk sampler: the score of a graph edge does not depend on the root, so a node's adjacency is
     * scored once per generator state and the (root, node) distributions of all roots gather from it */
    int64_t es_gathers;     /* (root, node) distributions that took their scores from the cache (no rows streamed) */
    int64_t es_nodes;       /* nodes whose whole adjacency was scored into the cache */
    int64_t score_gathers;  /* ... of the timed launches (the cadence of score_rows / score_dists) */
    int64_t score_nodes;
    int64_t walk_reruns;    /* sync-free walk launches that overflowed their learned buffer capacity / level count and were
                               repeated in sized mode (expected: the first launches of a workload, then none) */
} gg_counters;

typedef struct gg_ctx gg_ctx;

int gg_abi_version(void);
const char *gg_last_error(const gg_ctx *ctx);

/* ---- life cycle.  Replaces build_generator/build_discriminator + tf.Session init
 * (graph_gan.py:48-61, generator.py:11-15, discriminator.py:11-15): two [n_node, n_emb]
 * fp32 tables from the caller's init matrices, zero bias vectors, zero Adam slots. */
int gg_create(int32_t n_node, int32_t n_emb, const float *emb_gen, const float *emb_dis,
              const gg_config *cfg, gg_ctx **out);
int gg_destroy(gg_ctx *ctx);

/* ---- graph + BFS trees.
 * gg_set_graph_csr: the adjacency dict of utils.read_edges (utils.py:12-47) as CSR,
 * neighbour order = list order (it decides BFS child order and D-step positives). */
int gg_set_graph_csr(gg_ctx *ctx, const int64_t *rowptr /*[n_node+1]*/, const int32_t *col);

/* Trees cross the ABI in the REFERENCE'S SHAPE (graph_gan.py:90,96): for root slot r, node v's list
 * [father, child_0, ...] (root: [root, child...]) is nbr[nbr_base[r] + off[r*(n_node+1)+v] .. off[r*(n_node+1)+v+1]),
 * lists in node-id order; 2 * |component| - 1 entries per root.  Inside the context they live in BFS-order form
 * (pop order + first-child rank per rank, 8 bytes per node and root).
 * gg_host_build_trees: construct_trees (graph_gan.py:84-108) on host threads, no GPU, no ctx.
 * nbr == NULL sizes only.  Returns total entries (>= 0) or GG_E*. */
int64_t gg_host_build_trees(int32_t n_node, const int64_t *rowptr, const int32_t *col,
                            const int32_t *roots, int32_t n_roots,
                            int32_t *off /*[n_roots*(n_node+1)]*/, int32_t *nbr, int64_t *nbr_base /*[n_roots+1]*/,
                            int64_t cap, int32_t n_threads, int32_t *max_depth_out);

/* gg_build_trees: same, built in batches and uploaded into the context (replaces the pickle
 * cache load/construct branch, graph_gan.py:31-46).  Root slot i holds the tree of roots[i]. */
int gg_build_trees(gg_ctx *ctx, const int32_t *roots, int32_t n_roots, int32_t n_threads);
/* gg_build_trees_device: the same trees built ON THE GPU: one workgroup per root replays the reference's edge
 * stream (pop order x adjacency order) 4 096 edges at a time against a visited bitmap in LDS; a node is appended
 * at the first edge of the stream that reaches it.  Written straight into the resident BFS-order arrays. */
int gg_build_trees_device(gg_ctx *ctx, const int32_t *roots, int32_t n_roots);
/* gg_set_trees / gg_get_trees: upload / download trees in the reference's shape (tests, foreign caches; the
 * download shows the in-place D-mode mutations, graph_gan.py:258-259, as father entries of -1). */
/* LAZY trees (ABI 7).  The walks of one prepare call read a few hundred of a tree's children lists (graph_gan.py:249-250: the
 * list of the node the walk stands on) -- 5e-5 of what construct_trees (:84-108) writes per root on the 10^6-node bench graph --
 * and the tree of a root that cannot stay resident is built, walked twice and dropped.  In lazy mode gg_build_trees_device builds
 * a root's tree exactly only THROUGH the last level whose expansion is known to fit a node limit (nodes so far + adjacency entries
 * of the level <= limit); the children list of a deeper node is resolved by the first walk that stands on it, from the visited
 * set of the exact levels and their BFS ranks (a child of v = a neighbour outside the set whose first queue neighbour is v, in
 * adjacency order: the list the BFS would have appended, entry for entry) up to two levels below the exact ones.  A root whose
 * walks go deeper -- or whose pool of resolved lists is full -- gets its whole tree behind the scenes and the launch is repeated:
 * every gg_walk_sample / gg_prepare_* result is the whole trees' result, bit for bit (tested against the oracle's).
 * mode: 0 = whole trees (gg_get_trees / gg_save_trees / gg_get_tree_order need them), 1 = lazy, -1 (default) = lazy for graphs of
 * 2^18 nodes and more (GG_LZ_AUTO_NODES); GG_TREE_LAZY overrides.  node_cap: the limit per root; 0 = the default rule: 3/8 of the nodes
 * (at least 65 536; GG_LZ_CAP), except that the root's first two levels may expand up to the node count (GG_LZ_EASY levels), and a
 * root whose component holds <= 65 536 nodes gets its whole tree; node_cap > 0 = exactly that limit for every level and as the
 * whole-tree bound (tests).  gg_tree_info reports, for lazy trees, a depth no walk exceeds.
 * gg_lazy_stats: out24 = {resident trees are lazy, smallest exact level of the lazy slots, slots rebuilt whole so far, launches
 * repeated for it, exact nodes the BFS wrote, pool entries reserved by resolutions, lazy slots, deepest exact level; of the
 * resident build: lists resolved at depth 0 / 1 / 2, candidates judged, 16-entry scan rounds, most rounds of one list, longest
 * adjacency resolved, lists resolved by a whole workgroup (long adjacencies); lazy slots whose exact level is 0 .. 7}.
 * gg_get_lazy_trees (tests): the raw arrays -- info4[slot] = {first rank without a built list, exact ranks, their level,
 * capacity of the segment}, base[slot], then order / cstart / edge / pair over *n_entries entries (cstart: + n_roots). */
int gg_set_tree_mode(gg_ctx *ctx, int32_t mode, int64_t node_cap);
/* gg_debug_words (diagnostics): n of the context's 2 048 device counter words from `first` on. */
int gg_debug_words(gg_ctx *ctx, int32_t first, int32_t n, uint64_t *out);
int gg_lazy_stats(gg_ctx *ctx, int64_t *out24);
int gg_get_lazy_trees(gg_ctx *ctx, int64_t *n_entries, int32_t *info4, int64_t *base, int32_t *order, int32_t *cstart, int32_t *edge, uint64_t *pair);
int gg_set_trees(gg_ctx *ctx, const int32_t *roots, int32_t n_roots, const int32_t *off,
                 const int32_t *nbr, const int64_t *nbr_base, int32_t max_depth);
int gg_tree_info(const gg_ctx *ctx, int32_t *n_roots, int64_t *n_entries, int32_t *max_depth);
int gg_tree_roots(const gg_ctx *ctx, int32_t *roots /*[n_roots]*/);  /* root node of every resident slot */
/* gg_save_trees / gg_load_trees: the tree cache, replacing the pickle of graph_gan.py:31-46 (config.cache_filename).
 * One flat file holding the resident BFS-order arrays as built (roots, bases, pop order, first-child ranks; the
 * reference also pickles before any in-place mutation, :45) behind a header with the graph's fingerprint: a cache
 * built from another graph is refused with GG_EINVAL, a truncated or foreign file with GG_EIO (nothing loaded). */
int gg_save_trees(gg_ctx *ctx, const char *path);
int gg_load_trees(gg_ctx *ctx, const char *path);
int gg_get_trees(gg_ctx *ctx, int32_t *off, int32_t *nbr, int64_t *nbr_base);
/* gg_get_tree_order: the resident trees in their internal BFS-ORDER form, slot by slot: base[r] =
 * first entry of slot r (n_roots + 1 values); order[base[r] + i] = node of BFS pop rank i (the reference's queue,
 * graph_gan.py:96-107); cstart[base[r] + r + i] = rank of the first child of rank i (C_r + 1 values per slot); edge[base[r] + i]
 * = CSR index of the graph edge (father -> node) the BFS appended rank i at (-1 for the root) -- what the walk sampler's
 * edge-score cache is indexed by.  *edges_valid = 0 when the resident trees carry no edge indices (uploaded lists that are
 * no subgraph of the resident graph).  Any pointer may be NULL. */
int gg_get_tree_order(gg_ctx *ctx, int64_t *base, int32_t *order, int32_t *cstart, int32_t *edge, int32_t *edges_valid);

/* ---- K1 walk_sample.  Replaces GraphGAN.sample (graph_gan.py:225-270) including the
 * all_score fetch (:238, generator.py:21), utils.softmax (utils.py:131-133) and
 * np.random.choice (:262).  For each i: n_walks[i] walks on the tree in slot slots[i]
 * (D-mode: for_d = 1, n_walks = len(graph[root]), :190-191; G-mode: n_sample_gen, :210).
 * Walk w = walk_ptr[i] + j writes samples[w] (end node, -1 if the root aborted),
 * paths[w*stride ..] = [root, ..., end, prev-of-end], path_len[w] (0 if aborted).
 * Host output pointers may be NULL (results stay resident for gg_prepare_*).
 * Uniforms: Philox4x32-10(key = seed; counter = hop, j, root id, stream) -> results do not
 * depend on slot order, batching or GPU count.  Arithmetic: DESIGN.md section 3. */
int gg_walk_sample(gg_ctx *ctx, const int32_t *slots, const int32_t *n_walks, int32_t n_slots,
                   int32_t for_d, uint64_t seed, uint32_t stream,
                   int32_t *samples, int32_t *paths, int32_t *path_len, int32_t stride,
                   int32_t *root_status);

/* gg_walk_info / gg_get_walks: shape and contents of the walk outputs left resident by the last gg_walk_sample /
 * gg_prepare_d / gg_prepare_g call (what GraphGAN.sample returned for those roots, graph_gan.py:191,210): total walks,
 * path stride and slot count; then samples [total], paths [total * stride], path_len [total], root_status [n_slots]
 * (any pointer may be NULL).  Used by parity tests that check the walks INSIDE a prepare call. */
int gg_walk_info(const gg_ctx *ctx, int64_t *total_walks, int32_t *stride, int32_t *n_slots);
int gg_get_walks(gg_ctx *ctx, int32_t *samples, int32_t *paths, int32_t *path_len, int32_t *root_status);

/* ---- prepared sample buffers (device resident).
 * gg_prepare_d: prepare_data_for_d (graph_gan.py:182-202) for the given root slots: D-mode
 * walks (n_walks = CSR degree), then rows [pos..., neg...] per non-aborted root, in slot order.
 * gg_prepare_g: prepare_data_for_g (:204-223): n_sample walks per root, window pairs
 * (get_node_pairs_from_path, :272-291) and reward (discriminator.py:33-34) for all pairs.  Its walks read only the
 * generator's tables and the trees, so they are enqueued on a side stream and run beside a gg_d_pass that is still
 * in flight (gg_set_profiling); pairs and rewards follow on the main stream, behind that discriminator update. */
int gg_prepare_d(gg_ctx *ctx, const int32_t *slots, int32_t n_slots, uint64_t seed, uint32_t stream,
                 int64_t *n_rows_out, int32_t *root_status /*[n_slots] or NULL*/);
int gg_get_d_data(gg_ctx *ctx, int32_t *center, int32_t *neighbor, float *label);
int gg_prepare_g(gg_ctx *ctx, const int32_t *slots, int32_t n_slots, int32_t n_sample, uint64_t seed,
                 uint32_t stream, int64_t *n_pairs_out, int32_t *root_status);
/* Optional head start for the gg_prepare_g that follows a discriminator update: enqueue its walks NOW -- on the side stream,
 * without any host synchronisation -- typically between gg_prepare_d and gg_d_pass, so that they run beside the whole
 * discriminator pass instead of starting once the host has enqueued it (graph_gan.py:204-216 samples from the generator
 * only; the rewards of :220-222, which need the updated discriminator, stay in gg_prepare_g).  The next gg_prepare_g with
 * the SAME (slots, n_sample, seed, stream) adopts the launch and returns exactly what it would have returned without
 * this call.  Any other call that needs the walk buffers, the trees or the generator's tables -- another walk launch,
 * gg_get_walks, gg_get_g_data, gg_g_step / gg_g_pass, gg_set_embeddings / gg_set_bias(0), gg_load_state, a tree or graph
 * call -- first waits for the begun launch and drops it; a later gg_prepare_g then launches anew.  Allowed in between
 * without losing the head start: gg_d_step / gg_d_pass, gg_get_d_data, the getters and gg_synchronize. */
int gg_prepare_g_begin(gg_ctx *ctx, const int32_t *slots, int32_t n_slots, int32_t n_sample, uint64_t seed, uint32_t stream);
/* The (node_1, node_2) arrays of gg_prepare_g are expanded from its walks on first use -- gg_get_g_data, or a gg_g_pass in
 * minibatches -- which must therefore come before the next walk launch of the context (gg_prepare_d / gg_prepare_g /
 * gg_walk_sample); later it is GG_EINVAL.  Rewards, and whole-batch passes over the walks, do not need them. */
int gg_get_g_data(gg_ctx *ctx, int32_t *node_1, int32_t *node_2, float *reward);

/* ---- an epoch over ROOT BATCHES (ABI 5).  The reference keeps the tree of every root resident (self.trees, graph_gan.py:31-46)
 * and prepare_data_for_d / prepare_data_for_g visit every root (:188, :208) before the passes run over all rows (:149-157,
 * :168-176).  N trees are N^2 entries (12 TB at 10^6 nodes): where they cannot all be resident, the same schedule runs with
 * one batch of trees in HBM at a time:
 *   gg_epoch_begin   empty the accumulated discriminator rows (reset_d) and / or generator pairs (reset_g);
 *   gg_epoch_add     for the given ROOTS (node ids, not slots): their BFS trees are built on the GPU into slots 0 .. n_roots-1
 *                    (whatever was resident is replaced), the roots' Q3 bits -- father entries that D-mode walks of EARLIER
 *                    epochs removed for good, graph_gan.py:258-259 -- are restored from a store that outlives the trees; do_d:
 *                    gg_prepare_d on them (seed, stream_d), bits saved back, rows appended; do_g: gg_prepare_g (n_sample, seed,
 *                    stream_g), window pairs appended.  do_d and do_g in one call share the trees: the G-mode walks read only
 *                    the generator, the trees and the Q3 bits the D-mode walks of the same root have just set -- nothing the
 *                    discriminator's passes change -- so one BFS per root serves both phases of an outer epoch.  Batches are
 *                    appended in call order; *rows_total_out / *pairs_total_out = accumulated so far.  gg_get_walks /
 *                    gg_get_trees afterwards show the batch's last launch / its trees (mutations included).
 *   gg_epoch_commit  which = 1: the accumulated rows become the resident prepared data of gg_d_pass / gg_get_d_data;
 *                    which = 0: the rewards of ALL accumulated pairs are evaluated now (graph_gan.py:220-222, with the
 *                    discriminator as its passes left it) and the pairs become the data of gg_g_pass / gg_get_g_data.
 *                    With replicas this is where the ranks exchange their counts (once per epoch, not per batch).
 * Rows / pairs equal those of the all-resident calls over the same roots in the same order (tested bit for bit).
 * gg_q3_clear forgets all mutations (a fresh process of the reference: its pickle is written before any, :45);
 * gg_q3_get reads the store: word_off[v] .. word_off[v + 1] = the ceil(deg(v) / 32) words of root v, bit j = the father entry of
 * its (j + 1)-th tree child was removed.  The store is also what gg_epoch_add leaves in the resident slots' own Q3 rows. */
int gg_epoch_begin(gg_ctx *ctx, int32_t reset_d, int32_t reset_g);
int gg_epoch_add(gg_ctx *ctx, const int32_t *roots, int32_t n_roots, int32_t do_d, int32_t do_g, int32_t n_sample, uint64_t seed,
                 uint32_t stream_d, uint32_t stream_g, int64_t *rows_total_out, int64_t *pairs_total_out);
int gg_epoch_commit(gg_ctx *ctx, int32_t which, int64_t *n_out);
int gg_q3_clear(gg_ctx *ctx);
int gg_q3_get(gg_ctx *ctx, int64_t *word_off /*[n_node + 1] or NULL*/, uint32_t *words /*[word_off[n_node]] or NULL*/);

/* gg_d_pass / gg_g_pass: one inner epoch over the prepared rows = the minibatch loops
 * graph_gan.py:149-157 / :168-176: for each s in starts (the caller's shuffled start_list),
 * one optimizer step on rows [s, min(s+batch, n)). */
int gg_d_pass(gg_ctx *ctx, const int64_t *starts, int64_t n_batches, int32_t batch_size);
int gg_g_pass(gg_ctx *ctx, const int64_t *starts, int64_t n_batches, int32_t batch_size);

/* ---- the sess.run call sites with host buffers.
 * gg_pair_reward: sess.run(discriminator.reward) (graph_gan.py:220-222, discriminator.py:21-24,33-34)
 * gg_d_step:      sess.run(discriminator.d_updates) (graph_gan.py:154-157, discriminator.py:26-32)
 * gg_g_step:      sess.run(generator.g_updates)     (graph_gan.py:173-176, generator.py:22-31) */
int gg_pair_reward(gg_ctx *ctx, const int32_t *u, const int32_t *v, int64_t n, float *out);
int gg_d_step(gg_ctx *ctx, const int32_t *u, const int32_t *v, const float *label, int32_t n);
int gg_g_step(gg_ctx *ctx, const int32_t *u, const int32_t *v, const float *reward, int32_t n);

/* gg_all_score: sess.run(generator.all_score) (graph_gan.py:238, generator.py:21) for the given rows:
 * out[i * n_node + j] = g_rows[i] . g_j + b_g[j]; rows == NULL -> every row (n_rows ignored, out is
 * [n_node, n_node]).  Exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32, k-ordered fmaf chain).
 * The engine itself never needs it (gg_walk_sample scores tree neighbours on demand); it is here for
 * callers that still want score rows. */
int gg_all_score(gg_ctx *ctx, const int32_t *rows, int32_t n_rows, float *out);

/* gg_all_score_reduce: the same score rows STREAMED through a fused consumer instead of being materialised: for each
 * requested row i (rows == NULL: every node) the maximum of S[i, :], its column (first maximum, like numpy argmax) and
 * log sum_j exp(S[i, j]) -- the normaliser of the full softmax over all nodes (row_lse may be NULL with want_lse = 0).
 * Nothing of size n_rows x n_node exists at any time.  precision 0: exact fp32 scores on v_mfma_f32_32x32x2_f32;
 * precision 1: a bf16 copy of the table (round to nearest even) on v_mfma_f32_32x32x16_bf16, fp32 accumulate.
 * kernel_ms_out (may be NULL): HIP-event time of the tile stream.  Replaces what a caller of generator.all_score
 * (generator.py:21) would do with the N x N matrix at sizes where it cannot exist (BASELINE.json configs[4]). */
int gg_all_score_reduce(gg_ctx *ctx, const int32_t *rows, int32_t n_rows, int32_t precision, int32_t want_lse, float *row_max,
                        int32_t *row_argmax, float *row_lse, double *kernel_ms_out);

/* sess.run(embedding_matrix) (graph_gan.py:298); which: 0 = generator, 1 = discriminator
 * (config.modes order, config.py:1).  out is [n_node, n_emb] fp32, unpadded. */
int gg_get_embeddings(gg_ctx *ctx, int32_t which, float *out);
int gg_get_bias(gg_ctx *ctx, int32_t which, float *out);
/* gg_write_embeddings: write_embeddings_to_file (graph_gan.py:293-306) for one model, byte-identical to
 * the reference's text (header "N\td\n"; rows "id\tv0\t...\n"; str(float64(fp32)) per value), formatted by
 * host threads.  gg_host_write_embeddings: the same for a caller-supplied [n_node, n_emb] fp32 matrix. */
int gg_write_embeddings(gg_ctx *ctx, int32_t which, const char *path, int32_t n_threads);
int gg_host_write_embeddings(const float *emb, int64_t n_node, int32_t n_emb, const char *path, int32_t n_threads);
/* gg_write_embeddings_bin: binary side-car of that text for large N: {"GGEB", version 1 i32, n_emb i32, n_node i64} + fp32 rows. */
int gg_write_embeddings_bin(gg_ctx *ctx, int32_t which, const char *path);
/* gg_edge_scores: the evaluator's scores (src/evaluation/link_prediction.py:26-27: np.dot(emd[u], emd[v]) per test /
 * negative edge) computed on the device from the resident table of model `which`, in float64 like the reference. */
int gg_edge_scores(gg_ctx *ctx, int32_t which, const int32_t *u, const int32_t *v, int64_t n, double *out);
int gg_set_embeddings(gg_ctx *ctx, int32_t which, const float *emb);
int gg_set_bias(gg_ctx *ctx, int32_t which, const float *bias);

/* tf.train.Saver save/restore (graph_gan.py:55,124-127,137-138): all variables + Adam
 * slots + step counts, as one flat binary file (format: DESIGN.md). */
int gg_save_state(gg_ctx *ctx, const char *path);
int gg_load_state(gg_ctx *ctx, const char *path);

int gg_get_counters(gg_ctx *ctx, gg_counters *out);

/* Profiling cadence.  every_n = 1 (default): HIP events bracket every walk launch, every level_score_kernel launch
 * and every gg_*_pass, and the passes wait for theirs (synchronous, last_kernel_ms valid).  every_n = k > 1: only
 * every k-th walk launch carries events (an event pair costs ~6 us of stream bubble on each side, ~0.1 ms per walk
 * call), and gg_d_pass / gg_g_pass return as soon as their kernels are enqueued -- stream-ordered before whatever
 * is called next; the next call that returns data, gg_synchronize or gg_comm_barrier waits for them and reports
 * their errors.  every_n = 0: no events at all.  Also settable as GG_PROFILE_EVERY. */
int gg_set_profiling(gg_ctx *ctx, int32_t every_n);
/* gg_set_profiling_solo: solo = 1 (default): a profiled side-stream walk first waits for the main stream, so its kernels
 * are measured ALONE; solo = 0: it runs beside the discriminator update like every other launch (overlapped figure). */
int gg_set_profiling_solo(gg_ctx *ctx, int32_t solo);
/* Wait for everything enqueued on the context's stream. */
int gg_synchronize(gg_ctx *ctx);

/* ---- multi-GPU (no reference counterpart: single tf.Session, graph_gan.py:57-61).
 * One process per GPU.  Rank 0 calls gg_comm_unique_id, the 128 bytes travel by any side
 * channel, every rank calls gg_comm_init.  Afterwards each optimizer step sums the
 * gradients of all ranks (RCCL all-reduce on the context's stream) before the update,
 * so replicas stay identical. */
int gg_comm_unique_id(void *id128);
int gg_comm_init(gg_ctx *ctx, const void *id128, int32_t rank, int32_t world);
int gg_comm_barrier(gg_ctx *ctx);
/* gg_comm_stats: out4 = {optimizer steps that exchanged row packs (sparse), steps that exchanged the whole accumulators
 * (dense: reduce-scatter + all-gather), bytes this rank has sent for gradient exchanges so far, world size}. */
int gg_comm_stats(gg_ctx *ctx, int64_t *out4);
/* gg_comm_stats_ex (ABI 6): out8 = {steps that all-gathered fixed-capacity row packs, steps that took the owner-partitioned
 * exchange (send / recv to the rows' owners + all-gather of the reduced rows), dense steps, bytes sent, world size,
 * 1 if the owner-partitioned exchange moves bf16 rows (GG_COMM_BF16=1), 0, 0}.  The first two add up to out4[0]. */
int gg_comm_stats_ex(gg_ctx *ctx, int64_t *out8);

/* ---- edge-list ingest: utils.read_edges (utils.py:12-54) natively -- adjacency CSR in the reference's
 * list order from the train file (+ node ids of the test file); buffers are malloc'ed by the library and
 * released with gg_host_free_graph.  No GPU, no ctx. */
typedef struct gg_graph {
    int32_t n_node;
    int64_t nnz;            /* = 2 * n_train_edges */
    int64_t n_train_edges, n_test_edges;
    int64_t *rowptr;        /* [n_node + 1] */
    int32_t *col;           /* [nnz] */
} gg_graph;
int gg_host_read_edges(const char *train_path, const char *test_path /* may be NULL or "" */, gg_graph *out);
void gg_host_free_graph(gg_graph *g);

/* ---- synthetic power-law graphs for the benchmark configs (BASELINE.json configs[2..4];
 * recipe in SURVEY.md section 8d): Barabasi-Albert, m edges per new node, node ids permuted.
 * edges_out: [n_edges_cap][2] int32; returns the number of edges written or GG_E*. */
int64_t gg_synth_powerlaw(int32_t n_node, int32_t m, uint64_t seed_graph, uint64_t seed_perm,
                          int32_t *edges_out, int64_t n_edges_cap);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHGAN_HIP_H */
