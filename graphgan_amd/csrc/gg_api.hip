// gg_api.hip -- context life cycle, graph / tree residency, walk_sample entry point,
// variable fetch / restore.  The C ABI is include/graphgan_hip.h; each function there cites
// the reference interface it replaces.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>

#include "gg_internal.h"

namespace gg {

thread_local std::string g_last_error;

int fail(gg_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (ctx) ctx->err = buf;
    return code;
}

static void free_trees(gg_ctx *ctx) {
    discard_begun_walk(ctx);
    void *ps[] = {ctx->t_root, ctx->t_order, ctx->t_cstart, ctx->t_base, ctx->t_q3, ctx->t_q3off, ctx->t_edge};
    for (void *p : ps)
        if (p) (void)hipFree(p);
    ctx->t_root = ctx->t_order = ctx->t_cstart = ctx->t_edge = nullptr;
    ctx->t_edge_valid = false;
    ctx->t_base = ctx->t_q3off = nullptr;
    ctx->t_q3 = nullptr;
    ctx->t_cap_nodes = ctx->t_cap_roots = ctx->t_cap_q3 = 0;
    ctx->n_tree_roots = 0;
    ctx->tree_nodes = ctx->tree_entries = 0;
    ctx->tree_max_depth = ctx->tree_max_list = 0;
    ctx->t_lazy = false;
    ctx->lz_min_level = 0x7fffffff;
    ctx->h_troot.clear();
    ctx->h_tbase.clear();
    ctx->h_q3off.clear();
}

// Device arrays for the BFS-order trees of `roots`: node counts C_r (NULL: component sizes from a cached host sweep of
// the graph), Q3 bit rows sized by the roots' child counts (NULL: their degrees, an upper bound), zero-initialised.
int alloc_trees(gg_ctx *ctx, const int32_t *roots, int32_t n_roots, const int64_t *node_counts, const int64_t *root_children, int64_t extra_nodes) {
    discard_begun_walk(ctx);
    (void)hipDeviceSynchronize();  // walks of calls that returned early may still read the old trees
    ctx->dc_valid = false;
    ctx->t_lazy = false;  // (set by the lazy build once its arrays are filled)
    ctx->lz_min_level = 0x7fffffff;
    ctx->t_edge_valid = false;  // set by whoever fills the arrays (GPU BFS: in the same pass; otherwise derive_tree_edges)
    const int n = ctx->n_node;
    if (!node_counts && ctx->h_comp_size.empty()) component_sizes(n, ctx->h_rowptr.data(), ctx->h_col.data(), ctx->h_comp_size);
    ctx->h_tbase.assign(n_roots + 1, 0);
    ctx->h_q3off.assign(n_roots + 1, 0);
    for (int r = 0; r < n_roots; ++r) {
        ctx->h_tbase[r + 1] = ctx->h_tbase[r] + (node_counts ? node_counts[r] : ctx->h_comp_size[roots[r]]);
        const int64_t deg = root_children ? root_children[r] : ctx->h_rowptr[roots[r] + 1] - ctx->h_rowptr[roots[r]];
        ctx->h_q3off[r + 1] = ctx->h_q3off[r] + (deg + 31) / 32;
    }
    const int64_t seg_nodes = ctx->h_tbase[n_roots], q3w = ctx->h_q3off[n_roots];
    const int64_t nodes = seg_nodes + extra_nodes;  // (extra: the arena of whole trees behind the segments of a lazy build)
    const int nr = std::max(n_roots, 1);
    ctx->arena_next = seg_nodes;
    ctx->arena_end = nodes;
    // Trees are rebuilt per root batch when they cannot all stay resident (an epoch over 10^6 roots): keep the arrays
    // of the previous batch when they are large enough instead of returning 64 GB to the driver and asking for it again.
    auto regrow = [&](void **p, size_t bytes) -> hipError_t {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
        return hipMalloc(p, bytes);
    };
    if (nodes > ctx->t_cap_nodes || (ctx->t_cap_nodes > (64 << 20) && nodes < ctx->t_cap_nodes / 4) || nr > ctx->t_cap_roots) {
        // cstart holds nodes + roots entries: both capacities are tied to the same allocation
        const int64_t cn = std::max<int64_t>(nodes, 1), cr = std::max<int64_t>(nr, ctx->t_cap_roots);
        GG_HIP(ctx, regrow((void **)&ctx->t_order, sizeof(int32_t) * (size_t)cn));
        GG_HIP(ctx, regrow((void **)&ctx->t_edge, sizeof(int32_t) * (size_t)cn));
        GG_HIP(ctx, regrow((void **)&ctx->t_cstart, sizeof(int32_t) * (size_t)(cn + 2 * cr + 2)));  // (a slot's row starts at base + slot; arena rows too)
        GG_HIP(ctx, regrow((void **)&ctx->t_root, sizeof(int32_t) * (size_t)cr));
        GG_HIP(ctx, regrow((void **)&ctx->t_base, sizeof(int64_t) * (size_t)(cr + 1)));
        GG_HIP(ctx, regrow((void **)&ctx->t_q3off, sizeof(int64_t) * (size_t)(cr + 1)));
        ctx->t_cap_nodes = cn;
        ctx->t_cap_roots = cr;
    }
    if (q3w > ctx->t_cap_q3 || !ctx->t_q3) {  // the Q3 rows depend on the roots' degrees: leave room for the next batch
        const int64_t cq = std::max<int64_t>(q3w + q3w / 2, 1);
        GG_HIP(ctx, regrow((void **)&ctx->t_q3, sizeof(uint32_t) * (size_t)cq));
        ctx->t_cap_q3 = cq;
    }
    ctx->tree_max_depth = ctx->tree_max_list = 0;
    GG_HIP(ctx, hipMemcpyAsync(ctx->t_root, roots, sizeof(int32_t) * n_roots, hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, hipMemcpyAsync(ctx->t_base, ctx->h_tbase.data(), sizeof(int64_t) * (n_roots + 1), hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, hipMemcpyAsync(ctx->t_q3off, ctx->h_q3off.data(), sizeof(int64_t) * (n_roots + 1), hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, hipMemsetAsync(ctx->t_q3, 0, sizeof(uint32_t) * (size_t)std::max<int64_t>(q3w, 1), ctx->stream));
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_tree_roots = n_roots;
    ctx->tree_nodes = seg_nodes;
    ctx->tree_entries = 2 * seg_nodes - n_roots;
    ctx->h_troot.assign(roots, roots + n_roots);
    return GG_OK;
}

// The generator's tables changed (optimizer step, upload, restore): distributions cached by the D launch and every edge
// score are stale.  Edge scores are invalidated by moving on to the next epoch (stamps of older epochs never match).
void generator_changed(gg_ctx *ctx) {
    ctx->dc_valid = false;
    ctx->es_tick += 256;  // (launches enqueued earlier keep their own ticks: nothing they stamp stays valid)
    ctx->es_valid_from = ctx->es_tick;
}

static int upload_table(gg_ctx *ctx, float *dst, const float *src) {
    const int n = ctx->n_node, d = ctx->n_emb, ld = ctx->ld;
    if (ld == d) {
        GG_HIP(ctx, hipMemcpy(dst, src, sizeof(float) * (size_t)n * d, hipMemcpyHostToDevice));
    } else {
        GG_HIP(ctx, hipMemset(dst, 0, sizeof(float) * (size_t)n * ld));
        GG_HIP(ctx, hipMemcpy2D(dst, sizeof(float) * ld, src, sizeof(float) * d, sizeof(float) * d, n, hipMemcpyHostToDevice));
    }
    return GG_OK;
}

// gg_ctx::table_bad[which] = "a table of model `which` holds a non-finite value": recomputed whenever the host replaces a table
// (create, gg_set_embeddings / gg_set_bias, gg_load_state); between those, the optimizer kernels raise it when they write one.
__global__ __launch_bounds__(256) void table_finite_kernel(const float *E, const float *b, int64_t nE, int64_t nb, unsigned long long *bad) {
    bool any = false;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nE; i += stride) any |= !__builtin_isfinite(E[i]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) any |= !__builtin_isfinite(b[i]);
    if (any) *bad = 1ull;
}

int rescan_table_finite(gg_ctx *ctx, int which) {
    const Model &M = ctx->model[which];
    unsigned long long *w = ctx->table_bad.as<unsigned long long>() + which;
    GG_HIP(ctx, hipMemsetAsync(w, 0, sizeof(unsigned long long), ctx->stream));
    const int64_t nE = (int64_t)ctx->n_node * ctx->ld;
    hipLaunchKernelGGL(table_finite_kernel, dim3((unsigned)std::min<int64_t>(2048, (nE + 255) / 256)), dim3(256), 0, ctx->stream, M.E, M.b, nE, (int64_t)ctx->n_node, w);
    GG_HIP(ctx, hipGetLastError());
    // The scan is only ENQUEUED on the main stream.  A walk launch of the side stream (gg_prepare_g_begin) reads the generator's
    // flag in its reset kernel and is ordered behind the main stream only through ev_gen_pass / gen_dirty: without this it could
    // read the flag before the scan has written it (advisor, round 5).
    if (which == 0) ctx->gen_dirty = true;
    return GG_OK;
}

}  // namespace gg

using namespace gg;

// One walk launch on ctx->walk_stream; a side-stream launch is ordered behind the last generator update and
// hands its completion back to the main stream, where everything that follows the walk is enqueued.
static int launch_and_join(gg_ctx *ctx, int32_t n_slots, int64_t total, int32_t for_d, uint64_t seed, uint32_t stream, int32_t stride, bool join = true) {
    const bool side = ctx->walk_stream != ctx->stream;
    int rc = launch_walk_sample(ctx, n_slots, total, for_d, seed, stream, stride);
    if (rc != GG_OK) {
        if (side) (void)hipStreamSynchronize(ctx->walk_stream);  // nothing half-enqueued may outlive the failed call
        return rc;
    }
    if (side) {
        GG_HIP(ctx, hipEventRecord(ctx->ev_walk_done, ctx->walk_stream));
        if (join) GG_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_walk_done, 0));  // (gg_prepare_g_begin: joined by the adopting gg_prepare_g)
        if (!for_d) {  // behind the walks on their stream: the staging index of the generator pass that will consume them
            rc = enqueue_path_slots(ctx);
            if (rc != GG_OK) return rc;
        }
    }
    return GG_OK;
}

// A launch started by gg_prepare_g_begin that no gg_prepare_g adopted: wait for it and forget it.  What it stamped into the
// edge-score cache is dropped as well (its counter words are never read: whether it completed is unknown).
void gg::discard_begun_walk(gg_ctx *ctx) {
    if (!ctx->g_begun) return;
    (void)hipStreamSynchronize(ctx->stream2);
    ctx->g_begun = false;
    ctx->dc_request = 0;
    ctx->walk_stream = ctx->stream;
    ctx->w_nslots = 0;
    ctx->w_total = 0;
    generator_changed(ctx);
}

// Shared by gg_walk_sample and gg_prepare_*: stage the launch on device and enqueue it (no host
// synchronisation).  n_walks == NULL: CSR degree of each slot's root (D-mode, graph_gan.py:190-191).
int gg::walk_launch_async(gg_ctx *ctx, const int32_t *slots, const int32_t *n_walks, int32_t uniform_walks, int32_t n_slots,
                          int32_t for_d, uint64_t seed, uint32_t stream, int32_t stride, bool side_stream, bool defer_join) {
    discard_begun_walk(ctx);  // (a launch begun earlier uses the buffers this one is about to fill)
    // (an EMPTY launch needs no trees: a rank whose update_ratio draw selected no root still makes the call -- it ends with a
    // collective and resets the resident row count -- possibly before any tree was ever built)
    GG_CHECK(ctx, ctx->n_tree_roots > 0 || n_slots == 0, GG_EINVAL, "walk: no trees loaded (gg_build_trees / gg_set_trees)");
    GG_CHECK(ctx, n_slots >= 0 && (slots || n_slots == 0), GG_EINVAL, "walk: bad slots");
    GG_CHECK(ctx, stride >= 2, GG_ECAPACITY, "walk: stride %d < 2", stride);
    std::vector<int64_t> &ptr = ctx->h_ptr_new;
    ptr.assign(n_slots + 1, 0);
    for (int i = 0; i < n_slots; ++i) {
        GG_CHECK(ctx, slots[i] >= 0 && slots[i] < ctx->n_tree_roots, GG_EINVAL, "walk: slot %d out of range", slots[i]);
        int64_t nw;
        if (n_walks) nw = n_walks[i];
        else if (uniform_walks >= 0) nw = uniform_walks;
        else {
            GG_CHECK(ctx, !ctx->h_rowptr.empty(), GG_EINVAL, "walk: graph CSR needed for D-mode degrees");
            const int r = ctx->h_troot[slots[i]];
            nw = ctx->h_rowptr[r + 1] - ctx->h_rowptr[r];
        }
        GG_CHECK(ctx, nw >= 0, GG_EINVAL, "walk: negative n_walks");
        ptr[i + 1] = ptr[i] + nw;
    }
    const int64_t total = ptr[n_slots];
    GG_HIP(ctx, hipSetDevice(ctx->device));
    const int mode = for_d ? 1 : 0;
    ctx->w_mode = mode;
    GG_HIP(ctx, ctx->w_slots_m[mode].reserve(sizeof(int32_t) * (n_slots + 1)));
    GG_HIP(ctx, ctx->w_ptr_m[mode].reserve(sizeof(int64_t) * (n_slots + 1)));
    GG_HIP(ctx, ctx->w_status.reserve(sizeof(int32_t) * (n_slots + 1)));
    GG_HIP(ctx, ctx->w_abort.reserve(sizeof(int32_t) * (n_slots + 1)));
    GG_HIP(ctx, ctx->w_samples.reserve(sizeof(int32_t) * (total + 1)));
    GG_HIP(ctx, ctx->w_len.reserve(sizeof(int32_t) * (total + 1)));
    GG_HIP(ctx, ctx->w_first.reserve(sizeof(int32_t) * (total + 1)));
    GG_HIP(ctx, ctx->w_paths.reserve(sizeof(int32_t) * ((size_t)total * stride + 1)));
    ctx->walk_stream = side_stream ? ctx->stream2 : ctx->stream;
    if (side_stream && ctx->gen_dirty) {  // generator steps outside a completed pass: wait for everything on the main stream
        GG_HIP(ctx, hipEventRecord(ctx->ev_main_mark, ctx->stream));
        GG_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_main_mark, 0));
        ctx->gen_dirty = false;
    } else if (side_stream && ctx->gen_pass_recorded) {
        GG_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_gen_pass, 0));
    }
    // the lists of this mode's previous call are still resident: upload only what changed (the host shadows outlive the
    // enqueued copies: every public call ends with a stream sync)
    std::vector<int32_t> &hs = ctx->h_slots_m[mode];
    if (hs.size() != (size_t)n_slots || (n_slots && memcmp(hs.data(), slots, sizeof(int32_t) * n_slots) != 0)) {
        hs.assign(slots, slots + n_slots);
        if (n_slots) GG_HIP(ctx, hipMemcpyAsync(ctx->w_slots_m[mode].p, hs.data(), sizeof(int32_t) * n_slots, hipMemcpyHostToDevice, ctx->walk_stream));
    }
    if (ctx->h_ptr_m[mode] != ptr) {
        ctx->h_ptr_m[mode].swap(ptr);
        GG_HIP(ctx, hipMemcpyAsync(ctx->w_ptr_m[mode].p, ctx->h_ptr_m[mode].data(), sizeof(int64_t) * (n_slots + 1), hipMemcpyHostToDevice, ctx->walk_stream));
    }
    ctx->w_total = total;
    ctx->w_stride = stride;
    ctx->w_nslots = n_slots;
    ctx->w_uniform = (!n_walks && uniform_walks >= 0) ? uniform_walks : -1;
    ctx->w_args = {for_d, seed, stream};
    ctx->g_paths_valid = false;
    if (ctx->g_slots_ready) {  // an index being built for the walks this launch overwrites: let its kernels finish reading them
        GG_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_slots_done, 0));
        ctx->g_slots_ready = false;
        ctx->g_slots_unused++;  // nobody took it (the batch-64 schedule consumes the walks pair by pair: enqueue_path_slots)
    }
    if (n_slots == 0) {
        ctx->walk_stream = ctx->stream;
        return GG_OK;
    }
    return launch_and_join(ctx, n_slots, total, for_d, seed, stream, stride, !defer_join);
}

// Wait for the enqueued launch (and whatever the caller enqueued behind it), collect counters,
// rerun once in sized mode if the sync-free launch overflowed its learned buffer capacity.
int gg::walk_finalize(gg_ctx *ctx, bool *retried) {
    if (retried) *retried = false;
    if (ctx->w_nslots == 0) {
        GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return GG_OK;
    }
    const int64_t total = ctx->w_total;
    unsigned long long *c = ctx->h_pin;  // the launch's counter words, one asynchronous read-back into pinned memory
    constexpr int CW = gg_ctx::CTR_WORDS;
    GG_HIP(ctx, hipMemcpyAsync(c, ctx->dev_ctr, sizeof(unsigned long long) * 2 * CW, hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    harvest_timings(ctx);
    for (int attempt = 0; c[3] == 2ull || (c[2] != 0ull && ctx->t_lazy); ++attempt) {
        // nothing the voided launch wrote or counted is final (the D-mode post-pass is gated by the same flag,
        // the counter words are zeroed again by the new launch)
        GG_CHECK(ctx, attempt < 16, GG_EHIP, "walk: the launch was voided %d times in a row", attempt);
        if (c[2] != 0ull && ctx->t_lazy) {
            // LAZY trees: walks of some slots needed more than their lazy tree gives (lz_flag): those slots get their whole trees
            // (arena), everything is walked again -- the other slots' walks come out the same, theirs now complete
            int rebuilt = 0;
            int rc = lazy_fallback_rebuild(ctx, &rebuilt);
            if (rc == GG_ECAPACITY) rc = lazy_rebuild_whole(ctx);  // the arena is full: the whole batch as whole trees
            if (rc != GG_OK) return rc;
            if (ctx->tree_max_depth + 3 > ctx->w_stride) {  // a whole tree may be deeper than the lazy bound
                ctx->w_stride = ctx->tree_max_depth + 3;
                GG_HIP(ctx, ctx->w_paths.reserve(sizeof(int32_t) * ((size_t)total * ctx->w_stride + 1)));
            }
        } else {
            ctx->walk_force_sized = true;
        }
        ctx->ctr.walk_reruns += 1;
        generator_changed(ctx);  // nodes the aborted launch claimed were never scored: no stamp of it may stay valid
        int rc = launch_and_join(ctx, ctx->w_nslots, total, ctx->w_args.for_d, ctx->w_args.seed, ctx->w_args.stream, ctx->w_stride);
        ctx->walk_force_sized = false;
        if (rc != GG_OK) return rc;
        GG_HIP(ctx, hipMemcpyAsync(c, ctx->dev_ctr, sizeof(unsigned long long) * 2 * CW, hipMemcpyDeviceToHost, ctx->stream));
        GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (retried) *retried = true;
    }
    // a launch that ran as two halves counted per half (per-level and spread words, index 8 and up): fold the second block in
    const int n_half = ctx->w_split ? 2 : 1;
    if (n_half == 2)
        for (int i = 8; i < CW; ++i) c[i] += c[CW + i];
    {   // walks alive per streamed level, and behind the last one: where the next launch of this mode hands over to the finisher
        int64_t *prof = ctx->alive_prof[ctx->w_args.for_d ? 1 : 0];
        const int run = ctx->w_levels_run;
        for (int i = 0; i < 64; ++i) prof[i] = i < run ? (int64_t)c[8 + i] : -1;
        if (run > 0 && run < 64 && ctx->w_fin_follows) prof[run] = (int64_t)c[7];
        // the safety net caught more than a handful of walks: the next launches stream one level more
        if (ctx->w_net && c[7] > 2048ull && ctx->lv_levels_learned < ctx->tree_max_depth + 2) ctx->lv_levels_learned += 1;
    }
    // c[0], c[1], c[5]: counts of the per-walk finisher; the level pipeline's counts sit in 64 spread words each
    unsigned long long hops = c[0], reads = c[1], rows = c[5];
    unsigned long long gathers = 0, nodes = 0;
    for (int i = 0; i < 64; ++i) { hops += c[264 + i]; reads += c[328 + i]; rows += c[200 + i]; gathers += c[520 + i]; nodes += c[584 + i]; }
    ctx->ctr.es_gathers += (int64_t)gathers;
    ctx->ctr.es_nodes += (int64_t)nodes;
    ctx->ctr.hops += (int64_t)hops;
    ctx->ctr.nbr_reads += (int64_t)reads;
    ctx->ctr.rows_scored += (int64_t)rows;
    ctx->ctr.walks += total;
    if (total && ctx->walk_timed) {
        float ms = 0.f;
        GG_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        ctx->ctr.last_kernel_ms = ms;
        ctx->ctr.walk_kernel_ms += ms;
        ctx->ctr.walk_launches += 1;
        for (int i = 0; i < ctx->lv_ev_used; ++i) {
            float lms = 0.f;
            for (int k = 0; k < n_half; ++k) {  // the halves' score kernels are chained: their durations add up
                float e = 0.f;
                GG_HIP(ctx, hipEventElapsedTime(&e, ctx->lv_ev[4 * i + 2 * k], ctx->lv_ev[4 * i + 2 * k + 1]));
                lms += e;
            }
            ctx->ctr.score_kernel_ms += lms;
            ctx->ctr.score_launches += n_half;
            ctx->ctr.score_chunks += (int64_t)(c[136 + i] >> 32);
            if (getenv("GG_WALK_DEBUG"))
                fprintf(stderr, "[walk] for_d=%d level %d alive %llu score chunks %llu prefix chunks %llu big %llu small %llu rows %llu score %.1f us\n", ctx->w_args.for_d, i,
                        c[8 + i], c[136 + i] >> 32, c[136 + i] & 0xffffffffull, c[72 + i] & 0xffffffffull, c[72 + i] >> 32, c[456 + i], lms * 1e3);
        }
        if (ctx->lv_ev_used) {
            ctx->ctr.score_rows += (int64_t)(rows - c[5]);  // rows of the timed score launches
            unsigned long long dists = 0;
            for (int i = 0; i < 64; ++i) dists += c[392 + i];
            ctx->ctr.score_dists += (int64_t)dists;
            ctx->ctr.score_gathers += (int64_t)gathers;
            ctx->ctr.score_nodes += (int64_t)nodes;
        }
    }
    if (c[3] || c[6]) generator_changed(ctx);  // a failed launch leaves nothing behind that a later one may reuse
    if (c[3]) return fail(ctx, GG_ECAPACITY, "walk: a path needed more than stride=%d entries", ctx->w_stride);
    if (c[6]) return fail(ctx, GG_EINVAL, "walk: non-finite generator scores (a softmax had total weight 0, or the generator's tables hold a non-finite value)");
    return GG_OK;
}

int gg::walk_resident(gg_ctx *ctx, const int32_t *slots, const int32_t *n_walks, int32_t uniform_walks, int32_t n_slots,
                      int32_t for_d, uint64_t seed, uint32_t stream, int32_t stride) {
    int rc = walk_launch_async(ctx, slots, n_walks, uniform_walks, n_slots, for_d, seed, stream, stride);
    if (rc != GG_OK) return rc;
    return walk_finalize(ctx, nullptr);
}

extern "C" {

int gg_abi_version(void) { return GG_ABI_VERSION; }

const char *gg_last_error(const gg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int gg_create(int32_t n_node, int32_t n_emb, const float *emb_gen, const float *emb_dis, const gg_config *cfg,
              gg_ctx **out) {
    if (!out) return fail(nullptr, GG_EINVAL, "gg_create: out is NULL");
    *out = nullptr;
    if (n_node <= 0 || n_emb <= 0 || n_emb > 512 || !emb_gen || !emb_dis || !cfg)
        return fail(nullptr, GG_EINVAL, "gg_create: bad argument (n_node=%d n_emb=%d)", n_node, n_emb);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, GG_EHIP, "gg_create: no HIP device (%s); this engine has no CPU fallback",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, GG_EINVAL, "gg_create: device %d of %d", cfg->device, ndev);
    hipDeviceProp_t prop;
    GG_HIP(nullptr, hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, GG_EHIP, "gg_create: device %d is %s; kernels are built for gfx950 only", cfg->device, prop.gcnArchName);
    GG_HIP(nullptr, hipSetDevice(cfg->device));

    gg_ctx *ctx = new gg_ctx();
    ctx->n_node = n_node;
    ctx->n_emb = n_emb;
    ctx->ld = (n_emb + 3) / 4 * 4;
    ctx->cfg = *cfg;
    ctx->device = cfg->device;
    ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char *lv = getenv("GG_WALK_LEVELS")) ctx->walk_levels = atoi(lv);
    if (const char *pe = getenv("GG_PROFILE_EVERY")) ctx->profile_every = std::max(0, atoi(pe));
    if (const char *fw = getenv("GG_COMM_FAKE_WORLD")) ctx->fake_world = atoi(fw);
    if (const char *ow = getenv("GG_COMM_OWNER")) ctx->owner_exchange = atoi(ow);
    if (const char *bf = getenv("GG_COMM_BF16")) ctx->comm_bf16 = atoi(bf) != 0;
    if (const char *om = getenv("GG_COMM_OWNER_MIN")) ctx->owner_min_bound = atoll(om);
    if (const char *dt = getenv("GG_DETERMINISTIC")) ctx->deterministic = atoi(dt) != 0;
    if (const char *nc = getenv("GG_NO_DIST_CACHE")) ctx->dc_enabled = atoi(nc) == 0;
    if (const char *ft = getenv("GG_FIN_THRESHOLD")) ctx->fin_threshold = std::max(0, atoi(ft));
    if (const char *st = getenv("GG_STAGE_T")) ctx->sg_threshold = std::max(0, atoi(st));
    if (const char *ws = getenv("GG_WALK_SPLIT")) ctx->split_enabled = atoi(ws) != 0;
    if (const char *em = getenv("GG_ES_MODE")) ctx->es_mode = std::min(2, std::max(0, atoi(em)));
    if (const char *er = getenv("GG_ES_RATIO")) ctx->es_ratio_num = std::max(1, atoi(er));
    if (const char *eh = getenv("GG_ES_HUB")) ctx->es_hub = std::max(0, atoi(eh));
    if (const char *wm = getenv("GG_WALK_SPLIT_MIN")) ctx->split_min_walks = std::max(512, atoi(wm));
    for (auto &m : ctx->alive_prof) for (auto &v : m) v = -1;
    if (const char *dr = getenv("GG_COMM_DENSE_RATIO")) ctx->dense_exchange_ratio = (float)atof(dr);
#define GG_TRY(call)                        \
    do {                                    \
        int rc__ = (call);                  \
        if (rc__ != GG_OK) {                \
            g_last_error = ctx->err;        \
            gg_destroy(ctx);                \
            return rc__;                    \
        }                                   \
    } while (0)
    auto body = [&]() -> int {
        GG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        {
            // GG_WALK_PRIORITY=1: the side stream (G-mode walks beside the discriminator pass) at the highest stream priority --
            // those walks are the longer of the two branches the step joins behind
            int lo = 0, hi = 0;
            const char *wp = getenv("GG_WALK_PRIORITY");
            if (wp && atoi(wp) != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
                GG_HIP(ctx, hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, atoi(wp) > 0 ? hi : lo));
            else
                GG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        }
        GG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
        GG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        GG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        GG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_score[0], hipEventDisableTiming));
        GG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_score[1], hipEventDisableTiming));
        ctx->walk_stream = ctx->stream;
        GG_HIP(ctx, hipEventCreate(&ctx->ev0));
        GG_HIP(ctx, hipEventCreate(&ctx->ev1));
        GG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_walk_done, hipEventDisableTiming));
        GG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_gen_pass, hipEventDisableTiming));
        GG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_main_mark, hipEventDisableTiming));
        GG_HIP(ctx, hipHostMalloc((void **)&ctx->h_pin, sizeof(unsigned long long) * gg_ctx::PIN_WORDS, hipHostMallocDefault));
        memset(ctx->h_pin, 0, sizeof(unsigned long long) * gg_ctx::PIN_WORDS);
        const size_t tb = sizeof(float) * (size_t)n_node * ctx->ld, vb = sizeof(float) * (size_t)n_node;
        for (int m = 0; m < 2; ++m) {
            Model &M = ctx->model[m];
            GG_HIP(ctx, hipMalloc((void **)&M.E, tb));
            GG_HIP(ctx, hipMalloc((void **)&M.b, vb));
            GG_HIP(ctx, hipMemset(M.b, 0, vb));
            if (cfg->optimizer != GG_OPT_SGD) {
                GG_HIP(ctx, hipMalloc((void **)&M.mE, tb));
                GG_HIP(ctx, hipMalloc((void **)&M.vE, tb));
                GG_HIP(ctx, hipMalloc((void **)&M.mb, vb));
                GG_HIP(ctx, hipMalloc((void **)&M.vb, vb));
                GG_HIP(ctx, hipMemset(M.mE, 0, tb));
                GG_HIP(ctx, hipMemset(M.vE, 0, tb));
                GG_HIP(ctx, hipMemset(M.mb, 0, vb));
                GG_HIP(ctx, hipMemset(M.vb, 0, vb));
            }
            M.b1p = cfg->adam_beta1;
            M.b2p = cfg->adam_beta2;
            M.lr = m == 0 ? cfg->lr_gen : cfg->lr_dis;
            M.lambda = m == 0 ? cfg->lambda_gen : cfg->lambda_dis;
            int rc = upload_table(ctx, M.E, m == 0 ? emb_gen : emb_dis);
            if (rc != GG_OK) return rc;
        }
        GG_HIP(ctx, hipMalloc((void **)&ctx->gradE, tb));
        GG_HIP(ctx, hipMalloc((void **)&ctx->gradb, vb));
        GG_HIP(ctx, hipMemset(ctx->gradE, 0, tb));
        ctx->grad_elems_padded = (size_t)n_node * ctx->ld;
        GG_HIP(ctx, hipMemset(ctx->gradb, 0, vb));
        GG_HIP(ctx, hipMalloc((void **)&ctx->touched, sizeof(int32_t) * n_node));
        GG_HIP(ctx, hipMalloc((void **)&ctx->touched_list, sizeof(int32_t) * n_node));
        GG_HIP(ctx, hipMalloc((void **)&ctx->touched_cnt, sizeof(int32_t) * 4));
        GG_HIP(ctx, hipMemset(ctx->touched, 0, sizeof(int32_t) * n_node));
        GG_HIP(ctx, hipMemset(ctx->touched_cnt, 0, sizeof(int32_t) * 4));
        GG_HIP(ctx, hipMalloc((void **)&ctx->dev_ctr, sizeof(unsigned long long) * gg_ctx::PIN_WORDS));
        GG_HIP(ctx, hipMemset(ctx->dev_ctr, 0, sizeof(unsigned long long) * gg_ctx::PIN_WORDS));
        GG_HIP(ctx, ctx->table_bad.reserve(sizeof(unsigned long long) * 2));
        for (int m = 0; m < 2; ++m) {
            int rc = rescan_table_finite(ctx, m);
            if (rc != GG_OK) return rc;
        }
        GG_HIP(ctx, hipDeviceSynchronize());
        return GG_OK;
    };
    GG_TRY(body());
#undef GG_TRY
    *out = ctx;
    return GG_OK;
}

int gg_destroy(gg_ctx *ctx) {
    if (!ctx) return GG_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);  // (a launch begun by gg_prepare_g_begin may still run)
    comm_destroy(ctx);
    for (int m = 0; m < 2; ++m) {
        Model &M = ctx->model[m];
        float *ps[] = {M.E, M.b, M.mE, M.vE, M.mb, M.vb};
        for (float *p : ps)
            if (p) (void)hipFree(p);
    }
    void *ps[] = {ctx->gradE, ctx->gradb, ctx->touched, ctx->touched_list, ctx->touched_cnt, ctx->g_rowptr, ctx->g_col, ctx->dev_ctr,
                  ctx->es, ctx->es_stamp, ctx->g_rev};
    for (void *p : ps)
        if (p) (void)hipFree(p);
    free_trees(ctx);
    DevBuf *bufs[] = {&ctx->w_slots_m[0], &ctx->w_slots_m[1], &ctx->w_ptr_m[0], &ctx->w_ptr_m[1], &ctx->w_samples, &ctx->w_paths, &ctx->w_len, &ctx->w_status,
                      &ctx->w_first, &ctx->w_abort, &ctx->w_scratch, &ctx->d_center, &ctx->d_neighbor, &ctx->d_label, &ctx->d_cnt,
                      &ctx->d_ptr, &ctx->g_node1, &ctx->g_node2, &ctx->g_reward, &ctx->g_cnt, &ctx->g_ptr, &ctx->scan_tmp,
                      &ctx->step_u, &ctx->step_v, &ctx->step_x, &ctx->sg_cnt, &ctx->sg_off, &ctx->sg_slot, &ctx->sg_list, &ctx->sg_rows, &ctx->sg_bias, &ctx->sg_tot, &ctx->sg_key, &ctx->touched_ptr, &ctx->x_cnt, &ctx->x_send_ids, &ctx->x_send_rows,
                      &ctx->x_recv_ids, &ctx->x_recv_rows, &ctx->x_nglob, &ctx->x_own, &ctx->st_item, &ctx->st_item2, &ctx->st_cur, &ctx->st_prev, &ctx->st_len,
                      &ctx->st_alive, &ctx->st_rank, &ctx->bfs_key, &ctx->bfs_bm, &ctx->bfs_misc, &ctx->bfs_sparse, &ctx->bfs_rowptr32, &ctx->lv_pfx, &ctx->dc_keys, &ctx->dc_vals, &ctx->dc_words, &ctx->lv_beg, &ctx->lv_k, &ctx->lv_chunks, &ctx->lv_coff, &ctx->lv_scores, &ctx->lv_chunk_owner, &ctx->lv_prefix, &ctx->lv_big, &ctx->lv_fe, &ctx->fin_list, &ctx->table_bad, &ctx->sgp_cnt, &ctx->sgp_off, &ctx->sgp_slot, &ctx->sgp_list, &ctx->sgp_tot, &ctx->sgp_key, &ctx->sgp_scan,
                      &ctx->q3_store, &ctx->q3s_off, &ctx->ep_center, &ctx->ep_neighbor, &ctx->ep_label, &ctx->ep_node1, &ctx->ep_node2, &ctx->ep_reward};
    for (DevBuf *b : bufs) b->release();
    for (hipEvent_t e : ctx->lv_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &tr : ctx->tm_ev)
        for (hipEvent_t e : tr)
            if (e) (void)hipEventDestroy(e);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    for (hipEvent_t e : {ctx->ev_walk_done, ctx->ev_slots_done, ctx->ev_gen_pass, ctx->ev_main_mark, ctx->ev_fork, ctx->ev_join, ctx->ev_score[0], ctx->ev_score[1]})
        if (e) (void)hipEventDestroy(e);
    if (ctx->stream3) {
        (void)hipStreamSynchronize(ctx->stream3);
        (void)hipStreamDestroy(ctx->stream3);
    }
    if (ctx->stream2) {
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamDestroy(ctx->stream2);
    }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return GG_OK;
}

int gg_set_graph_csr(gg_ctx *ctx, const int64_t *rowptr, const int32_t *col) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, rowptr && rowptr[0] == 0, GG_EINVAL, "gg_set_graph_csr: rowptr[0] must be 0");
    discard_begun_walk(ctx);
    const int n = ctx->n_node;
    const int64_t nnz = rowptr[n];
    GG_CHECK(ctx, nnz >= 0 && (col || nnz == 0), GG_EINVAL, "gg_set_graph_csr: col is NULL");
    for (int v = 0; v < n; ++v) GG_CHECK(ctx, rowptr[v + 1] >= rowptr[v], GG_EINVAL, "gg_set_graph_csr: rowptr not monotone at %d", v);
    for (int64_t e = 0; e < nnz; ++e) GG_CHECK(ctx, col[e] >= 0 && col[e] < n, GG_EINVAL, "gg_set_graph_csr: col[%lld]=%d out of range", (long long)e, col[e]);
    GG_HIP(ctx, hipSetDevice(ctx->device));
    (void)hipDeviceSynchronize();
    for (void *p : {(void *)ctx->g_rowptr, (void *)ctx->g_col, (void *)ctx->es, (void *)ctx->es_stamp, (void *)ctx->g_rev})
        if (p) (void)hipFree(p);
    ctx->g_rowptr = nullptr;
    ctx->g_col = nullptr;
    ctx->es = nullptr;
    ctx->es_stamp = nullptr;
    ctx->g_rev = nullptr;
    ctx->t_edge_valid = false;  // edge indices of resident trees named the old graph
    ctx->q3_store_ready = false;  // (the persistent Q3 bits are laid out by the degrees of the graph: epoch.hip)
    ctx->bfs_rowptr32_valid = false;
    GG_HIP(ctx, hipMalloc((void **)&ctx->g_rowptr, sizeof(int64_t) * (n + 1)));
    GG_HIP(ctx, hipMalloc((void **)&ctx->g_col, sizeof(int32_t) * (std::max<int64_t>(nnz, 1) + 4)));  // (+ 16 B: the BFS reads adjacency in 16-byte quads that may start at the last entry)
    GG_HIP(ctx, hipMemcpy(ctx->g_rowptr, rowptr, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));
    if (nnz) GG_HIP(ctx, hipMemcpy(ctx->g_col, col, sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
    ctx->g_nnz = nnz;
    ctx->g_max_deg = 0;
    for (int v = 0; v < n; ++v) ctx->g_max_deg = (int32_t)std::min<int64_t>(std::max<int64_t>(ctx->g_max_deg, rowptr[v + 1] - rowptr[v]), 0x7fffffff);
    ctx->g_multi = true;
    ctx->h_rowptr.assign(rowptr, rowptr + n + 1);
    ctx->h_col.assign(col, col + nnz);
    ctx->h_comp_size.clear();
    // fingerprint of the adjacency (tree cache files are only valid for the graph they were built from): FNV-1a over 8-byte words
    uint64_t h = 1469598103934665603ull;
    for (int v = 0; v <= n; ++v) h = (h ^ (uint64_t)rowptr[v]) * 1099511628211ull;
    for (int64_t e = 0; e + 1 < nnz; e += 2) h = (h ^ ((uint64_t)(uint32_t)col[e] | ((uint64_t)(uint32_t)col[e + 1] << 32))) * 1099511628211ull;
    if (nnz & 1) h = (h ^ (uint64_t)(uint32_t)col[nnz - 1]) * 1099511628211ull;
    ctx->g_hash = h;
    // edge-score cache of the walk sampler (gg_internal.h): scores, per-node stamps, reverse-edge index
    if (nnz > 0 && nnz < (1ll << 31)) {
        // The cache is optional (launch_walk_sample scores privately when the pointers are NULL): if its 8 * nnz + 8 * n bytes
        // do not fit beside the graph, the graph stays usable without it instead of the call failing.
        hipError_t e = hipMalloc((void **)&ctx->es, sizeof(float) * (size_t)nnz);
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->es_stamp, sizeof(long long) * (size_t)n);
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->g_rev, sizeof(int32_t) * (size_t)nnz);
        if (e == hipSuccess) e = hipMemset(ctx->es_stamp, 0, sizeof(long long) * (size_t)n);
        generator_changed(ctx);
        int rc = e == hipSuccess ? compute_reverse_edges(ctx) : GG_ENOMEM;
        if (rc != GG_OK) {
            (void)hipGetLastError();
            for (void *p : {(void *)ctx->es, (void *)ctx->es_stamp, (void *)ctx->g_rev})
                if (p) (void)hipFree(p);
            ctx->es = nullptr;
            ctx->es_stamp = nullptr;
            ctx->g_rev = nullptr;
            if (getenv("GG_WALK_DEBUG")) fprintf(stderr, "[graph] edge-score cache not allocated (%s): walks score privately\n", e == hipSuccess ? ctx->err.c_str() : hipGetErrorString(e));
        }
    }
    return GG_OK;
}

int gg_build_trees(gg_ctx *ctx, const int32_t *roots, int32_t n_roots, int32_t n_threads) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, !ctx->h_rowptr.empty(), GG_EINVAL, "gg_build_trees: call gg_set_graph_csr first");
    GG_CHECK(ctx, n_roots >= 0 && (roots || n_roots == 0), GG_EINVAL, "gg_build_trees: bad roots");
    const int n = ctx->n_node;
    for (int r = 0; r < n_roots; ++r) GG_CHECK(ctx, roots[r] >= 0 && roots[r] < n, GG_EINVAL, "gg_build_trees: root %d out of range", roots[r]);
    GG_HIP(ctx, hipSetDevice(ctx->device));
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    int rc = alloc_trees(ctx, roots, n_roots, nullptr, nullptr);
    if (rc != GG_OK) return rc;
    // batches of roots bounded by ~2 GiB of host staging; roots of a batch are striped over threads
    const int64_t budget = 256ll << 20;  // (order + cstart) entries
    const std::vector<int64_t> &base = ctx->h_tbase;
    int32_t md = 0, ml = 0;
    std::vector<int32_t> order_h, cstart_h;
    for (int r0 = 0; r0 < n_roots;) {
        int r1 = r0 + 1;
        while (r1 < n_roots && base[r1 + 1] - base[r0] <= budget) ++r1;
        order_h.resize((size_t)(base[r1] - base[r0]));
        cstart_h.resize((size_t)(base[r1] - base[r0]) + (r1 - r0));
        const int nt = std::max(1, std::min(n_threads, r1 - r0));
        std::atomic<int> next(r0);
        std::vector<int32_t> tmd(nt, 0), tml(nt, 0);
        auto work = [&](int tid) {
            std::vector<uint32_t> stamp(n, 0u);
            uint32_t epoch = 0;
            for (;;) {
                const int r = next.fetch_add(1);
                if (r >= r1) break;
                int32_t depth = 0, mc = 0;
                (void)host_bfs_order(ctx->h_rowptr.data(), ctx->h_col.data(), roots[r], order_h.data() + (base[r] - base[r0]),
                                     cstart_h.data() + (base[r] - base[r0]) + (r - r0), stamp, epoch, &depth, &mc);
                tmd[tid] = std::max(tmd[tid], depth);
                tml[tid] = std::max(tml[tid], mc + 1);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &t : th) t.join();
        for (int t = 0; t < nt; ++t) { md = std::max(md, tmd[t]); ml = std::max(ml, tml[t]); }
        if (!order_h.empty()) GG_HIP(ctx, hipMemcpy(ctx->t_order + base[r0], order_h.data(), sizeof(int32_t) * order_h.size(), hipMemcpyHostToDevice));
        GG_HIP(ctx, hipMemcpy(ctx->t_cstart + base[r0] + r0, cstart_h.data(), sizeof(int32_t) * cstart_h.size(), hipMemcpyHostToDevice));
        r0 = r1;
    }
    ctx->tree_max_depth = md;
    ctx->tree_max_list = ml;
    return derive_tree_edges(ctx);
}

// Upload trees given in the reference's shape (per node the list [father, child...]): converted to BFS-order form.
int gg_set_trees(gg_ctx *ctx, const int32_t *roots, int32_t n_roots, const int32_t *off, const int32_t *nbr,
                 const int64_t *nbr_base, int32_t max_depth) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, n_roots >= 0 && roots && off && nbr_base && (nbr || nbr_base[n_roots] == 0), GG_EINVAL, "gg_set_trees: bad argument");
    const int n = ctx->n_node;
    GG_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int64_t> counts(n_roots), kids(n_roots);
    for (int r = 0; r < n_roots; ++r) {
        GG_CHECK(ctx, roots[r] >= 0 && roots[r] < n, GG_EINVAL, "gg_set_trees: root out of range");
        const int32_t *o = off + (size_t)r * (n + 1);
        GG_CHECK(ctx, o[0] == 0 && o[n] == nbr_base[r + 1] - nbr_base[r] && o[n] >= 1 && (o[n] & 1), GG_EINVAL,
                 "gg_set_trees: offsets of slot %d inconsistent (a tree over C nodes has 2C - 1 entries)", r);
        for (int v = 0; v < n; ++v) GG_CHECK(ctx, o[v + 1] >= o[v], GG_EINVAL, "gg_set_trees: offsets not monotone");
        counts[r] = (o[n] + 1) / 2;
        kids[r] = o[roots[r] + 1] - o[roots[r]];
    }
    int rc = alloc_trees(ctx, roots, n_roots, counts.data(), kids.data());
    if (rc != GG_OK) return rc;
    const std::vector<int64_t> &base = ctx->h_tbase;
    std::vector<int32_t> order_h((size_t)base[n_roots]), cstart_h((size_t)base[n_roots] + n_roots);
    std::vector<uint32_t> q3_h((size_t)std::max<int64_t>(ctx->h_q3off[n_roots], 1), 0u);
    int32_t md = 0, ml = 0;
    for (int r = 0; r < n_roots; ++r) {
        const int64_t C_expect = base[r + 1] - base[r];
        GG_CHECK(ctx, nbr_base[r + 1] - nbr_base[r] == 2 * C_expect - 1, GG_EINVAL,
                 "gg_set_trees: slot %d has %lld entries, the component of root %d needs %lld", r, (long long)(nbr_base[r + 1] - nbr_base[r]),
                 roots[r], (long long)(2 * C_expect - 1));
        int32_t depth = 0, mc = 0;
        const int32_t C = lists_to_order(n, roots[r], off + (size_t)r * (n + 1), nbr + nbr_base[r], order_h.data() + base[r],
                                         cstart_h.data() + base[r] + r, q3_h.data() + ctx->h_q3off[r],
                                         (int32_t)(ctx->h_q3off[r + 1] - ctx->h_q3off[r]), &depth, &mc);
        GG_CHECK(ctx, C == C_expect, GG_EINVAL, "gg_set_trees: the lists of slot %d are not a BFS tree of root %d", r, roots[r]);
        md = std::max(md, depth);
        ml = std::max(ml, mc + 1);
    }
    if (!order_h.empty()) GG_HIP(ctx, hipMemcpy(ctx->t_order, order_h.data(), sizeof(int32_t) * order_h.size(), hipMemcpyHostToDevice));
    if (!cstart_h.empty()) GG_HIP(ctx, hipMemcpy(ctx->t_cstart, cstart_h.data(), sizeof(int32_t) * cstart_h.size(), hipMemcpyHostToDevice));
    if (ctx->h_q3off[n_roots]) GG_HIP(ctx, hipMemcpy(ctx->t_q3, q3_h.data(), sizeof(uint32_t) * (size_t)ctx->h_q3off[n_roots], hipMemcpyHostToDevice));
    ctx->tree_max_depth = max_depth > 0 ? std::max(max_depth, md) : md;
    ctx->tree_max_list = ml;
    return derive_tree_edges(ctx);  // (lists that are no subgraph of the resident graph: walks score privately, no error)
}

int gg_tree_info(const gg_ctx *ctx, int32_t *n_roots, int64_t *n_entries, int32_t *max_depth) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    if (n_roots) *n_roots = ctx->n_tree_roots;
    if (n_entries) *n_entries = ctx->tree_entries;
    if (max_depth) *max_depth = ctx->tree_max_depth;
    return GG_OK;
}

int gg_tree_roots(const gg_ctx *ctx, int32_t *roots) {
    if (!ctx || !roots) return fail(nullptr, GG_EINVAL, "gg_tree_roots: NULL argument");
    for (int r = 0; r < ctx->n_tree_roots; ++r) roots[r] = ctx->h_troot[r];
    return GG_OK;
}

// Download the resident trees in the reference's shape, including the D-mode mutations (removed father entries = -1).
int gg_get_trees(gg_ctx *ctx, int32_t *off, int32_t *nbr, int64_t *nbr_base) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, !ctx->t_lazy, GG_EINVAL, "gg_get_trees: the resident trees are lazy (exact through a level only): build them whole (gg_set_tree_mode(ctx, 0, 0)) to export them");
    GG_CHECK(ctx, ctx->n_tree_roots > 0, GG_EINVAL, "gg_get_trees: no trees loaded");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    GG_HIP(ctx, hipDeviceSynchronize());
    const int n = ctx->n_node, R = ctx->n_tree_roots;
    const std::vector<int64_t> &base = ctx->h_tbase;
    if (nbr_base)
        for (int r = 0; r <= R; ++r) nbr_base[r] = 2 * base[r] - r;
    if (!off && !nbr) return GG_OK;
    GG_CHECK(ctx, off && nbr, GG_EINVAL, "gg_get_trees: off and nbr go together");
    std::vector<int32_t> order_h((size_t)base[R]), cstart_h((size_t)base[R] + R);
    std::vector<uint32_t> q3_h((size_t)std::max<int64_t>(ctx->h_q3off[R], 1), 0u);
    if (!order_h.empty()) GG_HIP(ctx, hipMemcpy(order_h.data(), ctx->t_order, sizeof(int32_t) * order_h.size(), hipMemcpyDeviceToHost));
    GG_HIP(ctx, hipMemcpy(cstart_h.data(), ctx->t_cstart, sizeof(int32_t) * cstart_h.size(), hipMemcpyDeviceToHost));
    if (ctx->h_q3off[R]) GG_HIP(ctx, hipMemcpy(q3_h.data(), ctx->t_q3, sizeof(uint32_t) * (size_t)ctx->h_q3off[R], hipMemcpyDeviceToHost));
    const int nt = (int)std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), (unsigned)R));
    std::atomic<int> next(0);
    auto work = [&]() {
        for (;;) {
            const int r = next.fetch_add(1);
            if (r >= R) break;
            order_to_lists(n, (int32_t)(base[r + 1] - base[r]), order_h.data() + base[r], cstart_h.data() + base[r] + r,
                           q3_h.data() + ctx->h_q3off[r], off + (size_t)r * (n + 1), nbr + (2 * base[r] - r));
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    return GG_OK;
}

int gg_get_tree_order(gg_ctx *ctx, int64_t *base, int32_t *order, int32_t *cstart, int32_t *edge, int32_t *edges_valid) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, !ctx->t_lazy, GG_EINVAL, "gg_get_tree_order: the resident trees are lazy (exact through a level only): build them whole (gg_set_tree_mode(ctx, 0, 0)) to export them");
    GG_CHECK(ctx, ctx->n_tree_roots > 0, GG_EINVAL, "gg_get_tree_order: no trees loaded");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    GG_HIP(ctx, hipDeviceSynchronize());
    const int R = ctx->n_tree_roots;
    const size_t nodes = (size_t)ctx->tree_nodes;
    if (base) memcpy(base, ctx->h_tbase.data(), sizeof(int64_t) * ((size_t)R + 1));
    if (order) GG_HIP(ctx, hipMemcpy(order, ctx->t_order, sizeof(int32_t) * nodes, hipMemcpyDeviceToHost));
    if (cstart) GG_HIP(ctx, hipMemcpy(cstart, ctx->t_cstart, sizeof(int32_t) * (nodes + R), hipMemcpyDeviceToHost));
    if (edge) GG_HIP(ctx, hipMemcpy(edge, ctx->t_edge, sizeof(int32_t) * nodes, hipMemcpyDeviceToHost));
    if (edges_valid) *edges_valid = ctx->t_edge_valid ? 1 : 0;
    return GG_OK;
}

// ---- tree cache (replaces the pickle of graph_gan.py:31-46).  Flat little-endian file:
//   {magic "GGTR", version 1, n_node i32, n_roots i32, nnz i64, graph fingerprint u64, max_depth i32, max_list i32, nodes i64}
//   roots i32[n_roots], base i64[n_roots + 1], order i32[nodes], cstart i32[nodes + n_roots]
// i.e. the resident BFS-order arrays as they are (no conversion on either side).  Like the reference's cache it holds
// the trees as built: the in-place D-mode mutations are not part of it (the pickle is written before any, :45).
namespace {
struct TreeHeader {
    char magic[4];
    int32_t version, n_node, n_roots;
    int64_t nnz;
    uint64_t graph_hash;
    int32_t max_depth, max_list;
    int64_t nodes;
};

// Structure check of trees that came from a file: the walk kernels index t_order / t_cstart without bounds tests, so a
// corrupt cache of the right size must not reach them.  One thread per (slot, rank): node ids in range, rank 0 = the
// root, child ranges ascending, behind their parent and inside [1, C], cstart[C] = C.
__global__ __launch_bounds__(256) void validate_trees_kernel(const int32_t *order, const int32_t *cstart, const int64_t *base, const int32_t *root,
                                                             int32_t n_roots, int32_t n_node, int32_t *bad) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= base[n_roots]) return;
    int lo = 0, hi = n_roots;  // slot r with base[r] <= j < base[r + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (base[mid] <= j) lo = mid; else hi = mid;
    }
    const int r = lo;
    const int64_t i = j - base[r], C = base[r + 1] - base[r];
    const int32_t nd = order[j];
    const int32_t *cs = cstart + base[r] + r;
    const int64_t c0 = cs[i], c1 = cs[i + 1];
    bool ok = nd >= 0 && nd < n_node && (i != 0 || (nd == root[r] && c0 == 1)) && c0 > i && c0 <= c1 && c1 <= C && (i + 1 != C || c1 == C);
    if (!ok) atomicOr(bad, 1);
}

int stream_dev(gg_ctx *ctx, FILE *f, int32_t *dev, size_t count, bool save) {
    const size_t step = 64u << 20;  // entries per staging round (256 MB)
    std::vector<int32_t> tmp(std::min(count, step));
    for (size_t o = 0; o < count; o += step) {
        const size_t n = std::min(step, count - o);
        if (save) {
            GG_HIP(ctx, hipMemcpy(tmp.data(), dev + o, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
            if (fwrite(tmp.data(), sizeof(int32_t), n, f) != n) return gg::fail(ctx, GG_EIO, "tree cache: short write");
        } else {
            if (fread(tmp.data(), sizeof(int32_t), n, f) != n) return gg::fail(ctx, GG_EIO, "tree cache: short read");
            GG_HIP(ctx, hipMemcpy(dev + o, tmp.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
        }
    }
    return GG_OK;
}
}  // namespace

int gg_save_trees(gg_ctx *ctx, const char *path) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, !ctx->t_lazy, GG_EINVAL, "gg_save_trees: the resident trees are lazy (exact through a level only): build them whole (gg_set_tree_mode(ctx, 0, 0)) to export them");
    GG_CHECK(ctx, path, GG_EINVAL, "gg_save_trees: path is NULL");
    GG_CHECK(ctx, ctx->n_tree_roots > 0, GG_EINVAL, "gg_save_trees: no trees loaded");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    GG_HIP(ctx, hipDeviceSynchronize());
    const int R = ctx->n_tree_roots;
    const std::string tmp_path = std::string(path) + ".tmp";
    FILE *f = fopen(tmp_path.c_str(), "wb");
    if (!f) return fail(ctx, GG_EIO, "gg_save_trees: cannot open %s", tmp_path.c_str());
    TreeHeader h{{'G', 'G', 'T', 'R'}, 1, ctx->n_node, R, ctx->g_nnz, ctx->g_hash, ctx->tree_max_depth, ctx->tree_max_list, ctx->tree_nodes};
    int rc = GG_OK;
    if (fwrite(&h, sizeof(h), 1, f) != 1 || fwrite(ctx->h_troot.data(), sizeof(int32_t), R, f) != (size_t)R ||
        fwrite(ctx->h_tbase.data(), sizeof(int64_t), R + 1, f) != (size_t)R + 1)
        rc = fail(ctx, GG_EIO, "gg_save_trees: short write");
    if (rc == GG_OK) rc = stream_dev(ctx, f, ctx->t_order, (size_t)ctx->tree_nodes, true);
    if (rc == GG_OK) rc = stream_dev(ctx, f, ctx->t_cstart, (size_t)ctx->tree_nodes + R, true);
    if (rc == GG_OK && (fflush(f) != 0 || fsync(fileno(f)) != 0)) rc = fail(ctx, GG_EIO, "gg_save_trees: cannot flush %s", tmp_path.c_str());
    fclose(f);
    if (rc == GG_OK && rename(tmp_path.c_str(), path) != 0) rc = fail(ctx, GG_EIO, "gg_save_trees: cannot rename %s to %s", tmp_path.c_str(), path);
    if (rc != GG_OK) (void)remove(tmp_path.c_str());
    return rc;
}

int gg_load_trees(gg_ctx *ctx, const char *path) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, path, GG_EINVAL, "gg_load_trees: path is NULL");
    GG_CHECK(ctx, !ctx->h_rowptr.empty(), GG_EINVAL, "gg_load_trees: call gg_set_graph_csr first");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    FILE *f = fopen(path, "rb");
    if (!f) return fail(ctx, GG_EIO, "gg_load_trees: cannot open %s", path);
    TreeHeader h;
    int rc = GG_OK;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "GGTR", 4) != 0 || h.version != 1)
        rc = fail(ctx, GG_EIO, "gg_load_trees: %s is not a GGTR v1 file", path);
    else if (h.n_node != ctx->n_node || h.nnz != ctx->g_nnz || h.graph_hash != ctx->g_hash)
        rc = fail(ctx, GG_EINVAL, "gg_load_trees: %s was built from another graph (%d nodes, %lld adjacency entries)", path, h.n_node, (long long)h.nnz);
    else if (h.n_roots <= 0 || h.nodes < h.n_roots) rc = fail(ctx, GG_EIO, "gg_load_trees: %s: bad header", path);
    std::vector<int32_t> roots;
    std::vector<int64_t> base, counts, kids;
    if (rc == GG_OK) {
        struct stat st;
        const size_t expect = sizeof(h) + sizeof(int32_t) * (size_t)h.n_roots + sizeof(int64_t) * ((size_t)h.n_roots + 1) +
                              sizeof(int32_t) * (2 * (size_t)h.nodes + (size_t)h.n_roots);
        if (fstat(fileno(f), &st) != 0 || (size_t)st.st_size != expect)
            rc = fail(ctx, GG_EIO, "gg_load_trees: %s is truncated (%lld bytes, expected %zu)", path, (long long)st.st_size, expect);
    }
    if (rc == GG_OK) {
        roots.resize(h.n_roots);
        base.resize((size_t)h.n_roots + 1);
        if (fread(roots.data(), sizeof(int32_t), h.n_roots, f) != (size_t)h.n_roots || fread(base.data(), sizeof(int64_t), (size_t)h.n_roots + 1, f) != (size_t)h.n_roots + 1)
            rc = fail(ctx, GG_EIO, "gg_load_trees: short read");
    }
    if (rc == GG_OK) {
        counts.resize(h.n_roots);
        for (int r = 0; r < h.n_roots && rc == GG_OK; ++r) {
            counts[r] = base[r + 1] - base[r];
            if (roots[r] < 0 || roots[r] >= ctx->n_node || counts[r] < 1 || counts[r] > ctx->n_node) rc = fail(ctx, GG_EIO, "gg_load_trees: %s: bad root table", path);
        }
        if (rc == GG_OK && (base[0] != 0 || base[h.n_roots] != h.nodes)) rc = fail(ctx, GG_EIO, "gg_load_trees: %s: bad root table", path);
    }
    if (rc == GG_OK) rc = alloc_trees(ctx, roots.data(), h.n_roots, counts.data(), nullptr);
    if (rc == GG_OK) rc = stream_dev(ctx, f, ctx->t_order, (size_t)h.nodes, false);
    if (rc == GG_OK) rc = stream_dev(ctx, f, ctx->t_cstart, (size_t)h.nodes + h.n_roots, false);
    fclose(f);
    if (rc == GG_OK) {
        int32_t *bad = (int32_t *)(ctx->dev_ctr + 2000);  // a spare word behind the walk launches' counters ([0, 2 * CTR_WORDS))
        int32_t h_bad = 0;
        hipError_t e = hipMemset(bad, 0, sizeof(int32_t));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(validate_trees_kernel, dim3((unsigned)cdiv(h.nodes, 256)), dim3(256), 0, ctx->stream, ctx->t_order, ctx->t_cstart,
                               ctx->t_base, ctx->t_root, h.n_roots, ctx->n_node, bad);
            e = hipStreamSynchronize(ctx->stream);
        }
        if (e == hipSuccess) e = hipMemcpy(&h_bad, bad, sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemset(bad, 0, sizeof(int32_t));
        if (e != hipSuccess) rc = fail(ctx, GG_EHIP, "gg_load_trees: %s", hipGetErrorString(e));
        else if (h_bad || h.max_depth < 0 || h.max_depth > ctx->n_node || h.max_list < 0 || h.max_list > ctx->n_node)
            rc = fail(ctx, GG_EIO, "gg_load_trees: %s: the tree arrays are corrupt (node id out of range or child ranges not a BFS order)", path);
    }
    if (rc != GG_OK) {  // nothing half-loaded stays resident
        ctx->n_tree_roots = 0;
        ctx->tree_nodes = 0;
        return rc;
    }
    ctx->tree_max_depth = h.max_depth;
    ctx->tree_max_list = h.max_list;
    return derive_tree_edges(ctx);
}

int gg_walk_sample(gg_ctx *ctx, const int32_t *slots, const int32_t *n_walks, int32_t n_slots, int32_t for_d,
                   uint64_t seed, uint32_t stream, int32_t *samples, int32_t *paths, int32_t *path_len, int32_t stride,
                   int32_t *root_status) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, n_walks || n_slots == 0, GG_EINVAL, "gg_walk_sample: n_walks is NULL");
    ctx->dc_request = 0;
    ctx->dc_valid = false;  // this launch reuses the prefix buffer from offset 0
    int rc = walk_resident(ctx, slots, n_walks, -1, n_slots, for_d ? 1 : 0, seed, stream, stride);
    if (rc != GG_OK) return rc;
    const int64_t total = ctx->w_total;
    if (samples && total) GG_HIP(ctx, hipMemcpy(samples, ctx->w_samples.p, sizeof(int32_t) * total, hipMemcpyDeviceToHost));
    if (path_len && total) GG_HIP(ctx, hipMemcpy(path_len, ctx->w_len.p, sizeof(int32_t) * total, hipMemcpyDeviceToHost));
    if (paths && total && ctx->w_stride == stride) GG_HIP(ctx, hipMemcpy(paths, ctx->w_paths.p, sizeof(int32_t) * (size_t)total * stride, hipMemcpyDeviceToHost));
    if (ctx->w_stride != stride && total) {
        // LAZY trees: slots that were rebuilt whole may hold deeper trees than the bound the caller sized its rows by
        std::vector<int32_t> len((size_t)total);
        GG_HIP(ctx, hipMemcpy(len.data(), ctx->w_len.p, sizeof(int32_t) * total, hipMemcpyDeviceToHost));
        for (int64_t w = 0; w < total; ++w)
            GG_CHECK(ctx, len[w] <= stride, GG_ECAPACITY, "walk: a path needed %d entries, stride is %d (gg_tree_info reports the depth of the trees as rebuilt)", len[w], stride);
        if (paths) GG_HIP(ctx, hipMemcpy2D(paths, sizeof(int32_t) * (size_t)stride, ctx->w_paths.p, sizeof(int32_t) * (size_t)ctx->w_stride, sizeof(int32_t) * (size_t)stride, (size_t)total, hipMemcpyDeviceToHost));
    }
    if (root_status && n_slots) GG_HIP(ctx, hipMemcpy(root_status, ctx->w_status.p, sizeof(int32_t) * n_slots, hipMemcpyDeviceToHost));
    return GG_OK;
}

int gg_debug_words(gg_ctx *ctx, int32_t first, int32_t n, uint64_t *out) {
    if (!ctx || !out || first < 0 || n < 0 || first + n > gg_ctx::PIN_WORDS) return fail(ctx, GG_EINVAL, "gg_debug_words: bad argument");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s;  // (its own stream: readable while a kernel of the context hangs)
    GG_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipError_t e = hipMemcpyAsync(out, ctx->dev_ctr + first, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    GG_HIP(ctx, e);
    return GG_OK;
}

int gg_set_tree_mode(gg_ctx *ctx, int32_t mode, int64_t node_cap) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, mode >= -1 && mode <= 1 && node_cap >= 0, GG_EINVAL, "gg_set_tree_mode: mode must be -1 (auto), 0 (whole trees) or 1 (lazy), node_cap >= 0");
    ctx->tree_mode = mode;
    ctx->lz_cap = node_cap;
    return GG_OK;
}

int gg_lazy_stats(gg_ctx *ctx, int64_t *out8) {
    if (!ctx || !out8) return fail(ctx, GG_EINVAL, "gg_lazy_stats: NULL argument");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < 24; ++i) out8[i] = 0;
    out8[0] = ctx->t_lazy ? 1 : 0;
    out8[1] = ctx->t_lazy ? ctx->lz_min_level : 0;
    out8[2] = ctx->lz_fallback_roots;
    out8[3] = ctx->lz_fallback_rounds;
    if (ctx->t_lazy && ctx->n_tree_roots > 0) {
        GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const int R = ctx->n_tree_roots;
        std::vector<int4> info((size_t)R);
        std::vector<int32_t> cur((size_t)R);
        GG_HIP(ctx, hipMemcpy(info.data(), ctx->lz_info.p, sizeof(int4) * (size_t)R, hipMemcpyDeviceToHost));
        GG_HIP(ctx, hipMemcpy(cur.data(), ctx->lz_cursor.p, sizeof(int32_t) * (size_t)R, hipMemcpyDeviceToHost));
        for (int r = 0; r < R; ++r) {
            out8[4] += info[r].y;                                  // exact nodes written by the BFS
            out8[5] += std::max(0, cur[r] - info[r].y);            // pool entries reserved by resolutions
            out8[6] += info[r].x < info[r].y ? 1 : 0;              // slots that are lazy (not built whole)
            out8[7] = std::max<int64_t>(out8[7], info[r].z);       // deepest exact level
            if (info[r].x < info[r].y && info[r].z >= 0 && info[r].z < 8) out8[16 + info[r].z] += 1;  // lazy slots by exact level
        }
        unsigned long long c[8];
        GG_HIP(ctx, hipMemcpy(c, ctx->dev_ctr + 1500, sizeof(c), hipMemcpyDeviceToHost));
        for (int i = 0; i < 8; ++i) out8[8 + i] = (int64_t)c[i];   // lists resolved at depth 0 / 1 / 2, candidates, scan rounds, most rounds of one list, longest adjacency resolved
    }
    return GG_OK;
}

int gg_get_lazy_trees(gg_ctx *ctx, int64_t *n_entries, int32_t *info4, int64_t *base, int32_t *order, int32_t *cstart, int32_t *edge, uint64_t *pair) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, ctx->t_lazy, GG_EINVAL, "gg_get_lazy_trees: the resident trees are not lazy");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    GG_HIP(ctx, hipDeviceSynchronize());
    const size_t R = (size_t)ctx->n_tree_roots, nodes = (size_t)ctx->arena_end;  // (segments and arena)
    if (n_entries) *n_entries = (int64_t)nodes;
    if (info4) GG_HIP(ctx, hipMemcpy(info4, ctx->lz_info.p, sizeof(int4) * R, hipMemcpyDeviceToHost));
    if (base) GG_HIP(ctx, hipMemcpy(base, ctx->t_base, sizeof(int64_t) * R, hipMemcpyDeviceToHost));
    if (order) GG_HIP(ctx, hipMemcpy(order, ctx->t_order, sizeof(int32_t) * nodes, hipMemcpyDeviceToHost));
    if (cstart) GG_HIP(ctx, hipMemcpy(cstart, ctx->t_cstart, sizeof(int32_t) * (nodes + R), hipMemcpyDeviceToHost));
    if (edge) GG_HIP(ctx, hipMemcpy(edge, ctx->t_edge, sizeof(int32_t) * nodes, hipMemcpyDeviceToHost));
    if (pair) GG_HIP(ctx, hipMemcpy(pair, ctx->lz_pair.p, sizeof(uint64_t) * std::min(nodes, ctx->lz_pair.bytes / sizeof(uint64_t)), hipMemcpyDeviceToHost));
    return GG_OK;
}

int gg_walk_info(const gg_ctx *ctx, int64_t *total_walks, int32_t *stride, int32_t *n_slots) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    if (total_walks) *total_walks = ctx->w_total;
    if (stride) *stride = ctx->w_stride;
    if (n_slots) *n_slots = ctx->w_nslots;
    return GG_OK;
}

int gg_get_walks(gg_ctx *ctx, int32_t *samples, int32_t *paths, int32_t *path_len, int32_t *root_status) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    // A launch that was only begun (gg_prepare_g_begin) has overwritten the walk buffers and has no results to hand out: the
    // caller asked for walks that no longer exist -- an error, not an empty answer.
    const bool begun = ctx->g_begun;
    discard_begun_walk(ctx);
    GG_CHECK(ctx, !begun, GG_EINVAL, "gg_get_walks: the walks of the last completed launch were overwritten by gg_prepare_g_begin (fetch them before it, or after the gg_prepare_g that adopts it)");
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t total = ctx->w_total;
    if (samples && total) GG_HIP(ctx, hipMemcpy(samples, ctx->w_samples.p, sizeof(int32_t) * total, hipMemcpyDeviceToHost));
    if (path_len && total) GG_HIP(ctx, hipMemcpy(path_len, ctx->w_len.p, sizeof(int32_t) * total, hipMemcpyDeviceToHost));
    if (paths && total) GG_HIP(ctx, hipMemcpy(paths, ctx->w_paths.p, sizeof(int32_t) * (size_t)total * ctx->w_stride, hipMemcpyDeviceToHost));
    if (root_status && ctx->w_nslots) GG_HIP(ctx, hipMemcpy(root_status, ctx->w_status.p, sizeof(int32_t) * ctx->w_nslots, hipMemcpyDeviceToHost));
    return GG_OK;
}

static int table_io(gg_ctx *ctx, int32_t which, float *out, const float *in, bool bias) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, which == 0 || which == 1, GG_EINVAL, "which must be 0 (gen) or 1 (dis)");
    GG_CHECK(ctx, out || in, GG_EINVAL, "buffer is NULL");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    Model &M = ctx->model[which];
    const int n = ctx->n_node, d = ctx->n_emb, ld = ctx->ld;
    if (in && which == 0) { discard_begun_walk(ctx); generator_changed(ctx); }
    if (bias) {
        if (out) GG_HIP(ctx, hipMemcpy(out, M.b, sizeof(float) * n, hipMemcpyDeviceToHost));
        else GG_HIP(ctx, hipMemcpy(M.b, in, sizeof(float) * n, hipMemcpyHostToDevice));
    } else if (out) {
        GG_HIP(ctx, hipMemcpy2D(out, sizeof(float) * d, M.E, sizeof(float) * ld, sizeof(float) * d, n, hipMemcpyDeviceToHost));
    } else {
        int rc = upload_table(ctx, M.E, in);
        if (rc != GG_OK) return rc;
    }
    if (in) return rescan_table_finite(ctx, which);
    return GG_OK;
}

int gg_get_embeddings(gg_ctx *ctx, int32_t which, float *out) { return table_io(ctx, which, out, nullptr, false); }
int gg_get_bias(gg_ctx *ctx, int32_t which, float *out) { return table_io(ctx, which, out, nullptr, true); }
int gg_set_embeddings(gg_ctx *ctx, int32_t which, const float *emb) { return table_io(ctx, which, nullptr, emb, false); }
int gg_set_bias(gg_ctx *ctx, int32_t which, const float *bias) { return table_io(ctx, which, nullptr, bias, true); }

int gg_get_counters(gg_ctx *ctx, gg_counters *out) {
    if (!ctx || !out) return fail(ctx, GG_EINVAL, "gg_get_counters: NULL argument");
    if (!ctx->tm_pending.empty()) {  // per-kernel timings still in flight: wait for them
        GG_HIP(ctx, hipSetDevice(ctx->device));
        GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        harvest_timings(ctx);
    }
    *out = ctx->ctr;
    return GG_OK;
}

int gg_set_profiling_solo(gg_ctx *ctx, int32_t solo) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    ctx->profile_solo = solo != 0;
    return GG_OK;
}

int gg_set_profiling(gg_ctx *ctx, int32_t every_n) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, every_n >= 0, GG_EINVAL, "gg_set_profiling: every_n < 0");
    ctx->profile_every = every_n;
    ctx->walk_call_index = 0;
    return GG_OK;
}

int gg_synchronize(gg_ctx *ctx) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->g_begun) GG_HIP(ctx, hipStreamSynchronize(ctx->stream2));  // (the begun launch stays adoptable)
    harvest_timings(ctx);
    return check_exchange_flag(ctx);
}

}  // extern "C"

// A row pack of the replica exchange did not fit its capacity (steps.hip: the capacity rule was violated, e.g. ranks
// passed inconsistent batch sizes): the step's gradients were truncated -- report it instead of training on.
int gg::check_exchange_flag(gg_ctx *ctx) {
    if (!ctx->comm && ctx->fake_world <= 1) return GG_OK;
    int32_t flag = 0;
    GG_HIP(ctx, hipMemcpy(&flag, ctx->touched_cnt + 2, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (flag) return fail(ctx, GG_ECOMM, "replica exchange: a rank touched more rows than the pack capacity of its step");
    return GG_OK;
}

// Event triples {before the gradient / reward kernel, between gradient and optimizer, after} recorded by profiled
// prepare / pass calls; folded into the counters once the stream has passed them (called after host synchronisations).
void gg::harvest_timings(gg_ctx *ctx) {
    size_t keep = 0;
    for (size_t i = 0; i < ctx->tm_pending.size(); ++i) {
        const gg_ctx::PendingTiming t = ctx->tm_pending[i];
        hipEvent_t *ev = ctx->tm_ev[t.slot];
        if (hipEventQuery(ev[2]) != hipSuccess) {
            ctx->tm_pending[keep++] = t;
            continue;
        }
        float a = 0.f, b = 0.f;
        (void)hipEventElapsedTime(&a, ev[0], ev[1]);
        (void)hipEventElapsedTime(&b, ev[1], ev[2]);
        const int64_t rows = t.has_rows ? (int64_t)ctx->h_pin[gg_ctx::H_ROWS + t.slot] : (int64_t)ctx->n_node;
        if (t.kind == 0) {
            ctx->ctr.reward_kernel_ms += a + b;
            ctx->ctr.reward_pairs_timed += t.units;
        } else if (t.kind == 1) {
            ctx->ctr.d_grad_ms += a; ctx->ctr.d_opt_ms += b;
            ctx->ctr.d_pairs_timed += t.units; ctx->ctr.d_rows_timed += rows; ctx->ctr.d_passes_timed += 1;
        } else {
            ctx->ctr.g_grad_ms += a; ctx->ctr.g_opt_ms += b;
            ctx->ctr.g_pairs_timed += t.units; ctx->ctr.g_rows_timed += rows; ctx->ctr.g_passes_timed += 1;
        }
    }
    ctx->tm_pending.resize(keep);
}

// A free event triple for a profiled call, or -1 (all in flight).
int gg::timing_slot(gg_ctx *ctx) {
    for (int s = 0; s < 8; ++s) {
        bool busy = false;
        for (const auto &t : ctx->tm_pending) busy |= t.slot == s;
        if (busy) continue;
        for (int k = 0; k < 3; ++k)
            if (!ctx->tm_ev[s][k] && hipEventCreate(&ctx->tm_ev[s][k]) != hipSuccess) return -1;
        return s;
    }
    return -1;
}

// ---- tf.train.Saver replacement (graph_gan.py:55,124-127,137-138).  Flat binary:
// header {magic "GGST", version, n_node, n_emb, ld, optimizer} then for gen, dis:
// {t, beta1_power, beta2_power, E[n*ld], b[n], (mE, vE, mb, vb unless SGD)} -- fp32 little endian.
namespace {
struct StateHeader {
    char magic[4];
    int32_t version, n_node, n_emb, ld, optimizer;
};

int io_dev(gg_ctx *ctx, FILE *f, float *dev, size_t count, bool save, std::vector<float> &tmp) {
    tmp.resize(count);
    if (save) {
        GG_HIP(ctx, hipMemcpy(tmp.data(), dev, sizeof(float) * count, hipMemcpyDeviceToHost));
        if (fwrite(tmp.data(), sizeof(float), count, f) != count) return gg::fail(ctx, GG_EIO, "state: short write");
    } else {
        if (fread(tmp.data(), sizeof(float), count, f) != count) return gg::fail(ctx, GG_EIO, "state: short read");
        GG_HIP(ctx, hipMemcpy(dev, tmp.data(), sizeof(float) * count, hipMemcpyHostToDevice));
    }
    return GG_OK;
}

int state_io(gg_ctx *ctx, const char *path, bool save) {
    if (!ctx) return gg::fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, path, GG_EINVAL, "state: path is NULL");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!save) discard_begun_walk(ctx);  // (the generator's tables are about to be replaced)
    const bool slots = ctx->cfg.optimizer != GG_OPT_SGD;
    const size_t ne = (size_t)ctx->n_node * ctx->ld, nb = (size_t)ctx->n_node;
    struct Scalars { int64_t t; float b1p, b2p; };
    const size_t per_model = sizeof(Scalars) + sizeof(float) * (ne + nb) * (slots ? 3 : 1);
    const size_t expect = sizeof(StateHeader) + 2 * per_model;
    // save: write a temporary next to the target, flush it to disk, then rename over the target -- a crash in the
    // middle leaves the previous checkpoint intact.  load: the whole file is validated (magic, shape, SIZE) before the
    // first byte of device memory changes, so a truncated file cannot leave the model half overwritten.
    const std::string tmp_path = std::string(path) + ".tmp";
    FILE *f = fopen(save ? tmp_path.c_str() : path, save ? "wb" : "rb");
    if (!f) return gg::fail(ctx, GG_EIO, "state: cannot open %s", save ? tmp_path.c_str() : path);
    StateHeader h{{'G', 'G', 'S', 'T'}, 1, ctx->n_node, ctx->n_emb, ctx->ld, ctx->cfg.optimizer};
    int rc = GG_OK;
    if (save) {
        if (fwrite(&h, sizeof(h), 1, f) != 1) rc = gg::fail(ctx, GG_EIO, "state: short write");
    } else {
        StateHeader g;
        if (fread(&g, sizeof(g), 1, f) != 1 || memcmp(g.magic, "GGST", 4) != 0 || g.version != 1)
            rc = gg::fail(ctx, GG_EIO, "state: %s is not a GGST v1 file", path);
        else if (g.n_node != h.n_node || g.n_emb != h.n_emb || g.ld != h.ld || (g.optimizer == GG_OPT_SGD) != (h.optimizer == GG_OPT_SGD))
            rc = gg::fail(ctx, GG_EINVAL, "state: shape/optimizer mismatch (file %dx%d opt %d)", g.n_node, g.n_emb, g.optimizer);
        if (rc == GG_OK) {
            struct stat st;
            if (fstat(fileno(f), &st) != 0 || (size_t)st.st_size != expect)
                rc = gg::fail(ctx, GG_EIO, "state: %s is truncated or has trailing bytes (%lld bytes, expected %zu); nothing was loaded", path,
                              (long long)st.st_size, expect);
        }
    }
    std::vector<float> tmp;
    Scalars loaded[2] = {};
    for (int m = 0; m < 2 && rc == GG_OK; ++m) {
        gg::Model &M = ctx->model[m];
        Scalars sc{M.t, M.b1p, M.b2p};
        if (save) {
            if (fwrite(&sc, sizeof(sc), 1, f) != 1) rc = gg::fail(ctx, GG_EIO, "state: short write");
        } else {
            if (fread(&sc, sizeof(sc), 1, f) != 1) rc = gg::fail(ctx, GG_EIO, "state: short read");
            else loaded[m] = sc;
        }
        if (rc == GG_OK) rc = io_dev(ctx, f, M.E, ne, save, tmp);
        if (rc == GG_OK) rc = io_dev(ctx, f, M.b, nb, save, tmp);
        if (slots) {
            if (rc == GG_OK) rc = io_dev(ctx, f, M.mE, ne, save, tmp);
            if (rc == GG_OK) rc = io_dev(ctx, f, M.vE, ne, save, tmp);
            if (rc == GG_OK) rc = io_dev(ctx, f, M.mb, nb, save, tmp);
            if (rc == GG_OK) rc = io_dev(ctx, f, M.vb, nb, save, tmp);
        }
    }
    if (save) {
        if (rc == GG_OK && (fflush(f) != 0 || fsync(fileno(f)) != 0)) rc = gg::fail(ctx, GG_EIO, "state: cannot flush %s", tmp_path.c_str());
        fclose(f);
        if (rc == GG_OK && rename(tmp_path.c_str(), path) != 0) rc = gg::fail(ctx, GG_EIO, "state: cannot rename %s to %s", tmp_path.c_str(), path);
        if (rc != GG_OK) (void)remove(tmp_path.c_str());
        return rc;
    }
    fclose(f);
    generator_changed(ctx);
    for (int m = 0; m < 2; ++m) {  // (the load path only: a save returns above)
        const int rs = gg::rescan_table_finite(ctx, m);
        if (rc == GG_OK) rc = rs;  // an enqueue failure would leave the finiteness flags stale
    }
    if (rc == GG_OK)  // step counts / beta powers only once every table arrived
        for (int m = 0; m < 2; ++m) { ctx->model[m].t = loaded[m].t; ctx->model[m].b1p = loaded[m].b1p; ctx->model[m].b2p = loaded[m].b2p; }
    return rc;
}
}  // namespace

extern "C" int gg_save_state(gg_ctx *ctx, const char *path) { return state_io(ctx, path, true); }
extern "C" int gg_load_state(gg_ctx *ctx, const char *path) { return state_io(ctx, path, false); }
