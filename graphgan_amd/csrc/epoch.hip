// epoch.hip -- an epoch over ROOT BATCHES (ABI 5: gg_epoch_begin / gg_epoch_add / gg_epoch_commit / gg_q3_*).
//
// The reference keeps the BFS tree of EVERY root resident (self.trees, src/GraphGAN/graph_gan.py:31-46) and its epoch
// visits every root: prepare_data_for_d / prepare_data_for_g loop over all of root_nodes (:188, :208), then the passes run
// over all rows (:149-157, :168-176).  N trees are N^2 entries -- 12 TB at 10^6 nodes -- so above a few 10^4 nodes only a
// batch of trees can be resident at a time.  This file keeps the reference's schedule with one batch of trees in HBM:
//
//   gg_epoch_begin        empty the accumulated D rows and / or G pairs
//   gg_epoch_add          for one batch of roots: BFS trees on the GPU (bfs_gpu.hip) into the resident slots, the roots' Q3
//                         bits restored from the persistent store, D-mode walks + rows (gg_prepare_d), the bits saved back,
//                         G-mode walks + window pairs (gg_prepare_g), rows and pairs APPENDED to the epoch's arrays
//   gg_epoch_commit       the accumulated arrays become the resident prepared data of gg_d_pass / gg_g_pass (for G the rewards,
//                         discriminator.py:33-34, are evaluated now: after the discriminator's passes, as :220-222 does)
//
// Q3 (graph_gan.py:258-259): D-mode walks delete the father entry of visited depth-1 children FOR GOOD -- the reference's
// trees live for the whole run.  With trees rebuilt per batch the bits cannot live in the tree slots: they are kept per
// (root node, child of the root) in a store of sum_v ceil(deg(v) / 32) words that outlives every rebuild.
//
// The G-mode walks of a batch may run in the SAME gg_epoch_add as its D-mode walks (do_d = do_g = 1): they read only the
// generator's tables, the trees and the Q3 bits the D-mode walks of the same root have just set -- none of which the
// discriminator's passes change -- so one BFS per root serves both phases of an outer epoch.
#include <algorithm>
#include <vector>

#include "gg_internal.h"

namespace gg {

int launch_pair_reward(gg_ctx *ctx, const int32_t *d_u, const int32_t *d_v, int64_t n, float *d_out);  // prepare.hip

// slot s holds root t_root[s]: its ceil(deg / 32) Q3 words <-> the root node's words of the store
__global__ __launch_bounds__(256) void q3_copy_kernel(const int32_t *t_root, const int64_t *t_q3off, uint32_t *t_q3, const int64_t *s_off,
                                                      uint32_t *store, int n_slots, int to_store) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const int v = t_root[s];
    const int64_t a = t_q3off[s], b = s_off[v], words = min(t_q3off[s + 1] - a, s_off[v + 1] - b);
    for (int64_t k = 0; k < words; ++k) {
        if (to_store) store[b + k] = t_q3[a + k];
        else t_q3[a + k] = store[b + k];
    }
}

static int ensure_q3_store(gg_ctx *ctx) {
    if (ctx->q3_store_ready) return GG_OK;
    GG_CHECK(ctx, !ctx->h_rowptr.empty(), GG_EINVAL, "epoch: call gg_set_graph_csr first");
    const int n = ctx->n_node;
    std::vector<int64_t> off(n + 1, 0);
    for (int v = 0; v < n; ++v) off[v + 1] = off[v] + (ctx->h_rowptr[v + 1] - ctx->h_rowptr[v] + 31) / 32;
    GG_HIP(ctx, ctx->q3s_off.reserve(sizeof(int64_t) * (n + 1)));
    GG_HIP(ctx, ctx->q3_store.reserve(sizeof(uint32_t) * (size_t)std::max<int64_t>(off[n], 1)));
    GG_HIP(ctx, hipMemcpy(ctx->q3s_off.p, off.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));
    GG_HIP(ctx, hipMemset(ctx->q3_store.p, 0, sizeof(uint32_t) * (size_t)std::max<int64_t>(off[n], 1)));
    ctx->q3_words = off[n];
    ctx->q3_store_ready = true;
    return GG_OK;
}

static int q3_copy(gg_ctx *ctx, int n_slots, bool to_store) {
    if (n_slots == 0) return GG_OK;
    hipLaunchKernelGGL(q3_copy_kernel, dim3(cdiv(n_slots, 256)), dim3(256), 0, ctx->stream, ctx->t_root, ctx->t_q3off, ctx->t_q3,
                       ctx->q3s_off.as<int64_t>(), ctx->q3_store.as<uint32_t>(), n_slots, to_store ? 1 : 0);
    GG_HIP(ctx, hipGetLastError());
    return GG_OK;
}

// grow a buffer that already holds `keep` bytes worth keeping
static int grow_keep(gg_ctx *ctx, DevBuf &b, size_t need, size_t keep) {
    if (need <= b.bytes) return GG_OK;
    DevBuf bigger;
    hipError_t e = bigger.reserve(need + need / 2);
    if (e != hipSuccess) return fail(ctx, GG_ENOMEM, "epoch: %zu bytes for the accumulated samples: %s", need, hipGetErrorString(e));
    if (keep && b.p) {
        GG_HIP(ctx, hipMemcpyAsync(bigger.p, b.p, keep, hipMemcpyDeviceToDevice, ctx->stream));
        GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    b.release();
    b = bigger;
    return GG_OK;
}

static int append(gg_ctx *ctx, DevBuf &acc, const DevBuf &src, int64_t have, int64_t add, size_t elem) {
    int rc = grow_keep(ctx, acc, elem * (size_t)(have + add + 1), elem * (size_t)have);
    if (rc != GG_OK) return rc;
    if (add) GG_HIP(ctx, hipMemcpyAsync((char *)acc.p + elem * (size_t)have, src.p, elem * (size_t)add, hipMemcpyDeviceToDevice, ctx->stream));
    return GG_OK;
}

}  // namespace gg

using namespace gg;

extern "C" {

int gg_epoch_begin(gg_ctx *ctx, int32_t reset_d, int32_t reset_g) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_q3_store(ctx);
    if (rc != GG_OK) return rc;
    if (reset_d) ctx->ep_rows = 0;
    if (reset_g) ctx->ep_pairs = 0;
    return GG_OK;
}

int gg_epoch_add(gg_ctx *ctx, const int32_t *roots, int32_t n_roots, int32_t do_d, int32_t do_g, int32_t n_sample, uint64_t seed,
                 uint32_t stream_d, uint32_t stream_g, int64_t *rows_total_out, int64_t *pairs_total_out) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, n_roots >= 0 && (roots || n_roots == 0), GG_EINVAL, "gg_epoch_add: bad roots");
    if (n_roots > 1 && (do_d || do_g)) {
        // the roots of a batch must be distinct: two slots of one root would restore from and save to the SAME words of the Q3
        // store, and the bits the first slot's D-mode walks set would be overwritten by the second slot's save
        std::vector<int32_t> sorted(roots, roots + n_roots);
        std::sort(sorted.begin(), sorted.end());
        const auto dup = std::adjacent_find(sorted.begin(), sorted.end());
        GG_CHECK(ctx, dup == sorted.end(), GG_EINVAL, "gg_epoch_add: root %d appears twice in one batch", dup == sorted.end() ? -1 : *dup);
    }
    GG_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_q3_store(ctx);
    if (rc != GG_OK) return rc;
    if (n_roots > 0 && (do_d || do_g)) {
        ctx->in_epoch_add = true;   // (also tells the build that these trees are walked twice and dropped: lazy by default on large graphs)
        rc = gg_build_trees_device(ctx, roots, n_roots);  // (waits for everything in flight; Q3 rows of the slots zeroed)
        ctx->in_epoch_add = false;
        if (rc != GG_OK) return rc;
        rc = q3_copy(ctx, n_roots, /*to_store=*/false);
        if (rc != GG_OK) return rc;
        std::vector<int32_t> &slots = ctx->ep_slots;
        if ((int32_t)slots.size() < n_roots) {
            const int32_t old = (int32_t)slots.size();
            slots.resize(n_roots);
            for (int32_t i = old; i < n_roots; ++i) slots[i] = i;
        }
        // (no replica collective per batch: the ranks' batch counts differ; gg_epoch_commit exchanges the totals)
        ctx->in_epoch_add = true;
        if (do_d) {
            int64_t rows = 0;
            rc = gg_prepare_d(ctx, slots.data(), n_roots, seed, stream_d, &rows, nullptr);
            if (rc == GG_OK) rc = q3_copy(ctx, n_roots, /*to_store=*/true);
            if (rc == GG_OK) rc = append(ctx, ctx->ep_center, ctx->d_center, ctx->ep_rows, rows, sizeof(int32_t));
            if (rc == GG_OK) rc = append(ctx, ctx->ep_neighbor, ctx->d_neighbor, ctx->ep_rows, rows, sizeof(int32_t));
            if (rc == GG_OK) rc = append(ctx, ctx->ep_label, ctx->d_label, ctx->ep_rows, rows, sizeof(float));
            if (rc == GG_OK) ctx->ep_rows += rows;
        }
        if (rc == GG_OK && do_g) {
            int64_t pairs = 0;
            rc = gg_prepare_g(ctx, slots.data(), n_roots, n_sample, seed, stream_g, &pairs, nullptr);
            if (rc == GG_OK) rc = ensure_g_pairs(ctx);  // (node_1, node_2) of the batch's walks (graph_gan.py:272-291)
            if (rc == GG_OK) rc = append(ctx, ctx->ep_node1, ctx->g_node1, ctx->ep_pairs, pairs, sizeof(int32_t));
            if (rc == GG_OK) rc = append(ctx, ctx->ep_node2, ctx->g_node2, ctx->ep_pairs, pairs, sizeof(int32_t));
            if (rc == GG_OK) ctx->ep_pairs += pairs;
        }
        ctx->in_epoch_add = false;
        if (rc != GG_OK) return rc;
        GG_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the appends read buffers the next batch overwrites
    }
    if (rows_total_out) *rows_total_out = ctx->ep_rows;
    if (pairs_total_out) *pairs_total_out = ctx->ep_pairs;
    return GG_OK;
}

int gg_epoch_commit(gg_ctx *ctx, int32_t which, int64_t *n_out) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, which == 0 || which == 1, GG_EINVAL, "gg_epoch_commit: which must be 0 (generator pairs) or 1 (discriminator rows)");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    discard_begun_walk(ctx);
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int rc;
    if (which == 1) {
        const int64_t rows = ctx->ep_rows;
        GG_HIP(ctx, ctx->ep_center.reserve(sizeof(int32_t)));  // (an epoch without rows still hands over valid buffers)
        GG_HIP(ctx, ctx->ep_neighbor.reserve(sizeof(int32_t)));
        GG_HIP(ctx, ctx->ep_label.reserve(sizeof(float)));
        std::swap(ctx->d_center, ctx->ep_center);
        std::swap(ctx->d_neighbor, ctx->ep_neighbor);
        std::swap(ctx->d_label, ctx->ep_label);
        ctx->d_rows = rows;
        ctx->ep_rows = 0;
        rc = exchange_count_max(ctx, ctx->d_rows, &ctx->d_rows_max);
        if (rc != GG_OK) return rc;
        if (n_out) *n_out = rows;
    } else {
        const int64_t pairs = ctx->ep_pairs;
        GG_HIP(ctx, ctx->ep_node1.reserve(sizeof(int32_t)));
        GG_HIP(ctx, ctx->ep_node2.reserve(sizeof(int32_t)));
        GG_HIP(ctx, ctx->ep_reward.reserve(sizeof(float) * (size_t)(pairs + 1)));
        // sess.run(discriminator.reward) over ALL pairs (graph_gan.py:220-222), with the discriminator as it is now
        rc = launch_pair_reward(ctx, ctx->ep_node1.as<int32_t>(), ctx->ep_node2.as<int32_t>(), pairs, ctx->ep_reward.as<float>());
        if (rc != GG_OK) return rc;
        GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::swap(ctx->g_node1, ctx->ep_node1);
        std::swap(ctx->g_node2, ctx->ep_node2);
        std::swap(ctx->g_reward, ctx->ep_reward);
        ctx->g_pairs = pairs;
        ctx->g_pairs_filled = true;   // the pair arrays ARE the prepared data
        ctx->g_paths_valid = false;   // (the resident walks are the last batch's only: no whole-walk pass over them)
        ctx->ep_pairs = 0;
        rc = exchange_count_max(ctx, ctx->g_pairs, &ctx->g_pairs_max);
        if (rc != GG_OK) return rc;
        if (n_out) *n_out = pairs;
    }
    return GG_OK;
}

int gg_q3_clear(gg_ctx *ctx) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_q3_store(ctx);
    if (rc != GG_OK) return rc;
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GG_HIP(ctx, hipMemset(ctx->q3_store.p, 0, sizeof(uint32_t) * (size_t)std::max<int64_t>(ctx->q3_words, 1)));
    return GG_OK;
}

int gg_q3_get(gg_ctx *ctx, int64_t *word_off /*[n_node + 1] or NULL*/, uint32_t *words /*[word_off[n_node]] or NULL*/) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_q3_store(ctx);
    if (rc != GG_OK) return rc;
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (word_off) GG_HIP(ctx, hipMemcpy(word_off, ctx->q3s_off.p, sizeof(int64_t) * (ctx->n_node + 1), hipMemcpyDeviceToHost));
    if (words && ctx->q3_words) GG_HIP(ctx, hipMemcpy(words, ctx->q3_store.p, sizeof(uint32_t) * (size_t)ctx->q3_words, hipMemcpyDeviceToHost));
    return GG_OK;
}

}  // extern "C"
