// comm.hip -- C1: gradient exchange between the per-GPU replicas over RCCL / xGMI.
//
// The reference is single-device (one tf.Session, src/GraphGAN/graph_gan.py:57-61); sharding
// roots over GPUs is new.  One process per GPU; every optimizer step sums
// the dense gradient accumulators (gradE [n_node*ld], gradb [n_node]) of all ranks with
// ncclAllReduce on the context's own HIP stream, so the update kernels that follow on the
// same stream see the global gradient and every replica applies the identical update.
//
// librccl is dlopen'ed on first use: a single-GPU run never loads it.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "gg_internal.h"

namespace gg {

typedef struct { char internal[128]; } RcclUniqueId;
typedef int (*fn_GetUniqueId)(RcclUniqueId *);
typedef int (*fn_CommInitRank)(void **, int, RcclUniqueId, int);
typedef int (*fn_CommDestroy)(void *);
typedef int (*fn_AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_AllGather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_ReduceScatter)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*fn_GetErrorString)(int);
typedef int (*fn_Send)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_Recv)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_Group)(void);

struct RcclApi {
    void *handle = nullptr;
    fn_GetUniqueId GetUniqueId = nullptr;
    fn_CommInitRank CommInitRank = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_AllReduce AllReduce = nullptr;
    fn_AllGather AllGather = nullptr;
    fn_ReduceScatter ReduceScatter = nullptr;
    fn_GetErrorString GetErrorString = nullptr;
    fn_Send Send = nullptr;            // (point-to-point: optional -- without them the owner-partitioned exchange is not offered)
    fn_Recv Recv = nullptr;
    fn_Group GroupStart = nullptr, GroupEnd = nullptr;
};

static RcclApi g_rccl;
constexpr int NCCL_FLOAT32 = 7;  // ncclFloat32
constexpr int NCCL_SUM = 0;      // ncclSum
constexpr int NCCL_MAX = 2;      // ncclMax
constexpr int NCCL_INT32 = 2;    // ncclInt32
constexpr int NCCL_INT64 = 4;    // ncclInt64

static int load_rccl(gg_ctx *ctx) {
    if (g_rccl.handle) return GG_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(ctx, GG_ECOMM, "cannot dlopen librccl: %s", dlerror());
    RcclApi api;
    api.handle = h;
    api.GetUniqueId = (fn_GetUniqueId)dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (fn_CommInitRank)dlsym(h, "ncclCommInitRank");
    api.CommDestroy = (fn_CommDestroy)dlsym(h, "ncclCommDestroy");
    api.AllReduce = (fn_AllReduce)dlsym(h, "ncclAllReduce");
    api.AllGather = (fn_AllGather)dlsym(h, "ncclAllGather");
    api.ReduceScatter = (fn_ReduceScatter)dlsym(h, "ncclReduceScatter");
    api.GetErrorString = (fn_GetErrorString)dlsym(h, "ncclGetErrorString");
    api.Send = (fn_Send)dlsym(h, "ncclSend");
    api.Recv = (fn_Recv)dlsym(h, "ncclRecv");
    api.GroupStart = (fn_Group)dlsym(h, "ncclGroupStart");
    api.GroupEnd = (fn_Group)dlsym(h, "ncclGroupEnd");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.AllGather || !api.ReduceScatter || !api.GetErrorString)
        return fail(ctx, GG_ECOMM, "librccl is missing a required symbol");
    g_rccl = api;
    return GG_OK;
}

#define GG_NCCL(ctx, call)                                                                              \
    do {                                                                                                \
        int r__ = (call);                                                                               \
        if (r__ != 0) return fail(ctx, GG_ECOMM, "%s -> %s", #call, g_rccl.GetErrorString(r__));        \
    } while (0)

// Dense exchange (GG_OPT_ADAM_DENSE, and the scale modes when the replicas together touch most of the table): the sum of
// the whole accumulators.  xGMI is point to point (7 links per GPU): the sum is issued as an in-place REDUCE-SCATTER --
// every rank ends up owning the sum of one 1/world slice, each slice arriving over its own link -- followed by an in-place
// ALL-GATHER of the slices, instead of one ring all-reduce that is bound by a single link.  The
// accumulator is padded to a multiple of the world size (gg_comm_init).  GG_COMM_DENSE=allreduce selects the plain call.
int comm_allreduce_grads(gg_ctx *ctx) {
    if (!ctx->comm) return GG_OK;
    const size_t ne = (size_t)ctx->n_node * ctx->ld;
    if (ctx->comm_rsag && ctx->world > 1) {
        const size_t chunk = ctx->grad_elems_padded / (size_t)ctx->world;
        float *mine = ctx->gradE + chunk * (size_t)ctx->rank;
        GG_NCCL(ctx, g_rccl.ReduceScatter(ctx->gradE, mine, chunk, NCCL_FLOAT32, NCCL_SUM, ctx->comm, ctx->stream));
        GG_NCCL(ctx, g_rccl.AllGather(mine, ctx->gradE, chunk, NCCL_FLOAT32, ctx->comm, ctx->stream));
    } else {
        GG_NCCL(ctx, g_rccl.AllReduce(ctx->gradE, ctx->gradE, ne, NCCL_FLOAT32, NCCL_SUM, ctx->comm, ctx->stream));
    }
    GG_NCCL(ctx, g_rccl.AllReduce(ctx->gradb, ctx->gradb, (size_t)ctx->n_node, NCCL_FLOAT32, NCCL_SUM, ctx->comm, ctx->stream));
    return GG_OK;
}

// bytes a rank sends for one dense exchange (statistics: gg_comm_stats)
size_t comm_dense_bytes(const gg_ctx *ctx) {
    if (!ctx->comm || ctx->world <= 1) return 0;
    const double f = (double)(ctx->world - 1) / (double)ctx->world;
    return (size_t)(2.0 * f * 4.0 * ((double)ctx->grad_elems_padded + (double)ctx->n_node));
}

// max of `count` int64 words over ranks (row / pair counts of a prepare call)
int comm_allreduce_max_i64(gg_ctx *ctx, int64_t *buf, size_t count) {
    if (!ctx->comm) return GG_OK;
    GG_NCCL(ctx, g_rccl.AllReduce(buf, buf, count, NCCL_INT64, NCCL_MAX, ctx->comm, ctx->stream));
    return GG_OK;
}

// sum of `count` int64 words over ranks (pair counts of a generator step)
int comm_allreduce_i64(gg_ctx *ctx, int64_t *buf, size_t count) {
    if (!ctx->comm) return GG_OK;
    GG_NCCL(ctx, g_rccl.AllReduce(buf, buf, count, NCCL_INT64, NCCL_SUM, ctx->comm, ctx->stream));
    return GG_OK;
}

// sum of the touched-row flags over ranks (the dense fall-back of the sparse exchange needs the union)
int comm_allreduce_flags(gg_ctx *ctx) {
    if (!ctx->comm) return GG_OK;
    GG_NCCL(ctx, g_rccl.AllReduce(ctx->touched, ctx->touched, (size_t)ctx->n_node, NCCL_INT32, NCCL_SUM, ctx->comm, ctx->stream));
    return GG_OK;
}

// all-gather of `count` elements per rank (bytes = count * elem); with GG_COMM_FAKE_WORLD (tests on one
// GPU) every "rank" is a copy of the local buffer
int comm_allgather(gg_ctx *ctx, const void *send, void *recv, size_t count, int elem_bytes) {
    if (ctx->comm) {
        const int dt = elem_bytes == 8 ? NCCL_INT64 : NCCL_FLOAT32;  // 4-byte payloads travel as fp32 words
        GG_NCCL(ctx, g_rccl.AllGather(send, recv, count, dt, ctx->comm, ctx->stream));
        return GG_OK;
    }
    for (int r = 0; r < ctx->fake_world; ++r)
        GG_HIP(ctx, hipMemcpyAsync((char *)recv + (size_t)r * count * elem_bytes, send, count * elem_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return GG_OK;
}

bool comm_has_p2p(const gg_ctx *ctx) {
    return !ctx->comm || (g_rccl.Send && g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd);
}

// Personalised exchange of 4-byte words: this rank sends send_cnt[r] words from send + send_off[r] to every peer r and
// receives recv_cnt[r] words from it at recv + recv_off[r] (one grouped batch of point-to-point calls: xGMI is point to
// point, every pair has its own link).  The caller handles r == rank itself.
int comm_exchange_v(gg_ctx *ctx, const float *send, const int64_t *send_off, const int64_t *send_cnt, float *recv, const int64_t *recv_off,
                    const int64_t *recv_cnt) {
    if (!ctx->comm) return GG_OK;
    GG_NCCL(ctx, g_rccl.GroupStart());
    // the group is ALWAYS closed: a Send / Recv that fails must not leave later collectives queued into a dangling group
    int first_err = 0;
    const char *what = "";
    for (int r = 0; r < ctx->world && first_err == 0; ++r) {
        if (r == ctx->rank) continue;
        if (send_cnt[r]) {
            first_err = g_rccl.Send(send + send_off[r], (size_t)send_cnt[r], NCCL_FLOAT32, r, ctx->comm, ctx->stream);
            what = "ncclSend";
        }
        if (first_err == 0 && recv_cnt[r]) {
            first_err = g_rccl.Recv(recv + recv_off[r], (size_t)recv_cnt[r], NCCL_FLOAT32, r, ctx->comm, ctx->stream);
            what = "ncclRecv";
        }
    }
    const int end_err = g_rccl.GroupEnd();
    if (first_err != 0) return fail(ctx, GG_ECOMM, "comm_exchange_v: %s -> %s (the other ranks are left in their half of the exchange: destroy the communicator)", what, g_rccl.GetErrorString(first_err));
    if (end_err != 0) return fail(ctx, GG_ECOMM, "comm_exchange_v: ncclGroupEnd -> %s", g_rccl.GetErrorString(end_err));
    return GG_OK;
}

void comm_destroy(gg_ctx *ctx) {
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->world = 1;
    ctx->rank = 0;
}

}  // namespace gg

using namespace gg;

extern "C" {

int gg_comm_unique_id(void *id128) {
    if (!id128) return fail(nullptr, GG_EINVAL, "gg_comm_unique_id: NULL");
    int rc = load_rccl(nullptr);
    if (rc != GG_OK) return rc;
    RcclUniqueId id;
    GG_NCCL(nullptr, g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return GG_OK;
}

int gg_comm_init(gg_ctx *ctx, const void *id128, int32_t rank, int32_t world) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, id128 && world >= 1 && rank >= 0 && rank < world, GG_EINVAL, "gg_comm_init: bad argument");
    GG_CHECK(ctx, !ctx->comm, GG_EINVAL, "gg_comm_init: already initialised");
    // world == 1 needs no communicator; GG_COMM_FORCE=1 creates a 1-rank one anyway so that the
    // RCCL plumbing (dlopen, enums, stream) can be exercised on a single-GPU box (tests).
    if (world == 1 && !getenv("GG_COMM_FORCE")) return GG_OK;
    int rc = load_rccl(ctx);
    if (rc != GG_OK) return rc;
    GG_HIP(ctx, hipSetDevice(ctx->device));
    RcclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    void *comm = nullptr;
    GG_NCCL(ctx, g_rccl.CommInitRank(&comm, world, id, rank));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->world = world;
    if (const char *m = getenv("GG_COMM_DENSE")) ctx->comm_rsag = strcmp(m, "allreduce") != 0;
    // the reduce-scatter needs equal slices: pad the gradient accumulator to a multiple of the world size
    const size_t ne = (size_t)ctx->n_node * ctx->ld;
    const size_t padded = (ne + (size_t)world - 1) / (size_t)world * (size_t)world;
    if (padded != ctx->grad_elems_padded) {
        GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->gradE) (void)hipFree(ctx->gradE);
        ctx->gradE = nullptr;
        GG_HIP(ctx, hipMalloc((void **)&ctx->gradE, sizeof(float) * padded));
        GG_HIP(ctx, hipMemset(ctx->gradE, 0, sizeof(float) * padded));
        ctx->grad_elems_padded = padded;
    }
    return GG_OK;
}

int gg_comm_barrier(gg_ctx *ctx) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->comm) {
        // a 4-byte all-reduce on the (idle) touched_cnt scratch word [3]
        float *w = (float *)(ctx->touched_cnt + 3);
        GG_NCCL(ctx, g_rccl.AllReduce(w, w, 1, NCCL_FLOAT32, NCCL_SUM, ctx->comm, ctx->stream));
    }
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    harvest_timings(ctx);
    return check_exchange_flag(ctx);
}

}  // extern "C"

// gg_comm_stats: see include/graphgan_hip.h.
extern "C" int gg_comm_stats_ex(gg_ctx *ctx, int64_t *out8) {
    if (!ctx || !out8) return fail(ctx, GG_EINVAL, "gg_comm_stats_ex: NULL argument");
    out8[0] = ctx->comm_steps_sparse - ctx->comm_steps_owner;
    out8[1] = ctx->comm_steps_owner;
    out8[2] = ctx->comm_steps_dense;
    out8[3] = ctx->comm_bytes_sent;
    out8[4] = ctx->world;
    out8[5] = ctx->comm_bf16 ? 1 : 0;
    out8[6] = out8[7] = 0;
    return GG_OK;
}

extern "C" int gg_comm_stats(gg_ctx *ctx, int64_t *out4) {
    if (!ctx || !out4) return fail(ctx, GG_EINVAL, "gg_comm_stats: NULL argument");
    out4[0] = ctx->comm_steps_sparse;
    out4[1] = ctx->comm_steps_dense;
    out4[2] = ctx->comm_bytes_sent;
    out4[3] = ctx->world;
    return GG_OK;
}
