// walk_sample.hip -- K1: the generator's graph-softmax walk sampler on gfx950.
//
// Replaces GraphGAN.sample (reference src/GraphGAN/graph_gan.py:225-270) together with the
// all_score fetch it performs per call (:238; generator.py:21, score = g_cur . g_j + b[j]),
// utils.softmax (src/utils.py:131-133) and np.random.choice (:262).
//
// Mapping: one 64-lane wavefront per walk, persistent grid, walks of a root
// kept on one XCD.  Per hop the wave
//   1. reads the tree list of (root, cur) from the tree CSR (scalar loads),
//   2. scores the k tree neighbours: four 16-lane groups, each streaming one neighbour row as
//      float4 chunks (256 B contiguous per group per load), fmaf chain per lane, xor-butterfly
//      over the 16 lanes (spec S1), + bias; scores parked in LDS (4 KiB per wave; a per-wave
//      HBM scratch row when k > 1024),
//   3. turns scores into exact fixed-point weights (S2, S3), wave-scans them (uint64) and
//      picks the first neighbour whose inclusive prefix exceeds floor(m * W / 2^53) (S4, S5).
// The kernel is HBM/L2-latency bound: algorithmic bytes per hop = 4k(d+2) + 4d + 12.
#include "gg_arith.h"
#include "gg_internal.h"

namespace gg {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int SCORE_CAP = 1024;  // fp32 scores per wave kept in LDS

struct WalkArgs {
    const float *E;
    const float *bias;
    int32_t n_node, ld, nchunk;  // nchunk = ld / 4 float4 chunks per row
    const int32_t *t_root;
    const int32_t *t_off;
    int32_t *t_nbr;
    const int64_t *t_base;
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
    int32_t *first_child;  // [total_walks]  D-mode: depth-1 child whose father entry this walk removes
    int32_t *abort_walk;   // [n_slots]      D-mode: smallest walk index that hit a leaf child
    float *scratch;        // per-wave score rows for k > SCORE_CAP
    int64_t scratch_stride;
    unsigned long long *ctr;  // [0] hops [1] nbr_reads [2] walks [3] error flag
};

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

template <int NCH>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void walk_sample_kernel(const WalkArgs a) {
    __shared__ float lds_scores[WAVES_PER_BLOCK][SCORE_CAP];
    __shared__ unsigned long long blk_ctr[2];

    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int t = lane & 15;  // virtual lane of spec S1
    const int q = lane >> 4;  // 16-lane group
    if (threadIdx.x < 2) blk_ctr[threadIdx.x] = 0;
    __syncthreads();

    // XCD-aware block order: hardware block b runs on XCD b % 8; give each XCD a contiguous
    // range of logical blocks so that the walks of one root share one L2.
    const int nblk = gridDim.x;
    const int lblock = (nblk % 8 == 0) ? (int)((blockIdx.x % 8) * (nblk / 8) + blockIdx.x / 8) : (int)blockIdx.x;
    const int64_t n_waves = (int64_t)nblk * WAVES_PER_BLOCK;
    float *const sbuf_lds = lds_scores[wib];
    float *const sbuf_glb = a.scratch + ((int64_t)blockIdx.x * WAVES_PER_BLOCK + wib) * a.scratch_stride;

    unsigned long long my_hops = 0, my_reads = 0;

    for (int64_t w = (int64_t)lblock * WAVES_PER_BLOCK + wib; w < a.total_walks; w += n_waves) {
        // walk -> (item i, walk-in-root j): upper_bound on walk_ptr
        int lo = 0, hi = a.n_slots;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.walk_ptr[mid] <= w) lo = mid; else hi = mid;
        }
        const int item = lo;
        const uint32_t j = (uint32_t)(w - a.walk_ptr[item]);
        const int slot = a.slots[item];
        const int root = a.t_root[slot];
        const int32_t *const o = a.t_off + (int64_t)slot * (a.n_node + 1);
        int32_t *const nb = a.t_nbr + a.t_base[slot];
        int32_t *const path = a.paths + w * (int64_t)a.stride;

        int cur = root, prev = -1, len = 1;
        uint32_t hop = 0;
        bool aborted = false, overflow = false;
        if (lane == 0) path[0] = root;
        if (a.for_d && lane == 0) a.first_child[w] = -1;

        for (;;) {
            int beg = o[cur];
            const int end = o[cur + 1];
            if (hop == 0) beg += 1;                 // tree[root][1:]  (graph_gan.py:250)
            else if (nb[beg] < 0) beg += 1;         // father entry removed earlier (Q3)
            int k = end - beg;
            if (k == 0) { aborted = true; break; }  // "the tree only has a root" (:252-253)
            if (a.for_d && hop == 1 && nb[beg] == root) {
                if (k == 1) {                       // node_neighbor == [root] (:255-257)
                    if (lane == 0) atomicMin(&a.abort_walk[item], (int)j);
                    aborted = true;
                    break;
                }
                if (lane == 0) a.first_child[w] = cur;  // node_neighbor.remove(root) (:258-259), applied by the post-pass
                beg += 1;
                k -= 1;
            }
            const int32_t *const ids = nb + beg;
            float *const sbuf = (k <= SCORE_CAP) ? sbuf_lds : sbuf_glb;

            // ---- current-node row, chunks t, t+16, ... (spec S1)
            float4 gc[NCH];
            const float4 *const crow = (const float4 *)(a.E + (int64_t)cur * a.ld);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = t + 16 * c;
                gc[c] = (ch < a.nchunk) ? crow[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
            }

            // ---- pass 1: scores
            float mx = -INFINITY;
            for (int j0 = 0; j0 < k; j0 += 64) {
                const int nblock = min(64, k - j0);
                const int myid = (lane < nblock) ? ids[j0 + lane] : 0;
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
                            if (t == 0) sbuf[j0 + s + u * 4 + q] = sc;
                        }
                    }
                }
            }
            mx = wave_max_f32(mx);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---- passes 2/3: exact fixed-point inverse-CDF sample
            const uint64_t m53 = uniform53(a.seed, a.stream, (uint32_t)root, j, hop);
            int idx = 0;
            if (k <= 64) {
                const uint64_t wgt = (lane < k) ? weight_fix40(exp_spec(sbuf[lane] - mx)) : 0ull;
                const uint64_t C = wave_incl_scan_u64(wgt, lane);
                const uint64_t W = __shfl(C, 63, 64);
                const uint64_t thr = threshold(m53, W);
                const unsigned long long bal = __ballot(C > thr);
                idx = __ffsll((long long)bal) - 1;
            } else {
                uint64_t part = 0;
                for (int jj = lane; jj < k; jj += 64) part += weight_fix40(exp_spec(sbuf[jj] - mx));
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
                const uint64_t thr = threshold(m53, part);
                uint64_t carry = 0;
                for (int j0 = 0; j0 < k; j0 += 64) {
                    const int jj = j0 + lane;
                    const uint64_t wgt = (jj < k) ? weight_fix40(exp_spec(sbuf[jj] - mx)) : 0ull;
                    const uint64_t C = carry + wave_incl_scan_u64(wgt, lane);
                    const unsigned long long bal = __ballot(C > thr);
                    if (bal) { idx = j0 + __ffsll((long long)bal) - 1; break; }
                    carry = __shfl(C, 63, 64);
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int nxt = ids[idx];
            my_hops += 1;
            my_reads += (unsigned long long)k;
            if (len >= a.stride) { overflow = true; break; }
            if (lane == 0) path[len] = nxt;
            len += 1;
            hop += 1;
            if (nxt == prev) break;  // terminating condition (:264-266): sample = cur
            prev = cur;
            cur = nxt;
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

    if (lane == 0) {
        atomicAdd(&blk_ctr[0], my_hops);
        atomicAdd(&blk_ctr[1], my_reads);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blk_ctr[0]) atomicAdd(&a.ctr[0], blk_ctr[0]);
        if (blk_ctr[1]) atomicAdd(&a.ctr[1], blk_ctr[1]);
    }
}

// D-mode post-pass: the reference walks sequentially and stops a root at its first aborting
// walk (graph_gan.py:255-257); only walks before it have mutated the tree (:258-259).
__global__ void walk_d_postpass_kernel(const WalkArgs a) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.total_walks) return;
    int lo = 0, hi = a.n_slots;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.walk_ptr[mid] <= w) lo = mid; else hi = mid;
    }
    const int item = lo;
    const int j = (int)(w - a.walk_ptr[item]);
    const int ab = a.abort_walk[item];
    const int slot = a.slots[item];
    if (j < ab) {
        const int c = a.first_child[w];
        if (c >= 0) {
            const int32_t *o = a.t_off + (int64_t)slot * (a.n_node + 1);
            a.t_nbr[a.t_base[slot] + o[c]] = -1;
        }
    }
    if (ab != 0x7fffffff) {
        a.samples[w] = -1;
        a.path_len[w] = 0;
        if (j == 0) a.status[item] = GG_ROOT_ABORTED;
    }
}

__global__ void walk_init_status_kernel(const WalkArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_slots) return;
    a.status[i] = (a.walk_ptr[i + 1] == a.walk_ptr[i]) ? GG_ROOT_EMPTY : GG_ROOT_OK;
    a.abort_walk[i] = 0x7fffffff;
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
    a.t_off = ctx->t_off;
    a.t_nbr = ctx->t_nbr;
    a.t_base = ctx->t_base;
    a.slots = ctx->w_slots.as<int32_t>();
    a.walk_ptr = ctx->w_ptr.as<int64_t>();
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

    hipLaunchKernelGGL(walk_init_status_kernel, dim3(cdiv(n_slots, 256)), dim3(256), 0, ctx->stream, a);
    if (total_walks == 0) return GG_OK;

    // persistent grid: <= 8 blocks of 4 waves per CU, multiple of 8 for the XCD mapping
    int64_t blocks = (total_walks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    const int64_t max_blocks = 256 * 8;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks >= 8) blocks -= blocks % 8;
    const int64_t need_stride = ctx->tree_max_list > SCORE_CAP ? ((ctx->tree_max_list + 63) / 64 * 64) : 0;
    GG_HIP(ctx, ctx->w_scratch.reserve((size_t)need_stride * blocks * WAVES_PER_BLOCK * sizeof(float) + 16));
    a.scratch = ctx->w_scratch.as<float>();
    a.scratch_stride = need_stride;

    const int nch = (a.nchunk + 15) / 16;
    GG_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    if (nch <= 1) hipLaunchKernelGGL(walk_sample_kernel<1>, dim3((unsigned)blocks), dim3(WAVES_PER_BLOCK * 64), 0, ctx->stream, a);
    else if (nch == 2) hipLaunchKernelGGL(walk_sample_kernel<2>, dim3((unsigned)blocks), dim3(WAVES_PER_BLOCK * 64), 0, ctx->stream, a);
    else if (nch <= 4) hipLaunchKernelGGL(walk_sample_kernel<4>, dim3((unsigned)blocks), dim3(WAVES_PER_BLOCK * 64), 0, ctx->stream, a);
    else if (nch <= 8) hipLaunchKernelGGL(walk_sample_kernel<8>, dim3((unsigned)blocks), dim3(WAVES_PER_BLOCK * 64), 0, ctx->stream, a);
    else return fail(ctx, GG_EINVAL, "n_emb %d not supported (max 512)", ctx->n_emb);
    GG_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    if (for_d)
        hipLaunchKernelGGL(walk_d_postpass_kernel, dim3(cdiv(total_walks, 256)), dim3(256), 0, ctx->stream, a);
    GG_HIP(ctx, hipGetLastError());
    return GG_OK;
}

}  // namespace gg
