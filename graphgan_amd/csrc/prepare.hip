// prepare.hip -- device-resident sample preparation:
//   K6 paths -> window pairs   (reference src/GraphGAN/graph_gan.py:272-291, get_node_pairs_from_path)
//   K2 pair_reward             (src/GraphGAN/discriminator.py:21-24,33-34 via graph_gan.py:220-222)
//   D rows [pos..., neg...]    (graph_gan.py:193-201)
// so that prepare_data_for_g / prepare_data_for_d (graph_gan.py:182-223) never round-trip
// node lists through the host between the walk kernel and the update passes.
#include <algorithm>

#include <cstring>

#include "gg_internal.h"

namespace gg {

// ------------------------------------------------------------------ exclusive scan (int32 -> int64)
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ int64_t block_excl_scan(int64_t v, int64_t *total, int64_t *sh /*[SCAN_THREADS/64]*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    int64_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_THREADS / 64; ++i) {
        if (i < wv) base += sh[i];
        tot += sh[i];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums(const int32_t *cnt, int64_t n, int64_t *tile_sum) {
    __shared__ int64_t sh[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) s += cnt[base + i];
    int64_t tot;
    (void)block_excl_scan(s, &tot, sh);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_offsets(int64_t *tile_sum, int64_t n_tiles, int64_t *total_out) {
    __shared__ int64_t sh[SCAN_THREADS / 64];
    int64_t carry = 0;
    for (int64_t t0 = 0; t0 < n_tiles; t0 += SCAN_THREADS) {
        const int64_t i = t0 + threadIdx.x;
        const int64_t v = i < n_tiles ? tile_sum[i] : 0;
        int64_t tot;
        const int64_t ex = block_excl_scan(v, &tot, sh);
        if (i < n_tiles) tile_sum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

// two independent tile-sum arrays in one launch (workgroup 0 / 1): the segment scan has two of them
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_offsets2(int64_t *tile_sum_a, int64_t *tile_sum_b, int64_t n_tiles, int64_t *total_a, int64_t *total_b) {
    __shared__ int64_t sh[SCAN_THREADS / 64];
    int64_t *const tile_sum = blockIdx.x == 0 ? tile_sum_a : tile_sum_b;
    int64_t carry = 0;
    for (int64_t t0 = 0; t0 < n_tiles; t0 += SCAN_THREADS) {
        const int64_t i = t0 + threadIdx.x;
        const int64_t v = i < n_tiles ? tile_sum[i] : 0;
        int64_t tot;
        const int64_t ex = block_excl_scan(v, &tot, sh);
        if (i < n_tiles) tile_sum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *(blockIdx.x == 0 ? total_a : total_b) = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply(const int32_t *cnt, int64_t n, const int64_t *tile_off, int64_t *ptr) {
    __shared__ int64_t sh[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = base + i < n ? cnt[base + i] : 0;
        s += v[i];
    }
    int64_t tot;
    int64_t run = tile_off[blockIdx.x] + block_excl_scan(s, &tot, sh);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) ptr[base + i] = run;
        run += v[i];
    }
}

// ptr[0..n] <- exclusive prefix sums of cnt[0..n); ptr[n] = total.
int device_exclusive_scan(gg_ctx *ctx, const int32_t *cnt, int64_t *ptr, int64_t n) {
    const int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    GG_HIP(ctx, ctx->scan_tmp.reserve(sizeof(int64_t) * (tiles + 2)));
    int64_t *ts = ctx->scan_tmp.as<int64_t>();
    if (n == 0) {
        GG_HIP(ctx, hipMemsetAsync(ptr, 0, sizeof(int64_t), ctx->stream));
        return GG_OK;
    }
    hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream, cnt, n, ts);
    hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, ts, tiles, ptr + n);
    hipLaunchKernelGGL(scan_apply, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream, cnt, n, ts, ptr);
    GG_HIP(ctx, hipGetLastError());
    return GG_OK;
}

// Flags -> ascending list of the set positions (the optimizer's touched rows, deterministic order) in the scan's own third
// pass: the tile offsets are all the compaction needs, so the [n + 1] offset array (8 B per node written, then read again by
// a separate compaction kernel) is never materialised.  *total_out = number of set flags.
__global__ __launch_bounds__(SCAN_THREADS) void flag_tile_sums(const int32_t *flag, int64_t n, int64_t *tile_sum) {
    __shared__ int64_t sh[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) s += flag[base + i] > 0 ? 1 : 0;
    int64_t tot;
    (void)block_excl_scan(s, &tot, sh);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void flag_compact(const int32_t *flag, int64_t n, const int64_t *tile_off, int32_t *list) {
    __shared__ int64_t sh[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    bool v[SCAN_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = base + i < n && flag[base + i] > 0;
        s += v[i] ? 1 : 0;
    }
    int64_t tot;
    int64_t run = tile_off[blockIdx.x] + block_excl_scan(s, &tot, sh);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (v[i]) list[run++] = (int32_t)(base + i);
}

int device_compact_flags(gg_ctx *ctx, const int32_t *flag, int64_t n, int32_t *list, int64_t *total_out) {
    const int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    GG_HIP(ctx, ctx->scan_tmp.reserve(sizeof(int64_t) * (tiles + 2)));
    int64_t *ts = ctx->scan_tmp.as<int64_t>();
    if (n == 0) {
        GG_HIP(ctx, hipMemsetAsync(total_out, 0, sizeof(int64_t), ctx->stream));
        return GG_OK;
    }
    hipLaunchKernelGGL(flag_tile_sums, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream, flag, n, ts);
    hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, ts, tiles, total_out);
    hipLaunchKernelGGL(flag_compact, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream, flag, n, ts, list);
    GG_HIP(ctx, hipGetLastError());
    return GG_OK;
}

// Row segments of the staged generator gradient (steps.hip): cnt[r] = gradient rows staged for table row r.  Rows with
// 0 < cnt <= T ("small") get a contiguous segment of the stage buffer: off[r] = exclusive sum of the small rows' counts,
// list = {row, off, cnt} of the small rows in ascending row order; totals[0] = small rows, totals[1] = their staged rows.  One scan, three launches.
__global__ __launch_bounds__(SCAN_THREADS) void seg_tile_sums(const int32_t *cnt, int64_t n, int T, int64_t *tile_rows, int64_t *tile_occ) {
    __shared__ int64_t sh[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t rows = 0, occ = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) {
            const int c = cnt[base + i];
            if (c > 0 && c <= T) { rows += 1; occ += c; }
        }
    int64_t tot;
    (void)block_excl_scan(rows, &tot, sh);
    if (threadIdx.x == 0) tile_rows[blockIdx.x] = tot;
    (void)block_excl_scan(occ, &tot, sh);
    if (threadIdx.x == 0) tile_occ[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void seg_apply(const int32_t *cnt, int64_t n, int T, const int64_t *tile_rows, const int64_t *tile_occ,
                                                          int32_t *off, int4 *list) {
    __shared__ int64_t sh[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int c[SCAN_ITEMS];
    int64_t rows = 0, occ = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int v = base + i < n ? cnt[base + i] : 0;
        c[i] = (v > 0 && v <= T) ? v : 0;
        rows += c[i] ? 1 : 0;
        occ += c[i];
    }
    int64_t tot;
    int64_t rrun = tile_rows[blockIdx.x] + block_excl_scan(rows, &tot, sh);
    int64_t orun = tile_occ[blockIdx.x] + block_excl_scan(occ, &tot, sh);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (c[i]) {
            list[rrun++] = make_int4((int32_t)(base + i), (int32_t)orun, c[i], 0);  // {row, first stage row, stage rows}
            off[base + i] = (int32_t)orun;
            orun += c[i];
        }
}

int device_segment_rows(gg_ctx *ctx, const int32_t *cnt, int64_t n, int T, int32_t *off, int4 *list, int64_t *totals, hipStream_t stream, DevBuf *scratch) {
    const int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (!scratch) scratch = &ctx->scan_tmp;  // (a launch beside the main stream's scans brings its own scratch)
    if (!stream) stream = ctx->stream;
    GG_HIP(ctx, scratch->reserve(sizeof(int64_t) * (2 * tiles + 4)));
    int64_t *tr = scratch->as<int64_t>(), *to = tr + tiles + 1;
    hipLaunchKernelGGL(seg_tile_sums, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, stream, cnt, n, T, tr, to);
    hipLaunchKernelGGL(scan_tile_offsets2, dim3(2), dim3(SCAN_THREADS), 0, stream, tr, to, tiles, totals, totals + 1);
    hipLaunchKernelGGL(seg_apply, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, stream, cnt, n, T, tr, to, off, list);
    GG_HIP(ctx, hipGetLastError());
    return GG_OK;
}

// ------------------------------------------------------------------ K6 window pairs
// pairs of path[:-1] with |i-j| <= window, i != j, in the reference's (i, then j) order.
// `flag` = the walk launch's status word: 2 means the launch is being rerun (its outputs are not
// final, path_len may be garbage) -> produce nothing.
__global__ void pair_count_kernel(const int32_t *path_len, int64_t n_walks, int window, int32_t *cnt, const unsigned long long *flag) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_walks) return;
    if (*flag == 2ull) { cnt[w] = 0; return; }
    const int L = path_len[w] - 1;  // the last element (back-step) is dropped (graph_gan.py:282)
    int c = 0;
    for (int i = 0; i < L; ++i) {
        const int lo = max(i - window, 0), hi = min(i + window + 1, L);
        c += hi - lo - 1;
    }
    cnt[w] = L > 0 ? c : 0;
}

__global__ void pair_fill_kernel(const int32_t *paths, const int32_t *path_len, int stride, int64_t n_walks, int window,
                                 const int64_t *ptr, int32_t *node1, int32_t *node2, const unsigned long long *flag) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_walks || *flag == 2ull) return;
    const int L = path_len[w] - 1;
    const int32_t *p = paths + w * (int64_t)stride;
    int64_t o = ptr[w];
    for (int i = 0; i < L; ++i) {
        const int lo = max(i - window, 0), hi = min(i + window + 1, L);
        const int c = p[i];
        for (int j = lo; j < hi; ++j) {
            if (j == i) continue;
            node1[o] = c;
            node2[o] = p[j];
            ++o;
        }
    }
}

// ------------------------------------------------------------------ K2 pair_reward
// reward = log(1 + exp(clip(d_u . d_v + b[v], -10, 10))).  One 16-lane group per pair:
// both rows streamed as float4 chunks, xor-butterfly reduce.  HBM-bound gather:
// algorithmic bytes per pair = 8d + 4 + 8 + 4.
__global__ __launch_bounds__(256) void pair_reward_kernel(const float *E, const float *bias, int ld, const int32_t *u,
                                                          const int32_t *v, int64_t n_host, const int64_t *n_dev, float *out) {
    const int64_t n = n_dev ? *n_dev : n_host;  // device-side count: no host round trip between pair expansion and rewards
    const int t = threadIdx.x & 15;
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int nchunk = ld >> 2;
    for (int64_t p = g0; p < n; p += ng) {
        const int a = u[p], b = v[p];
        const float4 *ra = (const float4 *)(E + (int64_t)a * ld);
        const float4 *rb = (const float4 *)(E + (int64_t)b * ld);
        float acc = 0.f;
        for (int c = t; c < nchunk; c += 16) {
            const float4 x = ra[c], y = rb[c];
            acc = __builtin_fmaf(x.x, y.x, acc);
            acc = __builtin_fmaf(x.y, y.y, acc);
            acc = __builtin_fmaf(x.z, y.z, acc);
            acc = __builtin_fmaf(x.w, y.w, acc);
        }
        acc += __shfl_xor(acc, 8, 64);
        acc += __shfl_xor(acc, 4, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 1, 64);
        if (t == 0) {
            float s = acc + bias[b];
            s = fminf(fmaxf(s, -10.0f), 10.0f);
            out[p] = logf(1.0f + expf(s));  // tf.log(1 + tf.exp(score)), fp32
        }
    }
}

// The same rewards for the window pairs of whole walks (prepare_g, window <= 2): one 16-lane group per walk slides a
// 5-row window over the path, so every discriminator row is read ONCE instead of once per pair (up to 8 times; the
// re-reads hit L2, but each still costs a 512-byte gather).  Pair order and arithmetic (lane t owns float4 chunks t, t+16,
// ...; fmaf chain, xor butterfly) are those of pair_fill_kernel + pair_reward_kernel: bit-identical rewards.
template <int NCH>
__global__ __launch_bounds__(256) void path_reward_kernel(const float *E, const float *bias, int ld, const int32_t *paths,
                                                          const int32_t *path_len, int stride, int64_t n_walks, int window,
                                                          const int64_t *ptr, float *out, const unsigned long long *flag) {
    const int t = threadIdx.x & 15;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (w >= n_walks || *flag == 2ull) return;
    const int L = path_len[w] - 1;
    if (L <= 1) return;
    const int32_t *p = paths + w * (int64_t)stride;
    const int nchunk = ld >> 2;
    const int64_t pi0 = ptr[w];
    // FORWARD pairs only: the centre c takes its neighbours c+1, c+2; one dot product serves the pair (c, c+d) [bias of c+d]
    // and its mirror (c+d, c) [bias of c], whose place in the pair list follows from the path length alone:
    // pairs of centre i = its min(i, W) backward then its min(L-1-i, W) forward neighbours.  The (up to four) scores of a
    // centre are spread over lanes 0..3, so the softplus runs once per centre, not once per pair.
    float4 R[3][NCH];  // slots 0..2 = path positions c, c+1, c+2
    float bv[3];
    bool have[3];
    auto fetch = [&](int pos, float4 (&row)[NCH], float &b, bool &ok) {
        ok = pos >= 0 && pos < L;
        const int nd = ok ? p[pos] : 0;
        b = ok ? bias[nd] : 0.f;
        const float4 *r = (const float4 *)(E + (int64_t)nd * ld);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = t + 16 * i;
            row[i] = (ok && c < nchunk) ? r[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0, R[0], bv[0], have[0]);
    fetch(1, R[1], bv[1], have[1]);
    fetch(2, R[2], bv[2], have[2]);
    auto n_pairs_of = [&](int i) { return min(i, window) + min(L - 1 - i, window); };
    int base = 0;  // pairs of the centres before c
    for (int c = 0; c < L; ++c) {
        float4 Rn[NCH];
        float nb;
        bool nh;
        fetch(c + 3, Rn, nb, nh);  // in flight while the centre's pairs are evaluated
        const int back_c = min(c, window);
        int base_d = base + n_pairs_of(c);  // first pair of centre c + 1 (then c + 2)
        float myval = 0.f;
        int myidx = -1;
#pragma unroll
        for (int d = 1; d <= 2; ++d) {
            if (d <= window && have[d]) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    acc = __builtin_fmaf(R[0][i].x, R[d][i].x, acc);
                    acc = __builtin_fmaf(R[0][i].y, R[d][i].y, acc);
                    acc = __builtin_fmaf(R[0][i].z, R[d][i].z, acc);
                    acc = __builtin_fmaf(R[0][i].w, R[d][i].w, acc);
                }
                acc += __shfl_xor(acc, 8, 64);
                acc += __shfl_xor(acc, 4, 64);
                acc += __shfl_xor(acc, 2, 64);
                acc += __shfl_xor(acc, 1, 64);
                if (t == 2 * d - 2) { myval = acc + bv[d]; myidx = base + back_c + (d - 1); }           // (c, c + d)
                if (t == 2 * d - 1) { myval = acc + bv[0]; myidx = base_d + (c - max(c + d - window, 0)); }  // (c + d, c)
            }
            base_d += n_pairs_of(c + d);
        }
        if (myidx >= 0) {
            const float sc = fminf(fmaxf(myval, -10.0f), 10.0f);
            out[pi0 + myidx] = logf(1.0f + expf(sc));
        }
        base += n_pairs_of(c);
#pragma unroll
        for (int i = 0; i < NCH; ++i) { R[0][i] = R[1][i]; R[1][i] = R[2][i]; R[2][i] = Rn[i]; }
        have[0] = have[1]; have[1] = have[2]; have[2] = nh;
        bv[0] = bv[1]; bv[1] = bv[2]; bv[2] = nb;
    }
}

// Evaluator scores (reference src/evaluation/link_prediction.py:26-27): np.dot of two embedding rows, in float64 like the
// reference computes them (its embeddings are float64 arrays holding the fp32 values): one 16-lane group per edge.
__global__ __launch_bounds__(256) void edge_dot_f64_kernel(const float *E, int ld, const int32_t *u, const int32_t *v, int64_t n, double *out) {
    const int t = threadIdx.x & 15;
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
    for (int64_t p = g0; p < n; p += ng) {
        const float *ra = E + (int64_t)u[p] * ld, *rb = E + (int64_t)v[p] * ld;
        double acc = 0.0;
        for (int k = t; k < ld; k += 16) acc = fma((double)ra[k], (double)rb[k], acc);
        acc += __shfl_xor(acc, 8, 64);
        acc += __shfl_xor(acc, 4, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 1, 64);
        if (t == 0) out[p] = acc;
    }
}

int launch_pair_reward(gg_ctx *ctx, const int32_t *d_u, const int32_t *d_v, int64_t n, float *d_out) {
    if (n == 0) return GG_OK;
    const Model &D = ctx->model[1];
    int64_t blocks = (n * 16 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(pair_reward_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, D.E, D.b, ctx->ld, d_u, d_v, n,
                       (const int64_t *)nullptr, d_out);
    GG_HIP(ctx, hipGetLastError());
    ctx->ctr.reward_pairs += n;
    return GG_OK;
}

// ------------------------------------------------------------------ D rows
__global__ void d_count_kernel(const int32_t *status, const int64_t *walk_ptr, int n_slots, int32_t *cnt, const unsigned long long *flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    if (*flag == 2ull) { cnt[i] = 0; return; }
    const int64_t deg = walk_ptr[i + 1] - walk_ptr[i];
    cnt[i] = (status[i] == GG_ROOT_OK && deg > 0) ? (int32_t)(2 * deg) : 0;
}

// one wavefront per root: [i]*deg + pos(graph[i]) + label 1, then [i]*deg + neg(samples) + label 0
__global__ __launch_bounds__(256) void d_fill_kernel(const int32_t *slots, const int32_t *t_root, const int64_t *g_rowptr,
                                                     const int32_t *g_col, const int32_t *samples, const int64_t *walk_ptr,
                                                     const int64_t *row_ptr, int n_slots, int32_t *center, int32_t *neighbor,
                                                     float *label, const unsigned long long *flag) {
    const int lane = threadIdx.x & 63;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n_slots || *flag == 2ull) return;
    const int64_t o = row_ptr[i], rows = row_ptr[i + 1] - o;
    if (rows == 0) return;
    const int64_t deg = rows / 2;
    const int root = t_root[slots[i]];
    const int64_t e0 = g_rowptr[root], w0 = walk_ptr[i];
    for (int64_t e = lane; e < deg; e += 64) {
        center[o + e] = root;
        neighbor[o + e] = g_col[e0 + e];
        label[o + e] = 1.0f;
        center[o + deg + e] = root;
        neighbor[o + deg + e] = samples[w0 + e];
        label[o + deg + e] = 0.0f;
    }
}

static int fill_g_pairs(gg_ctx *ctx) {
    const int64_t nw = ctx->w_total;
    hipLaunchKernelGGL(pair_fill_kernel, dim3(cdiv(nw, 256)), dim3(256), 0, ctx->stream, ctx->w_paths.as<int32_t>(),
                       ctx->w_len.as<int32_t>(), ctx->w_stride, nw, ctx->cfg.window_size, ctx->g_ptr.as<int64_t>(),
                       ctx->g_node1.as<int32_t>(), ctx->g_node2.as<int32_t>(), ctx->dev_ctr + 3);
    GG_HIP(ctx, hipGetLastError());
    ctx->g_pairs_filled = true;
    return GG_OK;
}

// K6 on demand: expand the resident walks into the (node_1, node_2) arrays (graph_gan.py:272-291) on the main stream.
int ensure_g_pairs(gg_ctx *ctx) {
    if (ctx->g_pairs_filled || ctx->g_pairs == 0) return GG_OK;
    GG_CHECK(ctx, ctx->g_paths_valid, GG_EINVAL,
             "the walks of the last gg_prepare_g were overwritten by a later walk launch before its pairs were read: fetch the pairs "
             "(gg_get_g_data) or run the minibatch passes before the next prepare call");
    return fill_g_pairs(ctx);
}

}  // namespace gg

using namespace gg;

extern "C" {

// rows of gg_prepare_d behind the walk on the same stream; capacity = 2 * (walks launched) >= rows
static int enqueue_d_rows(gg_ctx *ctx, int32_t n_slots) {
    hipLaunchKernelGGL(d_count_kernel, dim3(cdiv(n_slots, 256)), dim3(256), 0, ctx->stream, ctx->w_status.as<int32_t>(),
                       ctx->w_ptr_buf().as<int64_t>(), n_slots, ctx->d_cnt.as<int32_t>(), ctx->dev_ctr + 3);
    int rc = device_exclusive_scan(ctx, ctx->d_cnt.as<int32_t>(), ctx->d_ptr.as<int64_t>(), n_slots);
    if (rc != GG_OK) return rc;
    hipLaunchKernelGGL(d_fill_kernel, dim3(cdiv((int64_t)n_slots * 64, 256)), dim3(256), 0, ctx->stream,
                       ctx->w_slots_buf().as<int32_t>(), ctx->t_root, ctx->g_rowptr, ctx->g_col, ctx->w_samples.as<int32_t>(),
                       ctx->w_ptr_buf().as<int64_t>(), ctx->d_ptr.as<int64_t>(), n_slots, ctx->d_center.as<int32_t>(),
                       ctx->d_neighbor.as<int32_t>(), ctx->d_label.as<float>(), ctx->dev_ctr + 3);
    GG_HIP(ctx, hipMemcpyAsync(ctx->h_pin + gg_ctx::H_TOTAL, ctx->d_ptr.as<int64_t>() + n_slots, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(ctx, hipGetLastError());
    return GG_OK;
}

int gg_prepare_d(gg_ctx *ctx, const int32_t *slots, int32_t n_slots, uint64_t seed, uint32_t stream, int64_t *n_rows_out,
                 int32_t *root_status) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, ctx->g_rowptr, GG_EINVAL, "gg_prepare_d: call gg_set_graph_csr first");
    // a D path never exceeds tree depth + 2 entries
    const int stride = ctx->tree_max_depth + 3;
    ctx->d_rows = 0;
    // the distributions these walks evaluate are registered for the G-mode walks of the same step (walk_sample.hip)
    ctx->dc_valid = false;
    discard_begun_walk(ctx);
    ctx->dc_request = ctx->dc_enabled ? 1 : 0;
    int rc = walk_launch_async(ctx, slots, nullptr, -1, n_slots, 1, seed, stream, stride);
    if (rc != GG_OK) { ctx->dc_request = 0; return rc; }
    if (n_slots > 0) {
        const int64_t cap = 2 * ctx->w_total;  // every root contributes at most 2 * deg rows
        GG_HIP(ctx, ctx->d_cnt.reserve(sizeof(int32_t) * n_slots));
        GG_HIP(ctx, ctx->d_ptr.reserve(sizeof(int64_t) * (n_slots + 1)));
        GG_HIP(ctx, ctx->d_center.reserve(sizeof(int32_t) * (cap + 1)));
        GG_HIP(ctx, ctx->d_neighbor.reserve(sizeof(int32_t) * (cap + 1)));
        GG_HIP(ctx, ctx->d_label.reserve(sizeof(float) * (cap + 1)));
        rc = enqueue_d_rows(ctx, n_slots);
        if (rc != GG_OK) return rc;
    }
    bool retried = false;
    rc = walk_finalize(ctx, &retried);  // the only host synchronisation of the call (a rerun keeps the launch's cache mode)
    ctx->dc_request = 0;
    if (rc != GG_OK) return rc;
    if (n_slots > 0) {
        if (retried) {
            rc = enqueue_d_rows(ctx, n_slots);
            if (rc != GG_OK) return rc;
            GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        ctx->d_rows = (int64_t)ctx->h_pin[gg_ctx::H_TOTAL];
        ctx->dc_valid = ctx->dc_enabled && ctx->walk_levels > 0;
        if (root_status) GG_HIP(ctx, hipMemcpy(root_status, ctx->w_status.p, sizeof(int32_t) * n_slots, hipMemcpyDeviceToHost));
    }
    if (!ctx->in_epoch_add) {  // (an epoch over root batches exchanges the totals once, in gg_epoch_commit)
        rc = exchange_count_max(ctx, ctx->d_rows, &ctx->d_rows_max);  // replicas: capacity rule of the passes' row packs
        if (rc != GG_OK) return rc;
    }
    if (n_rows_out) *n_rows_out = ctx->d_rows;
    return GG_OK;
}

int gg_get_d_data(gg_ctx *ctx, int32_t *center, int32_t *neighbor, float *label) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    const int64_t n = ctx->d_rows;
    if (n == 0) return GG_OK;
    GG_HIP(ctx, hipSetDevice(ctx->device));
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (center) GG_HIP(ctx, hipMemcpy(center, ctx->d_center.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (neighbor) GG_HIP(ctx, hipMemcpy(neighbor, ctx->d_neighbor.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (label) GG_HIP(ctx, hipMemcpy(label, ctx->d_label.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    return GG_OK;
}

// pairs + rewards of gg_prepare_g behind the walk on the same stream; capacity = bound on pairs
static int enqueue_g_pairs(gg_ctx *ctx, int64_t nw, int64_t cap) {
    const int window = ctx->cfg.window_size;
    hipLaunchKernelGGL(pair_count_kernel, dim3(cdiv(nw, 256)), dim3(256), 0, ctx->stream, ctx->w_len.as<int32_t>(), nw, window,
                       ctx->g_cnt.as<int32_t>(), ctx->dev_ctr + 3);
    int rc = device_exclusive_scan(ctx, ctx->g_cnt.as<int32_t>(), ctx->g_ptr.as<int64_t>(), nw);
    if (rc != GG_OK) return rc;
    // The (node_1, node_2) arrays are written only for whoever reads them -- gg_get_g_data, minibatch passes, the per-pair
    // reward kernel (gg_ensure_g_pairs): the whole-walk kernels take the pairs from the paths themselves.
    const int nch = (ctx->ld / 4 + 15) / 16;
    const bool path_reward = window <= 2 && nch <= 4 && ctx->ld % 4 == 0 && !getenv("GG_NO_PATH_REWARD");
    ctx->g_pairs_filled = false;
    if (!path_reward) {
        rc = fill_g_pairs(ctx);
        if (rc != GG_OK) return rc;
    }
    // rewards for the device-side pair count (the host does not know it yet)
    const Model &D = ctx->model[1];
    const int ts = ctx->walk_timed ? timing_slot(ctx) : -1;  // profiled call: HIP events around the reward kernel
    if (ts >= 0) GG_HIP(ctx, hipEventRecord(ctx->tm_ev[ts][0], ctx->stream));
    if (path_reward) {
        const dim3 grid(cdiv(nw * 16, 256)), blk(256);
#define GG_PATH_REWARD(N)                                                                                                              \
    hipLaunchKernelGGL(path_reward_kernel<N>, grid, blk, 0, ctx->stream, D.E, D.b, ctx->ld, ctx->w_paths.as<int32_t>(),                 \
                       ctx->w_len.as<int32_t>(), ctx->w_stride, nw, window, ctx->g_ptr.as<int64_t>(), ctx->g_reward.as<float>(), ctx->dev_ctr + 3)
        if (nch <= 1) GG_PATH_REWARD(1);
        else if (nch == 2) GG_PATH_REWARD(2);
        else GG_PATH_REWARD(4);
#undef GG_PATH_REWARD
    } else {
        hipLaunchKernelGGL(pair_reward_kernel, dim3(256 * 16), dim3(256), 0, ctx->stream, D.E, D.b, ctx->ld, ctx->g_node1.as<int32_t>(),
                           ctx->g_node2.as<int32_t>(), (int64_t)-1, ctx->g_ptr.as<int64_t>() + nw, ctx->g_reward.as<float>());
    }
    if (ts >= 0) {
        GG_HIP(ctx, hipEventRecord(ctx->tm_ev[ts][1], ctx->stream));
        GG_HIP(ctx, hipEventRecord(ctx->tm_ev[ts][2], ctx->stream));
        // the pair count arrives with the call's one synchronisation: units are filled in by gg_prepare_g
        ctx->tm_pending.push_back({0, 0, ts, false});
    }
    GG_HIP(ctx, hipMemcpyAsync(ctx->h_pin + gg_ctx::H_TOTAL, ctx->g_ptr.as<int64_t>() + nw, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(ctx, hipGetLastError());
    (void)cap;
    return GG_OK;
}

// gg_prepare_g_begin: see include/graphgan_hip.h.  The launch is exactly the one gg_prepare_g would enqueue -- same stream,
// same arguments, same caches -- only earlier, and without the main stream waiting for it yet.
int gg_prepare_g_begin(gg_ctx *ctx, const int32_t *slots, int32_t n_slots, int32_t n_sample, uint64_t seed, uint32_t stream) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, n_sample >= 0, GG_EINVAL, "gg_prepare_g_begin: n_sample < 0");
    const int stride = ctx->tree_max_depth + 3;
    discard_begun_walk(ctx);
    ctx->dc_request = ctx->dc_valid ? 2 : 0;
    int rc = walk_launch_async(ctx, slots, nullptr, n_sample, n_slots, 0, seed, stream, stride, /*side_stream=*/true, /*defer_join=*/true);
    if (rc != GG_OK) { ctx->dc_request = 0; return rc; }
    if (n_slots == 0) { ctx->dc_request = 0; return GG_OK; }  // (nothing was enqueued: gg_prepare_g does the whole call)
    ctx->g_begun = true;
    ctx->g_begun_args.n_slots = n_slots;
    ctx->g_begun_args.n_sample = n_sample;
    ctx->g_begun_args.seed = seed;
    ctx->g_begun_args.stream = stream;
    return GG_OK;
}

int gg_prepare_g(gg_ctx *ctx, const int32_t *slots, int32_t n_slots, int32_t n_sample, uint64_t seed, uint32_t stream,
                 int64_t *n_pairs_out, int32_t *root_status) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, n_sample >= 0, GG_EINVAL, "gg_prepare_g: n_sample < 0");
    const int stride = ctx->tree_max_depth + 3;
    ctx->g_pairs = 0;
    int rc;
    const bool adopt = ctx->g_begun && ctx->g_begun_args.n_slots == n_slots && ctx->g_begun_args.n_sample == n_sample &&
                       ctx->g_begun_args.seed == seed && ctx->g_begun_args.stream == stream && slots &&
                       ctx->h_slots_m[0].size() == (size_t)n_slots && memcmp(ctx->h_slots_m[0].data(), slots, sizeof(int32_t) * n_slots) == 0;
    if (adopt) {
        // the launch gg_prepare_g_begin enqueued IS this call's: the main stream joins it here, everything else as below
        ctx->g_begun = false;
        GG_HIP(ctx, hipSetDevice(ctx->device));
        GG_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_walk_done, 0));
    } else {
        // G walks read the generator's tables and the trees only: beside the discriminator update still in flight
        // (a begun launch with other arguments is drained and dropped inside walk_launch_async)
        discard_begun_walk(ctx);
        ctx->dc_request = ctx->dc_valid ? 2 : 0;  // G-mode walks look their distributions up in what the D launch of the step left
        rc = walk_launch_async(ctx, slots, nullptr, n_sample, n_slots, 0, seed, stream, stride, /*side_stream=*/true);
        if (rc != GG_OK) { ctx->dc_request = 0; return rc; }
    }
    const int64_t nw = ctx->w_total;
    // a path of L = len - 1 <= stride - 1 nodes gives at most 2 * window * L pairs
    const int64_t cap = nw * 2 * ctx->cfg.window_size * (stride - 1);
    if (nw > 0) {
        GG_HIP(ctx, ctx->g_cnt.reserve(sizeof(int32_t) * nw));
        GG_HIP(ctx, ctx->g_ptr.reserve(sizeof(int64_t) * (nw + 1)));
        GG_HIP(ctx, ctx->g_node1.reserve(sizeof(int32_t) * (cap + 1)));
        GG_HIP(ctx, ctx->g_node2.reserve(sizeof(int32_t) * (cap + 1)));
        GG_HIP(ctx, ctx->g_reward.reserve(sizeof(float) * (cap + 1)));
        rc = enqueue_g_pairs(ctx, nw, cap);
        if (rc != GG_OK) return rc;
    }
    bool retried = false;
    const int64_t hops_before = ctx->ctr.hops;
    rc = walk_finalize(ctx, &retried);  // the only host synchronisation of the call
    ctx->dc_request = 0;
    if (rc != GG_OK) return rc;
    if (nw > 0) {
        if (retried) {
            // (LAZY trees: slots rebuilt whole may have deepened the paths -- the launch was repeated with a longer stride)
            const int64_t cap2 = std::max<int64_t>(cap, nw * 2 * ctx->cfg.window_size * (ctx->w_stride - 1));
            GG_HIP(ctx, ctx->g_node1.reserve(sizeof(int32_t) * (cap2 + 1)));
            GG_HIP(ctx, ctx->g_node2.reserve(sizeof(int32_t) * (cap2 + 1)));
            GG_HIP(ctx, ctx->g_reward.reserve(sizeof(float) * (cap2 + 1)));
            rc = enqueue_g_pairs(ctx, nw, cap2);
            if (rc != GG_OK) return rc;
            GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        ctx->g_pairs = (int64_t)ctx->h_pin[gg_ctx::H_TOTAL];
        ctx->ctr.reward_pairs += ctx->g_pairs;
        if (ctx->walk_timed) {  // the reward kernel's time is folded in by harvest_timings
            ctx->ctr.reward_pairs_timed += ctx->g_pairs;
            ctx->ctr.g_walk_nodes_timed += ctx->ctr.hops - hops_before;
        }
        ctx->g_paths_valid = true;
    }
    if (root_status && n_slots) GG_HIP(ctx, hipMemcpy(root_status, ctx->w_status.p, sizeof(int32_t) * n_slots, hipMemcpyDeviceToHost));
    if (!ctx->in_epoch_add) {
        rc = exchange_count_max(ctx, ctx->g_pairs, &ctx->g_pairs_max);
        if (rc != GG_OK) return rc;
    }
    if (n_pairs_out) *n_pairs_out = ctx->g_pairs;
    return GG_OK;
}

int gg_get_g_data(gg_ctx *ctx, int32_t *node_1, int32_t *node_2, float *reward) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    discard_begun_walk(ctx);  // (a begun launch overwrote the walks these pairs are expanded from: g_paths_valid is false)
    const int64_t n = ctx->g_pairs;
    if (n == 0) return GG_OK;
    GG_HIP(ctx, hipSetDevice(ctx->device));
    if (node_1 || node_2) {
        int rc = ensure_g_pairs(ctx);
        if (rc != GG_OK) return rc;
    }
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (node_1) GG_HIP(ctx, hipMemcpy(node_1, ctx->g_node1.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (node_2) GG_HIP(ctx, hipMemcpy(node_2, ctx->g_node2.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (reward) GG_HIP(ctx, hipMemcpy(reward, ctx->g_reward.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    return GG_OK;
}

int gg_pair_reward(gg_ctx *ctx, const int32_t *u, const int32_t *v, int64_t n, float *out) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, n >= 0 && (n == 0 || (u && v && out)), GG_EINVAL, "gg_pair_reward: bad argument");
    if (n == 0) return GG_OK;
    for (int64_t i = 0; i < n; ++i)
        GG_CHECK(ctx, u[i] >= 0 && u[i] < ctx->n_node && v[i] >= 0 && v[i] < ctx->n_node, GG_EINVAL, "gg_pair_reward: id out of range at %lld", (long long)i);
    GG_HIP(ctx, hipSetDevice(ctx->device));
    GG_HIP(ctx, ctx->step_u.reserve(sizeof(int32_t) * n));
    GG_HIP(ctx, ctx->step_v.reserve(sizeof(int32_t) * n));
    GG_HIP(ctx, ctx->step_x.reserve(sizeof(float) * n));
    GG_HIP(ctx, hipMemcpyAsync(ctx->step_u.p, u, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, hipMemcpyAsync(ctx->step_v.p, v, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_pair_reward(ctx, ctx->step_u.as<int32_t>(), ctx->step_v.as<int32_t>(), n, ctx->step_x.as<float>());
    if (rc != GG_OK) return rc;
    GG_HIP(ctx, hipMemcpyAsync(out, ctx->step_x.p, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GG_OK;
}

}  // extern "C"

// gg_edge_scores: see include/graphgan_hip.h.
extern "C" int gg_edge_scores(gg_ctx *ctx, int32_t which, const int32_t *u, const int32_t *v, int64_t n, double *out) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, (which == 0 || which == 1) && n >= 0 && (n == 0 || (u && v && out)), GG_EINVAL, "gg_edge_scores: bad argument");
    if (n == 0) return GG_OK;
    for (int64_t i = 0; i < n; ++i)
        GG_CHECK(ctx, u[i] >= 0 && u[i] < ctx->n_node && v[i] >= 0 && v[i] < ctx->n_node, GG_EINVAL, "gg_edge_scores: id out of range at %lld", (long long)i);
    GG_HIP(ctx, hipSetDevice(ctx->device));
    GG_HIP(ctx, ctx->step_u.reserve(sizeof(int32_t) * n));
    GG_HIP(ctx, ctx->step_v.reserve(sizeof(int32_t) * n));
    GG_HIP(ctx, ctx->step_x.reserve(sizeof(double) * n));
    GG_HIP(ctx, hipMemcpyAsync(ctx->step_u.p, u, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, hipMemcpyAsync(ctx->step_v.p, v, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    int64_t blocks = std::min<int64_t>((n * 16 + 255) / 256, 4096);
    hipLaunchKernelGGL(edge_dot_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->model[which].E, ctx->ld,
                       ctx->step_u.as<int32_t>(), ctx->step_v.as<int32_t>(), n, ctx->step_x.as<double>());
    GG_HIP(ctx, hipGetLastError());
    GG_HIP(ctx, hipMemcpyAsync(out, ctx->step_x.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GG_OK;
}
