// steps.hip -- K3 d_step, K4 g_step, K5 optimizers.
//
//   d_step  <- sess.run(discriminator.d_updates) (reference src/GraphGAN/graph_gan.py:154-157,
//              src/GraphGAN/discriminator.py:21-32): L = sum_b sigmoid_CE(s_b, y_b)
//              + lambda * 1/2 (|E_v|^2 + |E_u|^2 + b_v^2);  dL/ds_b = sigmoid(s_b) - y_b
//   g_step  <- sess.run(generator.g_updates) (graph_gan.py:173-176, src/GraphGAN/generator.py:22-31):
//              L = -mean_b(log(clip(sigmoid(s_b), 1e-5, 1)) * r_b) + lambda * 1/2 (|E_v|^2 + |E_u|^2);
//              dL/ds_b = -(r_b / B) (1 - sigmoid(s_b)) inside the clip range, else 0
//   Adam    <- tf.train.AdamOptimizer(lr).minimize (generator.py:30-31, discriminator.py:31-32),
//              tensorflow==1.8.0 sparse-apply semantics: duplicate rows summed, m and v decayed
//              over ALL rows, var moved over ALL rows, lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps outside.
//
// Two kernels per step: (A) one 16-lane group per pair accumulates the dense gradient with
// fp32 atomics (rows of E_u, E_v and b_v); (B) the optimizer sweep.  GG_OPT_ADAM_DENSE sweeps
// all N*(ld+1) elements (TF1 parity; 24 B/element, L2-resident on CA-GrQc); GG_OPT_ADAM_LAZY
// and GG_OPT_SGD sweep only the rows a step touched (scale mode: 48d+24 / 16d+20 B per pair).
// With gg_comm_init the gradient accumulators are all-reduced between (A) and (B).
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "gg_internal.h"

namespace gg {

struct StepArgs {
    float *E, *b;          // variables
    float *gE, *gb;        // dense gradient accumulators
    int32_t *touched, *touched_list, *touched_cnt;
    int ld, track;
    const int32_t *u, *v;
    const float *x;        // label (D) or reward (G)
    int n;
    float lambda, inv_n;
    const int64_t *n_glob; // multi-GPU G step: pairs of ALL ranks in this step (device word) -> inv_n = 1 / *n_glob
    int is_d;
    int ppg;               // consecutive pairs handled by one 16-lane group
    unsigned long long *prof;  // GG_DET_PROFILE: 100 MHz clock stamps of the small-batch kernel (wave 0 and the last wave), else null
    // STAGED (large fused batches on one replica, see path_count_kernel): stage row of the v-side gradient of pair p, and of
    // the u-side gradient of the run that ends at pair p (-2: no run ends there); -1 = hub row, atomics as before
    const int32_t *slot_v, *slot_u;
    float *stage, *stage_b;
};

// occurrences of every table row in a fused batch, as the gradient kernel will emit them: one per pair for v, one per
// run of equal u inside a group's ppg pairs for u; the returned rank becomes the row's slot inside its stage segment.
// The rank is the atomic's ARRIVAL order -- run dependent -- so every stage row also carries a key that names its source
// (2 p + 1: v side of pair p, 2 q: the u run ending at pair q); the reducing kernel adds a segment in ascending key order.
__global__ __launch_bounds__(256) void pair_occ_count_kernel(const int32_t *u, const int32_t *v, int n, int ppg, int32_t *cnt,
                                                             int32_t *slot_v, int32_t *slot_u) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    slot_v[p] = atomicAdd(&cnt[v[p]], 1);
    const bool run_end = p == n - 1 || (p + 1) % ppg == 0 || u[p + 1] != u[p];
    slot_u[p] = run_end ? atomicAdd(&cnt[u[p]], 1) : -2;
}

__global__ __launch_bounds__(256) void pair_occ_slot_kernel(const int32_t *u, const int32_t *v, int n, const int32_t *cnt, const int32_t *off,
                                                            int T, int32_t *slot_v, int32_t *slot_u, int32_t *key) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int iv = v[p], iu = u[p];
    const int sv = cnt[iv] <= T ? off[iv] + slot_v[p] : -1;
    slot_v[p] = sv;
    if (sv >= 0) key[sv] = 2 * p + 1;
    if (slot_u[p] != -2) {
        const int su = cnt[iu] <= T ? off[iu] + slot_u[p] : -1;
        slot_u[p] = su;
        if (su >= 0) key[su] = 2 * p;
    }
}

// One 16-lane group per run of PAIRS_PER_GROUP consecutive pairs.  The reference's batches are
// contiguous slices of the prepare-order lists (a13): a D slice is one root's rows (same u for
// deg(root) rows), a G slice is the pairs of consecutive path positions (same u for 2-4 rows).
// The u-side gradient is therefore accumulated in registers while u stays the same and flushed
// with one row of atomics per run; the v side goes out per pair.  Lane t owns floats t, t+16, ...
// so that one atomic instruction of the group covers one contiguous 64-byte line.
// Large fused batches on one replica stage their gradient rows instead (STAGED, see path_count_kernel).
// Measured and not kept (round 2, D pass beside the generator's walks): a grid capped at 6 workgroups per CU with a
// grid-stride loop (the walks of the other stream start no earlier, this kernel +10 %); plain stores instead of atomics only
// for rows the batch names once (no gain over atomics: the accumulator round trip stays).
constexpr int PAIRS_PER_GROUP = 16;  // large fused batches; small (B = 64) batches use 1 pair per group: latency, not contention, rules there

template <int NF, bool STAGED>  // NF = ceil(ld / 16) floats per lane
__global__ __launch_bounds__(256) void pair_grad_kernel(const StepArgs a) {
    const int t = threadIdx.x & 15;
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int p0 = g * a.ppg;
    if (p0 >= a.n) return;
    const int p1 = min(p0 + a.ppg, a.n);
    const int nchunk = a.ld >> 2;
    const float inv_n = a.n_glob ? 1.0f / (float)(*a.n_glob) : a.inv_n;
    float accu[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) accu[i] = 0.f;
    int run_u = a.u[p0];
    auto flush_u = [&](int q) {
        const int sl = STAGED ? a.slot_u[q] : -1;
        if (STAGED && sl >= 0) {
            float *g = a.stage + (int64_t)sl * a.ld;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int f = t + 16 * i;
                if (f < a.ld) g[f] = accu[i];
            }
            if (t == 0) a.stage_b[sl] = 0.f;
            return;
        }
        float *gu = a.gE + (int64_t)run_u * a.ld;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = t + 16 * i;
            if (f < a.ld) atomicAdd(gu + f, accu[i]);
        }
        if (t == 0 && a.track) a.touched[run_u] = 1;
    };
    for (int p = p0; p < p1; ++p) {
        const int iu = a.u[p], iv = a.v[p];
        if (iu != run_u) {  // flush the finished run of u (it ended at pair p - 1)
            flush_u(p - 1);
#pragma unroll
            for (int i = 0; i < NF; ++i) accu[i] = 0.f;
            run_u = iu;
        }
        const float4 *ru = (const float4 *)(a.E + (int64_t)iu * a.ld);
        const float4 *rv = (const float4 *)(a.E + (int64_t)iv * a.ld);
        float acc = 0.f;
        for (int c = t; c < nchunk; c += 16) {
            const float4 x = ru[c], y = rv[c];
            acc = __builtin_fmaf(x.x, y.x, acc);
            acc = __builtin_fmaf(x.y, y.y, acc);
            acc = __builtin_fmaf(x.z, y.z, acc);
            acc = __builtin_fmaf(x.w, y.w, acc);
        }
        acc += __shfl_xor(acc, 8, 64);
        acc += __shfl_xor(acc, 4, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 1, 64);
        const float bv = a.b[iv];
        const float s = acc + bv;
        const float sg = 1.0f / (1.0f + expf(-s));
        float ds;
        if (a.is_d) {
            ds = sg - a.x[p];
        } else {
            const bool inside = (sg >= 1e-5f) && (sg <= 1.0f);
            ds = inside ? -(a.x[p] * inv_n) * (1.0f - sg) : 0.0f;
        }
        const float *fu = (const float *)ru, *fv = (const float *)rv;  // rows were just read: L1/L2 hits
        const int slv = STAGED ? a.slot_v[p] : -1;
        const bool st = STAGED && slv >= 0;
        float *gv = st ? a.stage + (int64_t)slv * a.ld : a.gE + (int64_t)iv * a.ld;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = t + 16 * i;
            if (f < a.ld) {
                const float x = fu[f], y = fv[f];
                accu[i] += ds * y + a.lambda * x;
                const float gval = ds * x + a.lambda * y;
                if (st) gv[f] = gval;
                else atomicAdd(gv + f, gval);
            }
        }
        if (t == 0) {
            const float gbv = a.is_d ? ds + a.lambda * bv : ds;
            if (st) {
                a.stage_b[slv] = gbv;
            } else {
                atomicAdd(a.gb + iv, gbv);
                if (a.track) a.touched[iv] = 1;  // plain flag stores; the row list is built by a scan (no contended atomics)
            }
        }
    }
    flush_u(p1 - 1);
}

// The same step for LARGE fused batches (ppg = 16: a group owns 16 consecutive pairs), rebuilt around the memory latency that
// bounded the loop above (SQ_WAIT 90 %: per pair a chain ids -> rows -> dot -> sigmoid -> stores, sixteen times in a row):
// the group's 16 (u, v, x) triples and stage slots arrive with ONE coalesced load each and live in its lanes (shuffles, no
// loads and no branches between a row prefetch and its use), and the rows of pair j + 2 are in flight -- raw, from clamped
// addresses -- while pair j is evaluated.  Arithmetic and accumulation order per row are those of pair_grad_kernel (the
// staged sums are order-free anyway; the atomic path adds the same values).
template <int NF, bool STAGED>
__global__ __launch_bounds__(256) void pair_grad16_kernel(const StepArgs a) {
    const int t = threadIdx.x & 15;
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t p0 = g * 16;
    if (p0 >= a.n) return;
    const int np = (int)min((int64_t)16, (int64_t)a.n - p0);
    const float inv_n = a.n_glob ? 1.0f / (float)(*a.n_glob) : a.inv_n;
    const bool mine = t < np;
    const int myu = mine ? a.u[p0 + t] : 0, myv = mine ? a.v[p0 + t] : 0;
    const float myx = mine ? a.x[p0 + t] : 0.f;
    const int my_sv = (STAGED && mine) ? a.slot_v[p0 + t] : -1;
    const int my_su = (STAGED && mine) ? a.slot_u[p0 + t] : -1;
    const float mybv = a.b[myv];
    // end of a run of equal centres (the u-side gradient of the run is flushed there): as pair_occ_count_kernel defines it
    const int nextu = __shfl(myu, (t + 1) & 15, 16);
    const bool my_end = mine && (t == np - 1 || nextu != myu);
    auto fetch = [&](int j, float (&ru)[NF], float (&rv)[NF]) {  // rows of pair j (clamped: pairs behind the group's end re-read pair 0)
        const int jj = j < np ? j : 0;
        const float *su = a.E + (int64_t)__shfl(myu, jj, 16) * a.ld, *sv = a.E + (int64_t)__shfl(myv, jj, 16) * a.ld;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = t + 16 * i;
            const int fc = f < a.ld ? f : a.ld - 1;
            ru[i] = su[fc];
            rv[i] = sv[fc];
        }
    };
    float U[3][NF], V[3][NF], accu[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) accu[i] = 0.f;
    fetch(0, U[0], V[0]);
    fetch(1, U[1], V[1]);
#pragma unroll 1
    for (int j0 = 0; j0 < np; j0 += 3) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {  // buffer r holds pair j0 + r; the fetch of pair j + 2 goes to buffer (r + 2) % 3
            const int j = j0 + r;
            if (j >= np) break;
            fetch(j + 2, U[(r + 2) % 3], V[(r + 2) % 3]);
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const bool in = t + 16 * i < a.ld;
                const float x = in ? U[r][i] : 0.f, y = in ? V[r][i] : 0.f;
                U[r][i] = x; V[r][i] = y;
                acc = __builtin_fmaf(x, y, acc);
            }
            acc += __shfl_xor(acc, 8, 64);
            acc += __shfl_xor(acc, 4, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 1, 64);
            const float bv = __shfl(mybv, j, 16), xj = __shfl(myx, j, 16);
            const float s = acc + bv;
            const float sg = 1.0f / (1.0f + expf(-s));
            float ds;
            if (a.is_d) {
                ds = sg - xj;
            } else {
                const bool inside = (sg >= 1e-5f) && (sg <= 1.0f);
                ds = inside ? -(xj * inv_n) * (1.0f - sg) : 0.0f;
            }
            const int iv = __shfl(myv, j, 16), iu = __shfl(myu, j, 16);
            const int slv = __shfl(my_sv, j, 16), slu = __shfl(my_su, j, 16);
            const bool st = STAGED && slv >= 0;
            float *gv = st ? a.stage + (int64_t)slv * a.ld : a.gE + (int64_t)iv * a.ld;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int f = t + 16 * i;
                if (f < a.ld) {
                    accu[i] += ds * V[r][i] + a.lambda * U[r][i];
                    const float gval = ds * U[r][i] + a.lambda * V[r][i];
                    if (st) gv[f] = gval;
                    else atomicAdd(gv + f, gval);
                }
            }
            if (t == 0) {
                const float gbv = a.is_d ? ds + a.lambda * bv : ds;
                if (st) {
                    a.stage_b[slv] = gbv;
                } else {
                    atomicAdd(a.gb + iv, gbv);
                    if (a.track) a.touched[iv] = 1;
                }
            }
            if (__shfl((int)my_end, j, 16)) {  // the run of equal centres ends at this pair
                if (STAGED && slu >= 0) {
                    float *gu = a.stage + (int64_t)slu * a.ld;
#pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        const int f = t + 16 * i;
                        if (f < a.ld) gu[f] = accu[i];
                    }
                    if (t == 0) a.stage_b[slu] = 0.f;
                } else {
                    float *gu = a.gE + (int64_t)iu * a.ld;
#pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        const int f = t + 16 * i;
                        if (f < a.ld) atomicAdd(gu + f, accu[i]);
                    }
                    if (t == 0 && a.track) a.touched[iu] = 1;
                }
#pragma unroll
                for (int i = 0; i < NF; ++i) accu[i] = 0.f;
            }
        }
    }
}

// G pass over whole walks (fast mode: one fused batch = every prepared pair, graph_gan.py:168-176 with
// batch_size >= train_size).  The pairs of a walk are the window-2 pairs of its path
// (graph_gan.py:272-291), i.e. every node row is needed by up to 8 pairs.  One 16-lane group per
// walk slides a 5-row window over the path: each row is read ONCE, each node's gradient is
// accumulated in registers and flushed ONCE (instead of 8 row reads + 8 rows of atomics).
struct PathArgs {
    float *E, *b, *gE, *gb;
    int32_t *touched, *touched_list, *touched_cnt;
    int ld, track, window;
    const int32_t *paths, *path_len;
    int stride;
    const int64_t *pair_ptr;  // [n_walks + 1] first pair of each walk
    const float *reward;
    int64_t n_walks;
    float lambda, inv_n;
    const int64_t *n_glob;  // multi-GPU: pairs of all ranks (device word), see StepArgs
    // STAGED: slot[w * stride + pos] = row of the stage buffer that receives the gradient of that path node (plain
    // stores), or -1 = the node's table row collects too many gradients for one reducing group -> atomics as before
    const int32_t *slot;
    float *stage, *stage_b;
};

// Staged generator gradient (single replica, lazy Adam / SGD).  The gradient rows of a G pass used to be ADDED to the
// accumulator table with fp32 atomics -- 116 M lane-atomics per pass at the L2's ~270 G/s, 56 % of the kernel.  Now the
// path nodes are counted per table row first (path_count_kernel: one int atomic per node, returning its rank), rows with
// <= T gradients get a contiguous segment of a stage buffer (device_segment_rows), the gradient kernel STORES each node's
// row into its segment slot, and staged_opt_kernel sums a row's segment and applies the optimizer in the same pass: no
// accumulator round trip.  The slot inside a segment is the rank the counting atomic returned, i.e. arrival order, which
// differs from run to run; the sum does not: every stage row carries the path position it came from (sg_key) and the
// reducing group adds its segment in ascending key order -- a deterministic sum for those rows.  Hub rows (> T gradients)
// keep the atomic path (run-dependent fp32 order) and the flag -> list -> sparse_opt_kernel update.
// `flag` (early launches behind the walk, enqueue_path_slots): the walk launch's status word; 2 = the launch is being rerun, its
// paths are not final -> count nothing (the rerun brings its own launch of these kernels)
__global__ __launch_bounds__(256) void path_count_kernel(const int32_t *paths, const int32_t *path_len, int stride, int64_t n_walks,
                                                         int32_t *cnt, int32_t *slot, const unsigned long long *flag) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t w = idx / stride;
    if (w >= n_walks || (flag && *flag == 2ull)) return;
    const int c = (int)(idx - w * stride), L = path_len[w] - 1;
    if (L <= 1 || c >= L) return;  // walks without pairs flush nothing
    slot[idx] = atomicAdd(&cnt[paths[idx]], 1);
}

__global__ __launch_bounds__(256) void path_slot_kernel(const int32_t *paths, const int32_t *path_len, int stride, int64_t n_walks,
                                                        const int32_t *cnt, const int32_t *off, int T, int32_t *slot, int32_t *key, const unsigned long long *flag) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t w = idx / stride;
    if (w >= n_walks || (flag && *flag == 2ull)) return;
    const int c = (int)(idx - w * stride), L = path_len[w] - 1;
    if (L <= 1 || c >= L) return;
    const int nd = paths[idx];
    const int sl = cnt[nd] <= T ? off[nd] + slot[idx] : -1;
    slot[idx] = sl;
    if (sl >= 0) key[sl] = (int32_t)idx;  // n_pos < 2^31 (run_path_step)
}

// SHORT: every path has at most 16 nodes (stride <= 17: trees of depth <= 14), so the walk's node ids, stage slots and
// rewards live in the group's lanes and are handed out with shuffles -- no load, and above all no BRANCH, sits between a
// row prefetch and its use (with branches in the loop body the compiler falls back to s_waitcnt vmcnt(0) at every join,
// which drains the prefetches it was asked to keep in flight).
template <int NF, bool STAGED, bool SHORT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NF <= 8 ? 3 : 1, 4))) void path_grad_kernel(const PathArgs a) {
    const int t = threadIdx.x & 15;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (w >= a.n_walks) return;
    const int L = a.path_len[w] - 1;  // the back-step is dropped (graph_gan.py:282)
    if (L <= 1) return;               // no pairs
    const float inv_n = a.n_glob ? 1.0f / (float)(*a.n_glob) : a.inv_n;
    const int32_t *p = a.paths + w * (int64_t)a.stride;
    int64_t pi = a.pair_ptr[w];
    const int32_t *const wslot = STAGED ? a.slot + w * (int64_t)a.stride : nullptr;
    // the walk's ids / stage slots / rewards in the lanes of its group
    const int myid = (t < L) ? p[t] : -1;
    const int myslot = (STAGED && t < L) ? wslot[t] : -1;
    const int64_t pi0 = pi;
    const int n_pairs = (int)(a.pair_ptr[w + 1] - pi0);
    float rwl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rwl[q] = (16 * q + t < n_pairs) ? a.reward[pi0 + 16 * q + t] : 0.f;
    auto id_at = [&](int pos) -> int {
        if (SHORT) {
            const int v = __shfl(myid, pos & 15, 16);
            return (pos >= 0 && pos < L) ? v : -1;
        }
        return (pos >= 0 && pos < L) ? (pos < 16 ? __shfl(myid, pos, 16) : p[pos]) : -1;
    };
    auto slot_at = [&](int pos) -> int {
        if (!STAGED) return -1;
        if (SHORT) {
            const int v = __shfl(myslot, pos & 15, 16);
            return (pos >= 0 && pos < L) ? v : -1;
        }
        return (pos >= 0 && pos < L) ? (pos < 16 ? __shfl(myslot, pos, 16) : wslot[pos]) : -1;
    };
    auto reward_at = [&](int k, int64_t pidx) -> float {  // k may differ from lane to lane: fetch, THEN select
        const float v0 = __shfl(rwl[0], k & 15, 16), v1 = __shfl(rwl[1], k & 15, 16);
        const float v2 = __shfl(rwl[2], k & 15, 16), v3 = __shfl(rwl[3], k & 15, 16);
        const float v = k < 16 ? v0 : (k < 32 ? v1 : (k < 48 ? v2 : v3));
        if (SHORT) return v;  // <= 4 * 16 - 6 pairs
        return k < 64 ? v : a.reward[pidx];
    };
    // window slots 0..2 hold path positions c, c+1, c+2 of the current centre c: a centre handles its FORWARD neighbours,
    // one dot product per unordered pair {c, c+d} serving both ordered pairs (c, c+d) [bias of c+d] and (c+d, c) [bias of c]
    float R[3][NF], A[3][NF], bv[3], gb[3], npair[3];
    int node[3], sslot[3];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        node[sl] = -1; bv[sl] = 0.f; gb[sl] = 0.f; sslot[sl] = -1; npair[sl] = 0.f;
#pragma unroll
        for (int i = 0; i < NF; ++i) { R[sl][i] = 0.f; A[sl][i] = 0.f; }
    }
    // A row is fetched RAW -- unconditional loads from a clamped address (row 0 behind the path's end, the last float for
    // the lanes behind the row's end) -- and masked only where it enters the window, an iteration later: a select next
    // to the load would make the compiler wait for the load right there.
    auto fetch_raw = [&](int pos, float (&row)[NF], int &nd_out, int &slot_out, float &b_out) {
        const int nd = id_at(pos);
        const int ndc = nd >= 0 ? nd : 0;
        b_out = a.b[ndc];
        const float *src = a.E + (int64_t)ndc * a.ld;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = t + 16 * i;
            row[i] = src[f < a.ld ? f : a.ld - 1];
        }
        nd_out = nd;
        slot_out = slot_at(pos);
    };
    auto enter = [&](int sl, const float (&row)[NF], int nd, int slot, float b) {
        node[sl] = nd; sslot[sl] = slot; gb[sl] = 0.f; npair[sl] = 0.f;
        bv[sl] = nd >= 0 ? b : 0.f;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            R[sl][i] = (nd >= 0 && t + 16 * i < a.ld) ? row[i] : 0.f;
            A[sl][i] = 0.f;
        }
    };
    auto flush = [&](int sl) {
        const int nd = node[sl];
        if (nd < 0) return;
        const float lam_n = a.lambda * npair[sl];
#pragma unroll
        for (int i = 0; i < NF; ++i) A[sl][i] = __builtin_fmaf(lam_n, R[sl][i], A[sl][i]);
        if (STAGED && sslot[sl] >= 0) {
            float *g = a.stage + (int64_t)sslot[sl] * a.ld;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int f = t + 16 * i;
                if (f < a.ld) g[f] = A[sl][i];
            }
            if (t == 0) a.stage_b[sslot[sl]] = gb[sl];
            return;
        }
        float *g = a.gE + (int64_t)nd * a.ld;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = t + 16 * i;
            if (f < a.ld) atomicAdd(g + f, A[sl][i]);
        }
        if (t == 0) {
            if (gb[sl] != 0.f) atomicAdd(a.gb + nd, gb[sl]);
            if (a.track) a.touched[nd] = 1;
        }
    };
    // Rows are fetched TWO centres ahead (positions c+3 and c+4 are in flight while centre c is evaluated): with three
    // waves per SIMD one iteration of arithmetic does not cover an HBM round trip.
    float Rq[2][NF], qbv[2];
    int qnode[2], qslot[2];
    {
        float r0[NF], r1[NF], r2[NF], b0, b1, b2;
        int n0, n1, n2, s0, s1, s2;
        fetch_raw(0, r0, n0, s0, b0);
        fetch_raw(1, r1, n1, s1, b1);
        fetch_raw(2, r2, n2, s2, b2);
        fetch_raw(3, Rq[0], qnode[0], qslot[0], qbv[0]);
        enter(0, r0, n0, s0, b0);
        enter(1, r1, n1, s1, b1);
        enter(2, r2, n2, s2, b2);
    }
    auto n_pairs_of = [&](int i) { return min(i, a.window) + min(L - 1 - i, a.window); };
    int base = 0;  // pairs of the centres before c (the walk's pairs are listed centre by centre: backward, then forward neighbours)
    for (int c = 0; c < L; ++c) {
        fetch_raw(c + 4, Rq[1], qnode[1], qslot[1], qbv[1]);
        // the scores of the centre's (up to four) ordered pairs go to lanes 0..3: one sigmoid sequence per centre
        const int back_c = min(c, a.window);
        int base_d = base + n_pairs_of(c);  // first pair of centre c + 1 (then c + 2)
        float mys = 0.f;
        int myk = -1;  // index of the lane's pair inside the walk
#pragma unroll
        for (int d = 1; d <= 2; ++d) {
            if (d <= a.window && node[d] >= 0) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < NF; ++i) acc = __builtin_fmaf(R[0][i], R[d][i], acc);
                acc += __shfl_xor(acc, 8, 64);
                acc += __shfl_xor(acc, 4, 64);
                acc += __shfl_xor(acc, 2, 64);
                acc += __shfl_xor(acc, 1, 64);
                if (t == 2 * d - 2) { mys = acc + bv[d]; myk = base + back_c + (d - 1); }               // (c, c + d)
                if (t == 2 * d - 1) { mys = acc + bv[0]; myk = base_d + (c - max(c + d - a.window, 0)); }  // (c + d, c)
            }
            base_d += n_pairs_of(c + d);
        }
        const int kk = myk >= 0 ? myk : 0;
        const float rw = reward_at(kk, pi0 + kk);  // (shuffles: every lane takes part)
        float myds = 0.f;
        if (myk >= 0) {
            const float sg = 1.0f / (1.0f + expf(-mys));
            const bool inside = (sg >= 1e-5f) && (sg <= 1.0f);
            myds = inside ? -(rw * inv_n) * (1.0f - sg) : 0.0f;
        }
#pragma unroll
        for (int d = 1; d <= 2; ++d) {
            const float ds_f = __shfl(myds, 2 * d - 2, 16), ds_r = __shfl(myds, 2 * d - 1, 16);
            if (d <= a.window && node[d] >= 0) {
                // data term now; the l2 term lambda * row once per pair the row takes part in is added at the flush
                // (npair[] counts them) -- the kernel is VALU bound
                const float dsum = ds_f + ds_r;
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    A[0][i] = __builtin_fmaf(dsum, R[d][i], A[0][i]);
                    A[d][i] = __builtin_fmaf(dsum, R[0][i], A[d][i]);
                }
                npair[0] += 2.0f;
                npair[d] += 2.0f;
                gb[d] += ds_f;
                gb[0] += ds_r;
            }
        }
        base += n_pairs_of(c);
        flush(0);  // node c has met all its neighbours
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            node[sl] = node[sl + 1]; bv[sl] = bv[sl + 1]; gb[sl] = gb[sl + 1]; sslot[sl] = sslot[sl + 1]; npair[sl] = npair[sl + 1];
#pragma unroll
            for (int i = 0; i < NF; ++i) { R[sl][i] = R[sl + 1][i]; A[sl][i] = A[sl + 1][i]; }
        }
        enter(2, Rq[0], qnode[0], qslot[0], qbv[0]);
        qnode[0] = qnode[1]; qbv[0] = qbv[1]; qslot[0] = qslot[1];
#pragma unroll
        for (int i = 0; i < NF; ++i) Rq[0][i] = Rq[1][i];
    }
}

struct OptArgs {
    float *E, *b, *mE, *vE, *mb, *vb, *gE, *gb;
    const int64_t *touched_ptr;    // exclusive scan of the touched flags, [n_node + 1]
    const int64_t *touched_total;  // = touched_ptr + n_node
    int64_t nE;   // n_node * ld
    int n_node, ld;
    float lr_t, b1, b2, eps, lr;
    int32_t *touched, *touched_list, *touched_cnt;
    // staged generator gradient: per-row counts (reset by whoever updates the row), segment offsets, the small-row list,
    // {small rows, staged rows} totals, the stage
    int32_t *sg_cnt;
    const int4 *sg_list;  // {row, first stage row, stage rows} per small row
    const int64_t *sg_tot;
    const float *stage, *stage_b;
    const int32_t *stage_key;  // source of every stage row (unique inside a pass): a segment is summed in ascending key order
    unsigned long long *bad;   // set when a kernel writes a non-finite value into this model's tables (gg_ctx::table_bad)
};

// every optimizer kernel tests what it writes: a diverged table is reported by the next walk launch (walk_reset_kernel copies the
// generator's word into the launch's error flags) even when the walks reach the row only through hops that score nothing (a
// distribution with ONE candidate, a leaf's back-step)
__device__ __forceinline__ bool nonfinite4(const float4 &x) { return !(__builtin_isfinite(x.x) && __builtin_isfinite(x.y) && __builtin_isfinite(x.z) && __builtin_isfinite(x.w)); }

__device__ __forceinline__ void adam_elem(float &var, float &m, float &v, float g, const OptArgs &a) {
    m = m * a.b1;                          // m_t = assign(m, m * beta1)
    m = m + (1.0f - a.b1) * g;             // scatter_add(m, idx, (1 - beta1) * grad)
    v = v * a.b2;
    v = v + (g * g) * (1.0f - a.b2);
    var = var - (a.lr_t * m) / (sqrtf(v) + a.eps);
}

// Dense TF1 Adam over E (float4) and b; clears the gradient accumulators it consumed.
__global__ __launch_bounds__(256) void adam_dense_kernel(const OptArgs a) {
    const int64_t n4 = a.nE >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 var = ((float4 *)a.E)[i], m = ((float4 *)a.mE)[i], v = ((float4 *)a.vE)[i];
        const float4 g = ((const float4 *)a.gE)[i];
        adam_elem(var.x, m.x, v.x, g.x, a);
        adam_elem(var.y, m.y, v.y, g.y, a);
        adam_elem(var.z, m.z, v.z, g.z, a);
        adam_elem(var.w, m.w, v.w, g.w, a);
        bad |= nonfinite4(var);
        ((float4 *)a.E)[i] = var;
        ((float4 *)a.mE)[i] = m;
        ((float4 *)a.vE)[i] = v;
        if (g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f) ((float4 *)a.gE)[i] = z;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_node; i += stride) {
        float var = a.b[i], m = a.mb[i], v = a.vb[i];
        const float g = a.gb[i];
        adam_elem(var, m, v, g, a);
        bad |= !__builtin_isfinite(var);
        a.b[i] = var;
        a.mb[i] = m;
        a.vb[i] = v;
        if (g != 0.f) a.gb[i] = 0.f;
    }
    if (bad) *a.bad = 1ull;
}

// Lazy Adam / SGD on one touched row by one 16-lane group (clears the row's gradient and flag).
template <int SGD>
__device__ __forceinline__ void opt_row(const OptArgs &a, int row, int t, int nchunk) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int64_t o = ((int64_t)row * a.ld) >> 2;
        for (int c = t; c < nchunk; c += 16) {
            float4 var = ((float4 *)a.E)[o + c];
            const float4 g = ((const float4 *)a.gE)[o + c];
            if (SGD) {
                var.x -= a.lr * g.x; var.y -= a.lr * g.y; var.z -= a.lr * g.z; var.w -= a.lr * g.w;
            } else {
                float4 m = ((float4 *)a.mE)[o + c], v = ((float4 *)a.vE)[o + c];
                adam_elem(var.x, m.x, v.x, g.x, a);
                adam_elem(var.y, m.y, v.y, g.y, a);
                adam_elem(var.z, m.z, v.z, g.z, a);
                adam_elem(var.w, m.w, v.w, g.w, a);
                ((float4 *)a.mE)[o + c] = m;
                ((float4 *)a.vE)[o + c] = v;
            }
            if (nonfinite4(var)) *a.bad = 1ull;
            ((float4 *)a.E)[o + c] = var;
            ((float4 *)a.gE)[o + c] = z;
        }
        if (t == 0) {
            float var = a.b[row];
            const float g = a.gb[row];
            if (SGD) {
                var -= a.lr * g;
            } else {
                float m = a.mb[row], v = a.vb[row];
                adam_elem(var, m, v, g, a);
                a.mb[row] = m;
                a.vb[row] = v;
            }
            if (!__builtin_isfinite(var)) *a.bad = 1ull;
            a.b[row] = var;
            a.gb[row] = 0.f;
            a.touched[row] = 0;
            if (a.sg_cnt) a.sg_cnt[row] = 0;
        }
    }
}

// The reference's small batches (n <= DET_MAX_PAIRS, i.e. every step of the default schedule: batch 64,
// graph_gan.py:149-157,168-176) -- ONE workgroup of 64 sixteen-lane groups, NO atomics, rows resident in LDS:
//   phase 1  group p loads the two rows of pair p ONCE (global -> registers -> LDS), takes the dot product, the pair's
//            coefficient dL/ds (discriminator.py:21-30 / generator.py:22-29) goes to LDS;
//   phase 2  every distinct table row of the batch is owned by the first slot that names it (slots = [u_0..u_{n-1},
//            v_0..v_{n-1}]); its owner sums the row's contributions in ascending slot order out of LDS and STORES the
//            gradient row.
// Two runs of one build therefore give bit-identical tables (the fp32 atomics of pair_grad_kernel add in scheduling
// order), which is what lets the end-to-end tests compare whole schedules per seed, and the one dependent round of
// global loads makes it no slower than the atomic kernel.  Default for these batches; GG_DETERMINISTIC=0 selects the
// atomic kernel.  When 2 n rows do not fit the LDS (n > 64 with wide rows) the partner rows are re-read from the table.
// OPT != 0 (one replica, lazy Adam / SGD, rows in LDS): the owner of a row applies the optimizer to it right there, with the
// gradient in its registers -- ONE launch per step of the strict schedule at scale (every table read of the step happened
// in phase 1, before the barrier, so writing the table in phase 2 races with nothing); the per-element operation sequence
// is opt_row's, i.e. the same bits as gradient kernel + flag compaction + sparse_opt_kernel.
constexpr int DET_MAX_PAIRS = 256;
constexpr int DET_THREADS = 1024;
constexpr int DET_GROUPS = DET_THREADS / 16;
constexpr size_t DET_LDS_ROW_BYTES = 144 * 1024;  // of the CU's 160 KB

__device__ __forceinline__ unsigned group16_ballot(bool pred) {  // the 16 predicate bits of this lane's row of 16 lanes
    const uint64_t m = __ballot(pred);
    return (unsigned)(m >> (threadIdx.x & 48)) & 0xffffu;
}

// (What the kernel's time is made of -- it is ONE chain on ONE compute unit: pair ids -> the two rows -> coefficient -> barrier ->
// owners' sums -> stores.  The first version read the ids through LDS in front of the row loads, tested ownership and collected a
// row's contributions with one LDS read per step of a loop, and added the contributions one at a time, each behind its own LDS
// round trip: 10.8 us per launch against 5.9 for the atomic kernel -- a discriminator batch names its centre 64 times.  Now the
// groups take their pair's ids straight from global memory, the slot ids sit in registers (ownership and the contribution masks
// are ballots over them), and a row is summed by a whole wavefront, a lane per feature, four contributions at a time: see phase 2.)
// SMALLN: n <= 64 (the reference's batch): the 128 slot ids are two registers per lane; up to 256 pairs: eight.
template <int NF, bool ROWS_IN_LDS, int OPT, bool SMALLN>  // OPT: 0 = store the gradient rows, 1 = lazy Adam, 2 = SGD
__global__ __launch_bounds__(DET_THREADS) void pair_grad_det_kernel(const StepArgs a, const OptArgs o) {
    static_assert(OPT == 0 || ROWS_IN_LDS, "the fused update needs every table read in front of the barrier");
    extern __shared__ float det_rows[];  // [2 n][ld] when ROWS_IN_LDS
    __shared__ int32_t ids[2 * DET_MAX_PAIRS];
    __shared__ float coef[DET_MAX_PAIRS], coefb[DET_MAX_PAIRS];
    const int t = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int n = a.n, ld = a.ld, nchunk = ld >> 2;
    const float inv_n = a.n_glob ? 1.0f / (float)(*a.n_glob) : a.inv_n;
    const bool stamp = a.prof && blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == DET_THREADS - 64);
    unsigned long long *const pw = a.prof + (threadIdx.x ? 8 : 0);
    if (stamp) pw[0] = wall_clock64();
    for (int s = threadIdx.x; s < 2 * n; s += DET_THREADS) ids[s] = s < n ? a.u[s] : a.v[s - n];
    for (int p = g; p < n; p += DET_GROUPS) {
        const int iu = a.u[p], iv = a.v[p];  // (the group's own pair: not through the LDS copy, which is for phase 2)
        const float xp = a.x[p];
        const float bv = a.b[iv];
        const float4 *ru = (const float4 *)(a.E + (int64_t)iu * ld);
        const float4 *rv = (const float4 *)(a.E + (int64_t)iv * ld);
        float acc = 0.f;
        for (int c = t; c < nchunk; c += 16) {
            const float4 x = ru[c], y = rv[c];
            if (ROWS_IN_LDS) {
                ((float4 *)(det_rows + (size_t)p * ld))[c] = x;
                ((float4 *)(det_rows + (size_t)(n + p) * ld))[c] = y;
            }
            acc = __builtin_fmaf(x.x, y.x, acc);
            acc = __builtin_fmaf(x.y, y.y, acc);
            acc = __builtin_fmaf(x.z, y.z, acc);
            acc = __builtin_fmaf(x.w, y.w, acc);
        }
        acc += __shfl_xor(acc, 8, 64);
        acc += __shfl_xor(acc, 4, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 1, 64);
        const float sc = acc + bv;
        const float sg = 1.0f / (1.0f + expf(-sc));
        float ds;
        if (a.is_d) {
            ds = sg - xp;
        } else {
            const bool inside = (sg >= 1e-5f) && (sg <= 1.0f);
            ds = inside ? -(xp * inv_n) * (1.0f - sg) : 0.0f;
        }
        if (t == 0) {
            coef[p] = ds;
            coefb[p] = a.is_d ? ds + a.lambda * bv : ds;
        }
    }
    if (stamp) pw[1] = wall_clock64();
    __syncthreads();
    if (stamp) pw[2] = wall_clock64();
    // ---- phase 2: ONE WAVEFRONT per slot (wave w: slots w, w + 16, ...), everything about the slot wave-uniform (scalar code);
    // lane L owns the features L, L + 64, ...  The first slot that names a row owns it and adds the row's contributions in
    // ascending slot order, CB at a time (their LDS reads in flight together): the fp32 operation sequence per element is the
    // same whatever the hardware does.
    // (Measured on the way here, CA-GrQc batches: a 16-lane group per slot, the other three groups of its wavefront masked off while
    // it summed a centre row named by 20 - 64 pairs: 5.4 us of the kernel's 9.4 -- a wave64 instruction takes its issue cycles
    // however few lanes are alive; the four groups sharing a row's contributions and merging partial sums with 4 NF + 4 permutes: 2.9 us.)
    constexpr int NW = SMALLN ? 2 : 2 * DET_MAX_PAIRS / 64;  // 64-slot words of the id list
    constexpr int NFW = (NF + 3) / 4;                          // features per lane: ceil(ld / 64)
    constexpr int CB = 4;                                      // contributions fetched together (8: measured slower -- single-contribution rows pay for the empty reads)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int idr[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) idr[j] = (64 * j + lane < 2 * n) ? ids[64 * j + lane] : -1;
    // is slot s the first that names its row?  (row id and the row's slot masks come back with the answer: all wave-uniform)
    auto leader_of = [&](const int s, int &r, uint64_t (&m)[NW]) -> bool {
        // (slot s sits in lane s & 63 of register word s >> 6: a lane read, not an LDS round trip per slot)
        r = __builtin_amdgcn_readlane(idr[0], s & 63);
#pragma unroll
        for (int j = 1; j < NW; ++j) r = (s >> 6) == j ? __builtin_amdgcn_readlane(idr[j], s & 63) : r;
        bool earlier = false;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            m[j] = __ballot(idr[j] == r);  // bit = slot: the slots that name row r
            if (64 * j + 63 < s) earlier |= m[j] != 0ull;
            else if (64 * j <= s) earlier |= (m[j] & ((1ull << (s & 63)) - 1ull)) != 0ull;
        }
        return !earlier;
    };
    // the owner's work for row r (first named by slot s): the sum, then the gradient store (OPT == 0) or the optimizer.  PRE: the
    // row's moments and bias words were requested earlier (pm / pv per feature; pb = {bias, its m, its v} in lane 0)
    auto own_row = [&](const int s, const int r, const uint64_t (&m)[NW], const bool PRE, const float (&pm)[NFW], const float (&pv)[NFW], const float (&pb)[3]) {
        float own[NFW], lown[NFW], acc[NFW], accb = 0.f;
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            const int f = lane + 64 * i;
            own[i] = f < ld ? (ROWS_IN_LDS ? det_rows[(size_t)s * ld + f] : a.E[(int64_t)r * ld + f]) : 0.f;
            lown[i] = a.lambda * own[i];  // the l2 term of one occurrence
            acc[i] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            uint64_t cur = m[j];  // (no slot before s names r: the words in front of s are empty)
            while (cur) {
                int sq[CB];
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    if (cur) { sq[k] = 64 * j + __builtin_ctzll(cur); cur &= cur - 1ull; }
                    else sq[k] = -1;
                }
                float c[CB], cb[CB], pr[CB][NFW];
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    const int sk = sq[k] >= 0 ? sq[k] : s;
                    const int p = sk < n ? sk : sk - n;
                    const int ps = sk < n ? n + p : p;  // the partner's slot
                    c[k] = coef[p];
                    cb[k] = coefb[p];
                    const float *prow = ROWS_IN_LDS ? det_rows + (size_t)ps * ld : a.E + (int64_t)ids[ps] * ld;
#pragma unroll
                    for (int i = 0; i < NFW; ++i) {
                        const int f = lane + 64 * i;
                        pr[k][i] = f < ld ? prow[f] : 0.f;
                    }
                }
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    if (sq[k] >= 0) {  // (uniform)
#pragma unroll
                        for (int i = 0; i < NFW; ++i) acc[i] += c[k] * pr[k][i] + lown[i];
                        if (sq[k] >= n) accb += cb[k];
                    }
                }
            }
        }
        if (OPT == 0) {
            float *gr = a.gE + (int64_t)r * ld;
#pragma unroll
            for (int i = 0; i < NFW; ++i) {
                const int f = lane + 64 * i;
                if (f < ld) gr[f] = acc[i];
            }
            if (lane == 0) {
                a.gb[r] = accb;
                if (a.track) a.touched[r] = 1;
            }
        } else {
            const int64_t ro = (int64_t)r * ld;
#pragma unroll
            for (int i = 0; i < NFW; ++i) {
                const int f = lane + 64 * i;
                if (f < ld) {
                    float var = own[i];
                    if (OPT == 2) {
                        var -= o.lr * acc[i];
                    } else {
                        float m1 = PRE ? pm[i] : o.mE[ro + f], v1 = PRE ? pv[i] : o.vE[ro + f];
                        adam_elem(var, m1, v1, acc[i], o);
                        o.mE[ro + f] = m1;
                        o.vE[ro + f] = v1;
                    }
                    if (!__builtin_isfinite(var)) *o.bad = 1ull;
                    o.E[ro + f] = var;
                }
            }
            if (lane == 0) {
                float var = PRE ? pb[0] : o.b[r];
                if (OPT == 2) {
                    var -= o.lr * accb;
                } else {
                    float m1 = PRE ? pb[1] : o.mb[r], v1 = PRE ? pb[2] : o.vb[r];
                    adam_elem(var, m1, v1, accb, o);
                    o.mb[r] = m1;
                    o.vb[r] = v1;
                }
                if (!__builtin_isfinite(var)) *o.bad = 1ull;
                o.b[r] = var;
            }
        }
    };
    if constexpr (OPT != 0 && SMALLN && NFW <= 2) {
        // The fused variants read a row's moments (lazy Adam) and bias words from the table behind the sum: one global round trip
        // per owner round, eight rounds per wavefront one after the other (20 / 14 us per D / G step at d = 128 on the 1M graph).
        // So the wavefront FIRST decides which of its next four slots own a row and requests all their words, THEN sums and updates.
        constexpr int SPW = 2 * 64 / (DET_THREADS / 64);  // slots per wavefront: 8 ...
        constexpr int SPH = SPW / 2;                       // ... requested and processed four at a time: all eight at once measured the same and
                                                           // sat at the edge of the register allocator (125 VGPRs; 650+ bytes of scratch per lane with three more live values)
#pragma clang loop unroll(full)
        for (int h0 = 0; h0 < SPW; h0 += SPH) {
            float pm[SPH][NFW], pv[SPH][NFW], pb[SPH][3];  // (the k loops MUST unroll: an index that survives puts these arrays into scratch memory)
#pragma clang loop unroll(full)
            for (int k = 0; k < SPH; ++k) {
                const int s = wave + (h0 + k) * (DET_THREADS / 64);
                int r = 0;
                uint64_t m[NW];
#pragma unroll
                for (int i = 0; i < NFW; ++i) { pm[k][i] = 0.f; pv[k][i] = 0.f; }
                pb[k][0] = pb[k][1] = pb[k][2] = 0.f;
                if (s < 2 * n && leader_of(s, r, m)) {
                    const int64_t ro = (int64_t)r * ld;
#pragma unroll
                    for (int i = 0; i < NFW; ++i) {
                        const int f = lane + 64 * i;
                        if (OPT == 1 && f < ld) { pm[k][i] = o.mE[ro + f]; pv[k][i] = o.vE[ro + f]; }
                    }
                    if (lane == 0) {
                        pb[k][0] = o.b[r];
                        if (OPT == 1) { pb[k][1] = o.mb[r]; pb[k][2] = o.vb[r]; }
                    }
                }
            }
#pragma clang loop unroll(full)
            for (int k = 0; k < SPH; ++k) {
                const int s = wave + (h0 + k) * (DET_THREADS / 64);
                int r = 0;
                uint64_t m[NW];
                if (s < 2 * n && leader_of(s, r, m)) own_row(s, r, m, true, pm[k], pv[k], pb[k]);
            }
        }
    } else {
        // OPT == 0: the kernel only READS the table, so several workgroups may run it side by side -- each redoes phase 1 for itself
        // (128 rows out of the L2: no more latency than one workgroup's) and takes every gridDim.x-th round of slots in phase 2: a
        // wavefront's chain of owner rounds shrinks from eight to one without any synchronisation between workgroups.  (The fused
        // variants write the table in phase 2 and stay one workgroup: another workgroup could still be reading those rows.)
        const float none[NFW] = {}, none3[3] = {};
        for (int s = (int)blockIdx.x * (DET_THREADS / 64) + wave; s < 2 * n; s += (int)gridDim.x * (DET_THREADS / 64)) {
            int r = 0;
            uint64_t m[NW];
            if (leader_of(s, r, m)) own_row(s, r, m, false, none, none, none3);
        }
    }
    if (stamp) pw[3] = wall_clock64();  // (one clock read per phase: s_memrealtime is itself a slow scalar memory operation -- a read per owner round measured itself)
}

// workgroups of the table-read-only variant: one round of slots (16 wavefronts) each, at most 8
static unsigned det_workgroups(int n) {
    static const int cap = [] { const char *e = getenv("GG_DET_WORKGROUPS"); return e ? std::max(1, atoi(e)) : 8; }();
    return (unsigned)std::min(cap, std::max(1, (2 * n + DET_THREADS / 64 - 1) / (DET_THREADS / 64)));
}

// Whether this device grants a workgroup DET_LDS_ROW_BYTES of dynamic LDS (gfx950: yes).  Asked once per device with the call
// that raises the limit; a device that refuses (an ARCH override with 64 KB of LDS) takes the variant with the rows in L2.
static bool det_lds_granted(int device);
static bool det_rows_fit_lds(const gg_ctx *ctx, const StepArgs &s) {
    return (size_t)2 * s.n * s.ld * sizeof(float) <= DET_LDS_ROW_BYTES && det_lds_granted(ctx->device);
}

template <int NF, int OPT>
static hipError_t launch_pair_grad_det_lds(gg_ctx *ctx, const StepArgs &s, const OptArgs &o) {
    static bool raised_dev[64] = {};  // (the attribute belongs to the function on a device, not to a context)
    bool &raised = raised_dev[ctx->device & 63];
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void *)pair_grad_det_kernel<NF, true, OPT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DET_LDS_ROW_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)pair_grad_det_kernel<NF, true, OPT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DET_LDS_ROW_BYTES);
        if (e != hipSuccess) return e;
        raised = true;
    }
    const size_t dyn = (size_t)2 * s.n * s.ld * sizeof(float);
    const dim3 grid(OPT == 0 ? det_workgroups(s.n) : 1);
    if (s.n <= 64) hipLaunchKernelGGL((pair_grad_det_kernel<NF, true, OPT, true>), grid, dim3(DET_THREADS), dyn, ctx->stream, s, o);
    else hipLaunchKernelGGL((pair_grad_det_kernel<NF, true, OPT, false>), grid, dim3(DET_THREADS), dyn, ctx->stream, s, o);
    return hipSuccess;
}

static bool det_lds_granted(int device) {
    static int state[64] = {};  // 0 = not asked, 1 = granted, -1 = refused
    int &s = state[device & 63];
    if (s == 0) {
        const hipError_t e = hipFuncSetAttribute((const void *)pair_grad_det_kernel<4, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DET_LDS_ROW_BYTES);
        if (e != hipSuccess) (void)hipGetLastError();
        s = e == hipSuccess ? 1 : -1;
    }
    return s > 0;
}

// opt: 0 = gradient rows to the accumulators (the optimizer kernels follow), 1 / 2 = lazy Adam / SGD applied by the row owners
template <int NF>
static hipError_t launch_pair_grad_det(gg_ctx *ctx, const StepArgs &s, const OptArgs &o, int opt) {
    if (!det_rows_fit_lds(ctx, s)) {
        const dim3 grid(det_workgroups(s.n));
        if (s.n <= 64) hipLaunchKernelGGL((pair_grad_det_kernel<NF, false, 0, true>), grid, dim3(DET_THREADS), 0, ctx->stream, s, o);
        else hipLaunchKernelGGL((pair_grad_det_kernel<NF, false, 0, false>), grid, dim3(DET_THREADS), 0, ctx->stream, s, o);
        return hipSuccess;
    }
    if (opt == 1) return launch_pair_grad_det_lds<NF, 1>(ctx, s, o);
    if (opt == 2) return launch_pair_grad_det_lds<NF, 2>(ctx, s, o);
    return launch_pair_grad_det_lds<NF, 0>(ctx, s, o);
}

// ... over the compacted row list (flag array -> scan -> list: deterministic row order; the replica exchange packs the same list).
// (Measured and not kept: updating straight from the flags without the scan / compaction launches -- 30 us SLOWER per call:
// the list gives every group exactly one row; and summing one root's walk gradients in LDS before the global atomics --
// 80 us slower per G pass: the 20 walks of a root share too few rows below its first level.  Bound on what a sort + segmented
// reduce could buy: path_grad_kernel with plain stores in place of its fp32 atomics (wrong sums, timing only) runs 319 us
// against 728 us -- a staged gradient (319 us) + key sort + a reducing optimizer over 470 MB of staged rows would land
// within ~30 % of today's 728 + 233 us, for a deterministic sum; not built.)
template <int SGD>
__global__ __launch_bounds__(256) void sparse_opt_kernel(const OptArgs a) {
    const int t = threadIdx.x & 15;
    const int g0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int ng = (gridDim.x * blockDim.x) >> 4;
    const int cnt = (int)*a.touched_total;
    const int nchunk = a.ld >> 2;
    for (int r = g0; r < cnt; r += ng) opt_row<SGD>(a, a.touched_list[r], t, nchunk);
}

// Small rows of the staged gradient: one 16-lane group per row sums the row's segment of the stage buffer (contiguous
// 512-byte rows, four in flight) and applies lazy Adam (MODE 0) / SGD (MODE 1) with the sum in registers.
// ORDER OF THE SUM: the position of a gradient row inside its segment is the arrival order of the counting atomics
// (path_count_kernel / pair_occ_count_kernel) and changes from run to run, so the group first ranks the segment's source
// keys (stage_key: path position resp. pair side, unique inside a pass) -- segments of <= STAGE_SORT rows by counting
// smaller keys with shuffles (16 shuffles per 16 keys), the permutation goes through 256 B of LDS per group; longer
// segments (only with GG_STAGE_T > 64) by repeated minimum selection -- and adds the rows in ASCENDING KEY ORDER: the same
// fp32 operation sequence in every run, whatever the scheduling was.
// MODE 2 (replicas): the sum goes to the gradient accumulators instead -- plain stores, the row's flag set -- and the
// exchange + sparse_opt_kernel path takes it from there like an atomically accumulated gradient.
constexpr int STAGE_SORT = 64;

template <int MODE>
__global__ __launch_bounds__(256) void staged_opt_kernel(const OptArgs a) {
    constexpr bool SGD = MODE == 1;
    __shared__ int32_t perm_s[16][STAGE_SORT];
    const int t = threadIdx.x & 15;
    int32_t *const perm = perm_s[threadIdx.x >> 4];
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int64_t n_rows = a.sg_tot[0];
    const int nchunk = a.ld >> 2;  // <= 64: up to 4 float4 chunks per lane
    for (int64_t r = g0; r < n_rows; r += ng) {
        const int4 e = a.sg_list[r];
        const int row = e.x, n = e.z;
        const float4 *const seg = (const float4 *)(a.stage + (int64_t)e.y * a.ld);
        const float *const sb = a.stage_b + e.y;
        const int32_t *const key = a.stage_key + e.y;
        float4 g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        float gbias = 0.f;
        auto add_row = [&](const float4 *src) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = t + 16 * i;
                if (c < nchunk) {
                    const float4 x = src[c];
                    g[i].x += x.x; g[i].y += x.y; g[i].z += x.z; g[i].w += x.w;
                }
            }
        };
        if (n <= STAGE_SORT) {
            if (n > 1) {  // rank of every key = number of smaller keys of the segment
                int k[4], rk[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    k[i] = (t + 16 * i < n) ? key[t + 16 * i] : 0x7fffffff;
                    rk[i] = 0;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (16 * j < n) {
                        const int nl = min(n - 16 * j, 16);
                        for (int l = 0; l < nl; ++l) {
                            const int kk = __shfl(k[j], l, 16);
#pragma unroll
                            for (int i = 0; i < 4; ++i) rk[i] += kk < k[i];
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (t + 16 * i < n) perm[rk[i]] = t + 16 * i;
            } else if (t == 0) {
                perm[0] = 0;
            }
            // the group's lanes sit in one wavefront: its LDS operations execute in program order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // four stage rows in flight, added in key order
            int o = 0;
            for (; o + 4 <= n; o += 4) {
                const float4 *s0 = seg + (int64_t)perm[o] * nchunk, *s1 = seg + (int64_t)perm[o + 1] * nchunk;
                const float4 *s2 = seg + (int64_t)perm[o + 2] * nchunk, *s3 = seg + (int64_t)perm[o + 3] * nchunk;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = t + 16 * i;
                    if (c < nchunk) {
                        const float4 x0 = s0[c], x1 = s1[c], x2 = s2[c], x3 = s3[c];
                        g[i].x += x0.x; g[i].y += x0.y; g[i].z += x0.z; g[i].w += x0.w;
                        g[i].x += x1.x; g[i].y += x1.y; g[i].z += x1.z; g[i].w += x1.w;
                        g[i].x += x2.x; g[i].y += x2.y; g[i].z += x2.z; g[i].w += x2.w;
                        g[i].x += x3.x; g[i].y += x3.y; g[i].z += x3.z; g[i].w += x3.w;
                    }
                }
            }
            for (; o < n; ++o) add_row(seg + (int64_t)perm[o] * nchunk);
            for (int q = t; q < n; q += 16) gbias += sb[perm[q]];  // lane t: ranks t, t + 16, ... in order, then the butterfly
            __builtin_amdgcn_wave_barrier();  // the next row's permutation is written behind these reads
        } else {
            int last = -1;  // keys are >= 0
            for (int q = 0; q < n; ++q) {
                int best = 0x7fffffff, besto = 0;
                for (int o = t; o < n; o += 16) {
                    const int kk = key[o];
                    if (kk > last && kk < best) { best = kk; besto = o; }
                }
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) {
                    const int ob = __shfl_xor(best, m, 16), oo = __shfl_xor(besto, m, 16);
                    if (ob < best) { best = ob; besto = oo; }
                }
                add_row(seg + (int64_t)besto * nchunk);
                if ((q & 15) == t) gbias += sb[besto];
                last = best;
            }
        }
        gbias += __shfl_xor(gbias, 8, 64);
        gbias += __shfl_xor(gbias, 4, 64);
        gbias += __shfl_xor(gbias, 2, 64);
        gbias += __shfl_xor(gbias, 1, 64);
        const int64_t o4 = ((int64_t)row * a.ld) >> 2;
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = t + 16 * i;
                if (c < nchunk) ((float4 *)a.gE)[o4 + c] = g[i];
            }
            if (t == 0) {
                a.gb[row] = gbias;
                a.touched[row] = 1;
                a.sg_cnt[row] = 0;
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = t + 16 * i;
            if (c >= nchunk) continue;
            float4 var = ((float4 *)a.E)[o4 + c];
            if (SGD) {
                var.x -= a.lr * g[i].x; var.y -= a.lr * g[i].y; var.z -= a.lr * g[i].z; var.w -= a.lr * g[i].w;
            } else {
                float4 m = ((float4 *)a.mE)[o4 + c], v = ((float4 *)a.vE)[o4 + c];
                adam_elem(var.x, m.x, v.x, g[i].x, a);
                adam_elem(var.y, m.y, v.y, g[i].y, a);
                adam_elem(var.z, m.z, v.z, g[i].z, a);
                adam_elem(var.w, m.w, v.w, g[i].w, a);
                ((float4 *)a.mE)[o4 + c] = m;
                ((float4 *)a.vE)[o4 + c] = v;
            }
            if (nonfinite4(var)) *a.bad = 1ull;
            ((float4 *)a.E)[o4 + c] = var;
        }
        if (t == 0) {
            float var = a.b[row];
            if (SGD) {
                var -= a.lr * gbias;
            } else {
                float m = a.mb[row], v = a.vb[row];
                adam_elem(var, m, v, gbias, a);
                a.mb[row] = m;
                a.vb[row] = v;
            }
            if (!__builtin_isfinite(var)) *a.bad = 1ull;
            a.b[row] = var;
            a.sg_cnt[row] = 0;
        }
    }
}

__global__ void add_word_kernel(int64_t *dst, const int64_t *src) { *dst += *src; }

// ---- sparse gradient exchange between replicas (lazy / sgd modes).  A step touches a small part of
// the tables, so instead of all-reducing N*(ld+1) floats each rank packs its touched rows
// {row id, gradient row, bias gradient}, the packs are all-gathered, and every rank adds the packs
// in RANK ORDER with one launch per source rank (ids are unique inside a pack -> plain adds, no
// atomics): all replicas compute bit-identical sums, so they stay bit-identical.
// The pack has a fixed CAPACITY known to every rank's host before the step (2 * the most pairs any rank can bring:
// every pair touches at most two rows), so the collectives that move it need no row count from the device: entries
// behind this rank's rows carry id -1 and are skipped by the receivers.  `err` is raised if the rows do not fit
// (impossible while the capacity rule holds; checked at the next host synchronisation).
__global__ __launch_bounds__(256) void pack_rows_kernel(float *gE, float *gb, int32_t *touched, const int32_t *list, const int64_t *cnt_ptr,
                                                        int64_t cap, int ld, int32_t *out_ids, float *out_rows, int32_t *err) {
    const int t = threadIdx.x & 15;
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int64_t cnt = *cnt_ptr;
    if (cnt > cap && blockIdx.x == 0 && threadIdx.x == 0) *err = 1;
    for (int64_t r = g0; r < cap; r += ng) {
        if (r >= cnt) {
            if (t == 0) out_ids[r] = -1;
            continue;
        }
        const int row = list[r];
        float *src = gE + (int64_t)row * ld, *dst = out_rows + r * (ld + 1);
        for (int f = t; f < ld; f += 16) { dst[f] = src[f]; src[f] = 0.f; }
        if (t == 0) { dst[ld] = gb[row]; gb[row] = 0.f; out_ids[r] = row; touched[row] = 0; }
    }
}

// Exchange rows as fp32 words ((ld + 1) per row: gradient row + bias gradient) or, GG_COMM_BF16=1, as bf16 halves ((ld + 2) / 2
// words per row: ld is a multiple of 4; one half of padding): half the bytes on xGMI.  Sums are ALWAYS accumulated in fp32.
__device__ __forceinline__ uint16_t f32_to_bf16(float x) {  // round to nearest even; NaN stays NaN
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
static inline size_t pack_row_words(int ld, bool bf16) { return bf16 ? (size_t)(ld + 2) / 2 : (size_t)ld + 1; }
template <bool BF16> __device__ __forceinline__ float pack_get(const float *rows, int64_t r, int ld, int f) {
    if (BF16) return bf16_to_f32(((const uint16_t *)rows)[r * (ld + 2) + f]);
    return rows[r * (ld + 1) + f];
}
template <bool BF16> __device__ __forceinline__ void pack_put(float *rows, int64_t r, int ld, int f, float v) {
    if (BF16) ((uint16_t *)rows)[r * (ld + 2) + f] = f32_to_bf16(v);
    else rows[r * (ld + 1) + f] = v;
}

template <bool BF16 = false>
__global__ __launch_bounds__(256) void add_rows_kernel(float *gE, float *gb, int32_t *touched, const int32_t *ids, const float *rows, int64_t cnt,
                                                       int ld) {
    const int t = threadIdx.x & 15;
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
    for (int64_t r = g0; r < cnt; r += ng) {
        const int row = ids[r];
        if (row < 0) continue;  // padding behind the source rank's rows
        float *dst = gE + (int64_t)row * ld;
        for (int f = t; f < ld; f += 16) dst[f] += pack_get<BF16>(rows, r, ld, f);
        if (t == 0) { gb[row] += pack_get<BF16>(rows, r, ld, ld); touched[row] = 1; }
    }
}

// (touched flags -> row list in row order, deterministic: device_compact_flags, prepare.hip)

int apply_optimizer(gg_ctx *ctx, int which, int64_t n);
void step_done(gg_ctx *ctx, int which, int64_t n);
int run_path_step(gg_ctx *ctx);

static OptArgs make_opt_args(gg_ctx *ctx, int which) {
    Model &M = ctx->model[which];
    OptArgs o{};
    o.E = M.E; o.b = M.b; o.mE = M.mE; o.vE = M.vE; o.mb = M.mb; o.vb = M.vb; o.gE = ctx->gradE; o.gb = ctx->gradb;
    o.nE = (int64_t)ctx->n_node * ctx->ld;
    o.n_node = ctx->n_node; o.ld = ctx->ld;
    o.b1 = ctx->cfg.adam_beta1; o.b2 = ctx->cfg.adam_beta2; o.eps = ctx->cfg.adam_eps;
    o.lr = M.lr;
    // lr_t = lr * sqrt(1 - beta2_power) / (1 - beta1_power), all fp32 (TF keeps the powers as fp32 variables)
    o.lr_t = (M.lr * sqrtf(1.0f - M.b2p)) / (1.0f - M.b1p);
    o.touched = ctx->touched; o.touched_list = ctx->touched_list; o.touched_cnt = ctx->touched_cnt;
    o.bad = ctx->table_bad.as<unsigned long long>() + which;
    return o;
}

__global__ void set_count_kernel(int64_t *dst, int64_t v) { *dst = v; }

// The generator's loss is a MEAN over the batch (generator.py:28): with replicas the batch of a step is the union
// of the ranks' slices, so the 1/B of the data term must use the pairs of ALL ranks (the L2 term and the
// discriminator's loss are sums and need nothing).  One 8-byte all-reduce on the stream, read by the gradient
// kernel; returns the device word, or nullptr on a single replica (host-side 1/n, same arithmetic).
static int global_pair_count(gg_ctx *ctx, int64_t n_local, const int64_t **out) {
    *out = nullptr;
    const int world = ctx->comm ? ctx->world : ctx->fake_world;
    if (world <= 1) return GG_OK;
    GG_HIP(ctx, ctx->x_nglob.reserve(sizeof(int64_t) * 2));
    int64_t *w = ctx->x_nglob.as<int64_t>();
    // simulated ranks (GG_COMM_FAKE_WORLD): every rank holds this rank's batch
    hipLaunchKernelGGL(set_count_kernel, dim3(1), dim3(1), 0, ctx->stream, w, ctx->comm ? n_local : n_local * world);
    if (ctx->comm) {
        int rc = comm_allreduce_i64(ctx, w, 1);
        if (rc != GG_OK) return rc;
    }
    *out = w;
    return GG_OK;
}

// ---- staged gradient, host side (see path_count_kernel): buffers for up to n_occ staged rows ...
// n_occ: entries of the slot array (one per gradient the kernel may emit); n_stage: upper bound of the rows that are
// actually staged (<= n_occ).  Returns 1 (no error set) when the stage would not fit: the caller takes the atomic path.
static int64_t stage_bytes_cap() {
    static const int64_t cap = [] {
        const char *e = getenv("GG_STAGE_MAX_BYTES");
        return e ? (int64_t)atoll(e) : (int64_t)24 << 30;
    }();
    return cap;
}

static int staged_reserve(gg_ctx *ctx, int64_t n_occ, int64_t n_stage) {
    if (n_stage * (int64_t)(ctx->ld + 2) * 4 > stage_bytes_cap()) return 1;
    const size_t cnt_before = ctx->sg_cnt.bytes;
    GG_HIP(ctx, ctx->sg_cnt.reserve(sizeof(int32_t) * (size_t)ctx->n_node));
    // the counts are zero between passes -- every update resets the rows it applies -- unless the buffer is new or an earlier
    // staged pass failed half way (sg_cnt_dirty stays set): then they are cleared here, stale counts would mis-place stage rows
    if (ctx->sg_cnt.bytes != cnt_before || ctx->sg_cnt_dirty) GG_HIP(ctx, hipMemsetAsync(ctx->sg_cnt.p, 0, ctx->sg_cnt.bytes, ctx->stream));
    ctx->sg_cnt_dirty = true;
    GG_HIP(ctx, ctx->sg_off.reserve(sizeof(int32_t) * (size_t)ctx->n_node));
    GG_HIP(ctx, ctx->sg_list.reserve(sizeof(int4) * (size_t)ctx->n_node));
    GG_HIP(ctx, ctx->sg_slot.reserve(sizeof(int32_t) * (size_t)n_occ));
    GG_HIP(ctx, ctx->sg_bias.reserve(sizeof(float) * (size_t)n_stage));
    GG_HIP(ctx, ctx->sg_key.reserve(sizeof(int32_t) * (size_t)n_stage));
    GG_HIP(ctx, ctx->sg_tot.reserve(sizeof(int64_t) * 4));
    if (ctx->sg_rows.reserve(sizeof(float) * (size_t)n_stage * ctx->ld) != hipSuccess) {
        (void)hipGetLastError();  // out of memory for the stage: not an error, the atomic kernels need no stage
        return 1;
    }
    return GG_OK;
}

// ... and the update behind the gradient kernel: the small rows by the reducing optimizer, the hub rows through the flag
// list and sparse_opt_kernel (apply_optimizer, which also advances the step count), the updated-row total for the timing
static int staged_finish(gg_ctx *ctx, int which, int64_t n, int64_t n_occ, bool early_index = false) {
    const int opt = ctx->cfg.optimizer;
    OptArgs o = make_opt_args(ctx, which);  // before apply_optimizer advances the step count and the beta powers
    // the staging index: the pass's own (sg_*), or the one built on the side stream behind the G-mode walks (sgp_*)
    DevBuf &b_cnt = early_index ? ctx->sgp_cnt : ctx->sg_cnt, &b_list = early_index ? ctx->sgp_list : ctx->sg_list;
    DevBuf &b_tot = early_index ? ctx->sgp_tot : ctx->sg_tot, &b_key = early_index ? ctx->sgp_key : ctx->sg_key;
    o.sg_cnt = b_cnt.as<int32_t>(); o.sg_list = b_list.as<int4>(); o.sg_tot = b_tot.as<int64_t>();
    o.stage = ctx->sg_rows.as<float>(); o.stage_b = ctx->sg_bias.as<float>(); o.stage_key = b_key.as<int32_t>();
    int nb = cdiv((int64_t)std::min<int64_t>(n_occ, ctx->n_node) * 16, 256);
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    const bool replicas = ctx->comm || ctx->fake_world > 1;  // the summed rows go through the accumulators and the exchange
    if (replicas) hipLaunchKernelGGL(staged_opt_kernel<2>, dim3(nb), dim3(256), 0, ctx->stream, o);
    else if (opt == GG_OPT_SGD) hipLaunchKernelGGL(staged_opt_kernel<1>, dim3(nb), dim3(256), 0, ctx->stream, o);
    else hipLaunchKernelGGL(staged_opt_kernel<0>, dim3(nb), dim3(256), 0, ctx->stream, o);
    ctx->sg_active = true;  // the hub rows: flags -> list -> sparse_opt_kernel, which also resets their counts
    ctx->sg_cnt_active = o.sg_cnt;
    const int rc = apply_optimizer(ctx, which, n);
    ctx->sg_active = false;
    if (rc != GG_OK) return rc;
    // rows this pass updated (read back by the timing harvest): hub rows + small rows
    if (!replicas)
        hipLaunchKernelGGL(add_word_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->touched_ptr.as<int64_t>() + ctx->n_node, b_tot.as<int64_t>());
    GG_HIP(ctx, hipGetLastError());
    if (early_index) ctx->sgp_cnt_clean = true;  // every counted row was applied and reset
    else ctx->sg_cnt_dirty = false;
    return GG_OK;
}

// (the reducing kernel holds a row's sum in 4 float4 per lane: ld <= 256; wider tables keep the atomic kernels)
static bool staged_allowed(const gg_ctx *ctx) {
    return ctx->sg_threshold > 0 && ctx->ld <= 256 && ctx->cfg.optimizer != GG_OPT_ADAM_DENSE && !getenv("GG_NO_STAGED_GRAD");
}

// One optimizer step of model `which` on n device-resident rows.
// Strict mode (batch 64, dense TF1-Adam, the reference's default schedule) is two launches per step: the gradient kernel
// and the dense sweep, ~8.8 us together on CA-GrQc -- the cost of two dependent kernel boundaries.  Measured on the
// MI355X and NOT kept: (a) a persistent kernel per pass with the model resident in LDS over 64 workgroups and one grid
// barrier per step (agent-scope counter, next minibatch's rows published through memory): correct, 75 us per step --
// cross-XCD arrive-and-poll is an order of magnitude dearer than a kernel boundary; (b) one fused launch per step (every
// workgroup recomputes the 64 pair coefficients, variables double-buffered): correct, no faster (the chain ids -> rows ->
// update is as long as the boundary it saves); (c) hipGraph replay of the two launches (round 1): no gain.
// Considered and not built: (d) replaying TF1's decay-only updates lazily, when a row is next touched (bit-exact: the
// per-element operation sequence is unchanged) in ONE single-workgroup kernel per pass -- it removes every dense sweep, but the
// sweeps' N x steps element updates then run on one CU instead of 256: ~18 us per step on CA-GrQc.
int run_step(gg_ctx *ctx, int which, const int32_t *d_u, const int32_t *d_v, const float *d_x, int32_t n) {
    if (n <= 0) return GG_OK;
    Model &M = ctx->model[which];
    const int opt = ctx->cfg.optimizer;
    StepArgs s{};
    s.E = M.E; s.b = M.b; s.gE = ctx->gradE; s.gb = ctx->gradb;
    s.touched = ctx->touched; s.touched_list = ctx->touched_list; s.touched_cnt = ctx->touched_cnt;
    s.ld = ctx->ld;
    s.track = opt != GG_OPT_ADAM_DENSE;
    s.u = d_u; s.v = d_v; s.x = d_x; s.n = n;
    s.lambda = M.lambda;
    s.inv_n = 1.0f / (float)n;
    s.is_d = which == 1;
    if (which == 0) {
        int rc = global_pair_count(ctx, n, &s.n_glob);
        if (rc != GG_OK) return rc;
    }
    s.ppg = n >= 16384 ? PAIRS_PER_GROUP : (n >= 2048 ? 4 : 1);
    if (ctx->deterministic && n <= DET_MAX_PAIRS) {
        const int nfd = (ctx->ld + 15) / 16;
        // one replica, lazy Adam / SGD, the batch's rows in LDS: the row owners apply the optimizer themselves -- one launch per step
        const bool replicas = ctx->comm || ctx->fake_world > 1;
        const int fused = (!replicas && opt != GG_OPT_ADAM_DENSE && det_rows_fit_lds(ctx, s) && !getenv("GG_NO_FUSED_SMALL_STEP")) ? (opt == GG_OPT_SGD ? 2 : 1) : 0;
        const OptArgs o = make_opt_args(ctx, which);
        static unsigned long long *det_prof = nullptr;
        static const bool det_prof_on = getenv("GG_DET_PROFILE") != nullptr;
        if (det_prof_on) {
            if (!det_prof) GG_HIP(ctx, hipMalloc((void **)&det_prof, 16 * sizeof(unsigned long long)));
            GG_HIP(ctx, hipMemsetAsync(det_prof, 0, 16 * sizeof(unsigned long long), ctx->stream));
            s.prof = det_prof;
        }
        hipError_t e;
        if (nfd <= 4) e = launch_pair_grad_det<4>(ctx, s, o, fused);
        else if (nfd <= 8) e = launch_pair_grad_det<8>(ctx, s, o, fused);
        else if (nfd <= 16) e = launch_pair_grad_det<16>(ctx, s, o, fused);
        else e = launch_pair_grad_det<32>(ctx, s, o, fused);
        GG_HIP(ctx, e);
        if (det_prof_on) {  // debugging aid: one synchronisation per step; sums of the phase times per model, printed every 1 000 steps
            static double acc_t[2][2][4] = {};
            static long cnt[2] = {};
            unsigned long long h[16];
            GG_HIP(ctx, hipMemcpyAsync(h, det_prof, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
            GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (int w = 0; w < 2; ++w)
                for (int k = 0; k < 3; ++k) acc_t[which][w][k] += h[8 * w + k + 1] > h[8 * w + k] ? 10.0 * (double)(h[8 * w + k + 1] - h[8 * w + k]) : 0.0;
            if (++cnt[which] % 1000 == 0)
                for (int w = 0; w < 2; ++w)
                    fprintf(stderr, "[det] model %d n=%d wave %s: loads+coef %.0f ns, barrier %.0f, owners' sums + stores issued %.0f (means over %ld steps)\n", which, n, w ? "15" : "0",
                            acc_t[which][w][0] / cnt[which], acc_t[which][w][1] / cnt[which], acc_t[which][w][2] / cnt[which], cnt[which]);
        }
        if (fused) {
            GG_HIP(ctx, hipGetLastError());
            step_done(ctx, which, n);
            return GG_OK;
        }
        return apply_optimizer(ctx, which, n);
    }
    const int groups = cdiv(n, s.ppg);
    const int blocks = cdiv((int64_t)groups * 16, 256);
    const int nf = (ctx->ld + 15) / 16;
    // large fused batches on one replica: staged gradient rows + reducing optimizer instead of fp32 atomics (see path_count_kernel)
    bool staged = n >= 16384 && n < (1 << 30) && staged_allowed(ctx);
    if (staged) {
        int rc = staged_reserve(ctx, 2 * (int64_t)n, 2 * (int64_t)n);
        if (rc < 0) return rc;
        if (rc == 1) staged = false;  // the stage does not fit: atomics
    }
    if (staged) {
        int rc;
        int32_t *cnt = ctx->sg_cnt.as<int32_t>(), *off = ctx->sg_off.as<int32_t>(), *slot_v = ctx->sg_slot.as<int32_t>(), *slot_u = slot_v + n;
        hipLaunchKernelGGL(pair_occ_count_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, d_u, d_v, n, s.ppg, cnt, slot_v, slot_u);
        rc = device_segment_rows(ctx, cnt, ctx->n_node, ctx->sg_threshold, off, ctx->sg_list.as<int4>(), ctx->sg_tot.as<int64_t>());
        if (rc != GG_OK) return rc;
        hipLaunchKernelGGL(pair_occ_slot_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, d_u, d_v, n, cnt, off, ctx->sg_threshold, slot_v, slot_u,
                           ctx->sg_key.as<int32_t>());
        s.slot_v = slot_v; s.slot_u = slot_u;
        s.stage = ctx->sg_rows.as<float>(); s.stage_b = ctx->sg_bias.as<float>();
        // (staged implies ppg = 16 and ld <= 256)
        if (nf <= 4) hipLaunchKernelGGL((pair_grad16_kernel<4, true>), dim3(blocks), dim3(256), 0, ctx->stream, s);
        else if (nf <= 8) hipLaunchKernelGGL((pair_grad16_kernel<8, true>), dim3(blocks), dim3(256), 0, ctx->stream, s);
        else hipLaunchKernelGGL((pair_grad16_kernel<16, true>), dim3(blocks), dim3(256), 0, ctx->stream, s);
        if (ctx->tm_cur >= 0) GG_HIP(ctx, hipEventRecord(ctx->tm_ev[ctx->tm_cur][1], ctx->stream));  // gradient | optimizer
        return staged_finish(ctx, which, n, 2 * (int64_t)n);
    }
    if (s.ppg == PAIRS_PER_GROUP && nf <= 16 && !getenv("GG_OLD_PAIR_GRAD")) {  // large fused batch on the atomic path (replicas, dense Adam, wide rows excluded)
        if (nf <= 4) hipLaunchKernelGGL((pair_grad16_kernel<4, false>), dim3(blocks), dim3(256), 0, ctx->stream, s);
        else if (nf <= 8) hipLaunchKernelGGL((pair_grad16_kernel<8, false>), dim3(blocks), dim3(256), 0, ctx->stream, s);
        else hipLaunchKernelGGL((pair_grad16_kernel<16, false>), dim3(blocks), dim3(256), 0, ctx->stream, s);
    } else if (nf <= 4) hipLaunchKernelGGL((pair_grad_kernel<4, false>), dim3(blocks), dim3(256), 0, ctx->stream, s);
    else if (nf <= 8) hipLaunchKernelGGL((pair_grad_kernel<8, false>), dim3(blocks), dim3(256), 0, ctx->stream, s);
    else if (nf <= 16) hipLaunchKernelGGL((pair_grad_kernel<16, false>), dim3(blocks), dim3(256), 0, ctx->stream, s);
    else hipLaunchKernelGGL((pair_grad_kernel<32, false>), dim3(blocks), dim3(256), 0, ctx->stream, s);
    if (ctx->tm_cur >= 0) GG_HIP(ctx, hipEventRecord(ctx->tm_ev[ctx->tm_cur][1], ctx->stream));  // gradient | exchange + optimizer

    return apply_optimizer(ctx, which, n);
}

template <bool STAGED>
static void launch_path_grad(gg_ctx *ctx, const PathArgs &p, int blocks, int nf) {
    const dim3 g(blocks), b(256);
    if (p.stride <= 17) {  // every path has <= 16 nodes
        if (nf <= 4) hipLaunchKernelGGL((path_grad_kernel<4, STAGED, true>), g, b, 0, ctx->stream, p);
        else if (nf <= 8) hipLaunchKernelGGL((path_grad_kernel<8, STAGED, true>), g, b, 0, ctx->stream, p);
        else hipLaunchKernelGGL((path_grad_kernel<16, STAGED, true>), g, b, 0, ctx->stream, p);
    } else {
        if (nf <= 4) hipLaunchKernelGGL((path_grad_kernel<4, STAGED, false>), g, b, 0, ctx->stream, p);
        else if (nf <= 8) hipLaunchKernelGGL((path_grad_kernel<8, STAGED, false>), g, b, 0, ctx->stream, p);
        else hipLaunchKernelGGL((path_grad_kernel<16, STAGED, false>), g, b, 0, ctx->stream, p);
    }
}

// G step over whole walks of the resident prepare_g data (see path_grad_kernel).
int run_path_step(gg_ctx *ctx) {
    Model &M = ctx->model[0];
    const int64_t n = ctx->g_pairs;
    PathArgs p{};
    p.E = M.E; p.b = M.b; p.gE = ctx->gradE; p.gb = ctx->gradb;
    p.touched = ctx->touched; p.touched_list = ctx->touched_list; p.touched_cnt = ctx->touched_cnt;
    p.ld = ctx->ld;
    p.track = ctx->cfg.optimizer != GG_OPT_ADAM_DENSE;
    p.window = ctx->cfg.window_size;
    p.paths = ctx->w_paths.as<int32_t>();
    p.path_len = ctx->w_len.as<int32_t>();
    p.stride = ctx->w_stride;
    p.pair_ptr = ctx->g_ptr.as<int64_t>();
    p.reward = ctx->g_reward.as<float>();
    p.n_walks = ctx->w_total;
    p.lambda = M.lambda;
    p.inv_n = 1.0f / (float)n;
    {
        int rc = global_pair_count(ctx, n, &p.n_glob);
        if (rc != GG_OK) return rc;
    }
    const int blocks = cdiv(p.n_walks * 16, 256);
    const int nf = (ctx->ld + 15) / 16;
    const int64_t n_pos = p.n_walks * (int64_t)p.stride;
    bool staged = staged_allowed(ctx) && n_pos < (1ll << 31);
    // staged rows = path nodes of the walks that have pairs: a walk of L >= 2 nodes has 4L - 6 (window 2; L = 2: 2) resp.
    // 2L - 2 (window 1) pairs, so sum L <= (pairs + 6 walks) / 4 resp. (pairs + 2 walks) / 2 -- from host-side numbers, and
    // several times smaller than the walks x stride slot array
    const int64_t n_stage = std::min<int64_t>(n_pos, p.window >= 2 ? (n + 6 * p.n_walks) / 4 + 1 : (n + 2 * p.n_walks) / 2 + 1);
    if (staged) {
        const int rc = staged_reserve(ctx, n_pos, n_stage);
        if (rc < 0) return rc;
        if (rc == 1) staged = false;
    }
    if (!staged) {
        launch_path_grad<false>(ctx, p, blocks, nf);
        if (ctx->tm_cur >= 0) GG_HIP(ctx, hipEventRecord(ctx->tm_ev[ctx->tm_cur][1], ctx->stream));  // gradient | exchange + optimizer
        return apply_optimizer(ctx, 0, n);
    }
    // ---- staged gradient (see path_count_kernel)
    int rc;
    int32_t *cnt = ctx->sg_cnt.as<int32_t>(), *off = ctx->sg_off.as<int32_t>(), *slot = ctx->sg_slot.as<int32_t>();
    const dim3 pgrid((unsigned)cdiv(n_pos, 256));
    // the index of these walks may already be on its way (side stream, behind the walks: enqueue_path_slots)
    const bool early = ctx->g_slots_ready && ctx->g_slots_walks == p.n_walks && ctx->g_slots_stride == p.stride;
    ctx->g_slots_ready = false;
    ctx->g_slots_unused = 0;  // this caller takes whole-walk passes: keep (or resume) building the index early
    if (early) {
        ctx->sg_cnt_dirty = false;  // (staged_reserve left the pass's own count array clean, and this pass does not use it)
        GG_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_slots_done, 0));
        slot = ctx->sgp_slot.as<int32_t>();
    } else {
        hipLaunchKernelGGL(path_count_kernel, pgrid, dim3(256), 0, ctx->stream, p.paths, p.path_len, p.stride, p.n_walks, cnt, slot, (const unsigned long long *)nullptr);
        rc = device_segment_rows(ctx, cnt, ctx->n_node, ctx->sg_threshold, off, ctx->sg_list.as<int4>(), ctx->sg_tot.as<int64_t>());
        if (rc != GG_OK) return rc;
        hipLaunchKernelGGL(path_slot_kernel, pgrid, dim3(256), 0, ctx->stream, p.paths, p.path_len, p.stride, p.n_walks, cnt, off, ctx->sg_threshold, slot,
                           ctx->sg_key.as<int32_t>(), (const unsigned long long *)nullptr);
    }
    p.slot = slot;
    p.stage = ctx->sg_rows.as<float>();
    p.stage_b = ctx->sg_bias.as<float>();
    launch_path_grad<true>(ctx, p, blocks, nf);
    if (ctx->tm_cur >= 0) GG_HIP(ctx, hipEventRecord(ctx->tm_ev[ctx->tm_cur][1], ctx->stream));  // gradient | optimizer
    return staged_finish(ctx, 0, n, n_pos, early);
}

// The staging index of the G-mode walks that were just enqueued on the side stream, right behind them on that stream: per-row
// counts of the path nodes, segments of the small rows, the nodes' slots and keys (what run_path_step used to launch in front
// of its gradient kernel: ~0.12 ms of the step's critical path at the bench batch; here it runs beside the reward kernels).
// The index has buffers of its own: the discriminator pass is using sg_* on the main stream at this very time.
int enqueue_path_slots(gg_ctx *ctx) {
    ctx->g_slots_ready = false;
    if (ctx->walk_stream == ctx->stream || ctx->in_epoch_add || getenv("GG_NO_EARLY_SLOTS")) return GG_OK;
    const int64_t n_walks = ctx->w_total, n_pos = n_walks * (int64_t)ctx->w_stride;
    if (!staged_allowed(ctx) || ctx->cfg.window_size > 2 || n_walks == 0 || n_pos >= (1ll << 31)) return GG_OK;
    if (ctx->g_slots_unused >= 2) {  // the caller runs minibatch steps (run_step): the index would be built and dropped per prepare_g
        if (ctx->sgp_slot.p) {       // (hipFree waits for the device: once, when the schedule is recognised)
            for (DevBuf *b : {&ctx->sgp_cnt, &ctx->sgp_off, &ctx->sgp_slot, &ctx->sgp_list, &ctx->sgp_key, &ctx->sgp_scan}) b->release();
            ctx->sgp_cnt_clean = false;
        }
        return GG_OK;
    }
    hipStream_t st = ctx->walk_stream;
    const size_t cnt_before = ctx->sgp_cnt.bytes;
    GG_HIP(ctx, ctx->sgp_cnt.reserve(sizeof(int32_t) * (size_t)ctx->n_node));
    if (ctx->sgp_cnt.bytes != cnt_before || !ctx->sgp_cnt_clean) GG_HIP(ctx, hipMemsetAsync(ctx->sgp_cnt.p, 0, ctx->sgp_cnt.bytes, st));
    ctx->sgp_cnt_clean = false;
    GG_HIP(ctx, ctx->sgp_off.reserve(sizeof(int32_t) * (size_t)ctx->n_node));
    GG_HIP(ctx, ctx->sgp_list.reserve(sizeof(int4) * (size_t)ctx->n_node));
    GG_HIP(ctx, ctx->sgp_slot.reserve(sizeof(int32_t) * (size_t)n_pos));
    GG_HIP(ctx, ctx->sgp_key.reserve(sizeof(int32_t) * (size_t)n_pos));  // (staged rows <= path positions)
    GG_HIP(ctx, ctx->sgp_tot.reserve(sizeof(int64_t) * 4));
    if (!ctx->ev_slots_done) GG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_slots_done, hipEventDisableTiming));
    const unsigned long long *flag = ctx->dev_ctr + 3;
    const dim3 pgrid((unsigned)cdiv(n_pos, 256));
    int32_t *cnt = ctx->sgp_cnt.as<int32_t>(), *off = ctx->sgp_off.as<int32_t>(), *slot = ctx->sgp_slot.as<int32_t>();
    hipLaunchKernelGGL(path_count_kernel, pgrid, dim3(256), 0, st, ctx->w_paths.as<int32_t>(), ctx->w_len.as<int32_t>(), ctx->w_stride, n_walks, cnt, slot, flag);
    int rc = device_segment_rows(ctx, cnt, ctx->n_node, ctx->sg_threshold, off, ctx->sgp_list.as<int4>(), ctx->sgp_tot.as<int64_t>(), st, &ctx->sgp_scan);
    if (rc != GG_OK) return rc;
    hipLaunchKernelGGL(path_slot_kernel, pgrid, dim3(256), 0, st, ctx->w_paths.as<int32_t>(), ctx->w_len.as<int32_t>(), ctx->w_stride, n_walks, cnt, off,
                       ctx->sg_threshold, slot, ctx->sgp_key.as<int32_t>(), flag);
    GG_HIP(ctx, hipGetLastError());
    GG_HIP(ctx, hipEventRecord(ctx->ev_slots_done, st));
    ctx->g_slots_ready = true;
    ctx->g_slots_walks = n_walks;
    ctx->g_slots_stride = ctx->w_stride;
    return GG_OK;
}

__global__ void normalize_flags_kernel(int32_t *f, int n) {  // after the cross-rank sum: counts -> 0/1
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && f[i] > 1) f[i] = 1;
}

__global__ void scale_grads_kernel(float *gE, float *gb, int64_t nE, int n, float k) {  // simulated ranks only
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nE; i += stride) gE[i] *= k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) gb[i] *= k;
}

// ---- owner-partitioned sparse exchange (round 4; SURVEY.md section 8e "sparse", graph_gan.py:154-157,173-176).
// A fused pass touches a few 10^5 rows per rank; with P = 8 the ranks' packs together outgrow the row-pack rule
// (P * cap >= 1.5 N) and the step falls back to the dense exchange: 2 (P-1)/P * 4 N (ld + 1) bytes per rank whatever was
// touched.  Here every table row has an OWNER, row mod P:
//   1. the ranks count their touched rows per owner and all-gather the P x P count matrix (16 P^2 bytes, one host round trip:
//      point-to-point transfers need their sizes);
//   2. scatter: a rank's touched rows travel to their owners only, each pair of ranks over its own xGMI link;
//   3. the owner adds what it received to its own contribution IN RANK ORDER (plain adds: a rank names a row once) -- the
//      sum of a row is formed in one place, so all replicas receive the same bits;
//   4. gather: the owners' reduced rows -- the union over ranks of the touched rows, not N -- are all-gathered (capacity =
//      the largest owner's count, from a second 8-byte exchange) and every rank OVERWRITES its accumulator rows with them.
// Bytes sent per rank: (P-1)/P * own touched rows + (P-1) * (union / P) rows of (ld + 2) words, against N rows twice for the
// dense path.  Taken when the matrix says it is cheaper (all ranks see the same matrix: same branch everywhere).
// GG_COMM_FAKE_WORLD = k simulates it on one GPU: this rank plays every owner in turn, each "source rank" is a copy of it.
__global__ __launch_bounds__(256) void owner_count_kernel(const int32_t *list, const int64_t *cnt_ptr, int world, long long *cnt) {
    __shared__ int sh[64];
    if (threadIdx.x < 64) sh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t n = *cnt_ptr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) atomicAdd(&sh[list[i] % world], 1);
    __syncthreads();
    if ((int)threadIdx.x < world && sh[threadIdx.x]) atomicAdd((unsigned long long *)&cnt[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}

// touched rows -> per-owner segments of the send buffer (slots inside a segment by arrival: a rank names a row once, so the
// owner's sum does not depend on the slot order)
template <bool BF16>
__global__ __launch_bounds__(256) void owner_pack_kernel(const float *gE, const float *gb, const int32_t *list, const int64_t *cnt_ptr, int world,
                                                         const long long *seg_off, long long *fill, int ld, int32_t *out_ids, float *out_rows) {
    const int t = threadIdx.x & 15;
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int64_t cnt = *cnt_ptr;
    for (int64_t r = g0; r < cnt; r += ng) {
        const int row = list[r];
        const int q = row % world;
        long long slot = 0;
        if (t == 0) slot = seg_off[q] + (long long)atomicAdd((unsigned long long *)&fill[q], 1ull);
        slot = __shfl(slot, (threadIdx.x & 63) & ~15, 64);
        const float *src = gE + (int64_t)row * ld;
        for (int f = t; f < ld; f += 16) pack_put<BF16>(out_rows, slot, ld, f, src[f]);
        if (t == 0) {
            pack_put<BF16>(out_rows, slot, ld, ld, gb[row]);
            if (BF16) pack_put<BF16>(out_rows, slot, ld, ld + 1, 0.f);
            out_ids[slot] = row;
        }
    }
}

// owner side, after the adds: the flagged rows this rank owns -> gather pack (ids -1 behind them up to the capacity)
template <bool BF16>
__global__ __launch_bounds__(256) void owner_gather_pack_kernel(float *gE, float *gb, const int32_t *list, const int64_t *cnt_ptr, int world,
                                                                int me, long long *fill, int64_t cap, int ld, int32_t *out_ids, float *out_rows) {
    const int t = threadIdx.x & 15;
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int64_t cnt = *cnt_ptr;
    for (int64_t r = g0; r < cnt; r += ng) {
        const int row = list[r];
        if (row % world != me) continue;
        long long slot = 0;
        if (t == 0) slot = (long long)atomicAdd((unsigned long long *)fill, 1ull);
        slot = __shfl(slot, (threadIdx.x & 63) & ~15, 64);
        if (slot >= cap) continue;  // (cannot happen: the capacity is the largest owner's count)
        float *src = gE + (int64_t)row * ld;
        for (int f = t; f < ld; f += 16) {
            pack_put<BF16>(out_rows, slot, ld, f, src[f]);
            if (BF16) src[f] = pack_get<BF16>(out_rows, slot, ld, f);  // the owner keeps what the others receive: the rounded sum
        }
        if (t == 0) {
            pack_put<BF16>(out_rows, slot, ld, ld, gb[row]);
            if (BF16) { gb[row] = pack_get<BF16>(out_rows, slot, ld, ld); pack_put<BF16>(out_rows, slot, ld, ld + 1, 0.f); }
            out_ids[slot] = row;
        }
    }
}

__global__ void owner_count_owned_kernel(const int32_t *list, const int64_t *cnt_ptr, int world, int me, long long *out) {
    const int64_t n = *cnt_ptr;
    int c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += (list[i] % world == me) ? 1 : 0;
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd((unsigned long long *)out, (unsigned long long)c);
}

__global__ void fill_ids_kernel(int32_t *ids, int64_t n, int32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) ids[i] = v;
}

// receivers of the gather: the owner's sum REPLACES whatever this rank had accumulated for the row
template <bool BF16>
__global__ __launch_bounds__(256) void set_rows_kernel(float *gE, float *gb, int32_t *touched, const int32_t *ids, const float *rows, int64_t cnt, int ld) {
    const int t = threadIdx.x & 15;
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
    for (int64_t r = g0; r < cnt; r += ng) {
        const int row = ids[r];
        if (row < 0) continue;
        float *dst = gE + (int64_t)row * ld;
        for (int f = t; f < ld; f += 16) dst[f] = pack_get<BF16>(rows, r, ld, f);
        if (t == 0) { gb[row] = pack_get<BF16>(rows, r, ld, ld); touched[row] = 1; }
    }
}

// returns GG_OK with *taken = true when the exchange was done here; *taken = false: the caller takes the dense path
static int exchange_owner(gg_ctx *ctx, OptArgs &o, int world, bool *taken) {
    *taken = false;
    const int n = ctx->n_node, ld = ctx->ld;
    const bool fake = !ctx->comm;
    if (!ctx->owner_exchange || world < 2 || world > 64 || !comm_has_p2p(ctx)) return GG_OK;
    const bool bf16 = ctx->comm_bf16;
    const size_t row_f = pack_row_words(ld, bf16);  // 4-byte words per packed row
    // ---- 1. this rank's touched rows, counted per owner; the P x P matrix on every host
    int rc = device_compact_flags(ctx, ctx->touched, n, ctx->touched_list, ctx->touched_ptr.as<int64_t>() + n);
    if (rc != GG_OK) return rc;
    GG_HIP(ctx, ctx->x_own.reserve(sizeof(long long) * (size_t)(3 * world + world * world + 8)));
    long long *cnt = ctx->x_own.as<long long>(), *fill = cnt + world, *seg = cnt + 2 * world, *mat = cnt + 3 * world, *own = mat + world * world;
    GG_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(long long) * (size_t)(3 * world + world * world + 8), ctx->stream));
    hipLaunchKernelGGL(owner_count_kernel, dim3(256), dim3(256), 0, ctx->stream, ctx->touched_list, o.touched_total, world, cnt);
    rc = comm_allgather(ctx, cnt, mat, (size_t)world, 8);
    if (rc != GG_OK) return rc;
    std::vector<long long> M((size_t)world * world);
    GG_HIP(ctx, hipMemcpyAsync(M.data(), mat, sizeof(long long) * M.size(), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int me_real = fake ? 0 : ctx->rank;
    long long sum_t = 0, max_t = 0;
    for (int r = 0; r < world; ++r) {
        long long tr = 0;
        for (int q = 0; q < world; ++q) tr += M[(size_t)r * world + q];
        sum_t += tr;
        max_t = std::max(max_t, tr);
    }
    // rows sent: own rows to their owners + the owners' unions to everyone, against the table twice for the dense path
    const double f = (double)(world - 1) / world;
    if (f * (double)max_t + f * (double)std::min<long long>(n, sum_t) >= 0.9 * 2.0 * f * (double)n) return GG_OK;  // dense is no worse
    // ---- 2. pack by owner
    std::vector<long long> soff(world + 1, 0);
    for (int q = 0; q < world; ++q) soff[q + 1] = soff[q] + M[(size_t)me_real * world + q];
    const long long T = soff[world];
    GG_HIP(ctx, hipMemcpyAsync(seg, soff.data(), sizeof(long long) * world, hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, ctx->x_send_ids.reserve(sizeof(int32_t) * (size_t)std::max<long long>(T, 1)));
    GG_HIP(ctx, ctx->x_send_rows.reserve(sizeof(float) * (size_t)std::max<long long>(T, 1) * row_f));
    int nb = cdiv(std::max<long long>(T, 1) * 16, 256);
    if (nb > 4096) nb = 4096;
    if (bf16) hipLaunchKernelGGL(owner_pack_kernel<true>, dim3(nb), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched_list, o.touched_total, world, seg, fill, ld,
                                 ctx->x_send_ids.as<int32_t>(), ctx->x_send_rows.as<float>());
    else hipLaunchKernelGGL(owner_pack_kernel<false>, dim3(nb), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched_list, o.touched_total, world, seg, fill, ld,
                            ctx->x_send_ids.as<int32_t>(), ctx->x_send_rows.as<float>());
    long long bytes = 0;
    const int owners = fake ? world : 1;  // simulated ranks: this GPU plays every owner in turn
    for (int oi = 0; oi < owners; ++oi) {
        const int me = fake ? oi : ctx->rank;
        // ---- 3. scatter: what the other ranks have for owner `me`
        std::vector<int64_t> s_off(world, 0), s_cnt(world, 0), r_off(world, 0), r_cnt(world, 0), si_off(world, 0), si_cnt(world, 0), ri_off(world, 0), ri_cnt(world, 0);
        long long R = 0;
        for (int r = 0; r < world; ++r) {
            if (r == me) continue;
            const long long in = M[(size_t)r * world + me], out = M[(size_t)me * world + r];
            ri_off[r] = R; ri_cnt[r] = in;
            r_off[r] = R * (long long)row_f; r_cnt[r] = in * (long long)row_f;
            si_off[r] = soff[r]; si_cnt[r] = out;
            s_off[r] = soff[r] * (long long)row_f; s_cnt[r] = out * (long long)row_f;
            R += in;
            bytes += out * (long long)(row_f + 1) * 4;
        }
        GG_HIP(ctx, ctx->x_recv_ids.reserve(sizeof(int32_t) * (size_t)std::max<long long>(R, 1)));
        GG_HIP(ctx, ctx->x_recv_rows.reserve(sizeof(float) * (size_t)std::max<long long>(R, 1) * row_f));
        if (!fake) {
            rc = comm_exchange_v(ctx, (const float *)ctx->x_send_ids.p, si_off.data(), si_cnt.data(), (float *)ctx->x_recv_ids.p, ri_off.data(), ri_cnt.data());
            if (rc == GG_OK) rc = comm_exchange_v(ctx, ctx->x_send_rows.as<float>(), s_off.data(), s_cnt.data(), ctx->x_recv_rows.as<float>(), r_off.data(), r_cnt.data());
            if (rc != GG_OK) return rc;
        } else {
            for (int r = 0; r < world; ++r) {  // every simulated source holds this rank's segment for owner `me`
                if (r == me || !ri_cnt[r]) continue;
                GG_HIP(ctx, hipMemcpyAsync(ctx->x_recv_ids.as<int32_t>() + ri_off[r], ctx->x_send_ids.as<int32_t>() + soff[me], sizeof(int32_t) * (size_t)ri_cnt[r], hipMemcpyDeviceToDevice, ctx->stream));
                GG_HIP(ctx, hipMemcpyAsync(ctx->x_recv_rows.as<float>() + r_off[r], ctx->x_send_rows.as<float>() + soff[me] * (long long)row_f, sizeof(float) * (size_t)r_cnt[r], hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        // ---- owner's sum: own contribution (already in the accumulators) + the sources in rank order
        for (int r = 0; r < world; ++r) {
            if (r == me || !ri_cnt[r]) continue;
            int nbr = cdiv(ri_cnt[r] * 16, 256);
            if (nbr > 4096) nbr = 4096;
            if (bf16) hipLaunchKernelGGL(add_rows_kernel<true>, dim3(nbr), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched, ctx->x_recv_ids.as<int32_t>() + ri_off[r],
                                         ctx->x_recv_rows.as<float>() + r_off[r], (int64_t)ri_cnt[r], ld);
            else hipLaunchKernelGGL(add_rows_kernel<false>, dim3(nbr), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched, ctx->x_recv_ids.as<int32_t>() + ri_off[r],
                                    ctx->x_recv_rows.as<float>() + r_off[r], (int64_t)ri_cnt[r], ld);
        }
    }
    // ---- 4. gather the owners' reduced rows
    rc = device_compact_flags(ctx, ctx->touched, n, ctx->touched_list, ctx->touched_ptr.as<int64_t>() + n);  // (the adds flagged new rows)
    if (rc != GG_OK) return rc;
    long long umax = 0;
    if (!fake) {
        hipLaunchKernelGGL(owner_count_owned_kernel, dim3(256), dim3(256), 0, ctx->stream, ctx->touched_list, o.touched_total, world, ctx->rank, own);
        rc = comm_allreduce_max_i64(ctx, (int64_t *)own, 1);
        if (rc != GG_OK) return rc;
        GG_HIP(ctx, hipMemcpyAsync(&umax, own, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const size_t cap = (size_t)std::max<long long>(umax, 1);
        GG_HIP(ctx, ctx->x_send_ids.reserve(sizeof(int32_t) * cap));
        GG_HIP(ctx, ctx->x_send_rows.reserve(sizeof(float) * cap * row_f));
        GG_HIP(ctx, ctx->x_recv_ids.reserve(sizeof(int32_t) * cap * world));
        GG_HIP(ctx, ctx->x_recv_rows.reserve(sizeof(float) * cap * row_f * world));
        hipLaunchKernelGGL(fill_ids_kernel, dim3(256), dim3(256), 0, ctx->stream, ctx->x_send_ids.as<int32_t>(), (int64_t)cap, -1);
        GG_HIP(ctx, hipMemsetAsync(own + 1, 0, sizeof(long long), ctx->stream));
        int nbg = cdiv((int64_t)cap * 16, 256);
        if (nbg > 4096) nbg = 4096;
        if (bf16) hipLaunchKernelGGL(owner_gather_pack_kernel<true>, dim3(nbg), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched_list, o.touched_total, world, ctx->rank,
                                     own + 1, (int64_t)cap, ld, ctx->x_send_ids.as<int32_t>(), ctx->x_send_rows.as<float>());
        else hipLaunchKernelGGL(owner_gather_pack_kernel<false>, dim3(nbg), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched_list, o.touched_total, world, ctx->rank,
                                own + 1, (int64_t)cap, ld, ctx->x_send_ids.as<int32_t>(), ctx->x_send_rows.as<float>());
        rc = comm_allgather(ctx, ctx->x_send_ids.p, ctx->x_recv_ids.p, cap, 4);
        if (rc == GG_OK) rc = comm_allgather(ctx, ctx->x_send_rows.p, ctx->x_recv_rows.p, cap * row_f, 4);
        if (rc != GG_OK) return rc;
        for (int r = 0; r < world; ++r) {
            if (r == ctx->rank) continue;
            if (bf16) hipLaunchKernelGGL(set_rows_kernel<true>, dim3(nbg), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched, ctx->x_recv_ids.as<int32_t>() + (size_t)r * cap,
                                         ctx->x_recv_rows.as<float>() + (size_t)r * cap * row_f, (int64_t)cap, ld);
            else hipLaunchKernelGGL(set_rows_kernel<false>, dim3(nbg), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched, ctx->x_recv_ids.as<int32_t>() + (size_t)r * cap,
                                    ctx->x_recv_rows.as<float>() + (size_t)r * cap * row_f, (int64_t)cap, ld);
        }
        bytes += (long long)(world - 1) * (long long)cap * (long long)(row_f + 1) * 4;
    } else {
        // simulated: every owner's rows are already in this GPU's accumulators; run the pack / overwrite kernels once (owner 0)
        // so that they are exercised: overwriting a row with itself changes nothing
        hipLaunchKernelGGL(owner_count_owned_kernel, dim3(256), dim3(256), 0, ctx->stream, ctx->touched_list, o.touched_total, bf16 ? 1 : world, 0, own);
        GG_HIP(ctx, hipMemcpyAsync(&umax, own, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const size_t cap = (size_t)std::max<long long>(umax, 1);
        GG_HIP(ctx, ctx->x_send_ids.reserve(sizeof(int32_t) * cap));
        GG_HIP(ctx, ctx->x_send_rows.reserve(sizeof(float) * cap * row_f));
        hipLaunchKernelGGL(fill_ids_kernel, dim3(256), dim3(256), 0, ctx->stream, ctx->x_send_ids.as<int32_t>(), (int64_t)cap, -1);
        GG_HIP(ctx, hipMemsetAsync(own + 1, 0, sizeof(long long), ctx->stream));
        int nbg = cdiv((int64_t)cap * 16, 256);
        if (nbg > 4096) nbg = 4096;
        // (bf16 packs: every owner rounds the sums it hands out, so the simulation rounds ALL touched rows -- world = 1 makes every row "owned")
        const int wsim = bf16 ? 1 : world;
        if (bf16) {
            hipLaunchKernelGGL(owner_gather_pack_kernel<true>, dim3(nbg), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched_list, o.touched_total, wsim, 0, own + 1,
                               (int64_t)cap, ld, ctx->x_send_ids.as<int32_t>(), ctx->x_send_rows.as<float>());
            hipLaunchKernelGGL(set_rows_kernel<true>, dim3(nbg), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched, ctx->x_send_ids.as<int32_t>(),
                               ctx->x_send_rows.as<float>(), (int64_t)cap, ld);
        } else {
            hipLaunchKernelGGL(owner_gather_pack_kernel<false>, dim3(nbg), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched_list, o.touched_total, wsim, 0, own + 1,
                               (int64_t)cap, ld, ctx->x_send_ids.as<int32_t>(), ctx->x_send_rows.as<float>());
            hipLaunchKernelGGL(set_rows_kernel<false>, dim3(nbg), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched, ctx->x_send_ids.as<int32_t>(),
                               ctx->x_send_rows.as<float>(), (int64_t)cap, ld);
        }
    }
    GG_HIP(ctx, hipGetLastError());
    ctx->comm_steps_sparse += 1;
    ctx->comm_steps_owner += 1;
    if (ctx->comm) ctx->comm_bytes_sent += bytes;
    *taken = true;
    return GG_OK;
}

// Replica exchange of the lazy / sgd modes, WITHOUT host synchronisation.  `bound` = the most pairs any rank brings to
// this step (known on every host: the batch size of a minibatch step, or the max over ranks of the prepared rows that
// gg_prepare_* exchanged inside its own synchronisation), so cap = min(N, 2 * bound) rows bound every rank's pack.
//   world * cap <  ratio * N : fixed-capacity row packs {id, gradient row, bias gradient}, all-gathered (every rank's
//                              pack travels over its own xGMI link) and added in RANK ORDER with plain adds -- replicas
//                              compute bit-identical sums;
//   otherwise                : the replicas together may touch most of the table: dense reduce-scatter + all-gather of
//                              the accumulators and the union of the row flags.
// The choice depends on host-side numbers that are identical on every rank, so all ranks take the same branch.
static int exchange_sparse(gg_ctx *ctx, OptArgs &o, int world, int64_t bound) {
    const int n = ctx->n_node, ld = ctx->ld;
    const int64_t cap = std::min<int64_t>(n, 2 * std::max<int64_t>(bound, 0));
    if (cap == 0) return GG_OK;  // no rank has a pair in this step
    if ((double)world * (double)cap >= (double)ctx->dense_exchange_ratio * n) {
        // the packs of all ranks together may outgrow the table: first choice the owner-partitioned exchange of what was
        // really touched (two small host round trips; a fused pass has one step), else the whole accumulators
        if (bound >= ctx->owner_min_bound) {
            bool taken = false;
            int rc = exchange_owner(ctx, o, world, &taken);
            if (rc != GG_OK) return rc;
            if (taken) return GG_OK;
        }
        if (ctx->comm) {
            int rc = comm_allreduce_grads(ctx);
            if (rc == GG_OK) rc = comm_allreduce_flags(ctx);
            if (rc != GG_OK) return rc;
            ctx->comm_bytes_sent += (int64_t)comm_dense_bytes(ctx) + (int64_t)(2.0 * 4.0 * n * (world - 1) / world);
        } else {  // GG_COMM_FAKE_WORLD: every simulated rank holds this rank's gradient
            hipLaunchKernelGGL(scale_grads_kernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, (int64_t)n * ld, n, (float)world);
        }
        hipLaunchKernelGGL(normalize_flags_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, ctx->touched, n);
        GG_HIP(ctx, hipGetLastError());
        ctx->comm_steps_dense += 1;
        return GG_OK;
    }
    int rc = device_compact_flags(ctx, ctx->touched, n, ctx->touched_list, ctx->touched_ptr.as<int64_t>() + n);
    if (rc != GG_OK) return rc;
    const size_t row_f = (size_t)ld + 1;
    GG_HIP(ctx, ctx->x_send_ids.reserve(sizeof(int32_t) * cap));
    GG_HIP(ctx, ctx->x_send_rows.reserve(sizeof(float) * cap * row_f));
    GG_HIP(ctx, ctx->x_recv_ids.reserve(sizeof(int32_t) * cap * world));
    GG_HIP(ctx, ctx->x_recv_rows.reserve(sizeof(float) * cap * row_f * world));
    int nb = cdiv(cap * 16, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(nb), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched, ctx->touched_list,
                       o.touched_total, cap, ld, ctx->x_send_ids.as<int32_t>(), ctx->x_send_rows.as<float>(), ctx->touched_cnt + 2);
    rc = comm_allgather(ctx, ctx->x_send_ids.p, ctx->x_recv_ids.p, (size_t)cap, 4);
    if (rc != GG_OK) return rc;
    rc = comm_allgather(ctx, ctx->x_send_rows.p, ctx->x_recv_rows.p, (size_t)cap * row_f, 4);
    if (rc != GG_OK) return rc;
    for (int r = 0; r < world; ++r)
        hipLaunchKernelGGL(add_rows_kernel<false>, dim3(nb), dim3(256), 0, ctx->stream, ctx->gradE, ctx->gradb, ctx->touched,
                           ctx->x_recv_ids.as<int32_t>() + (size_t)r * cap, ctx->x_recv_rows.as<float>() + (size_t)r * cap * row_f, cap, ld);
    GG_HIP(ctx, hipGetLastError());
    ctx->comm_steps_sparse += 1;
    if (ctx->comm) ctx->comm_bytes_sent += (int64_t)((world - 1) * cap * (row_f + 1) * 4);
    return GG_OK;
}

// Max over ranks of the rows / pairs a prepare call produced (one 8-byte all-reduce and one synchronisation per PREPARE
// call on replicas -- not per step): the capacity rule of the steps' row packs.
int exchange_count_max(gg_ctx *ctx, int64_t local, int64_t *max_out) {
    *max_out = local;
    if (!ctx->comm || ctx->world <= 1) return GG_OK;
    GG_HIP(ctx, ctx->x_nglob.reserve(sizeof(int64_t) * 4));
    int64_t *w = ctx->x_nglob.as<int64_t>() + 2;
    ctx->h_pin[gg_ctx::H_TOTAL + 2] = (unsigned long long)local;
    GG_HIP(ctx, hipMemcpyAsync(w, ctx->h_pin + gg_ctx::H_TOTAL + 2, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    int rc = comm_allreduce_max_i64(ctx, w, 1);
    if (rc != GG_OK) return rc;
    GG_HIP(ctx, hipMemcpyAsync(ctx->h_pin + gg_ctx::H_TOTAL + 1, w, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *max_out = (int64_t)ctx->h_pin[gg_ctx::H_TOTAL + 1];
    return GG_OK;
}

// gradient exchange (multi-GPU) + optimizer kernel + step bookkeeping, after a gradient kernel
int apply_optimizer(gg_ctx *ctx, int which, int64_t n) {
    const int opt = ctx->cfg.optimizer;
    int rc = GG_OK;
    if (opt == GG_OPT_ADAM_DENSE) {
        rc = comm_allreduce_grads(ctx);
        if (ctx->comm && ctx->world > 1) { ctx->comm_steps_dense += 1; ctx->comm_bytes_sent += (int64_t)comm_dense_bytes(ctx); }
    }
    if (rc != GG_OK) return rc;

    OptArgs o = make_opt_args(ctx, which);
    if (ctx->sg_active) o.sg_cnt = ctx->sg_cnt_active;  // hub rows of a staged pass: the row's count is reset with its flag
    if (opt == GG_OPT_ADAM_DENSE) {
        int64_t nb = (o.nE / 4 + 255) / 256;
        if (nb > 2048) nb = 2048;
        if (nb < 1) nb = 1;
        hipLaunchKernelGGL(adam_dense_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, o);
    } else {
        // rows to update = rows flagged by the gradient kernel (all ranks' flags were summed with the gradients)
        GG_HIP(ctx, ctx->touched_ptr.reserve(sizeof(int64_t) * ((size_t)ctx->n_node + 1)));
        o.touched_ptr = ctx->touched_ptr.as<int64_t>();
        o.touched_total = o.touched_ptr + ctx->n_node;
        const int world = ctx->comm ? ctx->world : ctx->fake_world;
        if (world > 1 || (ctx->comm && ctx->world == 1)) {
            rc = exchange_sparse(ctx, o, world < 1 ? 1 : world, ctx->step_bound > 0 ? ctx->step_bound : n);
            if (rc != GG_OK) return rc;
        }
        rc = device_compact_flags(ctx, ctx->touched, ctx->n_node, ctx->touched_list, ctx->touched_ptr.as<int64_t>() + ctx->n_node);
        if (rc != GG_OK) return rc;
        int nb = cdiv((int64_t)std::min<int64_t>(2ll * n * ctx->world, ctx->n_node) * 16, 256);
        if (nb > 4096) nb = 4096;
        if (nb < 1) nb = 1;
        if (opt == GG_OPT_SGD) hipLaunchKernelGGL(sparse_opt_kernel<1>, dim3(nb), dim3(256), 0, ctx->stream, o);
        else hipLaunchKernelGGL(sparse_opt_kernel<0>, dim3(nb), dim3(256), 0, ctx->stream, o);
    }
    GG_HIP(ctx, hipGetLastError());
    step_done(ctx, which, n);
    return GG_OK;
}

// host-side bookkeeping behind one optimizer step of model `which` on n pairs
void step_done(gg_ctx *ctx, int which, int64_t n) {
    Model &M = ctx->model[which];
    if (which == 0) { ctx->gen_dirty = true; generator_changed(ctx); }  // the generator moved: cached distributions and edge scores are stale
    M.t += 1;
    M.b1p = M.b1p * ctx->cfg.adam_beta1;
    M.b2p = M.b2p * ctx->cfg.adam_beta2;
    if (which == 1) { ctx->ctr.d_pairs += n; ctx->ctr.d_steps += 1; }
    else { ctx->ctr.g_pairs += n; ctx->ctr.g_steps += 1; }
}

}  // namespace gg

using namespace gg;

static int host_step(gg_ctx *ctx, int which, const int32_t *u, const int32_t *v, const float *x, int32_t n) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    if (which == 0) discard_begun_walk(ctx);  // (walks begun by gg_prepare_g_begin read the tables this step writes)
    GG_CHECK(ctx, n >= 0 && (n == 0 || (u && v && x)), GG_EINVAL, "step: bad argument");
    if (n == 0) return GG_OK;
    for (int i = 0; i < n; ++i)
        GG_CHECK(ctx, u[i] >= 0 && u[i] < ctx->n_node && v[i] >= 0 && v[i] < ctx->n_node, GG_EINVAL, "step: id out of range at %d", i);
    GG_HIP(ctx, hipSetDevice(ctx->device));
    GG_HIP(ctx, ctx->step_u.reserve(sizeof(int32_t) * n));
    GG_HIP(ctx, ctx->step_v.reserve(sizeof(int32_t) * n));
    GG_HIP(ctx, ctx->step_x.reserve(sizeof(float) * n));
    GG_HIP(ctx, hipMemcpyAsync(ctx->step_u.p, u, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, hipMemcpyAsync(ctx->step_v.p, v, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(ctx, hipMemcpyAsync(ctx->step_x.p, x, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream));
    ctx->step_bound = n;
    if (ctx->comm && ctx->world > 1 && ctx->cfg.optimizer != GG_OPT_ADAM_DENSE) {
        // ranks may pass batches of different sizes: the pack capacity is the largest of them (this call is synchronous anyway)
        GG_HIP(ctx, ctx->x_nglob.reserve(sizeof(int64_t) * 4));
        int64_t *w = ctx->x_nglob.as<int64_t>() + 3;
        const int64_t mine = n;
        GG_HIP(ctx, hipMemcpyAsync(w, &mine, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        int rc0 = comm_allreduce_max_i64(ctx, w, 1);
        if (rc0 != GG_OK) return rc0;
        int64_t mx = n;
        GG_HIP(ctx, hipMemcpyAsync(&mx, w, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->step_bound = mx;
    }
    int rc = run_step(ctx, which, ctx->step_u.as<int32_t>(), ctx->step_v.as<int32_t>(), ctx->step_x.as<float>(), n);
    if (rc != GG_OK) return rc;
    GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GG_OK;
}

static int run_pass(gg_ctx *ctx, int which, const int64_t *starts, int64_t n_batches, int32_t batch_size) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    if (which == 0) discard_begun_walk(ctx);  // (walks begun by gg_prepare_g_begin read the tables this pass writes)
    GG_CHECK(ctx, batch_size > 0 && n_batches >= 0 && (starts || n_batches == 0), GG_EINVAL, "pass: bad argument");
    const int64_t rows = which == 1 ? ctx->d_rows : ctx->g_pairs;
    const int32_t *u = which == 1 ? ctx->d_center.as<int32_t>() : ctx->g_node1.as<int32_t>();
    const int32_t *v = which == 1 ? ctx->d_neighbor.as<int32_t>() : ctx->g_node2.as<int32_t>();
    const float *x = which == 1 ? ctx->d_label.as<float>() : ctx->g_reward.as<float>();
    GG_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t rows_max = std::max(rows, which == 1 ? ctx->d_rows_max : ctx->g_pairs_max);
    // A step this rank has no rows for (start < 0, or nothing prepared at all): with replicas it still takes part in the
    // step's collectives -- the other ranks are waiting in them -- and applies the summed update of the others.
    auto empty_step = [&]() -> int {
        if (!ctx->comm) return GG_OK;
        if (which == 0) {  // the generator's steps open with the all-reduce of the ranks' pair counts
            const int64_t *unused = nullptr;
            int rc = global_pair_count(ctx, 0, &unused);
            if (rc != GG_OK) return rc;
        }
        return apply_optimizer(ctx, which, 0);
    };
    if (rows == 0) {
        ctx->step_bound = std::min<int64_t>(batch_size, rows_max);
        for (int64_t k = 0; k < n_batches; ++k) {
            int rc = empty_step();
            if (rc != GG_OK) return rc;
        }
        if (ctx->comm && n_batches > 0) GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return GG_OK;
    }
    const bool timed = ctx->profile_every == 1;  // gg_set_profiling: otherwise no events and no wait
    if (timed) GG_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    const bool whole = n_batches == 1 && starts[0] == 0 && batch_size >= rows && rows > 0;
    // per-kernel timing (gradient kernel | exchange + optimizer kernel) of every profile_every-th fused pass; the events
    // are read at the next host synchronisation (harvest_timings), the pass itself does not wait for them
    ctx->tm_cur = -1;
    if (whole && ctx->profile_every > 0 && (ctx->pass_call_index[which]++ % ctx->profile_every) == 0 && !(ctx->deterministic && rows <= DET_MAX_PAIRS)) {
        ctx->tm_cur = timing_slot(ctx);
        if (ctx->tm_cur >= 0) GG_HIP(ctx, hipEventRecord(ctx->tm_ev[ctx->tm_cur][0], ctx->stream));
    }
    ctx->step_bound = std::min<int64_t>(batch_size, rows_max);
    if (which == 0 && whole && ctx->g_paths_valid && ctx->cfg.window_size <= 2 && ctx->ld <= 256 && !getenv("GG_NO_PATH_GRAD")) {
        int rc = run_path_step(ctx);
        if (rc != GG_OK) { ctx->tm_cur = -1; return rc; }
    } else {
        if (which == 0) {  // minibatches read the pair arrays
            int rc = ensure_g_pairs(ctx);
            if (rc != GG_OK) { ctx->tm_cur = -1; return rc; }
        }
        for (int64_t k = 0; k < n_batches; ++k) {
            const int64_t s = starts[k];
            if (s < 0) {  // this rank's share of the step is empty (the replicas' batch lists have different lengths)
                int rc = empty_step();
                if (rc != GG_OK) return rc;
                continue;
            }
            GG_CHECK(ctx, s < rows, GG_EINVAL, "pass: start %lld outside the %lld prepared rows", (long long)s, (long long)rows);
            const int32_t n = (int32_t)std::min<int64_t>(batch_size, rows - s);
            int rc = run_step(ctx, which, u + s, v + s, x + s, n);
            if (rc != GG_OK) return rc;
        }
    }
    if (ctx->tm_cur >= 0) {
        const int ts = ctx->tm_cur;
        ctx->tm_cur = -1;
        GG_HIP(ctx, hipEventRecord(ctx->tm_ev[ts][2], ctx->stream));
        const bool has_rows = ctx->cfg.optimizer != GG_OPT_ADAM_DENSE;
        if (has_rows)  // rows the optimizer kernel updated: device-side count, copied behind it into pinned memory
            GG_HIP(ctx, hipMemcpyAsync(ctx->h_pin + gg_ctx::H_ROWS + ts, ctx->touched_ptr.as<int64_t>() + ctx->n_node, sizeof(int64_t),
                                       hipMemcpyDeviceToHost, ctx->stream));
        ctx->tm_pending.push_back({which == 1 ? 1 : 2, rows, ts, has_rows});
    }
    if (which == 0) {  // walks of a later gg_prepare_g run on the side stream: they start behind this generator update
        GG_HIP(ctx, hipEventRecord(ctx->ev_gen_pass, ctx->stream));
        ctx->gen_pass_recorded = true;
        ctx->gen_dirty = false;
    }
    if (timed) {
        GG_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        GG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        harvest_timings(ctx);
        float ms = 0.f;
        GG_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        ctx->ctr.last_kernel_ms = ms;
    }
    return GG_OK;
}

extern "C" {

int gg_d_step(gg_ctx *ctx, const int32_t *u, const int32_t *v, const float *label, int32_t n) { return host_step(ctx, 1, u, v, label, n); }
int gg_g_step(gg_ctx *ctx, const int32_t *u, const int32_t *v, const float *reward, int32_t n) { return host_step(ctx, 0, u, v, reward, n); }
int gg_d_pass(gg_ctx *ctx, const int64_t *starts, int64_t n_batches, int32_t batch_size) { return run_pass(ctx, 1, starts, n_batches, batch_size); }
int gg_g_pass(gg_ctx *ctx, const int64_t *starts, int64_t n_batches, int32_t batch_size) { return run_pass(ctx, 0, starts, n_batches, batch_size); }

}  // extern "C"
