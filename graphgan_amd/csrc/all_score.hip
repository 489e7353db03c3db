// all_score.hip -- K7: rows of the generator's all-pairs score matrix on the matrix cores.
//
// Replaces sess.run(generator.all_score) (reference src/GraphGAN/graph_gan.py:238,
// src/GraphGAN/generator.py:21):  S[i, j] = g_i . g_j + b_g[j]  (bias added per COLUMN).
// The walk sampler never needs it (it scores tree neighbours on demand), but the call site
// exists in the reference, and all-pairs scoring is the one GEMM-shaped piece of the path
// (N x d . d x N), so it goes to MFMA: v_mfma_f32_32x32x2_f32, f32 in / f32 accumulate --
// bit-for-bit a k-ordered fmaf chain, i.e. exact fp32 like the reference's tf.matmul class of
// arithmetic (no bf16 rounding of the embeddings).
//
// Tiling: one 256-thread workgroup (4 wavefronts) per 32-row x 128-column tile of S; the 32 A
// rows and 128 B rows (both are rows of E) are staged through LDS in K-chunks of 32 floats with a
// +1 padded row stride (conflict-free fragment reads: lane l reads [l & 31][k + (l >> 5)]); each
// wavefront owns a 32 x 32 accumulator (16 registers per lane).  Memory-bound on the S write
// (4 N bytes per row) and the single sweep over E; never materialises more than the requested rows.
#include <math.h>

#include <algorithm>
#include <vector>

#include "gg_internal.h"

namespace gg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AS_KC = 32;

__global__ __launch_bounds__(256) void all_score_kernel(const float *E, const float *bias, int n_node, int ld, const int32_t *rows,
                                                        int n_rows, float *out) {
    __shared__ float As[32][AS_KC + 1];
    __shared__ float Bs[128][AS_KC + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c0 = blockIdx.x * 128, r0 = blockIdx.y * 32;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < ld; k0 += AS_KC) {
        // stage A: 32 rows x 32 k (one float4 per thread), B: 128 rows x 32 k (four float4 per thread)
        {
            const int r = tid >> 3, kk = (tid & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + r < n_rows && k0 + kk < ld) {
                const int node = rows ? rows[r0 + r] : r0 + r;
                v = *(const float4 *)(E + (int64_t)node * ld + k0 + kk);
            }
            As[r][kk] = v.x; As[r][kk + 1] = v.y; As[r][kk + 2] = v.z; As[r][kk + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 3) + 32 * i, kk = (tid & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 + r < n_node && k0 + kk < ld) v = *(const float4 *)(E + (int64_t)(c0 + r) * ld + k0 + kk);
            Bs[r][kk] = v.x; Bs[r][kk + 1] = v.y; Bs[r][kk + 2] = v.z; Bs[r][kk + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < AS_KC; kk += 2) {
            const float a = As[lane & 31][kk + (lane >> 5)];
            const float b = Bs[wv * 32 + (lane & 31)][kk + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int col = c0 + wv * 32 + (lane & 31);
    if (col < n_node) {
        const float bj = bias[col];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = r0 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (row < n_rows) out[(int64_t)row * n_node + col] = acc[reg] + bj;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Streamed all-pairs consumer: the rows of S = E . E^T + b are produced tile by tile on the matrix cores and CONSUMED
// in registers -- per requested row the maximum, its column (argmax) and log sum_j exp(S[i, j]), the normaliser of the
// full softmax over all nodes that the graph softmax of the walk sampler approximates -- so nothing of size N^2 (or
// n_rows x N) ever exists.  At N = 10^7 a materialised row block would be 40 MB per row; the stream needs 12 bytes.
//   fp32 (default): v_mfma_f32_32x32x2_f32, exact fp32 scores (the same arithmetic as gg_all_score);
//   bf16 (optional): v_mfma_f32_32x32x16_bf16 on a bf16 copy of the table (round to nearest even), fp32 accumulate:
//                    16x the matrix-core rate, scores differ from fp32 by the input rounding.
// Work split: grid = (column splits, row tiles); a workgroup walks its column range in 128-column tiles, its 4 wavefronts
// take 32 columns each; every lane keeps the running (max, argmax, sum exp) of the 16 (row, column-lane) cells it owns and
// the cross-lane / cross-wave merge happens once per workgroup.  Partials per (row, split) are merged on the host.
struct Running {
    float m, s;
    int arg;
};

__device__ __forceinline__ void run_update(Running &r, float x, int col, bool lse) {
    // branch-free: one exponential per score (exp of minus the distance to the new maximum), selects instead of branches
    const bool up = x > r.m;
    const float nm = up ? x : r.m;
    if (lse) {
        const float e = __expf((up ? r.m : x) - nm);  // exp(-inf) = 0 covers the first score of a cell
        r.s = up ? r.s * e + 1.0f : r.s + e;
    }
    r.arg = up ? col : r.arg;
    r.m = nm;
}

__device__ __forceinline__ void run_merge(Running &a, float m, float s, int arg, bool lse) {
    if (m == -INFINITY) return;  // the other cell saw no column at all
    if (a.m == -INFINITY) { a.m = m; a.s = s; a.arg = arg; return; }
    if (m > a.m || (m == a.m && arg < a.arg)) {
        if (lse) a.s = a.s * __expf(a.m - m) + s;
        a.m = m;
        a.arg = arg;
    } else if (lse) {
        a.s += s * __expf(m - a.m);
    }
}

// merge the 16 per-lane cells of a wave over its 32 column lanes (lanes l and l ^ 32 hold different rows) and the
// 4 waves of the workgroup through LDS; rows of the tile: row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
template <int NR>
__device__ __forceinline__ void tile_finish(Running (&run)[NR], bool lse, int rows_base, int n_rows, int split, int n_splits, float *part_max,
                                            int32_t *part_arg, float *part_sum, float (*sh_m)[32 * (NR / 16)], float (*sh_s)[32 * (NR / 16)],
                                            int (*sh_a)[32 * (NR / 16)]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
            const float m = __shfl_xor(run[i].m, off, 64), sx = __shfl_xor(run[i].s, off, 64);
            const int ar = __shfl_xor(run[i].arg, off, 64);
            run_merge(run[i], m, sx, ar, lse);
        }
        if ((lane & 31) == 0) {
            const int rb = i / 16, reg = i % 16;
            const int row = rb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            sh_m[wv][row] = run[i].m;
            sh_s[wv][row] = run[i].s;
            sh_a[wv][row] = run[i].arg;
        }
    }
    __syncthreads();
    const int tr = threadIdx.x;
    if (tr < 32 * (NR / 16) && rows_base + tr < n_rows) {
        Running r{sh_m[0][tr], sh_s[0][tr], sh_a[0][tr]};
#pragma unroll
        for (int w = 1; w < 4; ++w) run_merge(r, sh_m[w][tr], sh_s[w][tr], sh_a[w][tr], lse);
        const int64_t o = (int64_t)(rows_base + tr) * n_splits + split;
        part_max[o] = r.m;
        part_arg[o] = r.arg;
        part_sum[o] = r.s;
    }
}

__global__ __launch_bounds__(256) void all_score_reduce_f32_kernel(const float *E, const float *bias, int n_node, int ld, const int32_t *rows,
                                                                   int n_rows, int cols_per_split, int lse, float *part_max, int32_t *part_arg,
                                                                   float *part_sum) {
    extern __shared__ float As_all[];  // [32][ld + 1]: the tile's 32 rows, staged once
    __shared__ float Bs[128][AS_KC + 1];
    __shared__ float sh_m[4][32], sh_s[4][32];
    __shared__ int sh_a[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int split = blockIdx.x, r0 = blockIdx.y * 32;
    const int lda = ld + 1;
    for (int i = tid; i < 32 * (ld / 4); i += 256) {
        const int r = i / (ld / 4), kk = (i % (ld / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + r < n_rows) {
            const int node = rows ? rows[r0 + r] : r0 + r;
            v = *(const float4 *)(E + (int64_t)node * ld + kk);
        }
        float *d = As_all + r * lda + kk;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    Running run[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) run[i] = Running{-INFINITY, 0.f, 0x7fffffff};
    const int cbeg = split * cols_per_split, cend = min(n_node, cbeg + cols_per_split);
    for (int c0 = cbeg; c0 < cend; c0 += 128) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int k0 = 0; k0 < ld; k0 += AS_KC) {
            __syncthreads();  // also orders the A staging before its first use
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (tid >> 3) + 32 * i, kk = (tid & 7) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c0 + r < cend && k0 + kk < ld) v = *(const float4 *)(E + (int64_t)(c0 + r) * ld + k0 + kk);
                Bs[r][kk] = v.x; Bs[r][kk + 1] = v.y; Bs[r][kk + 2] = v.z; Bs[r][kk + 3] = v.w;
            }
            __syncthreads();
            const int kmax = min(AS_KC, ld - k0);
            for (int kk = 0; kk < kmax; kk += 2) {
                const float a = As_all[(lane & 31) * lda + k0 + kk + (lane >> 5)];
                const float b = Bs[wv * 32 + (lane & 31)][kk + (lane >> 5)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
        const int col = c0 + wv * 32 + (lane & 31);
        if (col < cend) {
            const float bj = bias[col];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) run_update(run[reg], acc[reg] + bj, col, lse != 0);
        }
    }
    __syncthreads();
    tile_finish<16>(run, lse != 0, r0, n_rows, split, gridDim.x, part_max, part_arg, part_sum, sh_m, sh_s, sh_a);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// fp32 table -> bf16 copy (round to nearest even, k zero padded to ld16 = 16 KS, rows zero padded to a multiple of 32), TILED for
// the matrix instruction's B operand: the 16-byte piece {k = 16 s + 8 h .. + 8} of row r sits at piece index
//     ((r / 32) * KS + s) * 64 + 32 h + r % 32,
// i.e. the 64 lanes of a wavefront that loads k-step s of a 32-column tile (lane = 32 h + column) read ONE CONTIGUOUS KILOBYTE.
// (Row-major, every such load touched 32 different cache lines for 32 bytes each: 4 wavefronts x 16 loads x 32 line look-ups per
// tile and CU were what the wide consumer waited for, not the HBM and not the matrix pipe.)
__device__ __forceinline__ int64_t bf16_piece(int64_t r, int s, int h, int KS) { return ((r >> 5) * KS + s) * 64 + 32 * h + (r & 31); }

__global__ void to_bf16_kernel(const float *E, int64_t n, int64_t n_pad, int ld, int ld16, __bf16 *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad * ld16) return;
    const int64_t r = i / ld16;
    const int k = (int)(i % ld16);
    const float v = (r < n && k < ld) ? E[r * ld + k] : 0.0f;
    out[bf16_piece(r, k >> 4, (k >> 3) & 1, ld16 / 16) * 8 + (k & 7)] = (__bf16)v;
}

// KS = ld16 / 16 k-steps, RB row blocks of 32 rows per workgroup.  No LDS for the operands: the A fragments of the
// tile's rows stay in registers for the whole column sweep, every lane streams the 8-element k-slices of its own column
// straight from the bf16 table (16 bytes per load; the two half-waves read the two halves of a 32-byte piece).
template <int KS, int RB>
__global__ __launch_bounds__(256) void all_score_reduce_bf16_kernel(const uint4 *Eb, const float *bias, int n_node, const int32_t *rows, int n_rows,
                                                                    int cols_per_split, int lse, float *part_max, int32_t *part_arg, float *part_sum) {
    __shared__ float sh_m[4][32 * RB], sh_s[4][32 * RB];
    __shared__ int sh_a[4][32 * RB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5;
    const int split = blockIdx.x, r0 = blockIdx.y * (32 * RB);
    union Frag { uint4 u; bf16x8 v; };
    Frag afrag[RB][KS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int r = r0 + rb * 32 + (lane & 31);
        const int node = r < n_rows ? (rows ? rows[r] : r) : -1;
#pragma unroll
        for (int s = 0; s < KS; ++s) afrag[rb][s].u = node >= 0 ? Eb[bf16_piece(node, s, half, KS)] : make_uint4(0u, 0u, 0u, 0u);
    }
    Running run[16 * RB];
#pragma unroll
    for (int i = 0; i < 16 * RB; ++i) run[i] = Running{-INFINITY, 0.f, 0x7fffffff};
    const int cbeg = split * cols_per_split, cend = min(n_node, cbeg + cols_per_split);
    // B fragments are double buffered in registers: the KS loads of the NEXT 32-column tile are issued before the matrix
    // instructions and the consumer of the current one, so that memory latency hides behind them (two waves per SIMD).
    Frag bcur[KS], bnxt[KS];
    auto load_tile = [&](Frag (&dst)[KS], int c0t) {
        // (tiles start at multiples of 32 columns; the padded rows behind the table's end are zeros; a prefetch behind the
        // split's end re-reads its first tile)
        const uint4 *brow = Eb + (int64_t)((c0t < cend ? c0t : cbeg) >> 5) * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) dst[s].u = brow[64 * s];
    };
    load_tile(bcur, cbeg + wv * 32);
    for (int c0 = cbeg + wv * 32; c0 < cend; c0 += 128) {
        const int col = c0 + (lane & 31);
        const bool ok = col < cend;
        load_tile(bnxt, c0 + 128);
        f32x16 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[rb][s].v, bcur[s].v, acc[rb], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) bcur[s] = bnxt[s];
        if (ok) {
            const float bj = bias[col];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) run_update(run[rb * 16 + reg], acc[rb][reg] + bj, col, lse != 0);
        }
    }
    tile_finish<16 * RB>(run, lse != 0, r0, n_rows, split, gridDim.x, part_max, part_arg, part_sum, sh_m, sh_s, sh_a);
}

// The same consumer for MANY rows: the 4 wavefronts of a workgroup take 4 DIFFERENT row blocks (RB x 32 rows each) and walk the
// SAME 32-column tiles, so one sweep of the table serves 128 x RB rows instead of 32 x RB -- in the variant above every
// 32 x RB rows re-stream the whole bf16 table from HBM (4 096 rows x 10^7 nodes: 128 sweeps of 5 GB = 8.3 TB/s at 266 TFLOP/s:
// it ran at the HBM roofline, not the matrix cores').  The four wavefronts load the same B fragments within a few cycles
// of each other: one of them misses, the others hit the CU's vector cache.  No cross-wave merge: a wave owns its rows.
// With one wavefront per SIMD the kernel runs at (bytes in flight) / (memory latency) until the matrix cores saturate: one
// 16 KB tile per wave at d = 256 -- two row blocks per wave (RB = 2) double the work per byte in flight.
// Round 3: (a) ONE set of B fragments -- the k-slice s of the next tile is loaded into b[s] right behind the matrix instructions
// that read the current tile's b[s]: a rolling prefetch one tile deep without the 64 register copies per tile of a double
// buffer; (b) a 7-instruction consumer instead of ~12: the log-sum-exp is accumulated WITHOUT a running reference,
// s += exp2(x log2 e) -- one fused multiply-add and one exponential per score, no rescale, no select -- and brought to the
// (max, sum exp(x - max)) form once, at the end; max / argmax compare the scores themselves (bit-identical to the other
// kernels).  The reference-free sum is finite and accurate while the scores stay inside (-85, 85) -- embedding dot products are
// a few units --; a cell whose sum overflows, or underflows to 0, raises `overflow` and the host repeats the call with the
// narrow kernel above (running maximum).
template <int KS, int RB, bool LSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void all_score_reduce_bf16_rows_kernel(const uint4 *Eb, const float *bias, int n_node, const int32_t *rows,
                                                                         int n_rows, int cols_per_split, float *part_max,
                                                                         int32_t *part_arg, float *part_sum, int32_t *overflow) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5;
    const int split = blockIdx.x, r0 = blockIdx.y * (128 * RB) + wv * (32 * RB);
    union Frag { uint4 u; bf16x8 v; };
    Frag afrag[RB][KS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int r = r0 + rb * 32 + (lane & 31);
        const int node = r < n_rows ? (rows ? rows[r] : r) : -1;
#pragma unroll
        for (int s = 0; s < KS; ++s) afrag[rb][s].u = node >= 0 ? Eb[bf16_piece(node, s, half, KS)] : make_uint4(0u, 0u, 0u, 0u);
    }
    // (scalar arrays instead of one array of structs: 64 x 12 bytes is more than the compiler promotes to registers)
    float rm[RB][16], rs[RB][16];
    int ra[RB][16];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) { rm[rb][i] = -INFINITY; rs[rb][i] = 0.f; ra[rb][i] = 0x7fffffff; }
    const int cbeg = split * cols_per_split, cend = min(n_node, cbeg + cols_per_split);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    union FragB { u32x4 q; bf16x8 v; };
    // The B fragments are loaded and awaited BY HAND (inline assembly): where these loads sit decides everything here -- one
    // wavefront per SIMD, nothing to switch to -- and the compiler moved them with every edit of the loop (to the top of the next
    // iteration = no prefetch at all; into a second register set = 64 copies per tile; behind a bias load whose wait drained
    // the whole queue).  The order is: slice s of the NEXT tile is requested into b[s] right behind the matrix instructions
    // that read the current b[s]; the memory counter retires in order and every request has exactly 15 younger ones when its
    // slice is needed, so `s_waitcnt vmcnt(15)` in front of a slice's matrix instructions is exact.  Each b[s] is a
    // read-write operand of both statements: it stays in one physical register quadruple, and neither the matrix
    // instructions nor a register copy can move across the wait.  (The bias refill below is the only compiler-issued
    // load inside the loop; its compiler-placed wait drains the queue: over-waiting is safe.)
    auto tile_of = [&](int c0t) -> const char * {  // (tiles start at multiples of 32 columns; the prefetch behind the split's end re-reads its first tile)
        return (const char *)(Eb + (int64_t)((c0t < cend ? c0t : cbeg) >> 5) * KS * 64);
    };
    constexpr float LOG2E = 1.44269504088896341f;
    int voff[(KS + 3) / 4];  // byte offset of this lane's 16 bytes in slices 4k .. 4k+3 (the instruction's immediate reaches 4 KB)
#pragma unroll
    for (int k = 0; k < (KS + 3) / 4; ++k) voff[k] = lane * 16 + 4096 * k;
    FragB b[KS];
#define GG_B_LOAD(S, BASE, CONSTRAINT)                                                                                            \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : CONSTRAINT(b[S].q) : "v"(voff[(S) >> 2]), "s"(BASE), "n"(((S) & 3) * 1024))
    {
        // (the A fragments are used once here, so that their compiler-placed wait stands in front of the loop and not -- merged over
        // the back edge -- as a drain of the queue inside every iteration)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int s = 0; s < KS; ++s) asm volatile("" ::"v"(afrag[rb][s].u.x), "v"(afrag[rb][s].u.w));
        const char *const first = tile_of(cbeg);
#pragma unroll
        for (int s = 0; s < KS; ++s) GG_B_LOAD(s, first, "=v");
    }
    // The bias of the columns comes through LDS, 2 048 columns at a time and private to the wave: a global bias load per tile sits
    // in the same in-order memory counter as the B prefetch.  An LDS read has its own counter; the refill stalls once per 64 tiles.
    constexpr int BIAS_CHUNK = 2048;
    __shared__ float bias_lds[4][BIAS_CHUNK];
    float *const wb = bias_lds[wv];
    // Measured and not kept (round 2): three B buffers (two tiles in flight) with the loop unrolled over them -- the unrolled
    // consumer bodies cost more than the extra tile in flight brings (390 -> 309 TFLOP/s).
    for (int c0 = cbeg; c0 < cend; c0 += 32) {
        const int col = c0 + (lane & 31);
        const int within = (c0 - cbeg) & (BIAS_CHUNK - 1);
        if (within == 0) {
#pragma unroll
            for (int i = 0; i < BIAS_CHUNK / 64; ++i) {
                const int c = c0 + i * 64 + lane;
                wb[i * 64 + lane] = c < cend ? bias[c] : -INFINITY;
            }
        }
        const float bj = wb[within + (lane & 31)];
        const char *const nxt = tile_of(c0 + 32);
        f32x16 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(b[s].q) : "n"(KS - 1));
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[rb][s].v, b[s].v, acc[rb], 0, 0, 0);
            GG_B_LOAD(s, nxt, "+v");
        }
        // (a column behind the split's end: bias -inf -> score -inf, its exponential 0, the compare false: no branch)
        const float bj2 = bj * LOG2E;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float a = acc[rb][reg];
                const float x = a + bj;
                if (LSE) rs[rb][reg] += __builtin_amdgcn_exp2f(__builtin_fmaf(a, LOG2E, bj2));
                const bool up = x > rm[rb][reg];
                rm[rb][reg] = up ? x : rm[rb][reg];
                ra[rb][reg] = up ? col : ra[rb][reg];
            }
    }
    // (the last tile's prefetch is still in flight INTO b[]: drain it before the registers are anyone else's)
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[s].q));
#undef GG_B_LOAD
    // (max, sum exp(x - max)) per cell; merge a row's 32 column lanes; lanes 0 and 32 then hold the wave's rows:
    // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            float sx = rs[rb][reg];
            if (LSE) {
                // inf / nan, or 0 although the cell saw a column: the scores left the range a reference-free sum can hold
                if (!(sx <= 3.0e38f) || (sx == 0.f && rm[rb][reg] > -INFINITY)) atomicOr(overflow, 1);
                sx = rm[rb][reg] > -INFINITY ? sx * __builtin_amdgcn_exp2f(-rm[rb][reg] * LOG2E) : 0.f;
            }
            Running x{rm[rb][reg], sx, ra[rb][reg]};
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {
                const float m = __shfl_xor(x.m, off, 64), so = __shfl_xor(x.s, off, 64);
                const int ar = __shfl_xor(x.arg, off, 64);
                run_merge(x, m, so, ar, LSE);
            }
            if ((lane & 31) == 0) {
                const int row = r0 + rb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                if (row < n_rows) {
                    const int64_t o = (int64_t)row * gridDim.x + split;
                    part_max[o] = x.m;
                    part_arg[o] = x.arg;
                    part_sum[o] = x.s;
                }
            }
        }
    }
}

}  // namespace gg

using namespace gg;

extern "C" int gg_all_score(gg_ctx *ctx, const int32_t *rows, int32_t n_rows, float *out) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, out && n_rows >= 0, GG_EINVAL, "gg_all_score: bad argument");
    if (!rows) n_rows = ctx->n_node;
    if (n_rows == 0) return GG_OK;
    const int n = ctx->n_node;
    if (rows)
        for (int i = 0; i < n_rows; ++i) GG_CHECK(ctx, rows[i] >= 0 && rows[i] < n, GG_EINVAL, "gg_all_score: row id %d out of range", rows[i]);
    GG_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf d_rows, d_out;
    hipError_t e = d_out.reserve(sizeof(float) * (size_t)n_rows * n);
    if (e == hipSuccess && rows) e = d_rows.reserve(sizeof(int32_t) * n_rows);
    if (e != hipSuccess) {
        d_rows.release(); d_out.release();
        return fail(ctx, GG_ENOMEM, "gg_all_score: %d x %d fp32 does not fit (%s)", n_rows, n, hipGetErrorString(e));
    }
    if (rows) (void)hipMemcpyAsync(d_rows.p, rows, sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, ctx->stream);
    const Model &G = ctx->model[0];
    hipLaunchKernelGGL(all_score_kernel, dim3(cdiv(n, 128), cdiv(n_rows, 32)), dim3(256), 0, ctx->stream, G.E, G.b, n, ctx->ld,
                       rows ? d_rows.as<int32_t>() : nullptr, n_rows, d_out.as<float>());
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out.p, sizeof(float) * (size_t)n_rows * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    d_rows.release();
    d_out.release();
    if (e != hipSuccess) return fail(ctx, GG_EHIP, "gg_all_score: %s", hipGetErrorString(e));
    return GG_OK;
}

// gg_all_score_reduce: see include/graphgan_hip.h.
extern "C" int gg_all_score_reduce(gg_ctx *ctx, const int32_t *rows, int32_t n_rows, int32_t precision, int32_t want_lse, float *row_max,
                                   int32_t *row_argmax, float *row_lse, double *kernel_ms_out) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, n_rows >= 0 && row_max && row_argmax && (row_lse || !want_lse), GG_EINVAL, "gg_all_score_reduce: bad argument");
    GG_CHECK(ctx, precision == 0 || precision == 1, GG_EINVAL, "gg_all_score_reduce: precision must be 0 (fp32) or 1 (bf16)");
    if (!rows) n_rows = ctx->n_node;
    if (n_rows == 0) return GG_OK;
    const int n = ctx->n_node, ld = ctx->ld;
    if (rows)
        for (int i = 0; i < n_rows; ++i) GG_CHECK(ctx, rows[i] >= 0 && rows[i] < n, GG_EINVAL, "gg_all_score_reduce: row id %d out of range", rows[i]);
    GG_HIP(ctx, hipSetDevice(ctx->device));
    // k-steps of 16 bf16 elements; the kernel is instantiated for 4 / 8 / 16 / 32 of them: the copy is zero padded to that
    const int ks_need = (ctx->n_emb + 15) / 16;
    const int KS = ks_need <= 4 ? 4 : ks_need <= 8 ? 8 : ks_need <= 16 ? 16 : 32;
    const int ld16 = 16 * KS;
    const int RB = (precision == 1 && KS <= 8) ? 2 : 1;
    // bf16, many rows: a workgroup's four wavefronts take four row blocks and share the table sweep (all_score_reduce_bf16_rows_kernel)
    static thread_local bool force_narrow = false;  // set for the repeat of a call whose wide kernel reported an overflowing sum
    const bool wide = precision == 1 && n_rows >= 512 && !getenv("GG_ALLPAIRS_NARROW") && !force_narrow;
    const int RBW = KS <= 8 ? 4 : (KS <= 16 ? 2 : 1);  // row blocks per wavefront of the wide kernel: as many as the registers of ONE wave per SIMD hold
    const int tile_rows = wide ? 128 * RBW : 32 * RB;
    const int row_tiles = cdiv(n_rows, tile_rows);
    // enough workgroups for the chip: split the columns when there are few row tiles (multiples of 128 columns)
    int splits = std::max(1, std::min(cdiv(n, 128), cdiv(wide ? 1024 : 2048, row_tiles)));
    int cps = cdiv(cdiv(n, splits), 128) * 128;
    splits = cdiv(n, cps);
    DevBuf d_rows, d_pm, d_pa, d_ps, d_bf, d_ovf;
    auto rel = [&]() { d_rows.release(); d_pm.release(); d_pa.release(); d_ps.release(); d_bf.release(); d_ovf.release(); };
    const size_t np = (size_t)n_rows * splits;
    hipError_t e = d_pm.reserve(sizeof(float) * np);
    if (e == hipSuccess) e = d_pa.reserve(sizeof(int32_t) * np);
    if (e == hipSuccess) e = d_ps.reserve(sizeof(float) * np);
    if (e == hipSuccess && rows) e = d_rows.reserve(sizeof(int32_t) * n_rows);
    const int64_t n_pad = ((int64_t)n + 31) / 32 * 32;  // the bf16 copy is tiled by 32 rows
    if (e == hipSuccess && precision == 1) e = d_bf.reserve(sizeof(uint16_t) * (size_t)n_pad * ld16);
    if (e == hipSuccess) e = d_ovf.reserve(sizeof(int32_t) * 4);
    if (e != hipSuccess) { rel(); return fail(ctx, GG_ENOMEM, "gg_all_score_reduce: %s", hipGetErrorString(e)); }
    (void)hipMemsetAsync(d_ovf.p, 0, sizeof(int32_t) * 4, ctx->stream);
    if (rows) (void)hipMemcpyAsync(d_rows.p, rows, sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, ctx->stream);
    const Model &G = ctx->model[0];
    const int32_t *dr = rows ? d_rows.as<int32_t>() : nullptr;
    const dim3 grid(splits, row_tiles);
    if (precision == 1) {
        const int64_t tot = n_pad * ld16;
        hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, G.E, (int64_t)n, n_pad, ld, ld16, (__bf16 *)d_bf.p);
    }
    (void)hipEventRecord(ctx->ev0, ctx->stream);
    if (precision == 0) {
        const size_t dyn = sizeof(float) * 32 * (size_t)(ld + 1);
        if (dyn > 48 * 1024) (void)hipFuncSetAttribute((const void *)all_score_reduce_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL(all_score_reduce_f32_kernel, grid, dim3(256), dyn, ctx->stream, G.E, G.b, n, ld, dr, n_rows, cps, want_lse,
                           d_pm.as<float>(), d_pa.as<int32_t>(), d_ps.as<float>());
    } else {
        const uint4 *Eb = (const uint4 *)d_bf.p;
#define GG_BF16_LAUNCH(KSV, RBV)                                                                                                    \
    hipLaunchKernelGGL((all_score_reduce_bf16_kernel<KSV, RBV>), grid, dim3(256), 0, ctx->stream, Eb, G.b, n, dr, n_rows, cps, want_lse, \
                       d_pm.as<float>(), d_pa.as<int32_t>(), d_ps.as<float>())
#define GG_BF16_ROWS(KSV, RBV)                                                                                                                            \
    do {                                                                                                                                                 \
        if (want_lse) hipLaunchKernelGGL((all_score_reduce_bf16_rows_kernel<KSV, RBV, true>), grid, dim3(256), 0, ctx->stream, Eb, G.b, n, dr, n_rows, cps, \
                                         d_pm.as<float>(), d_pa.as<int32_t>(), d_ps.as<float>(), d_ovf.as<int32_t>());                                     \
        else hipLaunchKernelGGL((all_score_reduce_bf16_rows_kernel<KSV, RBV, false>), grid, dim3(256), 0, ctx->stream, Eb, G.b, n, dr, n_rows, cps,        \
                                d_pm.as<float>(), d_pa.as<int32_t>(), d_ps.as<float>(), d_ovf.as<int32_t>());                                             \
    } while (0)
        if (wide) {
            if (KS <= 4) { GG_BF16_ROWS(4, 4); }
            else if (KS <= 8) { GG_BF16_ROWS(8, 4); }
            else if (KS <= 16) { GG_BF16_ROWS(16, 2); }
            else { GG_BF16_ROWS(32, 1); }
        } else if (KS <= 4) { GG_BF16_LAUNCH(4, 2); }
        else if (KS <= 8) { GG_BF16_LAUNCH(8, 2); }
        else if (KS <= 16) { GG_BF16_LAUNCH(16, 1); }
        else { GG_BF16_LAUNCH(32, 1); }
#undef GG_BF16_LAUNCH
#undef GG_BF16_ROWS
    }
    (void)hipEventRecord(ctx->ev1, ctx->stream);
    std::vector<float> pm(np), ps(np);
    std::vector<int32_t> pa(np);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(pm.data(), d_pm.p, sizeof(float) * np, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pa.data(), d_pa.p, sizeof(int32_t) * np, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ps.data(), d_ps.p, sizeof(float) * np, hipMemcpyDeviceToHost, ctx->stream);
    int32_t h_ovf = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_ovf, d_ovf.p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && h_ovf && wide) {  // scores outside the range of the wide kernel's reference-free sum: running-max kernel
        rel();
        force_narrow = true;
        const int rc = gg_all_score_reduce(ctx, rows, rows ? n_rows : 0, precision, want_lse, row_max, row_argmax, row_lse, kernel_ms_out);
        force_narrow = false;
        return rc;
    }
    float ms = 0.f;
    if (e == hipSuccess) (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    rel();
    if (e != hipSuccess) return fail(ctx, GG_EHIP, "gg_all_score_reduce: %s", hipGetErrorString(e));
    if (kernel_ms_out) *kernel_ms_out = ms;
    for (int r = 0; r < n_rows; ++r) {  // merge the column splits (ascending columns: the first maximum wins, like numpy argmax)
        float M = -INFINITY;
        int arg = 0x7fffffff;
        for (int sp = 0; sp < splits; ++sp) {
            const size_t o = (size_t)r * splits + sp;
            if (pm[o] > M || (pm[o] == M && pa[o] < arg)) { M = pm[o]; arg = pa[o]; }
        }
        double S = 0.0;
        if (want_lse)
            for (int sp = 0; sp < splits; ++sp) {
                const size_t o = (size_t)r * splits + sp;
                if (pm[o] > -INFINITY) S += (double)ps[o] * exp((double)pm[o] - (double)M);
            }
        row_max[r] = M;
        row_argmax[r] = arg;
        if (want_lse) row_lse[r] = (float)((double)M + log(S));
    }
    return GG_OK;
}
