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
        // (the accumulators START at the column's bias -- S = b + sum_k, the order the x32 kernel below computes in: same bits)
        const float bj = ok ? bias[col] : 0.f;
        f32x16 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][i] = bj;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[rb][s].v, bcur[s].v, acc[rb], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) bcur[s] = bnxt[s];
        if (ok) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) run_update(run[rb * 16 + reg], acc[rb][reg], col, lse != 0);
        }
    }
    tile_finish<16 * RB>(run, lse != 0, r0, n_rows, split, gridDim.x, part_max, part_arg, part_sum, sh_m, sh_s, sh_a);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The same consumer for MANY rows (>= 512), round 6: v_mfma_f32_32x32x16_bf16 with the REQUESTED ROWS ON THE LANES.  The matrix
// instruction's A operand is the table tile (32 nodes), its B operand a block of 32 requested rows, so a lane's 16 accumulator
// registers are 16 NODES of ONE requested row (row = lane & 31; nodes (i & 3) + 8 (i >> 2) + 4 (lane >> 5) of the tile): the
// running state of a row block is one (max, tile, sum) cell per lane instead of 16, and the consumer is per score
//     y = x log2 e;  s += exp2(y);  half a v_max3
// -- 3.5 vector instructions (rounds 3-5's kernel on v_mfma_f32_16x16x32_bf16, rows on the registers: 9.2, SQ_INSTS_VALU / scores
// in profiles/r5_pmc_k7_sq.json; that kernel and its hand-placed loads are in profiles/HISTORY.md), which is about what fits
// beside a 32-cycle matrix instruction (MI355X_MICROARCH.md: <= 5 single-issue instructions hidden per
// v_mfma_f32_32x32x16_bf16 and wavefront).  What makes that possible:
//  - the bias of a node is the INITIAL VALUE of its accumulator register (16-byte LDS reads straight into the accumulator
//    tuple), not an addition per score; the narrow kernel above starts its accumulators at the bias as well: same bits;
//  - max / argmax: the tile's 16 scores of a row fold into one maximum (v_max3 chain), and a row block's running cell keeps
//    (max, TILE) by one compare and two selects per tile; WHICH node of the winning tile is found after the sweep by
//    multiplying that tile once more (below) -- the sweep has no branch;
//  - the log-sum-exp is accumulated WITHOUT a running reference, s += exp2(x log2 e), and brought to the (max, sum exp(x - max))
//    form once, at the end: finite and accurate while the scores stay inside (-85, 85) -- embedding dot products are a few
//    units --; a cell whose sum overflows, or underflows to 0, raises `overflow` and the host repeats the call with the
//    narrow kernel above (running maximum per cell).
// Operands: a wavefront keeps RB blocks of 32 requested rows as B fragments in ACCUMULATION registers for the whole sweep
// (RB KS 4 registers: 128 at d = 256 with RB = 2); the table tile (KS KB) is staged ONCE per workgroup through LDS -- every
// wavefront requests 1 / NW of tile t + 3 by LDS-DMA in the middle of iteration t; four buffers; one bare s_barrier per tile
// -- and read from there as A fragments, one k-step ahead of its matrix instructions.  NW = 8 wavefronts (two per SIMD) of
// RB = 2 row blocks: a workgroup sweeps its column range once for 512 rows (8 sweeps of the bf16 table for 4 096 rows; the row
// tiles of one column split sit on ONE XCD -- workgroup id mod 8 = split mod 8 -- and share its L2); one workgroup per
// compute unit, 256 in all.  (One wavefront per SIMD with RB = 4, GG_K7_NW4: the same without the log-sum-exp, 3 % slower with it.)
// Pipeline: iteration t multiplies tile t into accumulator set t & 1 and consumes tile t - 1 from the other set; a consumed
// quad's registers are re-initialised with the bias of tile t + 1 (a ring of 4 bias tiles in LDS).  Tiles outside the split
// have a bias of -inf: x = -inf, exp2 = 0, never a maximum -- the pipeline's first and last iterations need no special case.
// Measured, 4 096 rows x 10^7 nodes x d = 256 (ms per call): 15.6 - 16.0 with the log-sum-exp (the 16x16x32 kernel: 21.2),
// 14.3 - 14.6 without; its matrix instructions alone (ablation 27: no staging, no LDS reads) 11.2 -- the chip sustains ~1.9 GHz
// under this load (SQ_BUSY_CYCLES / duration), at which 6.4 10^8 matrix instructions x 32 cycles are 10.5 ms.
// DBG (builds with -DGG_K7_ABLATIONS only; results WRONG): 1 no barrier / staging, 2 no table loads, 4 no argmax resolution, 8 no
// fragment reads, 16 no bias reads, 128 staging through registers instead of LDS-DMA
template <int KS, int RB, int NW, bool LSE, int DBG = 0>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void all_score_reduce_bf16_x32_kernel(
    const uint4 *Eb, const float *bias, int n_node, const int32_t *rows, int n_rows, int cols_per_split, float *part_max, int32_t *part_arg,
    float *part_sum, int32_t *overflow) {
    extern __shared__ __attribute__((aligned(16))) unsigned char x32_lds[];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    union Frag { u32x4 q; bf16x8 v; };
    constexpr int BUF = KS * 1024;                                      // a staged tile; behind the four of them a ring of 4 bias tiles
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int split = blockIdx.x, r0 = blockIdx.y * (32 * RB * NW) + wv * (32 * RB);
    constexpr int G = KS >= NW ? KS / NW : 1;  // k-steps of a tile each wavefront stages (KS < NW: the first KS wavefronts one each)
    const bool stager = KS >= NW || wv < KS;
    constexpr float LOG2E = 1.44269504088896341f;
    const int cbeg = split * cols_per_split, cend = min(n_node, cbeg + cols_per_split);
    const int n_tiles = (cend - cbeg + 31) >> 5;
    // requested rows: lane (row l31, k-half hi) of k-step s = the 16-byte piece the narrow kernel loads as its A fragment
    Frag rf[RB][KS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int r = r0 + 32 * rb + l31;
        const int node = r < n_rows ? (rows ? rows[r] : r) : -1;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            rf[rb][s].q = node >= 0 ? *(const u32x4 *)&Eb[bf16_piece(node, s, hi, KS)] : z;
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+a"(rf[rb][s].q));  // (into ACCUMULATION registers: a matrix instruction reads B from either file)
    float rm[RB], rs[RB], tmax[RB];
    int rt[RB];  // the TILE that holds the row's maximum; which of its nodes is found after the sweep (below)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { rm[rb] = -INFINITY; rs[rb] = 0.f; rt[rb] = -1; tmax[rb] = -INFINITY; }
    // Staging.  In the MIDDLE of iteration t, right behind the iteration's one barrier, every wavefront requests its k-steps
    // (wv G .. wv G + G) of tile t + 3 straight into LDS (global_load_lds_dwordx4: no registers, no store instructions) -- into
    // the buffer tile t - 1 has left: every wavefront that passed the barrier has finished iteration t - 1.  The tile is
    // published by the barrier of iteration t + 1 (each wavefront waits for its own requests in front of it: a whole iteration
    // old) and first read at the end of iteration t + 2.  Four buffers: tiles t, t + 1, t + 2 (in flight), t + 3.
    // The barrier is a bare s_barrier in the middle of the matrix instructions (a barrier + register-staged store at the END of an
    // iteration, with the drain of the LDS queue in front of it, cost 3.1 of 17.5 ms: the matrix pipe idles through it).
    // The bias of tile t + 4 (thread tid < 32: node tid) is requested in iteration t and stored into ring slot (t + 4) & 3 in
    // iteration t + 1; it initialises accumulators from the beginning of iteration t + 3 on.
    auto tile_src = [&](int t) -> const u32x4 * {  // (a tile behind the split re-reads its first one: finite numbers, bias -inf)
        return (const u32x4 *)Eb + ((int64_t)((cbeg >> 5) + (t < n_tiles ? t : 0)) * KS + (stager ? wv * G : 0)) * 64 + lane;
    };
    auto bias_of = [&](int t) -> float {
        const int c = cbeg + 32 * t + tid;
        return (tid < 32 && t < n_tiles && c < cend) ? bias[c] : -INFINITY;
    };
    constexpr bool DMA = (DBG & 128) == 0;  // LDS-DMA requests (default) or registers (measured: the same speed, 8 G registers more)
    u32x4 g[G];
    float gb;
    unsigned char *b_cur = x32_lds, *b_nxt = x32_lds + BUF, *b_n2 = x32_lds + 2 * BUF, *b_wr = x32_lds + 3 * BUF;  // tiles t .. t + 3
    float *const ring = (float *)(x32_lds + 4 * BUF);
#pragma unroll
    for (int t0 = 0; t0 < 2; ++t0) {
        const u32x4 *src = tile_src(t0);
#pragma unroll
        for (int k = 0; k < G; ++k) g[k] = src[64 * k];
        unsigned char *dst = t0 ? b_nxt : b_cur;
#pragma unroll
        for (int k = 0; k < G; ++k) if (stager) ((u32x4 *)dst)[(wv * G + k) * 64 + lane] = g[k];
    }
    {
        const float b0 = bias_of(0), b1 = bias_of(1), b2 = bias_of(2);
        if (tid < 32) { ring[tid] = b0; ring[32 + tid] = b1; ring[64 + tid] = b2; }
        const u32x4 *src = tile_src(2);
#pragma unroll
        for (int k = 0; k < G; ++k) g[k] = src[64 * k];
#pragma unroll
        for (int k = 0; k < G; ++k) if (stager && DMA) ((u32x4 *)b_n2)[(wv * G + k) * 64 + lane] = g[k];  // (!DMA: iteration 0 stores it)
        gb = bias_of(3);  // (iteration 0 stores it)
    }
    __syncthreads();
    f32x16 acc[2][RB];
    {
        const f32x4 *bq = (const f32x4 *)ring + hi;  // quad q of a lane: nodes 8 q + 4 hi .. + 4
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = bq[2 * q];
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[0][rb][4 * q + j] = b[j]; acc[1][rb][4 * q + j] = -INFINITY; }
            }
    }
    // The loop body is ONE basic block per tile, cut into KS RB SLOTS: one matrix instruction and what fits into its 32-cycle
    // shadow (a wavefront issues in order: whatever sits between two matrix instructions and takes longer than that idles the
    // pipe -- measured, one wavefront per SIMD, max / argmax only, of 15.8 ms: the 4 LDS-DMA requests issued back to back 2.2,
    // the fragment read in FRONT of a k-step's first matrix instruction 1.7, the 4 bias reads of a row block back to back 1.2;
    // the barrier itself: nothing).  So every slot gets at most one memory instruction: slot (s, 0) the next fragment's LDS
    // read, slots (KS/2 + k, 1) the k-th LDS-DMA request, and the consumer is a queue of 16 RB ELEMENTS spread evenly over the
    // slots -- element e (row block e / 16, register i = e % 16 of the finished tile): y = x log2 e, exp2(y); s += the PREVIOUS
    // element's exponential (its latency passes in the other slot); v_max3 behind every second element; behind every fourth the
    // LDS read that starts the quad's four registers at the next tile's bias; behind the 16th the row block's (max, tile) update.
    // The consumer finishes one k-step early (CS slots): a quad's re-initialising read is issued in slot (s, 0) of the k-step AFTER
    // the one that consumed its last element, in front of the fragment read -- both are then RB - 1 matrix instructions old when
    // the next k-step waits for the fragment (the compiler waits with lgkmcnt(0) everywhere in this loop: an LDS-DMA request
    // in flight makes it treat the LDS counter as out of order).
    constexpr int SLOTS = KS * RB, ELEMS = 16 * RB, CS = SLOTS - RB;
    // A fragments: a ring of PD + 1, k-step s + PD requested under the matrix instructions of k-step s (PD + 1 divides KS: the
    // ring's indices are the same in every iteration).  (Measured: PD = 3 against 1 with two matrix instructions per k-step --
    // the same, three alternating runs each; the second wavefront of the SIMD covers the LDS round trip.)
    constexpr int PD = 1, RING = PD + 1;
    Frag af[RING];
#pragma unroll
    for (int s = 0; s < PD; ++s) af[s].q = ((const u32x4 *)b_cur)[64 * s + lane];
    float epend = 0.f;  // exponential of the last element, not yet added (row block RB - 1 at a tile's start)
    float *const ring_dst = tid < 32 ? ring + tid : ring + 128 + tid;  // (every thread stores: no branch in the loop; 128 .. 128 + 64 NW: a dump)
    for (int tt = 0; tt <= n_tiles; tt += 2) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int t = tt + p;
            const u32x4 *const tl = (const u32x4 *)b_cur + lane, *const tl_next = (const u32x4 *)b_nxt + lane;
            const f32x4 *const bq = (const f32x4 *)(ring + 32 * ((t + 1) & 3)) + hi;
            const u32x4 *const dsrc = tile_src(t + 3);
            const int bc = min(cbeg + 32 * (t + 4) + tid, cend - 1);  // bias of tile t + 4, clamped instead of predicated
            const bool bok = tid < 32 && t + 4 < n_tiles && cbeg + 32 * (t + 4) + tid < cend;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int k = 0; k < RB; ++k) {
                    const int slot = s * RB + k;
                    if (s == KS / 2 && k == 0 && !(DBG & 1)) {
                        // (the tile requested one iteration ago has landed: nothing to wait for; the wait makes it formal)
                        if (DMA) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                        else asm volatile("s_barrier" ::: "memory");
                    }
                    acc[p][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s % RING].v, rf[k][s].v, acc[p][k], 0, 0, 0);
                    if (k == 0) {
#pragma unroll
                        for (int e = 3; e < ELEMS; e += 4) {  // quads whose last element was consumed during the previous k-step
                            if (s == 0 || e * CS / ELEMS < (s - 1) * RB || e * CS / ELEMS >= s * RB || (DBG & 16)) continue;
                            const f32x4 b = bq[2 * ((e & 15) >> 2)];
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[p ^ 1][e >> 4][(e & 12) + j] = b[j];
                        }
                        if (!(DBG & 8)) af[(s + PD) % RING].q = s + PD < KS ? tl[64 * (s + PD)] : tl_next[64 * (s + PD - KS)];
                    }
                    if (DMA && k == (RB > 1 ? 1 : 0) && s >= KS / 2 && s < KS / 2 + G && stager && !(DBG & 3)) {
                        // tile t + 3 straight into the buffer tile t - 1 has left; lane l's 16 bytes land at M0 + 16 l
                        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(dsrc + 64 * (s - KS / 2)),
                                                         (void __attribute__((address_space(3))) *)(b_wr + (wv * G + s - KS / 2) * 1024), 16, 0, 0);
                    }
                    if (!DMA && s >= KS / 2 && s < KS / 2 + G && stager && !(DBG & 1)) {
                        // tile t + 2 (requested one iteration ago) into its buffer, and the same registers take tile t + 3
                        if (k == (RB > 1 ? 1 : 0)) ((u32x4 *)b_n2)[(wv * G + s - KS / 2) * 64 + lane] = g[s - KS / 2];
                        if (k == RB - 1 && !(DBG & 2)) g[s - KS / 2] = dsrc[64 * (s - KS / 2)];
                    }
                    if (k == 0 && s == KS / 2 && !(DBG & 1)) ring_dst[32 * ((t + 3) & 3)] = gb;  // bias of tile t + 3 (in front of the DMA requests)
                    if (k == RB - 1 && s == KS / 2 + 1 && !(DBG & 3)) gb = bok ? bias[bc] : -INFINITY;  // ... of tile t + 4
#pragma unroll
                    for (int e = 0; e < ELEMS; ++e) {
                        if (e * CS / ELEMS != slot) continue;
                        const int rb = e >> 4, i = e & 15, rbp = (e + ELEMS - 1) % ELEMS >> 4;
                        f32x16 &c = acc[p ^ 1][rb];
                        if (LSE) {
                            rs[rbp] += epend;
                            epend = __builtin_amdgcn_exp2f(c[i] * LOG2E);
                        }
                        if (i & 1) tmax[rb] = __builtin_fmaxf(__builtin_fmaxf(i == 1 ? -INFINITY : tmax[rb], c[i - 1]), c[i]);
                        if (i == 15) {
                            const bool up = tmax[rb] > rm[rb];  // (strictly: the first tile keeps a tie)
                            rm[rb] = up ? tmax[rb] : rm[rb];
                            rt[rb] = up ? t - 1 : rt[rb];
                        }
                        // (anchor: keeps the chain in its slot -- left alone, the machine-sink pass and the scheduler move pure
                        // chains like s += exp2(..) to their last use)
                        asm volatile("" : "+v"(rs[rbp]), "+v"(epend), "+v"(tmax[rb]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            unsigned char *const o = b_cur;
            b_cur = b_nxt;
            b_nxt = b_n2;
            b_n2 = b_wr;
            b_wr = o;
        }
    }
    if (LSE) rs[RB - 1] += epend;
    // Which node of the winning tile?  The sweep kept no branch for it (a uniform branch per tile and row block that finds the
    // register holding a new maximum cost 1.1 of 17.4 ms -- ~10 instructions when NOT taken, a basic-block cut per row block, and
    // its per-wavefront variance in front of every barrier).  Now: for every lane that can still win its row, the tile is
    // multiplied ONCE MORE -- same instruction, same operands, same order: the same bits -- and the first register equal to the
    // maximum names the node.  ~34 lanes per row block (the two lanes of a row: the larger maximum, on a tie the earlier tile,
    // on a tie of both, both) x KS matrix instructions: ~1 % of the sweep.
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const float m_o = __shfl_xor(rm[rb], 32, 64);
        const int t_o = __shfl_xor(rt[rb], 32, 64);
        const bool need = rt[rb] >= 0 && (rm[rb] > m_o || (rm[rb] == m_o && rt[rb] <= t_o));
        unsigned long long todo = (DBG & 4) ? 0ull : __builtin_amdgcn_ballot_w64(need);
        int arg = 0x7fffffff;
        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int tj = __builtin_amdgcn_readlane(rt[rb], j);
            const u32x4 *src = (const u32x4 *)Eb + (int64_t)((cbeg >> 5) + tj) * KS * 64 + lane;
            f32x16 c;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int nd = cbeg + 32 * tj + 4 * hi + (i & 3) + 8 * (i >> 2);
                c[i] = nd < cend ? bias[nd] : -INFINITY;
            }
            constexpr int KC = KS < 16 ? KS : 16;  // fragments in flight
#pragma unroll
            for (int s0 = 0; s0 < KS; s0 += KC) {
                Frag a[KC];
#pragma unroll
                for (int s2 = 0; s2 < KC; ++s2) a[s2].q = src[64 * (s0 + s2)];
#pragma unroll
                for (int s2 = 0; s2 < KC; ++s2) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s2].v, rf[rb][s0 + s2].v, c, 0, 0, 0);
            }
            int first = 0x7fffffff;
#pragma unroll
            for (int i = 15; i >= 0; --i) first = c[i] == rm[rb] ? (i & 3) + 8 * (i >> 2) : first;  // (first register = lowest node)
            if (lane == j && first != 0x7fffffff) arg = cbeg + 32 * tj + 4 * hi + first;
        }
        // merge the two lanes of a row; lanes < 32 hold row l31 of the block
        float sx = rs[rb];
        if (LSE) {
            if (!(sx <= 3.0e38f) || (sx == 0.f && rm[rb] > -INFINITY)) atomicOr(overflow, 1);
            sx = rm[rb] > -INFINITY ? sx * __builtin_amdgcn_exp2f(-rm[rb] * LOG2E) : 0.f;
        }
        Running x{rm[rb], sx, arg};
        const float m = __shfl_xor(x.m, 32, 64), so = __shfl_xor(x.s, 32, 64);
        const int ar = __shfl_xor(x.arg, 32, 64);
        run_merge(x, m, so, ar, LSE);
        const int row = r0 + 32 * rb + l31;
        if (hi == 0 && row < n_rows) {
            const int64_t o = (int64_t)row * gridDim.x + split;
            part_max[o] = x.m;
            part_arg[o] = x.arg;
            part_sum[o] = x.s;
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
    // bf16, many rows: the x32 kernel -- a workgroup's wavefronts take 32 RB rows each and share ONE sweep of the table, one
    // workgroup per compute unit (all_score_reduce_bf16_x32_kernel)
    static thread_local bool force_narrow = false;  // set for the repeat of a call whose wide kernel reported an overflowing sum
    const bool wide = precision == 1 && n_rows >= 512 && !getenv("GG_ALLPAIRS_NARROW") && !force_narrow;
    static const bool one_wave = getenv("GG_K7_NW4") != nullptr;  // (A/B at d <= 256: one wavefront per SIMD with 4 row blocks)
    const int tile_rows = wide ? (KS <= 16 ? 512 : 256) : 32 * RB;  // (x32: 32 RB NW rows)
    const int row_tiles = cdiv(n_rows, tile_rows);
    // enough workgroups for the chip: split the columns when there are few row tiles (multiples of 128 columns)
    int splits = std::max(1, std::min(cdiv(n, 128), cdiv(wide ? 256 : 2048, row_tiles)));
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
#define GG_BF16_X32(KSV, RBV, NWV)                                                                                                                         \
    do {                                                                                                                                               \
        const size_t dyn = 4 * (KSV) * 1024 + 1024 + 256 * (NWV);                                                                                                   \
        if (want_lse) {                                                                                                                                \
            (void)hipFuncSetAttribute((const void *)all_score_reduce_bf16_x32_kernel<KSV, RBV, NWV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
            hipLaunchKernelGGL((all_score_reduce_bf16_x32_kernel<KSV, RBV, NWV, true>), grid, dim3(64 * (NWV)), dyn, ctx->stream, Eb, G.b, n, dr, n_rows, cps, \
                               d_pm.as<float>(), d_pa.as<int32_t>(), d_ps.as<float>(), d_ovf.as<int32_t>());                                             \
        } else {                                                                                                                                       \
            (void)hipFuncSetAttribute((const void *)all_score_reduce_bf16_x32_kernel<KSV, RBV, NWV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
            hipLaunchKernelGGL((all_score_reduce_bf16_x32_kernel<KSV, RBV, NWV, false>), grid, dim3(64 * (NWV)), dyn, ctx->stream, Eb, G.b, n, dr, n_rows, cps, \
                               d_pm.as<float>(), d_pa.as<int32_t>(), d_ps.as<float>(), d_ovf.as<int32_t>());                                             \
        }                                                                                                                                              \
    } while (0)
#ifdef GG_K7_ABLATIONS  // (make EXTRA=-DGG_K7_ABLATIONS: the timing ablations of profiles/HISTORY.md, results WRONG)
        static const int k7dbg = getenv("GG_K7_DBG") ? atoi(getenv("GG_K7_DBG")) : 0;
#else
        constexpr int k7dbg = 0;
#endif
        if (wide && k7dbg && KS == 16) {
#ifdef GG_K7_ABLATIONS
            const size_t dyn = 4 * 16 * 1024 + 1024 + 256 * 4;
#define GG_X32_DBG(D)                                                                                                                         \
    case D:                                                                                                                                   \
        (void)hipFuncSetAttribute((const void *)all_score_reduce_bf16_x32_kernel<16, 4, 4, false, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
        hipLaunchKernelGGL((all_score_reduce_bf16_x32_kernel<16, 4, 4, false, D>), grid, dim3(256), dyn, ctx->stream, Eb, G.b, n, dr, n_rows, cps,    \
                           d_pm.as<float>(), d_pa.as<int32_t>(), d_ps.as<float>(), d_ovf.as<int32_t>());                                        \
        break;
            switch (k7dbg) {
                GG_X32_DBG(1) GG_X32_DBG(2) GG_X32_DBG(4) GG_X32_DBG(8) GG_X32_DBG(16) GG_X32_DBG(27) GG_X32_DBG(128)
            }
#undef GG_X32_DBG
#endif
        } else if (wide) {
            if (KS <= 4) { GG_BF16_X32(4, 2, 8); }
            else if (KS <= 8) { GG_BF16_X32(8, 2, 8); }
            else if (KS <= 16) { if (one_wave) { GG_BF16_X32(16, 4, 4); } else { GG_BF16_X32(16, 2, 8); } }
            else { GG_BF16_X32(32, 1, 8); }
        } else if (KS <= 4) { GG_BF16_LAUNCH(4, 2); }
        else if (KS <= 8) { GG_BF16_LAUNCH(8, 2); }
        else if (KS <= 16) { GG_BF16_LAUNCH(16, 1); }
        else { GG_BF16_LAUNCH(32, 1); }
#undef GG_BF16_LAUNCH
#undef GG_BF16_X32
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
