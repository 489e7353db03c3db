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
