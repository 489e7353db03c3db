// gather_bw2.hip -- which ingredient of the score kernel costs bandwidth?  Variants add one
// ingredient at a time to the plain random-row gather of gather_bw.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_nt(const float4 *p) { const f32x4 v = __builtin_nontemporal_load((const f32x4 *)p); return make_float4(v.x, v.y, v.z, v.w); }

// V=0 plain gather+sum; V=1 + dot with a current row (fma) + butterfly; V=2 + 4-byte store per row;
// V=3 + bias gather; V=4 + ids through a per-64-row descriptor (extra dependent load)
template <int V, bool NT>
__global__ __launch_bounds__(256) void k(const float4 *E, const int *ids, const int4 *desc, const float *bias, long n, float *out) {
    const int t = threadIdx.x & 15;
    const long g0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long ng = ((long)gridDim.x * blockDim.x) >> 4;
    float accs = 0.f;
    for (long c = g0; c * 64 < n; c += ng) {  // one 64-row chunk per group
        const int *myids = ids + c * 64;
        int cur = (int)(c & 1023);
        if (V >= 4) { const int4 d = desc[c]; myids = ids + (long)d.z; cur = d.x; }
        float4 gc0 = E[(long)cur * 32 + t], gc1 = E[(long)cur * 32 + t + 16];
        for (int jb = 0; jb < 64; jb += 16) {
            const int myid = myids[jb + t];
            for (int j0 = 0; j0 < 16; j0 += 4) {
                float4 y[4][2]; int id[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    id[u] = __shfl(myid, j0 + u, 16);
                    y[u][0] = NT ? ld_nt(&E[(long)id[u] * 32 + t]) : E[(long)id[u] * 32 + t];
                    y[u][1] = NT ? ld_nt(&E[(long)id[u] * 32 + t + 16]) : E[(long)id[u] * 32 + t + 16];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float acc;
                    if (V >= 1) {
                        acc = 0.f;
                        acc = fmaf(gc0.x, y[u][0].x, acc); acc = fmaf(gc0.y, y[u][0].y, acc); acc = fmaf(gc0.z, y[u][0].z, acc); acc = fmaf(gc0.w, y[u][0].w, acc);
                        acc = fmaf(gc1.x, y[u][1].x, acc); acc = fmaf(gc1.y, y[u][1].y, acc); acc = fmaf(gc1.z, y[u][1].z, acc); acc = fmaf(gc1.w, y[u][1].w, acc);
                        acc += __shfl_xor(acc, 8, 64); acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 1, 64);
                    } else acc = y[u][0].x + y[u][1].w;
                    if (V >= 3) acc += bias[id[u]];
                    if (V >= 2) { if (t == 0) out[c * 64 + jb + j0 + u] = acc; } else accs += acc;
                }
            }
        }
    }
    if (accs == 123.456f) out[0] = accs;
}

// V5: the bias travels in the row: 576-byte rows (512 B of values + a 64-byte tail whose first float is the bias)
__global__ __launch_bounds__(256) void k5(const float4 *E2, const int *ids, const int4 *desc, long n, float *out) {
    const int t = threadIdx.x & 15;
    const long g0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long ng = ((long)gridDim.x * blockDim.x) >> 4;
    for (long c = g0; c * 64 < n; c += ng) {
        const int4 d = desc[c];
        const int *myids = ids + (long)d.z;
        const int cur = d.x;
        const float4 gc0 = E2[(long)cur * 36 + t], gc1 = E2[(long)cur * 36 + t + 16];
        for (int jb = 0; jb < 64; jb += 16) {
            const int myid = myids[jb + t];
            for (int j0 = 0; j0 < 16; j0 += 4) {
                float4 y[4][2]; float b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int id = __shfl(myid, j0 + u, 16);
                    y[u][0] = E2[(long)id * 36 + t];
                    y[u][1] = E2[(long)id * 36 + t + 16];
                    b[u] = ((const float *)(E2 + (long)id * 36 + 32))[0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float acc = 0.f;
                    acc = fmaf(gc0.x, y[u][0].x, acc); acc = fmaf(gc0.y, y[u][0].y, acc); acc = fmaf(gc0.z, y[u][0].z, acc); acc = fmaf(gc0.w, y[u][0].w, acc);
                    acc = fmaf(gc1.x, y[u][1].x, acc); acc = fmaf(gc1.y, y[u][1].y, acc); acc = fmaf(gc1.z, y[u][1].z, acc); acc = fmaf(gc1.w, y[u][1].w, acc);
                    acc += __shfl_xor(acc, 8, 64); acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 1, 64);
                    acc += b[u];
                    if (t == 0) out[c * 64 + jb + j0 + u] = acc;
                }
            }
        }
    }
}

template <int V, bool NT = false>
double run(const float4 *E, const int *ids, const int4 *desc, const float *bias, long n, float *out, int blocks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<V, NT><<<blocks, 256>>>(E, ids, desc, bias, n, out); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k<V, NT><<<blocks, 256>>>(E, ids, desc, bias, n, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 5.0 * n * 512.0 / (ms * 1e-3) / 1e12;
}

int main() {
    const long n = 8l << 20, rows = 1l << 20;
    float4 *E; int *ids; float *out, *bias; int4 *desc;
    hipMalloc(&E, rows * 512); hipMalloc(&ids, n * 4); hipMalloc(&out, n * 4); hipMalloc(&bias, rows * 4); hipMalloc(&desc, n / 64 * 16);
    hipMemset(E, 0, rows * 512); hipMemset(bias, 0, rows * 4);
    std::vector<int> h(n); std::mt19937_64 rg(1);
    for (long i = 0; i < n; ++i) h[i] = (int)(rg() % rows);
    hipMemcpy(ids, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<int4> hd(n / 64);
    for (long c = 0; c < n / 64; ++c) { long cc = (c * 7919) % (n / 64); hd[c] = make_int4((int)(rg() % rows), 64, (int)(cc * 64), 0); }
    hipMemcpy(desc, hd.data(), n / 64 * 16, hipMemcpyHostToDevice);
    {
        float4 *E2; (void)hipMalloc(&E2, rows * 576); (void)hipMemset(E2, 0, rows * 576);
        for (int blocks : {1536, 2048}) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            k5<<<blocks, 256>>>(E2, ids, desc, n, out); (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            for (int r = 0; r < 5; ++r) k5<<<blocks, 256>>>(E2, ids, desc, n, out);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf("blocks %d: V5 (bias in a 64-byte tail of 576-byte rows, dot + store + descriptors) %.2f TB/s of 512-byte rows\n", blocks, 5.0 * n * 512.0 / (ms * 1e-3) / 1e12);
        }
        (void)hipFree(E2);
    }
    for (int blocks : {2048}) {
        printf("blocks %d, rows with the non-temporal hint: V2 %.2f  V3(+bias) %.2f  V4(+desc) %.2f TB/s\n", blocks,
               run<2, true>(E, ids, desc, bias, n, out, blocks), run<3, true>(E, ids, desc, bias, n, out, blocks), run<4, true>(E, ids, desc, bias, n, out, blocks));
        printf("blocks %d: V0 %.2f  V1(dot) %.2f  V2(+store) %.2f  V3(+bias) %.2f  V4(+desc) %.2f TB/s\n", blocks,
               run<0>(E, ids, desc, bias, n, out, blocks), run<1>(E, ids, desc, bias, n, out, blocks), run<2>(E, ids, desc, bias, n, out, blocks),
               run<3>(E, ids, desc, bias, n, out, blocks), run<4>(E, ids, desc, bias, n, out, blocks));
    }
    return 0;
}
