// gather_bw.hip -- micro-benchmark: achievable bandwidth of random 512-byte row gathers on
// MI355X with the walk kernel's access shape (16-lane group per row, float4 per lane, NCH=2,
// U rows in flight per group).  Calibrates the practical roofline of K1.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bw.hip -o /tmp/gather_bw && /tmp/gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

template <int U>
__global__ __launch_bounds__(256) void gather(const float4 *E, const int *ids, long n, float *out) {
    const int t = threadIdx.x & 15;
    const long g0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long ng = ((long)gridDim.x * blockDim.x) >> 4;
    float acc = 0.f;
    for (long i = g0 * U; i < n; i += ng * U) {
        float4 y[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = (i + u < n) ? ids[i + u] : 0;
            y[u][0] = E[r * 32 + t];
            y[u][1] = E[r * 32 + t + 16];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += y[u][0].x + y[u][0].w + y[u][1].y + y[u][1].z;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int U>
double run(const float4 *E, const int *ids, long n, float *out, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    gather<U><<<blocks, 256>>>(E, ids, n, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) gather<U><<<blocks, 256>>>(E, ids, n, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 5.0 * n * 512.0 / (ms * 1e-3) / 1e12;
}

int main() {
    const long n = 16l << 20;
    for (long rows : {1l << 20, 4l << 20, 16l << 20}) {
        float4 *E; int *ids; float *out;
        hipMalloc(&E, rows * 512); hipMalloc(&ids, n * 4); hipMalloc(&out, 4);
        hipMemset(E, 0, rows * 512);
        std::vector<int> h(n);
        std::mt19937_64 rg(1);
        for (long i = 0; i < n; ++i) h[i] = (int)(rg() % rows);
        hipMemcpy(ids, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int blocks : {2048, 4096, 8192}) {
            printf("table %5ld MB blocks %5d : U=1 %.2f  U=2 %.2f  U=4 %.2f  U=8 %.2f TB/s\n", rows * 512 >> 20, blocks,
                   run<1>(E, ids, n, out, blocks), run<2>(E, ids, n, out, blocks), run<4>(E, ids, n, out, blocks), run<8>(E, ids, n, out, blocks));
        }
        // sequential ids = streaming upper bound
        for (long i = 0; i < n; ++i) h[i] = (int)(i % rows);
        hipMemcpy(ids, h.data(), n * 4, hipMemcpyHostToDevice);
        printf("table %5ld MB sequential rows      : U=4 %.2f TB/s\n", rows * 512 >> 20, run<4>(E, ids, n, out, 4096));
        hipFree(E); hipFree(ids); hipFree(out);
    }
    return 0;
}
