// gather_bw3.hip -- ceiling for the access shape of level_score_kernel: one 16-lane group per 16-row chunk,
// a 16-byte descriptor per chunk {current row, rows, offset of the ids}, ids read from a large array at
// scattered positions (the tree arrays are tens of GB), bias gather, 4-byte score store.  Reports TB/s with the
// kernel's own accounting: 4(d+3) bytes per row + 4d+16 per chunk (d = 128).
//   hipcc -O3 --offload-arch=gfx950 tools/gather_bw3.hip -o /tmp/gbw3 && /tmp/gbw3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>

__global__ __launch_bounds__(256) void k(const float4 *E, const int *ids, const int4 *desc, const float *bias, long n_chunks, float *out) {
    const int t = threadIdx.x & 15;
    const long g0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long ng = ((long)gridDim.x * blockDim.x) >> 4;
    for (long c = g0; c < n_chunks; c += ng) {
        const int4 d = desc[c];
        const int nrows = d.y;
        const int *myids = ids + (((long)d.w << 32) | (unsigned)d.z);
        const float4 gc0 = E[(long)d.x * 32 + t], gc1 = E[(long)d.x * 32 + t + 16];
        const int myid = t < nrows ? myids[t] : -1;
        const float mybias = t < nrows ? bias[myid] : 0.f;
        float mysc = 0.f;
        for (int j0 = 0; j0 < nrows; j0 += 4) {
            float4 y[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int id = __shfl(myid, j0 + u, 16);
                const bool v = id >= 0;
                y[u][0] = v ? E[(long)id * 32 + t] : make_float4(0, 0, 0, 0);
                y[u][1] = v ? E[(long)id * 32 + t + 16] : make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float acc = 0.f;
                acc = fmaf(gc0.x, y[u][0].x, acc); acc = fmaf(gc0.y, y[u][0].y, acc); acc = fmaf(gc0.z, y[u][0].z, acc); acc = fmaf(gc0.w, y[u][0].w, acc);
                acc = fmaf(gc1.x, y[u][1].x, acc); acc = fmaf(gc1.y, y[u][1].y, acc); acc = fmaf(gc1.z, y[u][1].z, acc); acc = fmaf(gc1.w, y[u][1].w, acc);
                acc += __shfl_xor(acc, 8, 64); acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 1, 64);
                if (t == j0 + u) mysc = acc + mybias;
            }
        }
        if (t < nrows) out[c * 16 + t] = mysc;
    }
}

int main() {
    const long rows = 1l << 20, n_chunks = 1l << 19;
    for (int variant = 0; variant < 4; ++variant) {
        const bool ragged = variant & 1, far_ids = variant & 2;
        const long ids_len = far_ids ? (4l << 30) : n_chunks * 16;  // 16 GB of ids vs a compact array
        float4 *E; int *ids; float *out, *bias; int4 *desc;
        hipMalloc(&E, rows * 512); hipMalloc(&ids, ids_len * 4); hipMalloc(&out, n_chunks * 64); hipMalloc(&bias, rows * 4); hipMalloc(&desc, n_chunks * 16);
        hipMemset(E, 0, rows * 512); hipMemset(bias, 0, rows * 4);
        std::mt19937_64 rg(1);
        std::vector<int4> hd(n_chunks);
        std::vector<int> blockids(16);
        long tot_rows = 0;
        // fill only the id blocks that are used (far variant: scattered 64-byte blocks)
        std::vector<int> hids;
        if (!far_ids) hids.resize(n_chunks * 16);
        for (long c = 0; c < n_chunks; ++c) {
            const int nr = ragged ? 9 + (int)(rg() % 8) : 16;  // 9..16 rows, mean 12.5
            const long off = far_ids ? (long)((rg() % (ids_len / 16)) * 16) : c * 16;
            hd[c] = make_int4((int)(rg() % rows), nr, (int)(off & 0xffffffffl), (int)(off >> 32));
            for (int i = 0; i < 16; ++i) blockids[i] = (int)(rg() % rows);
            if (far_ids) hipMemcpy(ids + off, blockids.data(), 64, hipMemcpyHostToDevice);
            else for (int i = 0; i < 16; ++i) hids[c * 16 + i] = blockids[i];
            tot_rows += nr;
        }
        if (!far_ids) hipMemcpy(ids, hids.data(), n_chunks * 64, hipMemcpyHostToDevice);
        hipMemcpy(desc, hd.data(), n_chunks * 16, hipMemcpyHostToDevice);
        const double bytes = 4.0 * 131 * tot_rows + (4.0 * 128 + 16) * n_chunks;
        for (int blocks : {1536, 2048}) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            k<<<blocks, 256>>>(E, ids, desc, bias, n_chunks, out); hipDeviceSynchronize();
            hipEventRecord(a);
            for (int r = 0; r < 5; ++r) k<<<blocks, 256>>>(E, ids, desc, bias, n_chunks, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("chunks of %s rows, ids %s, %d blocks: %.2f TB/s (%.1f us per launch, %.2f GB)\n", ragged ? "9..16" : "16", far_ids ? "scattered over 16 GB" : "compact",
                   blocks, 5.0 * bytes / (ms * 1e-3) / 1e12, ms / 5 * 1e3, bytes / 1e9);
        }
        hipFree(E); hipFree(ids); hipFree(out); hipFree(bias); hipFree(desc);
    }
    return 0;
}
