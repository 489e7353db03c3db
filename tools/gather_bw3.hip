// gather_bw3.hip -- ceiling for the access shape of level_score_kernel: one 16-lane group per work item of 16 / 32 / 64 rows,
// a 16-byte descriptor per chunk {current row, rows, offset of the ids}, ids read from a large array at
// scattered positions (the tree arrays are tens of GB), bias gather, 4-byte score store.  Reports TB/s with the
// kernel's own accounting: 4(d+3) bytes per row + 4d+16 per chunk (d = 128).
//   hipcc -O3 --offload-arch=gfx950 tools/gather_bw3.hip -o /tmp/gbw3 && /tmp/gbw3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>

template <int ITEM>  // rows per work item (16, 32 or 64): one descriptor and one read of the current row per item
__global__ __launch_bounds__(256) void k(const float4 *E, const int *ids, const int4 *desc, const float *bias, long n_items, float *out) {
    const int t = threadIdx.x & 15;
    const long g0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long ng = ((long)gridDim.x * blockDim.x) >> 4;
    for (long c = g0; c < n_items; c += ng) {
        const int4 d = desc[c];
        const int nrows = d.y;
        const int *myids = ids + (((long)d.w << 32) | (unsigned)d.z);
        const float4 gc0 = E[(long)d.x * 32 + t], gc1 = E[(long)d.x * 32 + t + 16];
        for (int sb = 0; sb * 16 < nrows; ++sb) {
            const int nb = min(16, nrows - sb * 16);
            const int myid = t < nb ? myids[sb * 16 + t] : -1;
            const float mybias = t < nb ? bias[myid] : 0.f;
            float mysc = 0.f;
            for (int j0 = 0; j0 < nb; j0 += 4) {
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
            if (t < nb) out[c * ITEM + sb * 16 + t] = mysc;
        }
    }
}

template <int ITEM>
void run_item(bool ragged, bool far_ids) {
    const long rows = 1l << 20, n_items = (1l << 23) / ITEM;   // 8 M row slots in all
    const long ids_len = far_ids ? (4l << 30) : n_items * ITEM;  // 16 GB of ids vs a compact array
    float4 *E; int *ids; float *out, *bias; int4 *desc;
    (void)hipMalloc(&E, rows * 512); (void)hipMalloc(&ids, ids_len * 4); (void)hipMalloc(&out, n_items * ITEM * 4); (void)hipMalloc(&bias, rows * 4); (void)hipMalloc(&desc, n_items * 16);
    (void)hipMemset(E, 0, rows * 512); (void)hipMemset(bias, 0, rows * 4);
    std::mt19937_64 rg(1);
    std::vector<int4> hd(n_items);
    std::vector<int> blockids(ITEM);
    long tot_rows = 0;
    std::vector<int> hids;
    if (!far_ids) hids.resize(n_items * ITEM);
    for (long c = 0; c < n_items; ++c) {
        const int nr = ragged ? ITEM / 2 + 1 + (int)(rg() % (ITEM / 2)) : ITEM;  // ITEM/2+1 .. ITEM rows
        const long off = far_ids ? (long)((rg() % (ids_len / ITEM)) * ITEM) : c * ITEM;
        hd[c] = make_int4((int)(rg() % rows), nr, (int)(off & 0xffffffffl), (int)(off >> 32));
        for (int i = 0; i < ITEM; ++i) blockids[i] = (int)(rg() % rows);
        if (far_ids) (void)hipMemcpy(ids + off, blockids.data(), ITEM * 4, hipMemcpyHostToDevice);
        else for (int i = 0; i < ITEM; ++i) hids[c * ITEM + i] = blockids[i];
        tot_rows += nr;
    }
    if (!far_ids) (void)hipMemcpy(ids, hids.data(), n_items * ITEM * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(desc, hd.data(), n_items * 16, hipMemcpyHostToDevice);
    const double bytes = 4.0 * 131 * tot_rows + (4.0 * 128 + 16) * n_items;
    const int blocks = 1536;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<ITEM><<<blocks, 256>>>(E, ids, desc, bias, n_items, out); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k<ITEM><<<blocks, 256>>>(E, ids, desc, bias, n_items, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("items of %s%d rows, ids %s: %.2f TB/s algorithmic (%.1f us per launch, %.2f GB; rows alone %.2f TB/s)\n", ragged ? "up to " : "", ITEM,
           far_ids ? "scattered over 16 GB" : "compact", 5.0 * bytes / (ms * 1e-3) / 1e12, ms / 5 * 1e3, bytes / 1e9, 5.0 * 512.0 * tot_rows / (ms * 1e-3) / 1e12);
    (void)hipFree(E); (void)hipFree(ids); (void)hipFree(out); (void)hipFree(bias); (void)hipFree(desc);
}

int main() {
    for (int ragged = 0; ragged < 2; ++ragged) {
        run_item<16>(ragged, false);
        run_item<32>(ragged, false);
        run_item<64>(ragged, false);
    }
    run_item<16>(true, true);
    run_item<64>(true, true);
    return 0;
}
