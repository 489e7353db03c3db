// mfma_peak.hip -- what the chip sustains on back-to-back bf16 matrix instructions from registers, per instruction shape and
// wavefronts per SIMD.  Round 3's DESIGN.md claimed "1.56 PFLOP/s is what this chip sustains"; that number came from an ablation
// of the all-pairs kernel at ONE wavefront per SIMD (amdgpu_waves_per_eu(1,1)) with its register traffic around the instructions --
// a property of that kernel, not of the chip (MI355X_MICROARCH.md measures 2 495 TFLOP/s on 32x32x16).  This tool settles it:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NACC independent accumulators per wavefront, ITERS rounds over them: 2 * 32 * 32 * 16 flop per 32x32x16 instruction
template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(blockIdx.x - i); }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][15];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(blockIdx.x - i); }
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
    if (s == 12345.678f) out[0] = s;
}

template <class F>
double time_ms(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 3.0;
}

int main() {
    float *out;
    hipMalloc(&out, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    printf("%s: %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
    // waves per SIMD = blocks per CU (a block of 256 threads = 4 wavefronts = one per SIMD)
    for (int wps : {1, 2, 4}) {
        const int blocks = cus * wps;
        const double m32 = time_ms([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        const double f32 = 2.0 * 32 * 32 * 16 * 4.0 * iters * (double)blocks * 4;
        const double m16 = time_ms([&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        const double f16 = 2.0 * 16 * 16 * 32 * 8.0 * iters * (double)blocks * 4;
        printf("%d wavefront(s) per SIMD: v_mfma_f32_32x32x16_bf16 (4 independent accumulators) %.0f TFLOP/s in %.2f ms; "
               "v_mfma_f32_16x16x32_bf16 (8 accumulators) %.0f TFLOP/s in %.2f ms\n", wps, f32 / (m32 * 1e-3) / 1e12, m32, f16 / (m16 * 1e-3) / 1e12, m16);
    }
    return 0;
}
