// Microbenchmark: v_mfma_f32_32x32x2_f32 throughput for dependent chains, 1 or 2 waves per SIMD,
// one or two independent accumulators per wave.  hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nacc = 1; nacc <= 2; ++nacc)
        for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
            int grid = 256 * wgs_per_cu, iters = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double mf = (double)grid * 4 * iters * 16 * nacc;  // MFMAs
            printf("nacc %d waves/SIMD %d: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD @2.4GHz\n", nacc, wgs_per_cu, ms,
                   mf * 4096 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024));
        }
    return 0;
}
